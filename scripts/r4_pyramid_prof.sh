#!/bin/bash
# pyramid family: plain line + per-kernel statistics of the same command
mkdir -p gpurun_out/r4pyr; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4pyr/prof -- python3 $R/bench.py --config pyramid --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-exact-f32 > $R/gpurun_out/r4pyr/prof.json 2>/dev/null
find $R/gpurun_out/r4pyr/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r4pyr/kernel_stats.csv \;
rm -rf $R/gpurun_out/r4pyr/prof
cut -c1-260 $R/gpurun_out/r4pyr/kernel_stats.csv | head -40
