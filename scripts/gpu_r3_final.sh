#!/bin/bash
# round-3 final pass on one GPU box: full GPU test suite, smoke, the bench / rocprof / PMC artefacts
# (scripts/gpu_r3_profiles.sh -> gpurun_out/r3p/), the instruction-rate micro-benchmark, the small-window kernel's SQ
# counters and its A/B against the round-2 kernel (jaeger_amd/libjaeger_hip_old.so, if shipped)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_r3.log 2>&1; tail -3 gpurun_out/full_gpu_r3.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_r3.log 2>&1; tail -2 gpurun_out/smoke_r3.log
bash scripts/gpu_r3_profiles.sh 2>&1 | tail -12
O=$GRAFT_REPO_ROOT/gpurun_out/r3p
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/valu_rates scripts/ubench/valu_rates.hip && timeout 120 /tmp/valu_rates > $O/valu_rates.txt
bash scripts/gpu_small_icache.sh 2>&1 | grep "per launch" > $O/small_net_counters.txt
SKIP_UBENCH=1 SKIP_TESTS=1 LIBS="libjaeger_hip.so libjaeger_hip_old.so" bash scripts/gpu_small2.sh 2>&1 | tail -6 > $O/small_ab.txt; cat $O/small_ab.txt
