#!/bin/bash
# the default bench line again, under rocprofv3 --kernel-trace --stats, with profiles/mfma_util.json + pmc_traffic.json of
# THIS kernel build in place (scripts/gpu_r3_profiles.sh collects them after its own bench run, so that run's line says
# pmc_stale); same for baseline500
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default2 -- python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
cp $O/prof_default2/*/*kernel_stats.csv $O/default_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small2 -- python3 $R/bench.py --config baseline500 > $O/bench_baseline500.json 2> $O/bench_baseline500.err
cp $O/prof_small2/*/*kernel_stats.csv $O/baseline500_kernel_stats.csv
python3 -c "
import json
for f in ('bench_default.json','bench_baseline500.json'):
    d=json.load(open('$O/'+f)); r=d['roofline']
    print(f, d['value'], r.get('frac'), r.get('mfma_busy_frac'), r.get('eff_clock_ghz'), r.get('traffic'), r.get('pmc_stale'), d.get('exact_f32_mbps'), d.get('e2e',{}).get('mbps'))
"
