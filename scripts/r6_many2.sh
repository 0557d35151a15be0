#!/bin/bash
# round 6, item 5 (after the wave-wide one-run check): repeat-scan tests, CLI tests, then the two timelines
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6many2
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_termini.py tests/test_gpu_cli.py tests/test_gpu_legacy.py -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/r5_e2e_timeline.py many 4 > $O/timeline_many.log 2>&1
python3 $R/scripts/r5_e2e_timeline.py 10k 3 > $O/timeline_10k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/scripts/r5_e2e_timeline.py many 2 > $O/prof_many.log 2>&1
cp $O/prof/*/*kernel_stats.csv $O/many_kernel_stats.csv 2>/dev/null
rm -rf $O/prof
grep -A1 "==" $O/timeline_many.log $O/timeline_10k.log | cut -c1-900
head -8 $O/many_kernel_stats.csv | cut -c1-160
