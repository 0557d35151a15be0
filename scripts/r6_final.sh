#!/bin/bash
# round 6, final tree: full GPU suite (wall time), smoke, 400 fuzz seeds
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f
mkdir -p $O
cd $R
START=$(date +%s)
python -m pytest tests/ -x -q -m gpu > $O/run.log 2>&1
echo "exit $? wall $(( $(date +%s) - START )) s" >> $O/run.log
tail -22 $O/run.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
START=$(date +%s)
JAEGER_FUZZ_SEEDS=400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/fuzz400.log 2>&1
echo "exit $? wall $(( $(date +%s) - START )) s" >> $O/fuzz400.log
tail -4 $O/fuzz400.log
