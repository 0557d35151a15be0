#!/bin/bash
# round 6: the full GPU suite with its wall time (<= 600 s) + smoke
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6s
mkdir -p $O
cd $R
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
START=$(date +%s)
python -m pytest tests/ -x -q -m gpu > $O/run.log 2>&1
echo "exit $? wall $(( $(date +%s) - START )) s" >> $O/run.log
tail -30 $O/run.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
