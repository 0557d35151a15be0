cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5b; exec > gpurun_out/r5b/run.log 2>&1
python scripts/r5_e2e_timeline.py 10k 4 2>&1 | grep "==\|@\|before"
