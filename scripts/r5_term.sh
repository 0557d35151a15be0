# round 5: terminal repeats - the one-run shortcut against the exact kernel, then the million-record end-to-end timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5t; exec > gpurun_out/r5t/run.log 2>&1
timeout 900 python -m pytest tests/test_gpu_termini.py -q -x 2>&1 | tail -12
python scripts/r5_e2e_timeline.py many 3 2>&1 | grep "=="
python scripts/r5_e2e_timeline.py 10k 3 2>&1 | grep "=="
