cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d; exec > gpurun_out/r5d/run.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "dicodon or encoder or brain_1500 or baseline500" 2>&1 | tail -15
