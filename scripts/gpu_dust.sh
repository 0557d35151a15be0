cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dust
timeout 900 python -m pytest tests/test_gpu_dust.py -m gpu -x -q 2>&1 | grep -v "Warning\|warn" | tail -30 | tee gpurun_out/dust/tests.log
