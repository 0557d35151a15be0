cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dust
timeout 1500 python -m pytest tests/test_gpu_dust.py tests/test_gpu_cli.py tests/test_gpu_legacy.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|^E  |Error" | tail -30 | tee gpurun_out/dust/tests.log
