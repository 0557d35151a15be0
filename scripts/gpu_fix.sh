cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fix
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_dust.py tests/test_gpu_configs.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E  " | tail -12 | tee gpurun_out/fix/tests.log
