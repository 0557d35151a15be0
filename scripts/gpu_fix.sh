cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fix
JAEGER_FUZZ_SEEDS=200 timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_perf_guards.py tests/test_gpu_legacy.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E  " | tail -12 | tee gpurun_out/fix/tests.log
