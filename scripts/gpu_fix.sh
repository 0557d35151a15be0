cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fix
JAEGER_FUZZ_SEEDS=120 timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_legacy.py tests/test_gpu_perf_guards.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|^E  |Mbp/s" | tail -16 | tee gpurun_out/fix/tests.log
