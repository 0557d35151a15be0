cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pyr
timeout 900 python scripts/gpu_pyramid.py pyramid 2000 16384 0.85 2>&1 | grep -vi "warn" | tee gpurun_out/pyr/pyramid.log | tail -30
JAEGER_FUZZ_SEEDS=300 timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_robustness.py tests/test_gpu_legacy.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E  " | tail -30 | tee gpurun_out/pyr/tests.log
