cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pyr
timeout 900 python scripts/gpu_pyramid.py pyramid 2000 16384 0.85 2>&1 | grep -vi "warn" | tee gpurun_out/pyr/pyramid.log | tail -30
JAEGER_FUZZ_SEEDS=64 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "Warning\|warn" | tail -30 | tee gpurun_out/pyr/tests.log
