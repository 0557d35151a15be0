cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pyr
JAEGER_FUZZ_SEEDS=300 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -v "Warning\|warn" | grep "^E  \|passed\|failed\|FAILED" | tail -40 | tee gpurun_out/pyr/fuzz.log
