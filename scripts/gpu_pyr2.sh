cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pyr
JAEGER_FUZZ_SEEDS=64 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "Warning\|warn" | tail -40 | tee gpurun_out/pyr/fuzz.log
