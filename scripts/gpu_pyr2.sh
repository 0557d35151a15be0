cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pyr
JAEGER_FUZZ_SEEDS=600 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 4 2>&1 | grep -E "passed|failed|FAILED|^E  " | tail -20 | tee gpurun_out/pyr/fuzz.log
