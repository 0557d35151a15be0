cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
JAEGER_FUZZ_SEEDS=4000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 4 2>&1 | grep -E "passed|failed|FAILED|^E  " | tail -30 | tee gpurun_out/fuzz/fuzz1500.log
