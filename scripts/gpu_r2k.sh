cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
JAEGER_FUZZ_SEEDS=300 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -rs 2>&1 | grep -vi "warn\|disabled\|eng = \|^$" > gpurun_out/r2k/fuzz.log; grep -E "^E   |passed|failed|SKIP|Fatal" gpurun_out/r2k/fuzz.log | grep -v "assert ((" | head -60
