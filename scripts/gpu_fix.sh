cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fix
JAEGER_FUZZ_SEEDS=120 timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E  " | tail -12 | tee gpurun_out/fix/tests.log
python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('baseline500', d['value'], d['roofline'].get('fused_small_kernel'))" | tee gpurun_out/fix/b500.log
