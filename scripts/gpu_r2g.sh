cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nmdmerge or default_precision or baseline500 or small" > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2g/pytest.log | tail -5
for i in 1 2; do
python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('v2', d['value'], d['roofline']['fused_small_kernel'])"
JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_exp.so JG_SMALL_V1=1 python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('v1', d['value'], d['roofline']['fused_small_kernel'])"
done
