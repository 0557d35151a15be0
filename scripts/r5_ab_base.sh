# interleaved A/B on ONE box: the library as of the morning's commit 14471b5 (before the phase-split / epilogue changes) against
# the current one, default config, 3 timed steps each, three pairs; then the pyramid / strided parity tests + fuzz on the current one
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5ab
for i in 1 2 3; do
  JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_r5base.so python bench.py --steps 3 --no-also --no-e2e --no-cpu-baseline --no-exact-f32 > gpurun_out/r5ab/base_$i.json 2>/dev/null
  python bench.py --steps 3 --no-also --no-e2e --no-cpu-baseline --no-exact-f32 > gpurun_out/r5ab/new_$i.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5ab/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["box"]["mfma_loop_tflops"])
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "pyramid or strided or mask_modes or narrow_residual" 2>&1 | tail -3
JAEGER_FUZZ_SEEDS=200 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2
python bench.py --config pyramid --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pyramid', d['value'], d['roofline']['frac'], d['box']['mfma_loop_tflops'])"
