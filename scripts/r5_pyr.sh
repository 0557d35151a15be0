# round 5: fused narrow residual blocks - placement + parity tests, fuzz, pyramid line (fused / layer by layer are compared inside the test)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c; exec > gpurun_out/r5c/run.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -s -k "pyramid or strided or mask_modes or narrow_residual" 2>&1 | grep -v Warning | tail -12
JAEGER_FUZZ_SEEDS=100 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -4
timeout 300 python bench.py --config pyramid --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --steps 3 > gpurun_out/r5c/pyramid.json 2> gpurun_out/r5c/pyramid.err
tail -3 gpurun_out/r5c/pyramid.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5c/pyramid.json").read().strip().splitlines()[-1])
print("pyramid", d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_convs_incl_table_kernel'], d.get('box',{}).get('mfma_loop_tflops'))
PY
bash scripts/r5_pyr_prof.sh 2>&1 | head -14
