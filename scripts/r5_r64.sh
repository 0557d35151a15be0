# round 5: 64-channel fused residual block (jg_resblock64.hip) - phase stamps when an experiment library built with
# -DR6_STAMP is present (hipcc ... -DR6_STAMP -c jg_resblock64.hip, linked with the other objects into
# jaeger_amd/libjaeger_hip_r6s.so), then the parity tests and the plain pyramid line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5r; exec > gpurun_out/r5r/run.log 2>&1
if [ -f jaeger_amd/libjaeger_hip_r6s.so ]; then
JAEGER_HIP_LIB=$PWD/jaeger_amd/libjaeger_hip_r6s.so timeout 300 python bench.py --config pyramid --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --no-box --steps 2 2>&1 >/dev/null | grep "r6 stamp" | tail -3
fi
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -s -k "narrow_residual or pyramid" 2>&1 | grep -v Warning | tail -8
timeout 300 python bench.py --config pyramid --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --steps 3 > gpurun_out/r5r/pyramid.json 2> gpurun_out/r5r/pyramid.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5r/pyramid.json").read().strip().splitlines()[-1])
print("pyramid", d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('box',{}).get('mfma_loop_tflops'))
PY
bash scripts/r5_pyr_prof.sh 2>&1 | grep -i "resblock\|^[0-9]"
