# exact-f32 conv: parity (both precisions, legacy tower, fuzz), phase stamps, interleaved A/B of the libraries in LIBS
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4i; exec > gpurun_out/r4i/ab.log 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_legacy.py tests/test_gpu_fuzz.py tests/test_gpu_reference_kats.py -m gpu -q -x 2>&1 | tail -3
[ -f jaeger_amd/libjaeger_hip_f32stamp.so ] && JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_f32stamp.so python bench.py --precision f32 --contigs 600 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e 2>&1 | grep F32STAMP | grep "k=5" | tail -3
bash scripts/gpu_f32_ab.sh
