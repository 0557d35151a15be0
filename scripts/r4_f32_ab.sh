# exact-f32 conv: parity of the build in the tree, then interleaved A/B (old = before the batched staging; wpf3 / wpf4 = weight quads 3 / 4 steps ahead)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4i; exec > gpurun_out/r4i/ab.log 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_legacy.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
LIBS="libjaeger_hip.so libjaeger_hip_old.so libjaeger_hip_wpf3.so libjaeger_hip_wpf4.so" bash scripts/gpu_f32_ab.sh
