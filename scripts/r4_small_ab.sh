# small_net_kernel: parity of the build in the tree, then interleaved A/B against jaeger_amd/libjaeger_hip_old.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4h; exec > gpurun_out/r4h/ab.log 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_perf_guards.py -m gpu -q -k "baseline500 or nmdmerge or small or fused or guard" 2>&1 | tail -4
bash scripts/gpu_ab_small.sh
