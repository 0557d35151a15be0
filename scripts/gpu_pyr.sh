cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pyr
timeout 900 python scripts/gpu_pyramid.py pyramid 2000 8192 0.85 2>&1 | grep -vi "warn" | tee gpurun_out/pyr/pyramid.log | tail -30
