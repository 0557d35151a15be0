cd $GRAFT_REPO_ROOT
python scripts/gpu_ingest_time.py 2>&1 | grep -v "amdgpu.ids"
