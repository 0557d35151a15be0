JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_stamp.so python bench.py --contigs 45 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep -E "STAMP|metric" | cut -c1-250 | tail -40
