# per-phase shader cycles of every wave (make stamp); 60 contigs = a few full chunks of 1 024 windows
JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_stamp.so python bench.py --contigs 60 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "STAMP" | grep "rows=6144" | cut -c1-250 | tail -8
