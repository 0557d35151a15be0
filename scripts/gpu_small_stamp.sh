# per-phase cycle stamps of the fused small-window kernel (experiment build, JG_SMALL_DBG bit 16)
cd $GRAFT_REPO_ROOT
export JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_exp.so
JG_SMALL_DBG=16 python bench.py --config baseline500 --contigs 100000 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>&1 | grep -E "STAMP" | tail -1
