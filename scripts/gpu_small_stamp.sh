# per-phase cycle stamps of the fused small-window kernels (experiment build, JG_SMALL_DBG bit 16): v2 (two waves per
# row) and, with JG_SMALL_V1=1, the one-wave-per-row kernel
cd $GRAFT_REPO_ROOT
export JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_exp.so
for v1 in 0 1; do
  if [ $v1 = 1 ]; then export JG_SMALL_V1=1; else unset JG_SMALL_V1; fi
  echo -n "v1=$v1 "
  JG_SMALL_DBG=16 python bench.py --config baseline500 --contigs 100000 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>&1 | grep -E "STAMP" | tail -1
done
