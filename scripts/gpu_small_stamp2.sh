cd $GRAFT_REPO_ROOT
export JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_abl.so
for dbg in 16 20 24 28 18; do
  echo -n "dbg=$dbg "
  JG_SMALL_DBG=$dbg python bench.py --config baseline500 --contigs 40000 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>&1 | grep -E "STAMP" | tail -1
done
