# A/B on one box: current library vs an older build (jaeger_amd/libjaeger_hip_old.so), baseline500 bench, interleaved
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in libjaeger_hip.so libjaeger_hip_old.so; do
    echo -n "$lib: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('fused_small_kernel'))"
  done
done
