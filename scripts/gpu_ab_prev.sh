# A/B on one box: current library vs the previous build (jaeger_amd/libjaeger_hip_prev.so), default bench, interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for lib in libjaeger_hip.so libjaeger_hip_prev.so; do
    echo -n "$lib: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline'].get('table_kernel'))"
  done
done 2>&1 | tee gpurun_out/ab/ab_prev.log
