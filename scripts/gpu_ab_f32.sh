cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for lib in libjaeger_hip.so libjaeger_hip_prev.so; do
    echo -n "$lib: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python bench.py --precision f32 --contigs 600 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
  done
done
