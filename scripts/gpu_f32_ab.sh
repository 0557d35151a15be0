#!/bin/bash
# exact-f32 path A/B: bench.py --precision f32 on a small contig sample, libraries listed in LIBS, interleaved
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for lib in ${LIBS:-libjaeger_hip.so libjaeger_hip_pf2.so}; do
    [ -f jaeger_amd/$lib ] || continue
    echo -n "$lib: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib timeout 600 python bench.py --precision f32 --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('frac'), d['roofline'].get('avg_launch_ms'))"
  done
done
