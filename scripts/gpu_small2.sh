#!/bin/bash
# small-window kernel round 3: instruction-rate micro-benchmark, the small-window parity tests, then an interleaved A/B
# of the baseline500 bench line between the tree's library and jaeger_amd/libjaeger_hip_old.so (and any other
# libjaeger_hip_<name>.so listed in LIBS)
cd $GRAFT_REPO_ROOT
O=gpurun_out/small2; mkdir -p $O
if [ -z "$SKIP_UBENCH" ]; then
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o /tmp/valu_rates scripts/ubench/valu_rates.hip && timeout 120 /tmp/valu_rates | tee $O/valu_rates.txt
fi
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small or baseline500 or nmdmerge or config3" 2>&1 | tail -5
fi
LIBS=${LIBS:-"libjaeger_hip.so libjaeger_hip_old.so"}
for i in 1 2 3; do
  for lib in $LIBS; do
    [ -f jaeger_amd/$lib ] || continue
    echo -n "$lib: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib timeout 300 python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('avg_launch_ms'))"
  done
done | tee $O/ab.txt
