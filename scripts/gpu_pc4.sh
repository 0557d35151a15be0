#!/bin/bash
mkdir -p gpurun_out
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e"
run() {  # label lib pc
  JAEGER_HIP_LIB=$2 timeout 300 $B --conv-pc $3 > gpurun_out/pc4_$1.json 2> gpurun_out/pc4_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/pc4_$1.json"))
print("$1:", d["value"], "Mbp/s frac", d["roofline"]["frac"], "avg ms", d["roofline"]["avg_launch_ms"])
PY
}
for r in 1 2; do
  run classic_$r jaeger_amd/libjaeger_hip.so 0
  for v in $VARIANTS; do run ${v}_$r jaeger_amd/libjaeger_hip_$v.so 1; done
done
for v in $STAMPS; do
JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_$v.so timeout 300 python bench.py --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --conv-pc 1 > gpurun_out/pc4_$v.json 2> gpurun_out/pc4_$v.err
echo $v; grep PCSTAMP gpurun_out/pc4_$v.err | grep "rows=12288" | tail -3
done
