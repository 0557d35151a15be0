#!/bin/bash
# producer / consumer conv: per-role cycle stamps, then interleaved A/B of build variants (one box)
mkdir -p gpurun_out
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32"
JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_stamp.so timeout 300 python bench.py --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --conv-pc 1 > gpurun_out/pc2_stamp.json 2> gpurun_out/pc2_stamp.err
grep PCSTAMP gpurun_out/pc2_stamp.err | sort | uniq -c | sort -rn | head -0
grep PCSTAMP gpurun_out/pc2_stamp.err | tail -14
run() {  # label lib pc
  JAEGER_HIP_LIB=$2 timeout 300 $B --conv-pc $3 > gpurun_out/pc2_$1.json 2> gpurun_out/pc2_$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/pc2_$1.json"))
print("$1:", d["value"], "Mbp/s frac", d["roofline"]["frac"], "avg ms", d["roofline"]["avg_launch_ms"])
PY
}
for r in 1 2; do
  run classic_$r jaeger_amd/libjaeger_hip.so 0
  run pc_$r jaeger_amd/libjaeger_hip.so 1
  run late0_$r jaeger_amd/libjaeger_hip_late0.so 1
  run prio1_$r jaeger_amd/libjaeger_hip_prio1.so 1
  run prio0_$r jaeger_amd/libjaeger_hip_prio0.so 1
done
