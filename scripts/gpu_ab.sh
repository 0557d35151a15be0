#!/bin/bash
# interleaved A/B of bench.py over library builds on one box: RUNS="label:lib:pc ..." (lib = suffix of libjaeger_hip<suffix>.so)
mkdir -p gpurun_out
B="python bench.py --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e ${BENCH_ARGS}"
for r in 1 2 ${EXTRA_ROUNDS}; do
  for spec in $RUNS; do
    IFS=: read label lib pc <<< "$spec"
    JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip${lib}.so timeout 300 $B --conv-pc $pc > gpurun_out/ab_${label}_$r.json 2> gpurun_out/ab_${label}_$r.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_${label}_$r.json"))
    print("${label} run $r:", d["value"], "Mbp/s frac", d["roofline"]["frac"], "avg ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("${label} run $r: FAILED", e); print(open("gpurun_out/ab_${label}_$r.err").read()[-800:])
PY
  done
done
