#!/bin/bash
# first GPU contact of the producer / consumer conv: bit-identity + repeatability tests, then interleaved A/B bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "producer_consumer" 2>&1 | tail -15 > gpurun_out/pc1_tests.log
cat gpurun_out/pc1_tests.log
if grep -q "passed" gpurun_out/pc1_tests.log && ! grep -q "failed" gpurun_out/pc1_tests.log; then
  for r in 1 2; do
    for v in 0 1; do
      timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --conv-pc $v > gpurun_out/pc1_b${v}_$r.json 2> gpurun_out/pc1_b${v}_$r.err
      python - <<PY
import json
d=json.load(open("gpurun_out/pc1_b${v}_$r.json"))
print("pc=$v run $r:", d["value"], "Mbp/s frac", d["roofline"]["frac"], "avg ms", d["roofline"]["avg_launch_ms"])
PY
    done
  done
fi
