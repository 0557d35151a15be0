#!/bin/bash
# copy the artefacts of scripts/gpu_r3_profiles.sh (gpurun_out/r3p/) into the tracked profiles/ directory
cd "$(dirname "$0")/.."
S=gpurun_out/r3p
cp $S/bench_default.json profiles/r3_default_bench.json
cp $S/default_kernel_stats.csv profiles/r3_default_bench_kernel_stats.csv
cp $S/bench_baseline500.json profiles/r3_baseline500_bench.json
cp $S/baseline500_kernel_stats.csv profiles/r3_baseline500_bench_kernel_stats.csv
cp $S/pmc_raw.json profiles/r3_pmc_raw.json
cp $S/pmc_traffic.json profiles/pmc_traffic.json
cp $S/mfma_util.json profiles/mfma_util.json
cp $S/pc_ab.txt profiles/r3_pc_ab.txt
[ -f $S/pc_stamps.txt ] && cp $S/pc_stamps.txt profiles/r3_pc_stamps.txt
python3 - <<PY
import json
d = json.load(open("profiles/r3_default_bench.json")); b = json.load(open("profiles/r3_baseline500_bench.json"))
m = json.load(open("profiles/mfma_util.json"))
print("default", d["value"], "frac", d["roofline"]["frac"], "e2e", d.get("e2e", {}).get("mbps"), "exact_f32", d.get("exact_f32_mbps"), "cpu", d["cpu_baseline"]["value"])
print("baseline500", b["value"], "e2e", b.get("e2e", {}).get("mbps"))
print("hash", m.get("kernel_hash")); print({k: (v.get("mfma_busy_frac"), v.get("eff_clock_ghz")) for k, v in m.items() if isinstance(v, dict)})
PY
