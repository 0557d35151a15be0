# round-4 final pass on one box: full GPU suite, smoke, the two plain (un-profiled) driver-style bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4z; exec > gpurun_out/r4z/run.log 2>&1
python -m pytest tests -m gpu -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
python bench.py > gpurun_out/r4z/r4_default_bench_plain.json 2> gpurun_out/r4z/default.err
python bench.py --config baseline500 > gpurun_out/r4z/r4_baseline500_bench_plain.json 2> gpurun_out/r4z/b500.err
python - <<'PY'
import json
for f in ("r4_default_bench_plain", "r4_baseline500_bench_plain"):
    d = json.loads(open(f"gpurun_out/r4z/{f}.json").read().strip().splitlines()[-1])
    e = d.get("e2e") or {}
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "e2e", e.get("mbps"), "ratio", round(e.get("mbps", 0) / d["value"], 4), e.get("seconds_each_run"), "f32", d.get("exact_f32_mbps"), "cpu", d["cpu_baseline"]["value"], json.dumps(e.get("stages")))
PY
