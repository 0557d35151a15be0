# one box: the two driver-style bench lines (default, baseline500) with their e2e legs, and three plain e2e repeats each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4g; exec > gpurun_out/r4g/run.log 2>&1
export JAEGER_NO_CPROFILE=1
python bench.py --no-cpu-baseline --no-exact-f32 --steps 2 > gpurun_out/r4g/default.json 2> gpurun_out/r4g/default.err
python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 --steps 3 > gpurun_out/r4g/b500.json 2> gpurun_out/r4g/b500.err
python - <<'PY'
import json
for f in ("default", "b500"):
    d = json.loads(open(f"gpurun_out/r4g/{f}.json").read().strip().splitlines()[-1])
    e = d.get("e2e") or {}
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "e2e", e.get("mbps"), "ratio", round(e.get("mbps", 0) / d["value"], 4), e.get("seconds_each_run"), json.dumps(e.get("stages")))
PY
python scripts/r4_e2e_prof.py both 3 2>&1 | grep "== \|GPU worker" | grep -v "^    "
