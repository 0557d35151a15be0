# one box: terminal-repeat tests, resident bench lines of both model families (with their e2e legs), e2e repeats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4d; exec > gpurun_out/r4d/run.log 2>&1
export JAEGER_NO_CPROFILE=1
python -m pytest tests/test_gpu_termini.py tests/test_gpu_reference_kats.py -m gpu -q 2>&1 | tail -5
python scripts/r4_e2e_prof.py both 3 2>&1 | grep "== \|GPU worker\|wall time" | grep -v "^    "
python bench.py --no-cpu-baseline --no-exact-f32 --steps 2 > gpurun_out/r4d/default.json 2> gpurun_out/r4d/default.err
python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 --steps 3 > gpurun_out/r4d/b500.json 2> gpurun_out/r4d/b500.err
python - <<'PY'
import json
for f in ("default", "b500"):
    d = json.loads(open(f"gpurun_out/r4d/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], "e2e", json.dumps(d.get("e2e"))[:900])
PY
tail -3 gpurun_out/r4d/*.err
