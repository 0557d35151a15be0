# round 5, first contact: the new driver-style line (headline + also.frag1m + also.baseline500 + box), then the cost of the
# in-region HIP-event profiling (interleaved A/B: events on / off, two pairs), then the box calibration three times over
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5a; exec > gpurun_out/r5a/run.log 2>&1
date
( time python bench.py ) > gpurun_out/r5a/default.json 2> gpurun_out/r5a/default.err
tail -3 gpurun_out/r5a/default.err
for i in 1 2; do
  python bench.py --steps 3 --no-also --no-e2e --no-cpu-baseline --no-exact-f32 > gpurun_out/r5a/prof_on_$i.json 2>> gpurun_out/r5a/ab.err
  python bench.py --steps 3 --no-also --no-e2e --no-cpu-baseline --no-exact-f32 --no-profile > gpurun_out/r5a/prof_off_$i.json 2>> gpurun_out/r5a/ab.err
done
python - <<'PY'
import json, glob
d = json.loads(open("gpurun_out/r5a/default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "box", json.dumps(d.get("box")))
for k, v in (d.get("also") or {}).items():
    print("also", k, json.dumps(v)[:1500])
print("e2e", json.dumps(d.get("e2e"))[:600])
for f in sorted(glob.glob("gpurun_out/r5a/prof_*.json")):
    x = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, x["value"], x["ms_per_step"], x["roofline"]["avg_launch_ms"], json.dumps(x.get("box")))
PY
date
