# round 5: phase-split stride-2 convs - placement + parity tests, 400 fuzz seeds, pyramid line under rocprofv3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c; exec > gpurun_out/r5c/run.log 2>&1
python -m pytest tests/test_gpu_parity.py -q -x -k "pyramid or strided or mask_modes" 2>&1 | tail -4
JAEGER_FUZZ_SEEDS=400 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5c/prof -o pyr -- python3 $GRAFT_REPO_ROOT/bench.py --config pyramid --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --steps 2 > $GRAFT_REPO_ROOT/gpurun_out/r5c/pyramid_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r5c/pyramid_prof.err
cd $GRAFT_REPO_ROOT
python bench.py --config pyramid --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --steps 3 > gpurun_out/r5c/pyramid.json 2> gpurun_out/r5c/pyramid.err
find gpurun_out/r5c/prof -name "*kernel_stats.csv" | head -2
python - <<'PY'
import json, glob, csv
for f in ("gpurun_out/r5c/pyramid.json", "gpurun_out/r5c/pyramid_prof.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_convs_incl_table_kernel'], d.get('box',{}).get('mfma_loop_tflops'))
fs=glob.glob("gpurun_out/r5c/prof/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(fs[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:120]}")
PY
