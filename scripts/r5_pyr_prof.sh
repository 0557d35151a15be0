# pyramid family: per-kernel statistics of the bench command (rocprofv3 --kernel-trace --stats), summary into gpurun_out/r5c
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r5c; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5c/prof -- python3 $R/bench.py --config pyramid --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-exact-f32 --no-also > $R/gpurun_out/r5c/r5_pyramid_bench.json 2>/dev/null
find $R/gpurun_out/r5c/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r5c/r5_pyramid_bench_kernel_stats.csv \;
rm -rf $R/gpurun_out/r5c/prof
cd $R
python3 - <<'PY'
import csv, json
d=json.loads(open("gpurun_out/r5c/r5_pyramid_bench.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['windows_per_gpu'])
rows=list(csv.DictReader(open("gpurun_out/r5c/r5_pyramid_bench_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f}% calls {r['Calls']:>6} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
