# round 5: one million 500-bp records end to end (500-bp model): timeline, then the kernel statistics of such a run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5m; exec > gpurun_out/r5m/run.log 2>&1
python scripts/r5_e2e_timeline.py many 3 2>&1 | grep "==\|@"
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2e -- python3 $R/scripts/r5_e2e_timeline.py many 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob("/tmp/prof_e2e/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"{float(r['TotalDurationNs'])/1e6/2:8.2f} ms/run calls/run {int(r['Calls'])/2:7.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
