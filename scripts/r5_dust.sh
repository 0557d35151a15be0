# round 5: DUST filter - parity with the host scan, then the 10 000-contig end-to-end timeline (500-bp model) and the kernel statistics of one such run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d; exec > gpurun_out/r5d/run.log 2>&1
timeout 600 python -m pytest tests/test_gpu_dust.py tests/test_gpu_termini.py -q -x 2>&1 | tail -3
python scripts/r5_dust_time.py 2>&1 | tail -3
python scripts/r5_e2e_timeline.py 10k 4 2>&1 | grep "=="
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2e -- python3 $R/scripts/r5_e2e_timeline.py 10k 3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob("/tmp/prof_e2e/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{float(r['TotalDurationNs'])/1e6/3:8.2f} ms/run calls/run {int(r['Calls'])/3:7.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:90]}")
PY
