cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
JG_DBG=${1:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ksplit -- python3 $R/bench.py --contigs 300 --steps 1 --warmup 1 --chunk 256 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/ksplit/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(r["Name"][-60:], r["Calls"], r["AverageNs"], r["Percentage"])
PY
