cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_')
  JG_DBG=$1 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1_$tag -- python3 $R/bench.py --contigs 150 --steps 1 --warmup 0 --chunk 128 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$R/gpurun_out/pmc_$1_*")):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(float); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            if "conv_f16x3" in row["Kernel_Name"]:
                agg[row["Counter_Name"]]+=float(row["Counter_Value"]); n[row["Counter_Name"]]+=1
        print(d.split("/")[-1], {k:(round(v/n[k],1), n[k]) for k,v in agg.items()})
PY
