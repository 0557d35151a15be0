# HBM traffic of the conv kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no other tracing)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$grp -- python3 $R/bench.py --contigs 1000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections,json
res={}
for d in sorted(glob.glob("$R/gpurun_out/pmc_*SIZE")):
    for f in sorted(glob.glob(d+"/*/*counter_collection.csv"))[-1:]:
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
        for row in csv.DictReader(open(f)):
            name=row["Kernel_Name"]
            k="conv_f16x3_kernel" if "conv_f16x3_kernel" in name else name.split("(")[0][:60]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[k][row["Counter_Name"]]+=1
        for k in agg:
            for c in agg[k]:
                res.setdefault(k,{})[c]={"mean":agg[k][c]/n[k][c],"launches":n[k][c]}
json.dump(res,open("$R/gpurun_out/pmc_traffic_raw.json","w"),indent=1)
print(res["conv_f16x3_kernel"])
PY
