set -x
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_f16x3.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --contigs 1000 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
cp $R/gpurun_out/prof_bench/*/*kernel_stats.csv $R/gpurun_out/f16x3_kernel_stats.csv
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$grp -- python3 $R/bench.py --contigs 1000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections,json
res={}
for d in sorted(glob.glob("$R/gpurun_out/pmc_*")):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
        for row in csv.DictReader(open(f)):
            name=row["Kernel_Name"]
            k="conv_f16x3_kernel" if "conv_f16x3_kernel" in name else name.split("(")[0][:60]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[k][row["Counter_Name"]]+=1
        for k in agg:
            for c in agg[k]:
                res.setdefault(k,{})[c]={"mean":agg[k][c]/n[k][c],"launches":n[k][c]}
json.dump(res,open("$R/gpurun_out/pmc_traffic_raw.json","w"),indent=1)
for k,v in res.items(): print(k,v)
PY
