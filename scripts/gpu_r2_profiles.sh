# round-2 profiles: default bench and baseline500 bench under rocprofv3 --kernel-trace --stats (bench line + kernel
# summary of the SAME command), then HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) for both kernels
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
cp $O/prof_default/*/*kernel_stats.csv $O/default_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -- python3 $R/bench.py --config baseline500 > $O/bench_baseline500.json 2> $O/bench_baseline500.err
cp $O/prof_small/*/*kernel_stats.csv $O/baseline500_kernel_stats.csv
for cfg in default baseline500; do
  n=1000; [ $cfg = baseline500 ] && n=200000
  for grp in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_${cfg}_$grp -- python3 $R/bench.py --config $cfg --contigs $n --steps 1 --warmup 0 --no-cpu-baseline --no-exact-f32 > /dev/null 2>&1
  done
done
python3 - <<PY
import csv,glob,collections,json
res={}
for cfg in ("default","baseline500"):
    for d in sorted(glob.glob("$O/pmc_%s_*SIZE" % cfg)):
        for f in sorted(glob.glob(d+"/*/*counter_collection.csv"))[-1:]:
            agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
            for row in csv.DictReader(open(f)):
                name=row["Kernel_Name"]
                k=name.split("(")[0].replace("void ","").replace("(anonymous namespace)::","")[:70]
                agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[k][row["Counter_Name"]]+=1
            for k in agg:
                for c in agg[k]:
                    res.setdefault(cfg,{}).setdefault(k,{})[c]={"mean":agg[k][c]/n[k][c],"launches":n[k][c]}
json.dump(res,open("$O/pmc_traffic_raw.json","w"),indent=1)
for cfg in res:
    for k,v in res[cfg].items():
        if "conv_f16x3" in k or "small_net" in k: print(cfg,k,v)
PY
tail -c 600 $O/bench_default.json; echo; tail -c 900 $O/bench_baseline500.json; echo
head -8 $O/default_kernel_stats.csv | cut -c1-160; head -6 $O/baseline500_kernel_stats.csv | cut -c1-160
rm -rf $O/prof_default $O/prof_small $O/pmc_*_SIZE
