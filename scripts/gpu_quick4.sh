for d in 24 25 2 0; do JG_DBG=$d python bench.py --contigs 500 --steps 1 --warmup 1 --chunk 256 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg $d', d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_$tag -- python3 $R/bench.py --contigs 150 --steps 1 --warmup 0 --chunk 128 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$R/gpurun_out/pmcf_*")):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(float); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            if "conv_f16x3" in row["Kernel_Name"]:
                agg[row["Counter_Name"]]+=float(row["Counter_Value"]); n[row["Counter_Name"]]+=1
        print({k:(round(v/n[k],1)) for k,v in agg.items()})
    for f in glob.glob(d+"/*/*kernel_trace.csv"):
        ds=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "conv_f16x3" in r["Kernel_Name"]]
        print("avg ns", sum(ds)/len(ds), len(ds))
PY
