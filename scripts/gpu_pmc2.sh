cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  JG_DBG=${1:-0} rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_$i -- python3 $R/bench.py --contigs 200 --steps 1 --warmup 0 --chunk 256 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$R/gpurun_out/pmcg_*")):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
        for row in csv.DictReader(open(f)):
            k=row["Kernel_Name"]
            if "conv_f16x3" in k:
                kk=k[k.index("kernel<"):k.index(">")+1]
                agg[kk][row["Counter_Name"]]+=float(row["Counter_Value"]); n[kk][row["Counter_Name"]]+=1
        for kk in sorted(agg): print(kk, {c:round(v/n[kk][c]/1e6,2) for c,v in agg[kk].items()})
PY
