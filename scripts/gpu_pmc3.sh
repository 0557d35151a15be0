cd /tmp && export TMPDIR=/tmp
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" | sort -u > $R/gpurun_out/counters.txt
wc -l $R/gpurun_out/counters.txt
for dbg in 0 1; do
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS"; do
  tag=$(echo $grp | cut -d' ' -f1-2 | tr ' ' '_')
  JG_DBG=$dbg rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc3_${dbg}_$tag -- python3 $R/bench.py --contigs 300 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$R/gpurun_out/pmc3_*")):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
        for row in csv.DictReader(open(f)):
            if "conv_f16x3" not in row["Kernel_Name"]: continue
            k=row["Kernel_Name"].split("conv_f16x3_kernel")[1][:10]
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); n[k][row["Counter_Name"]]+=1
        for k in sorted(agg):
            print(d.split("/")[-1], k, {c: round(agg[k][c]/n[k][c]) for c in agg[k]})
PY
