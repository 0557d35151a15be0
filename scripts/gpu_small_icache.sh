#!/bin/bash
# instruction-cache counters of small_net_kernel (baseline500): is the fully unrolled code (about 85 KB per row) fetch-bound?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/small_ic; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQC_ICACHE[A-Z_]*|SQ_IFETCH[A-Z_]*|SQ_INST_LEVEL[A-Z_]*|SQC_INST[A-Z_]*|SQ_WAIT_INST[A-Z_]*|SQ_INST_CYCLES[A-Z_]*)\b" | sort -u > $O/avail.txt
cat $O/avail.txt | tr '\n' ' '; echo
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/bench.py --config baseline500 --contigs 200000 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e > /dev/null 2> $O/g$i.err
done
python3 - <<PY
import csv, glob, collections
O = "$O"
for g in range(1, 8):
    cc = sorted(glob.glob("%s/g%d/*/*counter_collection.csv" % (O, g)))
    if not cc:
        print("group", g, "no output"); continue
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(cc[-1])):
        if "small_net_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in agg: print("%-32s %14.0f per launch (%d launches)" % (k, agg[k] / cnt[k], cnt[k]))
PY
