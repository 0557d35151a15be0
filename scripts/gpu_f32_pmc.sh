#!/bin/bash
# counters of the exact-f32 conv kernel (bench.py --precision f32 on a short workload)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/f32pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES"
i=0
for grp in "$G1" "$G2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -- python3 $R/bench.py --contigs 400 --steps 1 --warmup 1 --precision f32 --no-cpu-baseline --no-exact-f32 --no-e2e > $O/g$i.json 2> $O/g$i.err
done
python3 - <<PY
import csv, glob, collections
for g in (1, 2):
    d = "$O/g%d" % g
    cc = sorted(glob.glob(d + "/*/*counter_collection.csv")); kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"))
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[-1]))}
    rows = [r for r in csv.DictReader(open(cc[-1])) if "conv_f32_kernel" in r["Kernel_Name"]]
    gmax = max(int(r["Grid_Size"]) for r in rows)
    agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set(); t = 0
    for r in rows:
        if int(r["Grid_Size"]) != gmax: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); t += dur[r["Dispatch_Id"]]
    print("pass", g, "launches", len(seen), "avg ms", t / len(seen) / 1e6, "grid", gmax, "vgpr", rows[0].get("VGPR_Count"), "lds", rows[0].get("LDS_Block_Size"))
    for c in sorted(agg): print("   %-30s %.4g" % (c, agg[c] / cnt[c]))
PY
rm -rf $O/g?/
