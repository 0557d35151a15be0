#!/bin/bash
# SQ / GRBM counters of the k = 5 conv kernels, two-workgroup (pc=0) vs producer / consumer (pc=1): MFMA busy, wave
# cycles, waits, instruction mix, LDS activity, effective clock.  Counters in their own runs (no trace domains but
# --kernel-trace).  Output: gpurun_out/pcpmc/summary.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pcpmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LIB=${1:-$R/jaeger_amd/libjaeger_hip.so}
export JAEGER_HIP_LIB=$LIB
G1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
G2="GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES"
G3="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM"
for pc in 0 1; do
  i=0
  for grp in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pc${pc}_g$i -- python3 $R/bench.py --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --conv-pc $pc > $O/pc${pc}_g$i.json 2> $O/pc${pc}_g$i.err
  done
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for pc in (0, 1):
    for g in (1, 2, 3):
        d = "$O/pc%d_g%d" % (pc, g)
        cc = sorted(glob.glob(d + "/*/*counter_collection.csv"))
        kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"))
        if not cc: continue
        dur = {}
        if kt:
            for row in csv.DictReader(open(kt[-1])):
                dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
        durs = collections.defaultdict(float); seen = collections.defaultdict(set)
        for row in csv.DictReader(open(cc[-1])):
            name = row["Kernel_Name"]
            if "conv_pc_kernel" in name: k = "conv_pc"
            elif "conv_f16x3_kernel<5" in name: k = "conv_f16x3_k5"
            else: continue
            # only the big launches (2048 windows): grid 256 (pc) / 512 (two-workgroup)
            if int(row["Grid_Size"]) not in (256 * 512, 512 * 256): continue
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
            if row["Dispatch_Id"] not in seen[k]:
                seen[k].add(row["Dispatch_Id"]); durs[k] += dur.get(row["Dispatch_Id"], 0)
        for k in agg:
            o = out.setdefault("pc%d" % pc, {}).setdefault(k, {})
            for c in agg[k]: o[c] = agg[k][c] / cnt[k][c]
            o.setdefault("launches_g%d" % g, len(seen[k])); o["avg_ns_g%d" % g] = durs[k] / max(len(seen[k]), 1)
json.dump(out, open("$O/summary.json", "w"), indent=1)
for pc in out:
    for k, v in out[pc].items():
        print(pc, k)
        for c in sorted(v): print("   %-32s %.4g" % (c, v[c]))
PY
rm -rf $O/pc*_g*/
