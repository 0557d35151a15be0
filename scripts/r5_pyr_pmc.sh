#!/bin/bash
# round 5: matrix-core busy share of the pyramid's kernels (fused residual blocks, CW = 129 / 128 / 64 convs) from rocprofv3 PMC
# passes of their own (--pmc with --kernel-trace only, the program directly behind `--`): SQ_VALU_MFMA_BUSY_CYCLES + instruction
# counts in one pass, GRBM_GUI_ACTIVE (the clock) in another -> gpurun_out/r5q/r5_pyramid_pmc.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5q
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_g$i -- python3 $R/bench.py --config pyramid --contigs 800 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --no-box > /dev/null 2> $O/pmc_g$i.err
done
python3 - <<PY
import csv, glob, collections, json, re
O = "$O"
def kname(name):
    m = re.search(r"(resblock\d+_kernel<[^>]*>|conv_f16x3_kernel<[^>]*>)", name)
    return m.group(1) if m else None
raw = collections.defaultdict(dict)
for g in (1, 2):
    d = "%s/pmc_g%d" % (O, g)
    cc = sorted(glob.glob(d + "/*/*counter_collection.csv")); kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"))
    if not cc: continue
    dur = {}
    for row in csv.DictReader(open(kt[-1])):
        dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    rows = [r for r in csv.DictReader(open(cc[-1])) if kname(r["Kernel_Name"])]
    gmax = collections.defaultdict(int)
    for r in rows: gmax[kname(r["Kernel_Name"])] = max(gmax[kname(r["Kernel_Name"])], int(r["Grid_Size"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    seen = collections.defaultdict(set); durs = collections.defaultdict(float)
    for r in rows:
        k = kname(r["Kernel_Name"])
        if int(r["Grid_Size"]) != gmax[k]: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"]); durs[k] += dur.get(r["Dispatch_Id"], 0)
    for k in agg:
        for c in agg[k]:
            raw[k][c] = agg[k][c] / cnt[k][c]
            raw[k][c + "__avg_ns"] = durs[k] / max(len(seen[k]), 1)
            raw[k][c + "__launches"] = cnt[k][c]
out = {"_comment": "pyramid bench (800 contigs, 1 step + 1 warmup) under rocprofv3 --pmc, two passes (SQ group; GRBM_GUI_ACTIVE); mean over the largest-grid launches of each kernel; eff_clock_ghz = GRBM_GUI_ACTIVE / 8 / launch duration of that pass; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x clock x launch duration of the SQ pass)"}
for k, o in sorted(raw.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o:
        clk = o["GRBM_GUI_ACTIVE"] / 8.0 / o["GRBM_GUI_ACTIVE__avg_ns"]
        cyc = clk * o["SQ_VALU_MFMA_BUSY_CYCLES__avg_ns"]
        out[k] = {"launches": o["SQ_VALU_MFMA_BUSY_CYCLES__launches"], "avg_launch_us_sq_pass": round(o["SQ_VALU_MFMA_BUSY_CYCLES__avg_ns"] / 1e3, 1),
                  "eff_clock_ghz": round(clk, 3), "mfma_busy_frac": round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4),
                  "insts_valu": o.get("SQ_INSTS_VALU"), "insts_mfma": o.get("SQ_INSTS_MFMA"),
                  "valu_per_mfma": round(o.get("SQ_INSTS_VALU", 0) / max(o.get("SQ_INSTS_MFMA", 1), 1), 2),
                  "wait_any_over_wave_cycles": round(o.get("SQ_WAIT_ANY", 0) / max(o.get("SQ_WAVE_CYCLES", 1), 1), 3)}
json.dump(out, open(O + "/r5_pyramid_pmc.json", "w"), indent=1)
for k, v in out.items():
    if k[0] != "_": print(k[:70], v)
PY
rm -rf $O/pmc_g1 $O/pmc_g2
