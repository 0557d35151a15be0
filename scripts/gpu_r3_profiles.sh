#!/bin/bash
# round-3 profiles: the default and baseline500 bench lines under rocprofv3 --kernel-trace --stats
# (line + kernel summary of the SAME command), then counters in passes of their own (--pmc with --kernel-trace only, the
# program directly behind `--`): HBM bytes (FETCH_SIZE / WRITE_SIZE), matrix-core busy cycles and the instruction mix
# (SQ_*), the effective clock (GRBM_GUI_ACTIVE).  Summaries: profiles/pmc_traffic.json, profiles/mfma_util.json, both
# stamped with bench.kernel_hash() so that bench.py only quotes them for the build they were collected on.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
cp $O/prof_default/*/*kernel_stats.csv $O/default_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -- python3 $R/bench.py --config baseline500 > $O/bench_baseline500.json 2> $O/bench_baseline500.err
cp $O/prof_small/*/*kernel_stats.csv $O/baseline500_kernel_stats.csv
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
for cfg in default baseline500; do
  n=1500; [ $cfg = baseline500 ] && n=200000
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "$G_SQ" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_${cfg}_g$i -- python3 $R/bench.py --config $cfg --contigs $n --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e > /dev/null 2> $O/pmc_${cfg}_g$i.err
  done
done
# the producer / consumer kernel (JG_OPT_CONV_PC = 1) through the same SQ / GRBM passes, an interleaved A/B against the
# two-workgroup kernel on this box, and the per-role cycle stamps of its experiment build (if that library was shipped)
i=4
for grp in "$G_SQ" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_default_g$i -- python3 $R/bench.py --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --conv-pc 1 > /dev/null 2> $O/pmc_default_g$i.err
done
( cd $R
  for r in 1 2 3; do for pc in 0 1; do
    timeout 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --conv-pc $pc > $O/ab_pc${pc}_$r.json 2>/dev/null
    python3 -c "import json; d=json.load(open('$O/ab_pc${pc}_$r.json')); print('conv_pc=$pc round $r:', d['value'], 'Mbp/s  k5 frac', d['roofline']['frac'], ' avg launch ms', d['roofline']['avg_launch_ms'])"
  done; done > $O/pc_ab.txt
  cat $O/pc_ab.txt
  if [ -f jaeger_amd/libjaeger_hip_stamp.so ]; then
    JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_stamp.so timeout 300 python3 bench.py --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --conv-pc 1 > /dev/null 2> $O/pc_stamp.err
    grep PCSTAMP $O/pc_stamp.err | grep "rows=12288" | sort | uniq -c | sort -rn | head -0
    grep PCSTAMP $O/pc_stamp.err | grep "rows=12288" | tail -6 > $O/pc_stamps.txt; cat $O/pc_stamps.txt
  fi )
HASH=$(cd $R && python3 -c "import bench; print(bench.kernel_hash())")
python3 - <<PY
import csv, glob, collections, json
O = "$O"
def kname(name):
    if "conv_pc_kernel" in name: return "conv_pc_kernel"
    if "conv_f16x3_kernel<5" in name: return "conv_f16x3_kernel"
    if "small_net_kernel" in name: return "small_net_kernel"
    return None
raw = {}
for cfg in ("default", "baseline500"):
    for g in (1, 2, 3, 4, 5, 6):
        d = "%s/pmc_%s_g%d" % (O, cfg, g)
        cc = sorted(glob.glob(d + "/*/*counter_collection.csv")); kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"))
        if not cc: continue
        dur = {}
        if kt:
            for row in csv.DictReader(open(kt[-1])):
                dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        # the big launches only: the largest grid of each kernel
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
            o = raw.setdefault(k, {})
            for c in agg[k]:
                o[c] = agg[k][c] / cnt[k][c]
                o[c + "__launches"] = cnt[k][c]
                o[c + "__avg_ns"] = durs[k] / max(len(seen[k]), 1)
json.dump(raw, open(O + "/pmc_raw.json", "w"), indent=1)
traffic = {"_comment": "HBM traffic of the dominant kernels from rocprofv3 PMC passes (scripts/gpu_r3_profiles.sh): FETCH_SIZE and WRITE_SIZE in separate --pmc runs of bench.py --contigs 1500 --steps 1 --warmup 1 (baseline500: 200000), mean over the largest-grid launches of each kernel. rocprofv3 reports KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half the bytes of wide (16 B/lane) streaming reads -> doubled for the conv kernels, left as is for the byte-wide id loads of small_net_kernel; WRITE_SIZE is exact.", "kernel_hash": "$HASH"}
util = {"_comment": "matrix-core utilisation from rocprofv3 PMC passes (scripts/gpu_r3_profiles.sh): SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's 1024 SIMDs; = 32 x v_mfma_f32_32x32x16 instructions) and GRBM_GUI_ACTIVE (summed over the 8 XCDs) in passes of their own; eff_clock_ghz = GRBM_GUI_ACTIVE / 8 / launch duration of that pass; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x eff_clock x launch duration of the SQ pass)", "kernel_hash": "$HASH"}
for k, o in raw.items():
    if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
        mult = 1.0 if k == "small_net_kernel" else 2.0
        traffic[k] = {"precision": "f16x3", "fsize": 500 if k == "small_net_kernel" else 1500,
                      "chunk": 0 if k == "small_net_kernel" else 2048,
                      "fetch_size_kib_mean": o["FETCH_SIZE"], "write_size_kib_mean": o["WRITE_SIZE"],
                      "launches": o["FETCH_SIZE__launches"],
                      "traffic_bytes_per_launch": int((mult * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024)}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "GRBM_GUI_ACTIVE" in o:
        clk = o["GRBM_GUI_ACTIVE"] / 8.0 / o["GRBM_GUI_ACTIVE__avg_ns"]          # cycles per ns = GHz
        cyc = clk * o["SQ_VALU_MFMA_BUSY_CYCLES__avg_ns"]
        util[k] = {"launches": o["SQ_VALU_MFMA_BUSY_CYCLES__launches"], "avg_launch_ms_sq_pass": o["SQ_VALU_MFMA_BUSY_CYCLES__avg_ns"] / 1e6,
                   "avg_launch_ms_grbm_pass": o["GRBM_GUI_ACTIVE__avg_ns"] / 1e6, "eff_clock_ghz": round(clk, 3),
                   "mfma_busy_cycles": o["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_busy_frac": round(o["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4),
                   "wave_cycles_quad": o.get("SQ_WAVE_CYCLES"), "wait_any_quad": o.get("SQ_WAIT_ANY"), "wait_inst_any_quad": o.get("SQ_WAIT_INST_ANY"),
                   "active_inst_any_quad": o.get("SQ_ACTIVE_INST_ANY"), "insts_valu": o.get("SQ_INSTS_VALU"), "insts_mfma": o.get("SQ_INSTS_MFMA"),
                   "sq_busy_cycles": o.get("SQ_BUSY_CYCLES")}
json.dump(traffic, open(O + "/pmc_traffic.json", "w"), indent=1)
json.dump(util, open(O + "/mfma_util.json", "w"), indent=1)
print(json.dumps(util, indent=1)); print(json.dumps({k: v for k, v in traffic.items() if k[0] != "_"}, indent=1))
PY
tail -c 1500 $O/bench_default.json; echo
head -8 $O/default_kernel_stats.csv | cut -c1-170
rm -rf $O/prof_default $O/prof_small $O/pmc_*_g?/
