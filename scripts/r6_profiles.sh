#!/bin/bash
# round-6 profiles.  (1) counters of the dominant kernels in passes of their own (--pmc with --kernel-trace only, the program
# directly behind `--`): HBM bytes (FETCH_SIZE / WRITE_SIZE), matrix-core busy cycles and the instruction mix (SQ_*), the
# effective clock (GRBM_GUI_ACTIVE) -> pmc_traffic.json / mfma_util.json stamped with bench.kernel_hash(), copied into
# profiles/ of THIS snapshot; (2) the default and baseline500 bench lines under rocprofv3 --kernel-trace --stats (line +
# kernel summary of the SAME command), which then quote the counters of the build they time; (3) kernel statistics of an
# end-to-end run of both families (DUST, repeat scan and network side by side).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
for cfg in default baseline500; do
  n=1500; [ $cfg = baseline500 ] && n=200000
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "$G_SQ" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_${cfg}_g$i -- python3 $R/bench.py --config $cfg --contigs $n --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --no-box > /dev/null 2> $O/pmc_${cfg}_g$i.err
  done
done
HASH=$(cd $R && python3 -c "import bench; print(bench.kernel_hash())")
python3 - <<PY
import csv, glob, collections, json
O = "$O"
def kname(name):
    if "conv_f16x3_kernel<5" in name: return "conv_f16x3_kernel"
    if "small_net_kernel" in name: return "small_net_kernel"
    return None
raw = {}
for cfg in ("default", "baseline500"):
    for g in (1, 2, 3, 4):
        d = "%s/pmc_%s_g%d" % (O, cfg, g)
        cc = sorted(glob.glob(d + "/*/*counter_collection.csv")); kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"))
        if not cc: continue
        dur = {}
        if kt:
            for row in csv.DictReader(open(kt[-1])):
                dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        rows = [r for r in csv.DictReader(open(cc[-1])) if kname(r["Kernel_Name"])]
        gmax = collections.defaultdict(int)                      # the full-size launches only: the largest grid of each kernel
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
json.dump(raw, open(O + "/r6_pmc_raw.json", "w"), indent=1)
traffic = {"_comment": "HBM traffic of the dominant kernels from rocprofv3 PMC passes (scripts/r6_profiles.sh): FETCH_SIZE and WRITE_SIZE in separate --pmc runs of bench.py --contigs 1500 --steps 1 --warmup 1 (baseline500: 200000), mean over the largest-grid (full-size) launches of each kernel. rocprofv3 reports KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half the bytes of wide (16 B/lane) streaming reads -> doubled for the conv kernels, left as is for the byte-wide id loads of small_net_kernel; WRITE_SIZE is exact.", "kernel_hash": "$HASH"}
util = {"_comment": "matrix-core utilisation from rocprofv3 PMC passes (scripts/r6_profiles.sh): SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's 1024 SIMDs; = 32 x v_mfma_f32_32x32x16 instructions) and GRBM_GUI_ACTIVE (summed over the 8 XCDs) in passes of their own; eff_clock_ghz = GRBM_GUI_ACTIVE / 8 / launch duration of that pass; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x eff_clock x launch duration of the SQ pass)", "kernel_hash": "$HASH"}
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
cp $O/pmc_traffic.json $O/mfma_util.json $R/profiles/           # (this snapshot's copy: the lines below quote them)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py > $O/r6_default_bench.json 2> $O/bench_default.err
cp $O/prof_default/*/*kernel_stats.csv $O/r6_default_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -- python3 $R/bench.py --config baseline500 --no-also > $O/r6_baseline500_bench.json 2> $O/bench_baseline500.err
cp $O/prof_small/*/*kernel_stats.csv $O/r6_baseline500_bench_kernel_stats.csv
# kernel statistics of an end-to-end run (10 000-contig FASTA) of both families
JAEGER_NO_CPROFILE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e_brain -- python3 $R/scripts/r4_e2e_prof.py brain 2 > $O/e2e_brain.log 2>&1
cp $O/prof_e2e_brain/*/*kernel_stats.csv $O/r6_e2e_default_kernel_stats.csv
JAEGER_NO_CPROFILE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e_b500 -- python3 $R/scripts/r4_e2e_prof.py baseline500 3 > $O/e2e_b500.log 2>&1
cp $O/prof_e2e_b500/*/*kernel_stats.csv $O/r6_e2e_baseline500_kernel_stats.csv
python3 -c "
import json
for f in ('r6_default_bench.json','r6_baseline500_bench.json'):
    d=json.loads(open('$O/'+f).read().strip().splitlines()[-1]); r=d['roofline']; e=d.get('e2e') or {}
    print(f, d['value'], r.get('frac'), (r.get('pmc_other_device') or {}).get('mfma_busy_frac'), (r.get('pmc_other_device') or {}).get('eff_clock_ghz'), r.get('traffic'), r.get('pmc_stale'), d.get('exact_f32_mbps'), e.get('mbps'), e.get('stages'))
"
grep '== ' $O/e2e_brain.log $O/e2e_b500.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pyr -- python3 $R/bench.py --config pyramid --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-exact-f32 --no-also > $O/r6_pyramid_bench.json 2>/dev/null
cp $O/prof_pyr/*/*kernel_stats.csv $O/r6_pyramid_bench_kernel_stats.csv
cd $R
python3 bench.py > $O/r6_default_bench_plain.json 2> $O/plain_default.err
python3 bench.py --config baseline500 --no-also > $O/r6_baseline500_bench_plain.json 2> $O/plain_b500.err
python3 bench.py --config pyramid --no-also --no-e2e --no-cpu-baseline --steps 3 > $O/r6_pyramid_bench_plain.json 2> $O/plain_pyr.err
python3 -c "
import json
for f in ('r6_default_bench_plain.json','r6_baseline500_bench_plain.json','r6_pyramid_bench_plain.json','r6_pyramid_bench.json'):
    d=json.loads(open('$O/'+f).read().strip().splitlines()[-1]); r=d['roofline']; e=d.get('e2e') or {}
    print(f, d['value'], r.get('frac'), r.get('avg_launch_ms'), r.get('traffic'), r.get('pmc_stale'), d.get('exact_f32_mbps'), e.get('mbps') if isinstance(e, dict) else e, (d.get('box') or {}).get('mfma_loop_tflops'))
    for k, v in (d.get('also') or {}).items():
        print('   also', k, v.get('value'), (v.get('roofline') or {}).get('frac'), (v.get('e2e') or {}).get('mbps') if isinstance(v.get('e2e'), dict) else v.get('e2e'), (v.get('e2e_records') or {}).get('mbps'))
"
rm -rf $O/prof_pyr $O/prof_default $O/prof_small $O/prof_e2e_brain $O/prof_e2e_b500 $O/pmc_*_g?/
