#!/bin/bash
# round 6, review item 1 step 1: do the residual stack's convs run faster when a pass's live tensors stay in the 256 MiB
# Infinity Cache?  Windows per pass (--chunk) 42 = 504 tiles = ONE round of the 512-workgroup persistent grid, 64.5 MB per
# activation tensor (x, h, y: 193 MB live); 84 / 126 / 168 / 252 = 2 / 3 / 4 / 6 rounds whose tensors no longer fit;
# 2048 = the shipped pass (48 rounds, 3.1 GB per tensor).  Observables per chunk: (a) the k = 5 launches' duration from
# the kernel trace -> time per tile round (a line through the 2 - 6-round points prices the fixed cost of a launch; the
# one-round point against that line is the cache's effect), (b) whole-step Mbp/s (launch gaps included), (c) MFMA-busy
# share and effective clock from --pmc passes of their own, (d) board power sampled beside the plain runs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6mall
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --no-also"
CHUNKS="${CHUNKS:-42 84 126 168 252 2048}"
# (b) + (d): plain runs, two interleaved rounds, power sampled every 0.25 s
for round in 1 2; do
  for c in $CHUNKS; do
    ( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $O/power_c${c}_r$round.log &
    SM=$!
    python3 $R/bench.py $COMMON --chunk $c > $O/plain_c${c}_r$round.json 2> $O/plain_c${c}_r$round.err
    kill $SM; wait $SM 2>/dev/null
  done
done
# (a): kernel trace, 1 500 contigs (60 000 windows)
for c in $CHUNKS; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c$c -- python3 $R/bench.py $COMMON --no-box --contigs 1500 --chunk $c > $O/kt_c$c.json 2> $O/kt_c$c.err
  cp $O/kt_c$c/*/*kernel_stats.csv $O/kernel_stats_c$c.csv
  cp $O/kt_c$c/*/*kernel_trace.csv $O/kernel_trace_c$c.csv
done
# (c): counters, SQ group and GRBM in passes of their own
G_SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"
for c in 42 84 2048; do
  i=0
  for grp in "$G_SQ" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmc_c${c}_g$i -- python3 $R/bench.py $COMMON --no-box --contigs 600 --steps 1 --chunk $c > /dev/null 2> $O/pmc_c${c}_g$i.err
  done
done
python3 - <<PY
import csv, glob, json, collections, statistics
O = "$O"
out = {}
def k5(name): return "conv_f16x3_kernel<5" in name
for c in "$CHUNKS".split():
    rec = out.setdefault(c, {})
    for r in (1, 2):
        try:
            d = json.loads(open(f"{O}/plain_c{c}_r{r}.json").read().strip().splitlines()[-1])
            rec[f"plain_r{r}"] = {"mbps": d["value"], "ms_per_step": d["ms_per_step"], "k5_avg_launch_ms_hip_events": d["roofline"]["avg_launch_ms"],
                                  "k5_launches": d["roofline"]["launches"], "frac": d["roofline"]["frac"],
                                  "box_tflops": (d.get("box") or {}).get("mfma_loop_tflops"), "box_clock": (d.get("box") or {}).get("clock_ghz")}
        except Exception as e:
            rec[f"plain_r{r}"] = {"error": str(e)}
        pw = []
        try:
            for ln in open(f"{O}/power_c{c}_r{r}.log"):
                try:
                    j = json.loads(ln)
                except ValueError:
                    continue
                for card in j.values():
                    for k, v in card.items():
                        if "ower" in k and "(W)" in k:
                            try: pw.append(float(v))
                            except ValueError: pass
            if pw:
                pw.sort()
                rec[f"power_r{r}"] = {"samples": len(pw), "median_w": statistics.median(pw), "p90_w": pw[int(0.9 * (len(pw) - 1))], "max_w": pw[-1]}
        except OSError:
            pass
    try:
        rows = [r for r in csv.DictReader(open(f"{O}/kernel_trace_c{c}.csv")) if k5(r["Kernel_Name"])]
        by_grid = collections.defaultdict(list)
        for r in rows:
            by_grid[int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        g = max(by_grid, key=lambda k: len(by_grid[k]))
        d = sorted(by_grid[g])
        rec["k5_trace"] = {"launches_full": len(d), "grid": g, "median_us": d[len(d) // 2] / 1e3, "mean_us": sum(d) / len(d) / 1e3,
                           "p10_us": d[len(d) // 10] / 1e3, "p90_us": d[len(d) * 9 // 10] / 1e3}
    except Exception as e:
        rec["k5_trace"] = {"error": str(e)}
    pm = {}
    for g in (1, 2, 3, 4):
        d = f"{O}/pmc_c{c}_g{g}"
        cc = sorted(glob.glob(d + "/*/*counter_collection.csv")); kt = sorted(glob.glob(d + "/*/*kernel_trace.csv"))
        if not cc or not kt: continue
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[-1]))}
        rows = [r for r in csv.DictReader(open(cc[-1])) if k5(r["Kernel_Name"])]
        if not rows: continue
        gmax = collections.Counter(int(r["Grid_Size"]) for r in rows).most_common(1)[0][0]
        agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set(); ns = 0
        for r in rows:
            if int(r["Grid_Size"]) != gmax: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); ns += dur.get(r["Dispatch_Id"], 0)
        for k in agg:
            pm[k] = agg[k] / cnt[k]
            pm[k + "__avg_ns"] = ns / max(len(seen), 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and "GRBM_GUI_ACTIVE" in pm:
        clk = pm["GRBM_GUI_ACTIVE"] / 8.0 / pm["GRBM_GUI_ACTIVE__avg_ns"]
        busy = pm["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * clk * pm["SQ_VALU_MFMA_BUSY_CYCLES__avg_ns"])
        rec["pmc"] = {"eff_clock_ghz": round(clk, 3), "mfma_busy_frac": round(busy, 4), "busy_x_clock": round(busy * clk, 4),
                      "avg_us_sq_pass": pm["SQ_VALU_MFMA_BUSY_CYCLES__avg_ns"] / 1e3, "avg_us_grbm_pass": pm["GRBM_GUI_ACTIVE__avg_ns"] / 1e3,
                      "fetch_kib": pm.get("FETCH_SIZE"), "write_kib": pm.get("WRITE_SIZE"),
                      "valu_per_mfma": (pm.get("SQ_INSTS_VALU", 0) / pm["SQ_INSTS_MFMA"]) if pm.get("SQ_INSTS_MFMA") else None}
json.dump(out, open(O + "/r6_mall_step1.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/kt_c* $O/pmc_c*_g?/ $O/kernel_trace_c*.csv
