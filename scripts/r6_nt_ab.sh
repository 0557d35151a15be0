#!/bin/bash
# round 6, item 1 step 1b: is it the non-temporal hint on the activation streams that keeps a 42-window pass's tensors out
# of the Infinity Cache?  Experiment builds with (libjaeger_hip_exp.so) and without (libjaeger_hip_exp_nont.so,
# -DJG_EXP_NO_NT) the hint, windows per pass 42 / 84 / 2048, interleaved, k = 5 launch durations from the kernel trace.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6nt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --no-also --no-box --contigs 1500"
for round in 1 2; do
  for lib in exp exp_nont; do
    for c in 42 84 2048; do
      JAEGER_HIP_LIB=$R/jaeger_amd/libjaeger_hip_$lib.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt_${lib}_c${c}_r$round -- python3 $R/bench.py $COMMON --chunk $c > $O/${lib}_c${c}_r$round.json 2> $O/${lib}_c${c}_r$round.err
    done
  done
done
python3 - <<PY
import csv, glob, json
O = "$O"
out = {}
for lib in ("exp", "exp_nont"):
    for c in (42, 84, 2048):
        for r in (1, 2):
            kt = sorted(glob.glob(f"{O}/kt_{lib}_c{c}_r{r}/*/*kernel_trace.csv"))
            if not kt: continue
            d = sorted(int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in csv.DictReader(open(kt[-1])) if "conv_f16x3_kernel<5" in x["Kernel_Name"])
            line = json.loads(open(f"{O}/{lib}_c{c}_r{r}.json").read().strip().splitlines()[-1])
            out[f"{lib}_c{c}_r{r}"] = {"k5_launches": len(d), "median_us": d[len(d) // 2] / 1e3, "mean_us": sum(d) / len(d) / 1e3, "mbps": line["value"]}
json.dump(out, open(O + "/r6_nt_ab.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/kt_*
