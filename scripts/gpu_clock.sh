# effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration.
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
# One warm-up step with the full kernel (real activations in the buffers), then one timed step under the ablation
# mask: the first half of each kernel's launches is the full kernel, the second half the ablated one.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 1 2; do
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_$dbg -- python3 $R/bench.py --contigs 1000 --steps 1 --warmup 1 --timed-dbg $dbg --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
d="$R/gpurun_out/clk_$dbg"
f=glob.glob(d+"/*/*counter_collection.csv")[0]
rows=collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if row["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
    k=row["Kernel_Name"].split("(")[0][-40:]
    dur=float(row["End_Timestamp"])-float(row["Start_Timestamp"])
    rows[k].append((float(row["Start_Timestamp"]),float(row["Counter_Value"]),dur))
for k,v in rows.items():
    v.sort()
    if sum(x[2] for x in v)/len(v)<2e5: continue
    h=len(v)//2
    for name,part in (("full kernel",v[:h]),("timed JG_DBG=$dbg",v[h:])):
        c=sum(x[1] for x in part); t=sum(x[2] for x in part)
        print(k,name,len(part),"launches  clock GHz=%.3f"%(c/8/t),"avg ms=%.3f"%(t/len(part)/1e6))
PY
done
