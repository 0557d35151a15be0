# effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 0 2; do
JG_DBG=$dbg rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_$dbg -- python3 $R/bench.py --contigs 1000 --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
d="$R/gpurun_out/clk_$dbg"
f=glob.glob(d+"/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda:[0.0,0.0,0])
for row in csv.DictReader(open(f)):
    if row["Counter_Name"]!="GRBM_GUI_ACTIVE": continue
    k=row["Kernel_Name"].split("(")[0][-40:]
    dur=float(row["End_Timestamp"])-float(row["Start_Timestamp"])
    if dur<2e5: continue
    a=agg[k]; a[0]+=float(row["Counter_Value"]); a[1]+=dur; a[2]+=1
for k,(c,t,n) in agg.items(): print("dbg=$dbg",k,n,"launches  clock GHz=%.3f"%(c/8/t), "avg ms=%.3f"%(t/n/1e6))
PY
done
