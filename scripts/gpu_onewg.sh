# one workgroup per CU (JG_ONE_WG=1) vs two, on real data (warm-up with the full kernel, ablation mask on the timed step)
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
one() { python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
  for dbg in 0 1 2; do
    echo -n "two WG/CU JG_DBG=$dbg: "; one $dbg
    echo -n "one WG/CU JG_DBG=$dbg: "; JG_ONE_WG=1 one $dbg
  done
done
