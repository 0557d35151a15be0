# (ablation masks are applied after a full-kernel warm-up step - bench.py --timed-dbg - so that the timed step reads real activations)
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
# what a one-workgroup-per-CU design could reach: lone workgroup with the epilogue parts removed
for one in 0 1; do for dbg in 0 64 96 224 1; do
  echo -n "one_wg=$one JG_DBG=$dbg: "
  if [ $one = 1 ]; then export JG_ONE_WG=1; else unset JG_ONE_WG; fi
  python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg $dbg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"
done; done
