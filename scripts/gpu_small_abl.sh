# ablations of the fused small-window kernel (experiment build: JG_SMALL_DBG bits 1 no table phase, 2 no MFMA, 4 GELU = identity, 8 no LDS stores)
cd $GRAFT_REPO_ROOT
export JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_exp.so
for dbg in 0 1 2 4 8 3 6 7 15; do
  echo -n "JG_SMALL_DBG=$dbg: "
  JG_SMALL_DBG=$dbg python bench.py --config baseline500 --contigs 300000 --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline']['fused_small_kernel']; print(d['value'], 'Mbp/s', f['avg_launch_ms'], 'ms/launch')"
done
