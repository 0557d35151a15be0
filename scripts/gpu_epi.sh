# (ablation masks are applied after a full-kernel warm-up step - bench.py --timed-dbg - so that the timed step reads real activations)
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
# epilogue ablations with the matrix cores running (JG_DBG bits: 1 no epilogue at all, 32 no stage arithmetic, 64 no stores, 128 no shortcut loads)
for dbg in 0 1 32 64 128 96 192 224; do
  echo -n "JG_DBG=$dbg: "
  python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg $dbg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"
done
cd /tmp && export TMPDIR=/tmp
for dbg in 0 64; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/epi_$dbg -- python3 $GRAFT_REPO_ROOT/bench.py --contigs 1000 --steps 1 --warmup 1 --no-cpu-baseline --timed-dbg $dbg > /dev/null 2>&1
echo "dbg=$dbg"; head -5 $GRAFT_REPO_ROOT/gpurun_out/epi_$dbg/*/*kernel_stats.csv | cut -c1-120
done
