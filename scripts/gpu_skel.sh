# (ablation masks are applied after a full-kernel warm-up step - bench.py --timed-dbg - so that the timed step reads real activations)
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
# skeleton ablations of the split-f16 conv (JG_DBG bits: 1 no epilogue, 2 no LDS reads + MFMA, 8 no X DMA, 16 no W DMA, 64 no stores)
for dbg in 0 2 18 10 26 3 27 1 66; do
  echo -n "JG_DBG=$dbg: "
  python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg $dbg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"
done
