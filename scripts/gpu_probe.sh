# (ablation masks are applied after a full-kernel warm-up step - bench.py --timed-dbg - so that the timed step reads real activations)
# VALU-in-the-MFMA-shadow probe: main loop only (JG_DBG=1), one or two workgroups per CU, N dummy FMAs per 12 MFMAs
for one in 1 0; do
  if [ $one = 1 ]; then export JG_ONE_WG=1; else unset JG_ONE_WG; fi
  for lib in libjaeger_hip.so libjaeger_hip_probe_60_0.so libjaeger_hip_probe_60_1.so libjaeger_hip_probe_120_0.so libjaeger_hip_probe_120_1.so; do
    echo -n "one_wg=$one $lib: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"
  done
done
