# VALU-in-the-MFMA-shadow probe: main loop only (timed JG_DBG=1 after a full-kernel warm-up on real data), one or two
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
# workgroups per CU, 60 dummy FMAs per 12 MFMAs.  The probe sits in the one-barrier-per-tap loop: both libraries are
# built with -DJG_PAIRED=0 (libjaeger_hip_np.so; libjaeger_hip_probe_60_0.so adds -DJG_VALU_PROBE=60).
for one in 1 0; do
  if [ $one = 1 ]; then export JG_ONE_WG=1; else unset JG_ONE_WG; fi
  for lib in libjaeger_hip_np.so libjaeger_hip_probe_60_0.so; do
    for dbg in 1 0; do
      echo -n "one_wg=$one $lib timed JG_DBG=$dbg: "
      JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg $dbg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"
    done
  done
done
