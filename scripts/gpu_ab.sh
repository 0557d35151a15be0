# A/B of two builds in the same session (same box, interleaved): libjaeger_hip_A.so vs libjaeger_hip_B.so
JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_B.so python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  for v in A B; do
    echo -n "$v: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_$v.so python bench.py --no-cpu-baseline --contigs 3000 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
  done
done
