for lib in libjaeger_hip_prev.so libjaeger_hip.so; do
  echo "== $lib"
  JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python -m pytest tests/test_gpu_parity.py -q -k repeatable 2>&1 | grep -E "repeats differ|passed|failed" | tail -6
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
