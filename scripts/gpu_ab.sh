# A/B of builds in the same session (same box, interleaved): bash scripts/gpu_ab.sh libA.so libB.so ...  (names under jaeger_amd/)
# first the parity tests on every non-default build, then 3 interleaved bench rounds
for lib in "$@"; do
  [ $lib = libjaeger_hip.so ] && continue
  echo "== parity with $lib"
  JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1
done
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "
    JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python bench.py --no-cpu-baseline --contigs 3000 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"
  done
done
