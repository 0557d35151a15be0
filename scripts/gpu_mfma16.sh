# Build the probe library first (not part of `make`):
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
#   make -C jaeger_amd/csrc -j6 BUILD=build_m16 OUT=$PWD/jaeger_amd/libjaeger_hip_m16.so \
#     "FLAGS=-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I$PWD/include -I. -Wall -Wno-pass-failed -DJG_MFMA16_PROBE"
# MFMA-shape probe (timing only): libjaeger_hip_m16.so (-DJG_MFMA16_PROBE) replaces every 32x32x16 MFMA of the main
# loop by two 16x16x32 (same operand registers, same LDS reads, same MACs; results are garbage of the same
# statistics).  Warm-up with the full kernel so that the activation buffers hold random-looking data (the chip's clock
# under MFMA load depends on the operand data), then the timed step with the ablation mask (1 = main loop only).
one() { python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3; do
  for dbg in 1 0; do
    echo -n "32x32x16 timed JG_DBG=$dbg: "; one $dbg
    echo -n "16x16x32 timed JG_DBG=$dbg: "; JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/libjaeger_hip_m16.so one $dbg
  done
done
# the ablation table on real data (regular library)
for dbg in 0 1 64 32 128 2 3 65; do echo -n "regular timed JG_DBG=$dbg: "; one $dbg; done
