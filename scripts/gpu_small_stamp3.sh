#!/bin/bash
# per-phase cycle stamps of small_net_kernel for the experiment libraries listed in LIBS (JG_SMALL_DBG bit 16)
cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-libjaeger_hip_exph.so libjaeger_hip_expw.so}; do
  echo -n "$lib: "
  JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib JG_SMALL_DBG=16 timeout 300 python bench.py --config baseline500 --contigs 100000 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e 2>&1 | grep -E "STAMP" | sed -n "3p;8p;$p"
done
