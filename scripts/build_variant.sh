#!/bin/bash
# scripts/build_variant.sh NAME "FLAGS" [file.hip ...]: libjaeger_hip_NAME.so = the shipped objects with the listed
# sources (default jg_conv_pc.hip) recompiled with FLAGS (experiment builds for interleaved A/B on one GPU box;
# select with JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_NAME.so)
set -e
name=$1; flags=$2; shift 2
files=${@:-jg_conv_pc.hip}
cd "$(dirname "$0")/../jaeger_amd/csrc"
make -j8 >/dev/null
objs=$(ls build/*.o | grep -v "_stamp.o\|_exp.o\|_v_")
for f in $files; do
  b=${f%.hip}
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -Wno-pass-failed $flags -c -o build/${b}_v_${name}.o $f
  objs=$(echo "$objs" | grep -v "build/${b}.o"); objs="$objs build/${b}_v_${name}.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libjaeger_hip_${name}.so $objs
echo built jaeger_amd/libjaeger_hip_${name}.so
