#!/bin/bash
# per-role cycle stamps of experiment builds: STAMPS="name ..." (jaeger_amd/libjaeger_hip_<name>.so)
mkdir -p gpurun_out
for v in $STAMPS; do
JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_$v.so timeout 300 python bench.py --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e --conv-pc 1 > gpurun_out/st_$v.json 2> gpurun_out/st_$v.err
echo $v; grep PCSTAMP gpurun_out/st_$v.err | grep "rows=12288" | tail -3
done
