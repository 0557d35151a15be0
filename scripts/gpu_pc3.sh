#!/bin/bash
mkdir -p gpurun_out
JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_stamp.so timeout 300 python bench.py --contigs 1500 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 --conv-pc 1 > gpurun_out/pc3_stamp.json 2> gpurun_out/pc3_stamp.err
python -c "
import json; d=json.load(open('gpurun_out/pc3_stamp.json')); print('stamp build:', d['value'], d['roofline']['avg_launch_ms'])"
grep PCSTAMP gpurun_out/pc3_stamp.err | grep "rows=12288" | tail -12
