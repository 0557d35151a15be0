python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
python bench.py --no-cpu-baseline --contigs 3000 --steps 1 2>&1 | tail -1 | cut -c1-400
JG_NO_LUT=1 python bench.py --no-cpu-baseline --contigs 3000 --steps 1 2>&1 | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q7 -- python3 $GRAFT_REPO_ROOT/bench.py --contigs 1000 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-100
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_q7/*/*kernel_stats.csv | cut -c1-150
