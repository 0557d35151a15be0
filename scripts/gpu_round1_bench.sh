set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r1_f16x3.json
python bench.py --precision f32 --steps 1 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_r1_f32.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_f16 -- python3 $GRAFT_REPO_ROOT/bench.py --contigs 1000 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
cp $GRAFT_REPO_ROOT/gpurun_out/prof_f16/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r1_f16x3_kernel_stats.csv
