python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
python bench.py --no-cpu-baseline 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q6 -- python3 $GRAFT_REPO_ROOT/bench.py --contigs 1000 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
head -8 $GRAFT_REPO_ROOT/gpurun_out/prof_q6/*/*kernel_stats.csv | cut -c1-150
