set -x
python __graft_entry__.py smoke 2>&1 | tail -8
for c in 32 64 128 256 1024; do python bench.py --contigs 1000 --steps 1 --warmup 1 --chunk $c --no-cpu-baseline 2>&1 | tail -1; done
python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r1_first.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --contigs 300 --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | tail -2
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof1 | head
