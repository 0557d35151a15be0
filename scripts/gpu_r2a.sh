# round 2, first GPU pass: full GPU suite, then the default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/r2a/pytest.log
timeout 600 python bench.py > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
cat gpurun_out/r2a/bench.json
