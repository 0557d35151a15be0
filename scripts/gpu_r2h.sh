cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/r2h/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
timeout 600 python bench.py > gpurun_out/r2h/bench_default.json 2>/dev/null; cut -c1-300 gpurun_out/r2h/bench_default.json
timeout 600 python bench.py --config baseline500 > gpurun_out/r2h/bench_baseline500.json 2>/dev/null; cut -c1-200 gpurun_out/r2h/bench_baseline500.json
timeout 600 python bench.py --config frag1m --no-cpu-baseline > gpurun_out/r2h/bench_frag1m.json 2>/dev/null; cut -c1-200 gpurun_out/r2h/bench_frag1m.json
