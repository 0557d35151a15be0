cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline500 or small_window or default_precision or nmdmerge" -s > gpurun_out/r2b/pytest_small.log 2>&1; echo "pytest rc=$?"
grep -v Warning gpurun_out/r2b/pytest_small.log | tail -40
timeout 600 python bench.py --config baseline500 --steps 3 --warmup 1 > gpurun_out/r2b/bench_small.json 2> gpurun_out/r2b/bench_small.err; echo "bench rc=$?"; tail -3 gpurun_out/r2b/bench_small.err
cat gpurun_out/r2b/bench_small.json
timeout 600 python bench.py --config baseline500 --steps 2 --warmup 1 --precision f32 --no-cpu-baseline > gpurun_out/r2b/bench_small_f32.json 2>/dev/null; cat gpurun_out/r2b/bench_small_f32.json | cut -c1-600
