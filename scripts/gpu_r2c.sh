cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "baseline500 or small_window or default_precision" > gpurun_out/r2c/pytest_small.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2c/pytest_small.log
python bench.py --config baseline500 --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'Mbp/s', d['roofline']['fused_small_kernel'])"
bash scripts/gpu_small_stamp.sh 2>&1 | grep STAMP | tail -1
