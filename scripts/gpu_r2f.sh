cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "nmdmerge or default_precision or baseline500 or small" > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|nmdmerge500 500|Error" gpurun_out/r2f/pytest.log | tail
python bench.py --config baseline500 --no-cpu-baseline 2>/dev/null | cut -c1-1500
