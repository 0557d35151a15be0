cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 900 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_cli.py tests/test_gpu_legacy.py -m gpu -x -q -s -k "robust or scale_sweep or prophage or savedmodel" > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^\||passed|failed|Error|error|logits" gpurun_out/r2d/pytest.log | tail -40
