cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"
grep -vi "warning\|disabled\|eng = \|^$" gpurun_out/r2e/pytest.log | tail -40
