cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E  |Error" | tail -30 | tee gpurun_out/full/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee gpurun_out/full/smoke.log
