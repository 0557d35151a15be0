cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_perf_guards.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|^E  |Mbp/s" | tail
