cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_cli.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|^E  " | tail
