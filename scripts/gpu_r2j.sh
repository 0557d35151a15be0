cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "sharded" 2>&1 | grep -vi "warn\|disabled\|eng = " | tail -25
