cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "random_window" 2>&1 | grep -vi "warn\|disabled\|eng = \|^$" | tail -25
