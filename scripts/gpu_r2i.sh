cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nmdmerge or default_precision or baseline500 or small" 2>&1 | grep -E "passed|failed" | tail -2
bash scripts/gpu_ab_small.sh 2>&1 | tail -4
