cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --durations=25 2>&1 | grep -E "passed|failed|s call|s setup" | head -40
