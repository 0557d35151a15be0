# full GPU suite (with the slowest tests listed) + smoke on the current tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5s; exec > gpurun_out/r5s/run.log 2>&1
python -m pytest tests -m gpu -q -x --durations=25 2>&1 | grep -v "Warning\|^  \|^$\|warnings.warn" | tail -45
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
