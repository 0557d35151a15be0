# full GPU suite + smoke on the current tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5s; exec > gpurun_out/r5s/run.log 2>&1
python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
