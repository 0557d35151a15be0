cd $GRAFT_REPO_ROOT
python bench.py --gpus 8 --oversubscribe --contigs 400 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['config']['windows_per_gpu'], d['config']['parallelism'], d['roofline']['launches'])"
