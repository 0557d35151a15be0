cd $GRAFT_REPO_ROOT
for c in 2048 4096 3072 2048 4096 3072; do
  python bench.py --chunk $c --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('chunk', $c, d['value'], d['roofline']['frac'], d['roofline']['launches'])"
done
