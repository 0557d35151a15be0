cd $GRAFT_REPO_ROOT
for c in 0 12288 24576 0 12288 24576; do
  python bench.py --config baseline500 --chunk $c --no-cpu-baseline --no-exact-f32 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('chunk', $c, d['value'], d['ms_per_step'], d['roofline']['launches'], d['roofline']['avg_launch_ms'])"
done
