cd $GRAFT_REPO_ROOT
for cfg in default baseline500 pyramid; do
  python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['config']['name'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('exact_f32_mbps'))"
done
