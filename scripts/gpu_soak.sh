cd $GRAFT_REPO_ROOT
python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['exact_f32_mbps'])"
