python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for d in 0 1 2 3; do JG_DBG=$d python bench.py --contigs 500 --steps 1 --warmup 1 --chunk 256 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg $d', d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"; done
