for ch in ${CHUNKS:-1024 512 256 170 128 85 64}; do
  echo -n "chunk $ch: "
  python bench.py --no-cpu-baseline --contigs 2000 --steps 1 --warmup 1 --chunk $ch 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done
