# cost of the per-launch HIP events that bench.py's roofline numbers come from
for rep in 1 2 3; do
  for f in "" "--no-profile"; do
    echo -n "bench $f: "
    python bench.py --no-cpu-baseline --contigs 3000 --steps 2 --warmup 1 $f 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
