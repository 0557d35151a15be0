# conv v3 prototype (main loop only, JG_DBG bit 256) against the shipped kernel's main loop (JG_DBG=1), real activations
one() { python bench.py --no-cpu-baseline --contigs 1500 --steps 1 --warmup 1 --timed-dbg $1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('avg conv launch ms', d['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3; do
  echo -n "shipped kernel, main loop only: "; one 1
  echo -n "v3 prototype,   main loop only: "; one 257
done
