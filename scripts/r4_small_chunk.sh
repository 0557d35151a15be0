#!/bin/bash
# baseline500 resident step against windows per pass (default 6 144 at 165 codons): the per-pass tail launches
# (small_pool_final, dense heads) are 2.9 % of a step; fewer, larger passes shrink that share.  One box, interleaved.
mkdir -p gpurun_out/r4chunk
for rep in 1 2; do
  for c in 0 12288 24576 49152 65536; do
    python bench.py --config baseline500 --chunk $c --steps 3 --warmup 1 --no-cpu-baseline --no-exact-f32 --no-e2e \
      2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chunk $c', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"
  done
done | tee gpurun_out/r4chunk/run.log
