#!/bin/bash
# baseline500: program-specialised small-window kernel vs its generic form (bench.py --small-generic), interleaved
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for f in "" "--small-generic"; do
    echo -n "small_net_kernel ${f:-program-specialised}: "
    timeout 300 python bench.py --config baseline500 --no-cpu-baseline --no-exact-f32 --no-e2e $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('avg_launch_ms'))"
  done
done
