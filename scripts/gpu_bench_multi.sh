cd $GRAFT_REPO_ROOT
for cfg in baseline500 pyramid frag1m; do
  echo "== $cfg"
  python bench.py --gpus 2 --oversubscribe --config $cfg --contigs 3000 --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | tail -4 | cut -c1-600
done
