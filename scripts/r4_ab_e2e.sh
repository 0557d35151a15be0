# one box: resident bench figure, then end-to-end runs under different span budgets (the host-buffer pipeline)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4c; exec > gpurun_out/r4c/ab.log 2>&1
export JAEGER_NO_CPROFILE=1
python bench.py --no-cpu-baseline --no-e2e --no-exact-f32 --steps 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', d['value'], d['ms_per_step'])"
for sb in 33554432 134217728 1073741824 8388608; do
  echo "---- stream bytes $sb"
  JAEGER_STREAM_BYTES=$sb python scripts/r4_e2e_prof.py brain 3 2>&1 | grep "== \|GPU worker" | grep -v "^    "
done
echo "---- no pipeline"
JAEGER_NO_PIPELINE=1 python scripts/r4_e2e_prof.py brain 2 2>&1 | grep "== \|GPU worker" | grep -v "^    "
python bench.py --no-cpu-baseline --no-e2e --no-exact-f32 --steps 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', d['value'], d['ms_per_step'])"
python -m pytest tests/test_gpu_reference_kats.py tests/test_gpu_configs.py tests/test_gpu_cli.py tests/test_gpu_dust.py -m gpu -q 2>&1 | tail -8
