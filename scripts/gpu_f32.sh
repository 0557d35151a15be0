python -m pytest tests/test_gpu_parity.py tests/test_gpu_legacy.py -x -q 2>&1 | tail -2
python bench.py --precision f32 --steps 1 --warmup 1 --contigs 2000 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('brain f32:', d['value'], 'Mbp/s', d['roofline']['achieved'], 'TF')"
bash scripts/gpu_configs.sh 2>&1 | grep -E "baseline500|legacy default" 
