# one box: where the 0.2 s between the resident step and the end-to-end forward go (DUST stream, repeat scan, nothing)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4f; exec > gpurun_out/r4f/ab.log 2>&1
export JAEGER_NO_CPROFILE=1
res() { python bench.py --no-cpu-baseline --no-e2e --no-exact-f32 --steps 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('resident', d['value'], d['ms_per_step'])"; }
run() { echo "---- $1"; env $2 python scripts/r4_e2e_prof.py brain 3 2>&1 | grep "== \|GPU worker" | grep -v "^    " | tail -4; }
res
run "default (DUST on the copy stream)" "JAEGER_DUST_STREAM=1"
run "DUST on the compute stream" "JAEGER_DUST_STREAM=0"
run "no DUST" "JAEGER_NO_DUST=1"
run "default again" "JAEGER_DUST_STREAM=1"
res
