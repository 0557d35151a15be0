# end-to-end CLI runs with DUST on the GPU (default) vs --dust-host, brain 1 500 bp and baseline500 500 bp; DUST kernel rate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/e2e
exec > >(tee gpurun_out/e2e/e2e2.log) 2>&1
python - <<'PY'
import numpy as np, sys, time
sys.path.insert(0, '.')
from bench import synth_contigs
rng = np.random.Generator(np.random.PCG64(20260923))
lengths, bases = synth_contigs(rng, 10000)
with open('/tmp/synth10k.fasta', 'wb') as fh:
    off = 0
    for i, l in enumerate(lengths):
        fh.write(b'>contig_%d len=%d\n' % (i, l))
        s = bases[off:off + l].tobytes(); off += l
        fh.write(b'\n'.join(s[j:j + 80] for j in range(0, l, 80)) + b'\n')
print('wrote', off / 1e6, 'Mbp')
from pathlib import Path
sys.path.insert(0, 'tests')
from conftest import make_model_dir
make_model_dir(Path('/tmp/model_brain'))
make_model_dir(Path('/tmp/model_b500'), name="baseline500", model_name="jaeger_500bp_baseline")
# DUST kernel rate on the device-resident buffer
from jaeger_amd.engine import HipDevice
from jaeger_amd import fragment as frag
d = HipDevice(0)
offs = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs[1:])
p = d.upload(bases)
d.dust_mask(p, bases.size, offs)
t = time.time(); n = d.dust_mask(p, bases.size, offs); dt = time.time() - t
print(f"jg_dust_mask_device: {bases.size / 1e6:.1f} Mbp in {dt * 1e3:.1f} ms = {bases.size / dt / 1e9:.2f} Gbp/s ({n} masked)")
d.free(p)
fa = frag.FastaBatch([""] * len(lengths), bases.copy(), offs)
t = time.time(); n = frag.dust_mask(fa); dt = time.time() - t
print(f"jg_dust_mask (host, all quota cores): {dt * 1e3:.1f} ms = {bases.size / dt / 1e9:.2f} Gbp/s ({n} masked)")
d.close()
PY
run() { local t0=$(date +%s%N); python -m jaeger_amd predict -i /tmp/synth10k.fasta -o /tmp/out_$1 --model_path $2 --fsize $3 --stride $3 -f $4 2>&1 | grep -E "processed|rror|DUST|GPU worker|Traceback"; echo "process wall: $(( ($(date +%s%N) - t0) / 1000000 )) ms"; }
for rep in 1 2; do
echo "== brain 1500, --dust-host"; run b /tmp/model_brain 1500 "--dust-host"
echo "== brain 1500, DUST on the GPU"; run a /tmp/model_brain 1500 ""
done
cmp /tmp/out_a/*/synth10k.tsv /tmp/out_b/*/synth10k.tsv && echo "TSVs identical"
for rep in 1 2; do
echo "== baseline500 500, --dust-host"; run d /tmp/model_b500 500 "--dust-host"
echo "== baseline500 500, DUST on the GPU"; run c /tmp/model_b500 500 ""
echo "== baseline500 500, --no-dustmask"; run e /tmp/model_b500 500 "--no-dustmask"
done
cmp /tmp/out_c/*/synth10k.tsv /tmp/out_d/*/synth10k.tsv && echo "TSVs identical"
