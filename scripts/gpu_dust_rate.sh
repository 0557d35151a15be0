cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, sys, time
sys.path.insert(0, '.')
from bench import synth_contigs
rng = np.random.Generator(np.random.PCG64(20260923))
lengths, bases = synth_contigs(rng, 10000)
from jaeger_amd.engine import HipDevice
from jaeger_amd import fragment as frag
d = HipDevice(0)
offs = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs[1:])
p = d.upload(bases)
d.dust_mask(p, bases.size, offs)
for w in (64, 64, 34, 12, 4):
    t = time.time(); n = d.dust_mask(p, bases.size, offs, window=w); dt = time.time() - t
    print(f"jg_dust_mask_device window {w}: {bases.size / 1e6:.1f} Mbp in {dt * 1e3:.1f} ms = {bases.size / dt / 1e9:.2f} Gbp/s ({n} masked)")
d.free(p); d.close()
PY
timeout 600 python -m pytest tests/test_gpu_dust.py -m gpu -x -q 2>&1 | tail -2
