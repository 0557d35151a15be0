cd $GRAFT_REPO_ROOT
python - <<'PY'
import numpy as np, sys, time
sys.path.insert(0, '.')
from bench import synth_contigs
from jaeger_amd.engine import HipDevice
from jaeger_amd import fragment as frag
from jaeger_amd.termini import terminal_repeat_table
rng = np.random.Generator(np.random.PCG64(20260923))
lengths, bases = synth_contigs(rng, 10000)
offs = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs[1:])
fa = frag.FastaBatch([f"c{i}" for i in range(len(lengths))], bases, offs)
d = HipDevice(0)
terminal_repeat_table(d, fa, 1500)
for _ in range(3):
    t = time.time(); tab = terminal_repeat_table(d, fa, 1500); dt = time.time() - t
    print(f"terminal repeats of {len(fa)} contigs: {dt * 1e3:.0f} ms  (checksum {int(np.asarray(tab, np.int64).sum())})")
d.close()
PY
timeout 600 python -m pytest tests/test_gpu_termini.py -m gpu -q 2>&1 | tail -2
