# > 4 GiB id tensor in one call (1.6 M windows x 2 988 B): logits must equal those of the same windows run in two halves
python - <<'PY'
import sys, time, numpy as np, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_model_cfg
from bench import synth_contigs
from jaeger_amd.engine import JaegerHipEngine
from jaeger_amd.fragment import build_window_table
from jaeger_amd.plan import build_plan
from jaeger_amd.weights import random_weights
rng = np.random.Generator(np.random.PCG64(3))
lengths, bases = synth_contigs(rng, 60000)
off = np.zeros(lengths.size + 1, np.int64); np.cumsum(lengths, out=off[1:])
tab = build_window_table(lengths, 1500, 1500)
starts = off[tab.contig] + tab.start
cfg = load_model_cfg("brain")
eng = JaegerHipEngine(model_cfg=cfg, weights=random_weights(build_plan(cfg), 1))
n = len(tab)
print(n, "windows,", bases.size / 1e9, "Gbp, id tensor", n * 2988 / 2**30, "GiB")
t = time.time(); full = eng.predict_windows(bases, starts, tab.length, 1500, want=("prediction", "reliability")); dt = time.time() - t
print("one call: %.1f s, %.1f Mbp/s" % (dt, n * 1500 / dt / 1e6))
h = (n // 2048) * 1024                          # split on a chunk boundary: same chunking, so bit-identical
a = eng.predict_windows(bases, starts[:h], tab.length[:h], 1500, want=("prediction", "reliability"))
b = eng.predict_windows(bases, starts[h:], tab.length[h:], 1500, want=("prediction", "reliability"))
for k in ("prediction", "reliability", "counts"):
    two = np.concatenate([a[k], b[k]])
    print(k, "identical:", bool(np.array_equal(full[k], two)), "finite:", bool(np.isfinite(full[k].astype(np.float64)).all()))
eng.close()
PY
