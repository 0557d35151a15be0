# BASELINE configs[3]: 500-bp baseline model, windows per device pass (chunk) sweep + kernel time shares
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
rng = np.random.Generator(np.random.PCG64(1))
lengths, bases = synth_contigs(rng, 3000)
off = np.zeros(lengths.size + 1, np.int64); np.cumsum(lengths, out=off[1:])
cfg = load_model_cfg("baseline500")
tab = build_window_table(lengths, 500, 500)
starts = off[tab.contig] + tab.start
for chunk in (1024, 4096, 16384, 65536):
    eng = JaegerHipEngine(model_cfg=cfg, weights=random_weights(build_plan(cfg), 1), chunk=chunk)
    eng.predict_windows(bases, starts[:chunk], tab.length[:chunk], 500, want=("prediction",))
    t = time.time(); eng.predict_windows(bases, starts, tab.length, 500, want=("prediction",)); dt = time.time() - t
    print(f"chunk {chunk}: {len(tab)} windows {dt:.3f} s  {len(tab)*500/dt/1e6:.1f} Mbp/s [{eng.model.precision}]")
    eng.close()
PY
