# throughput of the other model families / window sizes (host buffers in, logits out), for DESIGN.md
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
from jaeger_amd import legacy
rng = np.random.Generator(np.random.PCG64(1))
lengths, bases = synth_contigs(rng, 3000)
off = np.zeros(lengths.size + 1, np.int64); np.cumsum(lengths, out=off[1:])
def run(name, eng, fsize, stride, pw):
    tab = build_window_table(lengths, fsize, stride)
    starts = off[tab.contig] + tab.start
    pw(bases, starts[:3000], tab.length[:3000], fsize)
    t = time.time(); pw(bases, starts, tab.length, fsize); dt = time.time() - t
    uniq = float(np.minimum(lengths, (lengths - fsize) // stride * stride + fsize).sum()) if stride < fsize else len(tab) * fsize
    print(f"{name:28s} fsize {fsize} stride {stride}: {len(tab)} windows {dt:.2f} s  {len(tab)*fsize/dt/1e6:.1f} Mbp/s of windows ({uniq/dt/1e6:.1f} Mbp/s of unique bases)  [{eng.model.precision}]")
for name, fsize, stride in (("brain", 1500, 1500), ("brain", 2000, 1500), ("brain", 2000, 2000), ("zeus", 1500, 1500), ("baseline500", 500, 500),
                            ("nmdmerge500", 500, 500), ("pyramid", 2000, 2000)):
    cfg = load_model_cfg(name)
    wts = random_weights(build_plan(cfg), 1)
    if name == "pyramid":
        wts = {k: (v * np.float32(0.85) if k.startswith("rep/") and k.endswith("/kernel") else v) for k, v in wts.items()}
    eng = JaegerHipEngine(model_cfg=cfg, weights=wts)
    print("   placement:", eng.model.placement())
    run(name, eng, fsize, stride, lambda b, s, l, f: eng.predict_windows(b, s, l, f, want=("prediction", "reliability")))
    eng.close()
# SURVEY 8(d) config 2's variant: 1 % of the bases in N runs of length 1 - 20 (masks at work in every layer)
bases_n = bases.copy()
rn = np.random.Generator(np.random.PCG64(2))
pos = rn.integers(0, bases_n.size - 20, int(bases_n.size * 0.01 / 10.5))
for p_, l_ in zip(pos.tolist(), rn.integers(1, 21, pos.size).tolist()):
    bases_n[p_:p_ + l_] = ord("N")
cfg = load_model_cfg("brain")
eng = JaegerHipEngine(model_cfg=cfg, weights=random_weights(build_plan(cfg), 1))
keep = bases
bases = bases_n
run("brain, 1 % N runs", eng, 1500, 1500, lambda b, s, l, f: eng.predict_windows(b, s, l, f, want=("prediction", "reliability")))
bases = keep
eng.close()
w = legacy.load_legacy_h5('tests/golden/legacy_data/models/default/WRes_1024.h5')
for prec in ("f32", "f16x3"):
    eng = legacy.LegacyHipEngine(w, precision=prec)
    print("   placement:", eng.model.placement())
    run("legacy default", eng, 2000, 1500, eng.predict_windows)
    eng.close()
PY
