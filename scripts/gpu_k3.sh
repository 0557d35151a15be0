cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time, copy, numpy as np, warnings
warnings.simplefilter("ignore")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_model_cfg
from jaeger_amd.engine import JaegerHipEngine
from oracle import forward as ofwd
A = np.frombuffer(b"ACGT", np.uint8)
rng = np.random.Generator(np.random.PCG64(3))
def rate(cfg, fsize, n_win, precision):
    w = ofwd.random_weights(cfg, seed=38341)
    bases = A[rng.integers(0, 4, fsize * n_win, dtype=np.uint8)]
    starts = (np.arange(n_win) * fsize).astype(np.int64); lens = np.full(n_win, fsize, np.int32)
    eng = JaegerHipEngine(model_cfg=cfg, weights=w, device_id=0, precision=precision)
    want = ("prediction", "reliability")
    eng.predict_windows(bases[:fsize * 256], starts[:256], lens[:256], fsize, want=want)
    t = time.time(); eng.predict_windows(bases, starts, lens, fsize, want=want); dt = time.time() - t
    pl = eng.model.placement(); eng.close()
    return n_win * fsize / dt / 1e6, pl
b500 = load_model_cfg("baseline500")
for prec in ("f16x3", "f32"):
    r, pl = rate(b500, 1500, 16384, prec)
    print(f"baseline500 at 1500 bp (layer by layer) {prec}: {r:.0f} Mbp/s {pl}")
brain3 = copy.deepcopy(load_model_cfg("brain"))
for layer in brain3["representation_learner"]["hidden_layers"]:
    if layer["name"] == "residual_block":
        layer["config"].pop("kernel_size", None)
for prec in ("f16x3", "f32"):
    r, pl = rate(brain3, 1500, 8192 if prec == "f16x3" else 2048, prec)
    print(f"brain with 3-tap blocks {prec}: {r:.1f} Mbp/s {pl}")
PY
