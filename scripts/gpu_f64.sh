python - <<'PY'
import sys, numpy as np, warnings, torch
warnings.simplefilter("ignore")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_model_cfg
from jaeger_amd.engine import JaegerHipEngine, frame_length
from oracle import encoder as oenc, forward as ofwd
cfg = load_model_cfg("brain"); w = ofwd.random_weights(cfg, seed=38341)
rng = np.random.Generator(np.random.PCG64(5)); fsize, n = 1500, 96
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, fsize * n)].copy()
starts = np.arange(n, dtype=np.int64) * fsize; lens = np.full(n, fsize, np.int32)
ids = oenc.encode_windows([seq[s:s + fsize].tobytes() for s in starts], fsize, pad_to=frame_length(fsize))
torch.set_num_threads(32)
r64 = ofwd.forward(cfg, w, ids, dtype=torch.float64); r32 = ofwd.forward(cfg, w, ids)
eng = JaegerHipEngine(model_cfg=cfg, weights=w)
g16 = eng.predict_windows(seq, starts, lens, fsize); eng.model.set_precision("f32"); g32 = eng.predict_windows(seq, starts, lens, fsize); eng.close()
for name, x in (("torch-CPU f32 oracle", r32), ("GPU split-f16", g16), ("GPU exact-f32", g32)):
    print(f"{name:22s} max |logit - f64 truth| = {np.abs(x['prediction'] - r64['prediction']).max():.2e}   (logit range +-{np.abs(r64['prediction']).max():.1f})")
PY
