"""Two-strand nucleotide model (nn_config_500bp_dvf architecture, seeded weights): resident-input throughput of
jg_predict_windows on 200 000 windows of 500 bp, both arithmetic modes, and where the convolution was placed."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import warnings
from conftest import load_model_cfg  # noqa: E402
from jaeger_amd.engine import JaegerHipEngine  # noqa: E402
from jaeger_amd.plan import build_plan  # noqa: E402
from jaeger_amd.weights import random_weights  # noqa: E402
cfg = load_model_cfg("dvf500")
w = random_weights(build_plan(cfg))
n, fsize = 200_000, 500
rng = np.random.Generator(np.random.PCG64(7))
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * fsize, dtype=np.uint8)]
starts = (np.arange(n) * fsize).astype(np.int64); lens = np.full(n, fsize, np.int32)
for prec in ("f16x3", "f32"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            eng = JaegerHipEngine(model_cfg=cfg, weights=w, device_id=0, precision=prec)
        except Exception as e:
            print(prec, "not available:", e); continue
    print(prec, eng.model.precision, eng.model.placement()); print(eng.model.describe())
    d_b = eng.device.upload(seq); d_s = eng.device.upload(starts); d_l = eng.device.upload(lens)
    out = {k: eng.device.alloc(n * wd * 4) for k, wd in eng.model.widths.items() if wd}
    for rep in range(3):
        t0 = time.perf_counter()
        eng.model.predict_windows_raw(d_b, seq.size, d_s, d_l, n, fsize, eng.lut, eng.encode_flags, fsize, out)
        eng.device.sync(); dt = time.perf_counter() - t0
        print(f"  {prec}: {dt * 1e3:.1f} ms = {n * fsize / dt / 1e6:.0f} Mbp/s")
    eng.close()
