"""Pyramid ResNet (reference train_config/nn_config_baseline.yaml: widths 32/64/128/256, stride-2 blocks, dilations
1-8, 2 000-bp windows) on the GPU: parity vs the oracle and throughput per precision."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_model_cfg  # noqa: E402

from jaeger_amd.engine import JaegerHipEngine, frame_length  # noqa: E402
from oracle import encoder as oenc  # noqa: E402
from oracle import forward as ofwd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pyramid"
fsize = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n_big = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
cfg = load_model_cfg(name)
weights = ofwd.random_weights(cfg, seed=38341)
gain = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
for key in weights:                      # He-uniform kernels in a 36-conv residual pyramid blow the logits up to +-900
    if key.startswith("rep/") and key.endswith("/kernel"):
        weights[key] = weights[key] * np.float32(gain)
rng = np.random.Generator(np.random.PCG64(7))
acgt = np.frombuffer(b"ACGT", np.uint8)
n_win = 6
seq = acgt[rng.integers(0, 4, fsize * n_win, dtype=np.uint8)]
seq[100:160] = ord("N")
starts = (np.arange(n_win) * fsize).astype(np.int64)
lens = np.full(n_win, fsize, np.int32)
lens[2] = fsize * 2 // 3
ids = oenc.encode_windows([seq[s:s + n].tobytes() for s, n in zip(starts, lens)], fsize, pad_to=frame_length(fsize))
ref = ofwd.forward(cfg, weights, ids)
big = acgt[rng.integers(0, 4, fsize * n_big, dtype=np.uint8)]
bstarts = (np.arange(n_big) * fsize).astype(np.int64)
blens = np.full(n_big, fsize, np.int32)
for prec in ("f32", "f16x3"):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=0, precision=prec)
    got = eng.predict_windows(seq, starts, lens, fsize)
    errs = {k: float(np.abs(got[k] - r).max()) for k, r in ref.items()}
    print(name, prec, "->", eng.model.precision, {k: f"{v:.2e}" for k, v in errs.items()},
          "max|logit|", float(np.abs(ref["prediction"]).max()), flush=True)
    want = ("prediction", "reliability")
    eng.predict_windows(big[:fsize * 512], bstarts[:512], blens[:512], fsize, want=want)
    n = n_big if prec == "f16x3" else n_big // 4
    eng.device.profile_enable(True)
    eng.device.profile_read()
    t0 = time.perf_counter()
    eng.predict_windows(big[:fsize * n], bstarts[:n], blens[:n], fsize, want=want)
    dt = time.perf_counter() - t0
    print(f"  {n} windows x {fsize} bp: {dt:.3f} s = {n * fsize / dt / 1e6:.1f} Mbp/s (host buffers in, logits out)", flush=True)
    try:
        prof = eng.device.profile_read()
        for k, v in prof.items():
            if isinstance(v, dict) and v["launches"]:
                print(f"    {k}: {v['ms']:.1f} ms, {v['launches']} launches, {v['flops'] / v['ms'] / 1e9:.1f} TFLOP/s")
    except Exception as ex:  # noqa: BLE001
        print("  (no profile)", ex)
    eng.close()
