"""Relative-throughput guards, measured on the box the suite runs on: each model family must stay on its fast kernels.
(Two placement / register-allocation regressions of round 2 - the fused small-window kernel switched off by a wider
eligibility rule, the DyT stack-end epilogue spilling 576 bytes per lane - passed every parity test and were 4 - 5x
slower; ratios against a reference run on the same GPU catch that class of bug without depending on the box.)"""
import time
import warnings

import numpy as np
import pytest

from conftest import load_model_cfg

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", np.uint8)


def _rate(name, fsize, n_win, precision=None, gain=None):
    from jaeger_amd.engine import JaegerHipEngine
    from oracle import forward as ofwd
    cfg = load_model_cfg(name)
    weights = ofwd.random_weights(cfg, seed=38341)
    if gain is not None:
        weights = {k: (v * np.float32(gain) if k.startswith("rep/") and k.endswith("/kernel") else v)
                   for k, v in weights.items()}
    rng = np.random.Generator(np.random.PCG64(3))
    bases = ACGT[rng.integers(0, 4, fsize * n_win, dtype=np.uint8)]
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    lens = np.full(n_win, fsize, np.int32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=0, precision=precision)
    try:
        want = ("prediction", "reliability")
        eng.predict_windows(bases[:fsize * 256], starts[:256], lens[:256], fsize, want=want)
        best = 0.0
        for _ in range(2):
            t0 = time.perf_counter()
            eng.predict_windows(bases, starts, lens, fsize, want=want)
            best = max(best, n_win * fsize / (time.perf_counter() - t0) / 1e6)
        mode = eng.model.precision
    finally:
        eng.close()
    return best, mode


def test_zeus_dyt_epilogues_keep_pace_with_brain():
    brain, m1 = _rate("brain", 1500, 6144)
    zeus, m2 = _rate("zeus", 1500, 6144)
    print(f"brain {brain:.1f} Mbp/s, zeus {zeus:.1f} Mbp/s")
    assert m1 == m2 == "f16x3"
    assert zeus >= 0.5 * brain, (zeus, brain)          # measured 0.95; the spilling epilogue gave 0.27


def test_small_window_family_runs_fused():
    fused, m = _rate("baseline500", 500, 98_304)
    layered, _ = _rate("baseline500", 500, 24_576, precision="f32")
    print(f"baseline500 fused {fused:.0f} Mbp/s, exact-f32 layer by layer {layered:.0f} Mbp/s")
    assert m == "f16x3" and fused >= 2.0 * layered, (fused, layered)        # measured 5 - 6x with host buffers; unfused 1.3x


def test_pyramid_runs_on_the_split_f16_kernels():
    fast, m = _rate("pyramid", 2000, 6144, gain=0.85)
    exact, _ = _rate("pyramid", 2000, 1536, precision="f32", gain=0.85)
    print(f"pyramid split-f16 {fast:.1f} Mbp/s, exact-f32 {exact:.1f} Mbp/s")
    assert m == "f16x3" and fast >= 2.2 * exact, (fast, exact)              # measured 4.6x; with spilling kernels 1.8x
