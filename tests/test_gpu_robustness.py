"""Split-f16 ("f16x3") robustness: weight / batch-norm scale sweeps against the f64 evaluation of the same network.

The fast path carries every f32 operand as an f16 pair (22 significant bits) and keeps activations in the f16 exponent
range; real checkpoints have other dynamic ranges than the He-uniform stand-in weights.  For every case below the
engine must do one of two things - never a third:
  * stay within the 1e-4 logit gate of the f64 value (or within 4x the distance the torch-CPU *f32* oracle itself has
    from f64, where f32 arithmetic cannot reach 1e-4 because the logits are huge), or
  * trip its range guard and rerun on the exact-f32 kernels (precision reads "f32" afterwards), with the same bound.
The table this prints is kept in DESIGN.md section 3.1."""
import copy

import numpy as np
import pytest

from conftest import load_model_cfg

pytestmark = pytest.mark.gpu


def _cases(base, rng):
    def scaled(kernel_s=1.0, compensate=False, emb_s=1.0, heavy=False, gamma_lo=None, var=None):
        w = {k: v.copy() for k, v in base.items()}
        for k in w:
            leaf = k.rsplit("/", 1)[1]
            if leaf == "kernel" and k.startswith("rep/") and w[k].ndim == 3 and not (compensate and "/block" not in k):
                if heavy:
                    t = rng.standard_t(2.5, w[k].shape).astype(np.float32)
                    w[k] = (t * np.abs(base[k]).mean() * 0.8).astype(np.float32)
                w[k] = w[k] * np.float32(kernel_s)
                if compensate:          # the norm behind the conv sees moving statistics scaled alike: same function,
                    # conv outputs kernel_s times larger before the affine
                    for cand in (k.replace("conv1/kernel", "bn1/"), k.replace("conv2/kernel", "bn2/")):
                        if cand != k and cand + "moving_mean" in w:
                            w[cand + "moving_mean"] = base[cand + "moving_mean"] * np.float32(kernel_s)
                            w[cand + "moving_variance"] = base[cand + "moving_variance"] * np.float32(kernel_s) ** 2
            if leaf == "embeddings":
                w[k] = w[k] * np.float32(emb_s)
            if leaf == "gamma" and gamma_lo is not None:
                w[k] = np.exp(rng.uniform(np.log(gamma_lo), np.log(1.0 / gamma_lo), w[k].shape)).astype(np.float32)
            if leaf == "moving_variance" and var is not None:
                w[k] = np.exp(rng.uniform(np.log(var), np.log(1.0 / var), w[k].shape)).astype(np.float32)
        return w
    yield "stand-in weights", scaled()
    for s in (0.01, 0.1, 10.0, 100.0, 1000.0):
        yield f"block kernels x{s:g}, norms compensate", scaled(kernel_s=s, compensate=True)
    for s in (0.1, 3.0, 10.0, 30.0):
        yield f"all conv kernels x{s:g}", scaled(kernel_s=s)
    for s in (0.1, 10.0, 30.0):
        yield f"embedding x{s:g}", scaled(emb_s=s)
    yield "heavy-tailed kernels (Student t, 2.5 dof)", scaled(heavy=True)
    yield "gamma log-uniform 1e-2 .. 1e2", scaled(gamma_lo=1e-2)
    yield "gamma log-uniform 1e-3 .. 1e3", scaled(gamma_lo=1e-3)
    yield "moving variance log-uniform 1e-3 .. 1e3", scaled(var=1e-3)


def test_f16x3_scale_sweep_against_f64():
    import torch
    from jaeger_amd.engine import JaegerHipEngine, frame_length
    from oracle import encoder as oenc
    from oracle import forward as ofwd
    cfg = load_model_cfg("brain")
    base = ofwd.random_weights(cfg, seed=38341)
    rng = np.random.Generator(np.random.PCG64(77))
    fsize, n_win = 1500, 8
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=fsize * n_win).copy()
    seq[700:760] = ord("N")
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    lens = np.full(n_win, fsize, np.int32)
    ids = oenc.encode_windows([seq[s:s + fsize].tobytes() for s in starts], fsize, pad_to=frame_length(fsize))
    rows, bad = [], []
    for name, w in _cases(base, rng):
        ref64 = ofwd.forward(cfg, w, ids, dtype=torch.float64)["prediction"]
        ref32 = ofwd.forward(cfg, w, ids)["prediction"]
        with pytest.warns(UserWarning):
            eng = JaegerHipEngine(model_cfg=copy.deepcopy(cfg), weights=w, precision="f16x3")
        got = eng.predict_windows(seq, starts, lens, fsize, want=("prediction",))["prediction"]
        mode = eng.model.precision
        eng.close()
        mag = float(np.abs(ref64).max())
        finite = bool(np.isfinite(got).all())
        e_gpu = float(np.abs(got - ref64).max()) if finite else float("inf")
        e_cpu = float(np.abs(ref32 - ref64).max())
        ok = finite and (e_gpu <= 1e-4 or e_gpu <= 4.0 * e_cpu)
        rows.append((name, mag, mode, e_gpu, e_cpu, ok))
        if not ok:
            bad.append(name)
    print("\n| case | max abs logit | path after the run | GPU vs f64 | torch-CPU f32 vs f64 |")
    print("|---|---|---|---|---|")
    for name, mag, mode, e_gpu, e_cpu, ok in rows:
        print(f"| {name} | {mag:.3g} | {mode}{'' if ok else ' **FAIL**'} | {e_gpu:.2e} | {e_cpu:.2e} |")
    assert not bad, bad


def test_range_guard_inside_the_streamed_pipeline():
    """The host -> HBM pipeline of jg_predict_windows keeps two groups in flight and reads the f16 range guard back per
    group: when it trips (here: conv kernels x10, logits ~1e13), the pipeline drains, falls back to the exact-f32 kernels
    and restarts from the group that tripped it - every window of the call must then equal an exact-f32 run, bit for bit,
    whichever group it was in, and the progress mark must have reached the end."""
    from jaeger_amd.engine import JaegerHipEngine
    from oracle import forward as ofwd
    cfg = load_model_cfg("brain")
    base = ofwd.random_weights(cfg, seed=38341)
    w = {k: (v * np.float32(10.0) if (k.startswith("rep/") and k.endswith("/kernel") and v.ndim == 3) else v.copy())
         for k, v in base.items()}
    rng = np.random.Generator(np.random.PCG64(5))
    fsize, n_win = 1500, 700
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=fsize * n_win).copy()
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    lens = np.full(n_win, fsize, np.int32)
    want = ("prediction", "reliability")
    with pytest.warns(UserWarning):
        ref_eng = JaegerHipEngine(model_cfg=copy.deepcopy(cfg), weights=w, precision="f32")
    ref = ref_eng.predict_windows(seq, starts, lens, fsize, want=want)
    ref_eng.close()
    with pytest.warns(UserWarning):
        eng = JaegerHipEngine(model_cfg=copy.deepcopy(cfg), weights=w, precision="f16x3")
    eng.chunk = 64
    eng.device.set_stream_bytes(128 * 1024)            # ~85 windows per span: a dozen groups, 1 - 2 passes each
    got = eng.predict_windows(seq, starts, lens, fsize, want=want)
    stats = eng.device.stream_stats()
    assert stats["groups"] >= 6 and eng.model.precision == "f32"          # streamed, and the guard tripped
    assert eng.device.windows_done() == n_win
    for k in want + ("counts",):
        np.testing.assert_array_equal(got[k], ref[k])
    # a healthy model through the same pipeline stays on the fast path
    with pytest.warns(UserWarning):
        ok = JaegerHipEngine(model_cfg=copy.deepcopy(cfg), weights=base, precision="f16x3")
    ok.chunk = 64
    ok.device.set_stream_bytes(128 * 1024)
    a = ok.predict_windows(seq, starts, lens, fsize, want=want)
    assert ok.model.precision == "f16x3" and ok.device.stream_stats()["groups"] >= 6
    ok.device.set_stream_bytes(1 << 30)
    b = ok.predict_windows(seq, starts, lens, fsize, want=want)
    assert ok.device.stream_stats()["groups"] == 0
    for k in want + ("counts",):
        np.testing.assert_array_equal(a[k], b[k])
    eng.close()
    ok.close()
