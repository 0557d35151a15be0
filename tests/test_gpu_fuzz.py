"""Differential fuzz of the program compiler + format planner + kernels: seeded random conv-family architectures
(kernel sizes, dilations, strides, widths - constant or changing from block to block like the reference's pyramid
ResNet, 32 to 256 channels -, norm types, mask modes, 1x1 bypasses, NMD taps in either position, return_nmd, pooling) run on the split-f16 / mixed path AND on the exact-f32 path, both against the CPU oracle."""
import copy

import numpy as np
import pytest

from conftest import load_model_cfg

pytestmark = pytest.mark.gpu
TOL = 1e-4


def random_model(rng):
    base = copy.deepcopy(load_model_cfg("brain"))
    width = int(rng.choice([128, 128, 128, 64, 32]))
    pyramid = rng.random() < 0.4                             # widths change from block to block
    emb = int(rng.choice([16, 64, 128]))
    base["embedding"]["embedding_size"] = emb
    layers = []
    k0 = int(rng.choice([3, 5, 7, 9]))
    layers.append({"name": "masked_conv1d", "config": {"filters": width, "kernel_size": k0,
                                                         "padding": str(rng.choice(["same", "valid"])),
                                                         "mask_mode": str(rng.choice(["any", "any", "majority", "strict"]))}})
    n_nmd = 0

    def tail(allow_return_nmd=True):
        nonlocal n_nmd
        norm = str(rng.choice(["masked_batchnorm", "masked_batchnorm", "masked_dyt", "masked_layernorm", "none"]))
        tap = str(rng.choice(["front", "back", "none", "return"]))
        if tap == "front":
            layers.append({"name": "nmd", "config": {}})
            n_nmd += 1
        if norm != "none":
            cfg = {}
            if tap == "return" and norm == "masked_batchnorm" and allow_return_nmd:
                cfg["return_nmd"] = True
                n_nmd += 1
            layers.append({"name": norm, "config": cfg})
        if rng.random() < 0.85:
            layers.append({"name": "activation", "config": {"activation": "gelu"}})

    tail()
    for _ in range(int(rng.integers(1, 4))):
        prev_width = width
        if pyramid and rng.random() < 0.7:
            width = int(rng.choice([32, 64, 128, 128, 256]))
        if rng.random() < 0.8:
            nt = str(rng.choice(["masked_batchnorm", "masked_batchnorm", "masked_dyt", "masked_layernorm"]))
            cfg = {"filters": width, "kernel_size": int(rng.choice([2, 3, 4, 5, 5, 5, 7])), "block_size": int(rng.integers(1, 3)),
                   "dilation_rate": int(rng.choice([1, 2, 3, 4, 8])), "strides": int(rng.choice([1, 1, 1, 2])),
                   "use_1x1conv": bool(rng.random() < 0.3), "norm_type": nt}
            if nt == "masked_batchnorm" and rng.random() < 0.3:
                cfg["return_nmd"] = True
                n_nmd += 1
            if width != prev_width and cfg["strides"] == 1:
                cfg["use_1x1conv"] = True                    # (the reference adds x itself otherwise: layers.py:1905-1912)
            layers.append({"name": "residual_block", "config": cfg})
        else:
            layers.append({"name": "masked_conv1d", "config": {"filters": width, "kernel_size": int(rng.choice([1, 3, 5])),
                                                                 "padding": "same", "dilation_rate": int(rng.choice([1, 2]))}})
        if rng.random() < 0.7:
            if rng.random() < 0.5:
                layers.append({"name": "nmd", "config": {}})
                n_nmd += 1
            else:
                tail()
    if n_nmd == 0:
        layers.append({"name": "nmd", "config": {}})
        n_nmd = 1
    base["representation_learner"]["hidden_layers"] = layers
    base["representation_learner"]["pooling"] = str(rng.choice(["max", "average"]))
    base["classifier"]["hidden_layers"] = [{"name": "dense", "config": {"units": 6, "activation": None}}]
    base["classifier"]["input_shape"] = width
    base["reliability_model"]["hidden_layers"] = [{"name": "dense", "config": {"units": 8, "activation": "gelu"}},
                                                  {"name": "dense", "config": {"units": 1, "activation": None}}]
    base["reliability_model"].pop("input_shape", None)
    base["reliability_model"]["mode"] = "nmd"
    if rng.random() < 0.25:
        base["reliability_model"]["mode"] = "nmd_plus_signals"
        base["reliability_model"]["signals"] = [str(x) for x in rng.permutation(["max_prob", "entropy", "energy", "margin",
                                                                                "nmd_norm"])[:int(rng.integers(1, 6))]]
    if rng.random() < 0.2:
        base["use_masking"] = False
    if rng.random() < 0.2:                                   # hidden dense layer in the classifier
        base["classifier"]["hidden_layers"] = [{"name": "dense", "config": {"units": 24, "activation": "gelu"}},
                                               {"name": "dense", "config": {"units": 6, "activation": None}}]
    if rng.random() < 0.15:                                  # translated one-hot input
        base["embedding"].update(use_embedding_layer=False, embedding_size=int(rng.choice([0, 32])),
                                 input_shape=[6, None, 64])
        base["string_processor"]["seq_onehot"] = True
    if rng.random() < 0.15 and width == 32 and not pyramid:                  # the small-window family shape: k = 3 blocks, batch norms
        layers = [layers[0], {"name": "masked_batchnorm", "config": {}}, {"name": "activation", "config": {"activation": "gelu"}},
                  {"name": "residual_block", "config": {"filters": 32, "kernel_size": 3,
                                                        "block_size": int(rng.integers(1, 3))}},
                  {"name": "masked_batchnorm", "config": {}}, {"name": "activation", "config": {"activation": "gelu"}}]
        if rng.random() < 0.5:
            layers.append({"name": "nmd", "config": {}})
        else:
            base.pop("reliability_model", None)
        base["representation_learner"]["hidden_layers"] = layers
    # round 6: NMDMerge over several taps (nmd.py:93-155) - projecting modes, now and then with an activated projection
    taps = sum(1 for l in base["representation_learner"]["hidden_layers"]
               if l["name"] == "nmd" or (l.get("config") or {}).get("return_nmd"))
    if "reliability_model" in base and taps >= 2 and rng.random() < 0.4:
        merge = {"mode": str(rng.choice(["sum", "mean", "max", "weighted"])), "target_dim": int(rng.choice([8, 24, 40]))}
        if rng.random() < 0.5:
            merge["projection_kwargs"] = {"activation": str(rng.choice(["gelu", "relu", "tanh"]))}
        base["reliability_model"]["merge"] = merge
    return base


import os

@pytest.mark.parametrize("seed", list(range(int(os.environ.get("JAEGER_FUZZ_SEEDS", "24")))))
def test_random_architecture(seed):
    from jaeger_amd.engine import JaegerHipEngine, frame_length
    from oracle import encoder as oenc
    from oracle import forward as ofwd
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    cfg = random_model(rng)
    weights = ofwd.random_weights(cfg, seed=seed)
    # 2000-2500 bp: the window-packed tiling of the k = 5 convs; small chunks: several launch groups per call
    fsize, n_win = int(rng.choice([450, 600, 900, 900, 1500, 2000, 2500])), 7
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=fsize * n_win).copy()
    for s in rng.integers(0, seq.size, 6):
        seq[s:s + rng.integers(1, 30)] = ord("N")
    starts = (np.arange(n_win) * fsize).astype(np.int64)
    lens = np.full(n_win, fsize, np.int32)
    lens[1::3] = rng.integers(fsize // 2, fsize, lens[1::3].size)
    ids = oenc.encode_windows([seq[s:s + n].tobytes() for s, n in zip(starts, lens)], fsize, pad_to=frame_length(fsize))
    ref = ofwd.forward(cfg, weights, ids)
    names = [l["name"] + (str(l["config"].get("strides", "")) if l["name"] == "residual_block" else "")
             for l in cfg["representation_learner"]["hidden_layers"]]
    from jaeger_amd._lib import JaegerHipError
    from jaeger_amd.plan import UnsupportedLayer
    try:
        with pytest.warns(UserWarning):
            eng = JaegerHipEngine(model_cfg=cfg, weights=weights, chunk=int(rng.choice([0, 0, 3, 5])))
    except (UnsupportedLayer, JaegerHipError) as e:        # a loud refusal is fine; a wrong number is not
        pytest.skip(f"architecture refused: {e}")
    modes = [eng.model.precision] + (["f32"] if eng.model.precision != "f32" else [])
    for mode in modes:
        eng.model.set_precision(mode)
        got = eng.predict_windows(seq, starts, lens, fsize)
        again = eng.predict_windows(seq, starts, lens, fsize)
        for k, r in ref.items():
            scale = max(1.0, float(np.abs(r).max()) / 8)
            err = float(np.abs(got[k] - r).max())
            assert got[k].shape == r.shape and err <= TOL * scale, (seed, mode, k, err, names)
            np.testing.assert_array_equal(got[k], again[k])
    eng.close()
