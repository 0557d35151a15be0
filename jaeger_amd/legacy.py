"""The reference's legacy ``default`` model (``jaeger predict -m default``, BASELINE config #1) on
the MI355X engine.

* encoder: ``preprocess/v1/convert.py:56-125`` - the same 6-frame slicing as the modern encoder but
  amino-acid ids 1..21 (``preprocess/v1/maps.py`` TRIMER_INT, 0 = unknown trimer), no upper-casing
  (soft-masked bases break their codons) -> ``jg_encode`` with the v1 table and the case-sensitive
  flag;
* network: ``nnlib/v1/layers.py:154-207,399-423`` (see ``oracle/legacy.py`` for the statement) compiled
  to the same op program the modern models use: conv ops with fused bias / exact-erf GELU /
  batch-norm stages, ``MAXPOOL1D``, ``FRAMESUM``, an unmasked global max pool and three dense ops; it
  runs on the split-f16 conv kernels (first conv by table lookup, ``MaxPool`` on the F16S tensors, erf-GELU
  in the compiled epilogue, window-packed tiles for its short frames) or, with ``precision="f32"``, on the
  exact-f32 MFMA kernels;
* weights: Keras-2.5 ``WRes_1024.h5`` read with :mod:`jaeger_amd.hdf5_lite`;
* outputs keyed like ``JaegerModel.predict`` (``nnlib/inference.py:69-75``): ``y_hat.output``,
  ``y_hat.embedding``, ``meta``.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np

from . import _lib as L
from .maps import V1_TRIMER_INT
from .program import Program, _Blob, pack_conv_kernel

BN_EPS = 1e-3            # tf.keras.layers.BatchNormalization default (nnlib/v1/layers.py:49)
VOCAB, EMB_DIM, WIDTH, N_CLASSES = 22, 4, 128, 4


def tower_layers():
    """(conv, bn, kernel, dilation, pool_after, extra_gelu_after) in graph order."""
    rows = [("block1_0", "bn_block1_1", 9, 1, True, False), ("block1_1", "bn_block1_2", 5, 2, True, False)]
    for n in range(5):
        rows.append((f"block2_{n}1", f"bn_block2_{n}1", 5, 3 + n, False, False))
        rows.append((f"block2_{n}2", f"bn_block2_{n}2", 5, 3 + n, False, True))
    return rows


def weight_shapes() -> dict[str, tuple]:
    shp: dict[str, tuple] = {"aa/embeddings": (VOCAB, EMB_DIM)}
    cin = EMB_DIM
    for conv, bn, k, _, _, _ in tower_layers():
        shp[f"{conv}/kernel"], shp[f"{conv}/bias"] = (k, cin, WIDTH), (WIDTH,)
        for leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
            shp[f"{bn}/{leaf}"] = (WIDTH,)
        cin = WIDTH
    for name, cout in (("augdense-1", WIDTH), ("augdense-2", WIDTH), ("outdense", N_CLASSES)):
        shp[f"{name}/kernel"], shp[f"{name}/bias"] = (WIDTH, cout), (cout,)
    return shp


def load_legacy_h5(path) -> dict[str, np.ndarray]:
    """``WRes_1024.h5`` -> canonical names.  The file stores the first conv as ``conv1d`` (its
    ``layer_names`` order is what Keras' by-order ``load_weights`` follows, so it lands on
    ``block1_0``, ``commands/predict_legacy.py:221``)."""
    from .hdf5_lite import read_datasets
    raw = read_datasets(path)
    out = {}
    for key, arr in raw.items():
        parts = key.strip("/").split("/")
        if len(parts) != 3:
            continue
        layer, leaf = parts[0], parts[2].split(":")[0]
        out[f"{'block1_0' if layer == 'conv1d' else layer}/{leaf}"] = np.asarray(arr, np.float32)
    want = weight_shapes()
    missing = [n for n in want if n not in out]
    if missing:
        raise KeyError(f"{path}: legacy weights missing {missing[:5]}")
    for n, s in want.items():
        if tuple(out[n].shape) != s:
            raise ValueError(f"{path}: {n} has shape {out[n].shape}, expected {s}")
    return {n: out[n] for n in want}


def load_legacy_bundle(graph_dir) -> dict[str, np.ndarray]:
    """The legacy tower's weights out of a SavedModel's ``variables/`` bundle (``<name>_graph/``, the artefact
    ``nnlib/inference.py:307-325`` executes): owner groups in object-graph order and variable attribute names
    (:func:`jaeger_amd.weights.assign_groups`) against the tower's layers in graph order."""
    from pathlib import Path as _P

    from .savedmodel_lite import bundle_layer_groups
    from .weights import assign_groups
    order: list[tuple[str, list[str]]] = [("aa", ["embeddings"])]
    for conv, bn, _, _, _, _ in tower_layers():
        order.append((conv, ["kernel", "bias"]))
        order.append((bn, ["gamma", "beta", "moving_mean", "moving_variance"]))
    for name in ("augdense-1", "augdense-2", "outdense"):
        order.append((name, ["kernel", "bias"]))
    vdir = _P(graph_dir) / "variables"
    got = assign_groups(bundle_layer_groups(vdir), order, weight_shapes(), str(vdir))
    return {n: got[n] for n in weight_shapes()}


def random_weights(seed: int = 1) -> dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in sorted(weight_shapes().items()):
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            lim = np.sqrt(6.0 / int(np.prod(shp[:-1])))
            v = rng.uniform(-lim, lim, shp)
        elif leaf in ("gamma", "moving_variance"):
            v = rng.uniform(0.5, 1.5, shp)
        elif leaf == "embeddings":
            v = rng.normal(0, 1.0, shp)
        else:
            v = rng.normal(0, 0.1, shp)
        out[name] = v.astype(np.float32)
    return out


def compile_legacy(weights: dict[str, np.ndarray]) -> Program:
    blob = _Blob()
    ops = []

    def op(kind, **kw):
        o = L.JgOp()
        o.kind = kind
        for f in ("in_buf", "out_buf", "in_mask", "out_mask", "in_vec", "out_vec"):
            setattr(o, f, -1)
        o.w_off = o.b_off = -1
        o.stride = o.dilation = 1
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def stage(kind, arg=0, p0=-1, p1=-1, p2=-1, p3=-1):
        st = L.JgStage()
        st.kind, st.arg, st.p0, st.p1, st.p2, st.p3, st.f0 = kind, arg, p0, p1, p2, p3, 0.0
        return st

    emb_off = blob.add(weights["aa/embeddings"])
    buf, cin = L.JG_BUF_IDS, EMB_DIM
    for conv, bn, k, d, pool, extra in tower_layers():
        inv_std = (np.float32(1.0) / np.sqrt(weights[f"{bn}/moving_variance"].astype(np.float32)
                                             + np.float32(BN_EPS))).astype(np.float32)
        stages = [stage(L.ST_BIAS, p0=blob.add(weights[f"{conv}/bias"])),
                  stage(L.ST_ACT, arg=L.ACT_GELU_ERF),
                  stage(L.ST_BN, p0=blob.add(weights[f"{bn}/moving_mean"]), p1=blob.add(inv_std),
                        p2=blob.add(weights[f"{bn}/gamma"]), p3=blob.add(weights[f"{bn}/beta"]))]
        if extra:
            stages.append(stage(L.ST_ACT, arg=L.ACT_GELU_ERF))
        out = next(s for s in range(2) if s != buf)
        o = op(L.OP_CONV, in_buf=buf, out_buf=out, k=k, cin=cin, cout=WIDTH, dilation=d, padding=L.PAD_SAME,
               mask_mode=L.MASK_ANY, w_off=blob.add(pack_conv_kernel(weights[f"{conv}/kernel"])))
        if buf == L.JG_BUF_IDS:
            o.b_off = emb_off
        o.n_stages = len(stages)
        for i, st in enumerate(stages):
            o.stages[i] = st
        ops.append(o)
        buf, cin = out, WIDTH
        if pool:
            out = next(s for s in range(2) if s != buf)
            ops.append(op(L.OP_MAXPOOL1D, in_buf=buf, out_buf=out, cout=WIDTH))
            buf = out
    out = next(s for s in range(2) if s != buf)
    ops.append(op(L.OP_FRAMESUM, in_buf=buf, out_buf=out, cout=WIDTH))
    pooled, hidden = L.VEC_SCRATCH0, L.VEC_SCRATCH0 + 1
    ops.append(op(L.OP_POOL, in_buf=out, out_vec=pooled, vec_off=0, cout=WIDTH, arg=L.POOL_MAX))
    for name, src, dst, cout, act in (("augdense-1", pooled, hidden, WIDTH, L.ACT_GELU_ERF),
                                      ("augdense-2", hidden, L.VEC_EMBEDDING, WIDTH, L.ACT_GELU_ERF),
                                      ("outdense", L.VEC_EMBEDDING, L.VEC_PREDICTION, N_CLASSES, L.ACT_NONE)):
        ops.append(op(L.OP_DENSE, in_vec=src, out_vec=dst, vec_off=0, cin=WIDTH, cout=cout, arg=act,
                      w_off=blob.add(weights[f"{name}/kernel"]), b_off=blob.add(weights[f"{name}/bias"])))
    return Program(ops, blob.finish(), VOCAB, N_CLASSES, False, 0, WIDTH)


class LegacyHipEngine:
    """``JaegerModel`` (``nnlib/inference.py:20-75``) stand-in for the ``default`` model."""

    def __init__(self, weights: dict[str, np.ndarray] | str | Path, device_id: int = 0, chunk: int = 0,
                 precision: str | None = None):
        from .engine import HipDevice, HipModel, codon_lut
        if not isinstance(weights, dict):
            # a SavedModel directory (``<name>_graph/`` - what the reference executes) or the Keras-2.5 H5 beside it
            weights = load_legacy_bundle(weights) if Path(weights).is_dir() else load_legacy_h5(weights)
        self.program = compile_legacy(weights)
        self.device = HipDevice(device_id)
        # split-f16 by default like the modern models (f32-accurate; measured 186 Mbp/s of 2000-bp windows
        # vs 75 on the exact-f32 kernels, both within 2e-5 of the oracle); precision="f32" forces the latter
        self.model = HipModel(self.device, self.program)
        if precision is not None:
            self.model.set_precision(precision)
        self.chunk = chunk
        self.lut = codon_lut([v - 1 for v in V1_TRIMER_INT])
        self.encode_flags = 2                          # no upper-casing in the v1 string processor

    def predict_windows(self, bases, win_start, win_len, fsize: int, pre_cased: bool = False) -> dict:
        bases = np.ascontiguousarray(bases, np.uint8)
        ws = np.ascontiguousarray(win_start, np.int64)
        wl = np.ascontiguousarray(win_len, np.int32)
        out = self.model.predict_windows(bases, bases.size, ws, wl, ws.size, fsize, self.lut,
                                         self.encode_flags | (1 if pre_cased else 0), None, self.chunk,
                                         want=("prediction", "embedding"))
        return {"output": out["prediction"], "embedding": out["embedding"], "counts": out["counts"]}

    def forward_ids(self, ids: np.ndarray) -> dict:
        out = self.model.forward(ids, chunk=self.chunk, want=("prediction", "embedding"))
        return {"output": out["prediction"], "embedding": out["embedding"]}

    def close(self):
        self.model.close()
        self.device.close()
