"""Model weights for the MI355X engine: canonical-name dicts (see ``plan.weight_shapes``).

``random_weights`` is the seeded stand-in used when a published checkpoint cannot be
shipped (He-uniform kernels, BN gamma~U[0.5,1.5], beta/mu~N(0,0.1),
var~U[0.5,1.5], seed 38341).  ``load_weights`` reads a model directory entry as produced by
``AvailableModels`` (utils/misc.py:346-392): a canonical ``.npz`` (written by
``save_npz``) is read directly; Keras-3 ``.weights.h5`` files need the optional ``h5py``
module (layout per scripts/convert_legacy_classifier_checkpoint.py:29-175 of the reference:
``layers/<layer>/vars/<i>`` in variable-creation order).
"""

from __future__ import annotations

import math
from pathlib import Path

import numpy as np

from .plan import Conv, Dense, ModelPlan, Nmd, Norm, ResBlock, weight_shapes


def random_weights(plan: ModelPlan, seed: int = 38341) -> dict[str, np.ndarray]:
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in sorted(weight_shapes(plan).items()):
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            lim = math.sqrt(6.0 / int(np.prod(shp[:-1])))
            v = rng.uniform(-lim, lim, shp)
        elif leaf == "embeddings":
            v = rng.normal(0.0, 1.0 / math.sqrt(shp[1]) * 4.0, shp)
        elif leaf in ("gamma", "moving_variance"):
            v = rng.uniform(0.5, 1.5, shp)
        elif leaf == "alpha":
            v = np.full(shp, 0.5)
        else:  # bias, beta, moving_mean
            v = rng.normal(0.0, 0.1, shp)
        out[name] = v.astype(np.float32)
    return out


def save_npz(path, weights: dict[str, np.ndarray]) -> None:
    np.savez(path, **{k.replace("/", "."): v for k, v in weights.items()})


def load_npz(path) -> dict[str, np.ndarray]:
    z = np.load(path)
    return {k.replace(".", "/"): z[k] for k in z.files}


def _layer_order(plan: ModelPlan) -> list[tuple[str, list[str]]]:
    """(canonical prefix, variable leaves in Keras add_weight order) for every weighted layer in
    graph order: Embedding, then rep / classifier / reliability layers (layers.py:1196-1211
    conv, :828-855 batch-norm, :412-429 DyT, :321-331 layer-norm; nmd.py:34-40)."""
    norm_vars = {"masked_batchnorm": ["gamma", "beta", "moving_mean", "moving_variance"],
                 "masked_dyt": ["alpha", "gamma", "beta"], "masked_layernorm": ["gamma", "beta"]}
    out: list[tuple[str, list[str]]] = [("embedding", ["embeddings"])]

    def conv(c: Conv):
        out.append((c.name, ["kernel", "bias"] if c.use_bias else ["kernel"]))

    for seq in (plan.rep, plan.classifier, plan.reliability or []):
        for layer in seq:
            if isinstance(layer, Conv):
                conv(layer)
            elif isinstance(layer, Norm):
                out.append((layer.name, norm_vars[layer.kind]))
            elif isinstance(layer, Nmd):
                out.append((layer.name, ["moving_mean"]))
            elif isinstance(layer, ResBlock):       # sublayer creation order, layers.py:1839-1876
                conv(layer.conv1)
                conv(layer.conv2)
                if layer.conv3 is not None:
                    conv(layer.conv3)
                    out.append((layer.bn3.name, norm_vars[layer.bn3.kind]))
                out.append((layer.bn1.name, norm_vars[layer.bn1.kind]))
                out.append((layer.bn2.name, norm_vars[layer.bn2.kind]))
            elif isinstance(layer, Dense):
                out.append((layer.name, ["kernel", "bias"] if layer.use_bias else ["kernel"]))
    return out


def load_keras3_h5(path, plan: ModelPlan) -> dict[str, np.ndarray]:
    """Keras-3 ``.weights.h5``: walk ``layers/**/vars/<i>`` groups in file order and assign them
    to the plan's weighted layers by type and order (names in the file are Keras-generated and
    differ between export generations - the reference's own converter maps by order too,
    scripts/convert_legacy_classifier_checkpoint.py:76-175)."""
    try:
        import h5py
    except ImportError as e:          # pragma: no cover - h5py is optional
        raise RuntimeError(
            "reading Keras .weights.h5 needs the optional 'h5py' module; alternatively convert the "
            "weights once to the canonical .npz with jaeger_amd.weights.save_npz") from e
    shapes = weight_shapes(plan)
    groups: list[list[np.ndarray]] = []
    with h5py.File(path, "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Group) and name.endswith("/vars") and len(obj):
                groups.append([np.asarray(obj[str(i)]) for i in range(len(obj))])
        f.visititems(visit)
    order = _layer_order(plan)
    if len(groups) != len(order):
        raise ValueError(f"{path}: {len(groups)} weighted layers in the file, the plan has {len(order)}")
    out = {}
    used = [False] * len(groups)
    for prefix, leaves in order:
        want = [tuple(shapes[f"{prefix}/{v}"]) for v in leaves]
        for gi, g in enumerate(groups):          # first unused group with matching shapes
            if not used[gi] and [tuple(a.shape) for a in g] == want:
                used[gi] = True
                for v, a in zip(leaves, g):
                    out[f"{prefix}/{v}"] = a.astype(np.float32)
                break
        else:
            raise ValueError(f"{path}: no weight group matches {prefix} {want}")
    return out


def load_weights(path_dict: dict, plan: ModelPlan) -> dict[str, np.ndarray]:
    w = path_dict.get("weights")
    if w is None:
        raise FileNotFoundError("model entry has no weights file (*.weights.h5 or canonical *.npz)")
    w = Path(w)
    if w.suffix == ".npz":
        return load_npz(w)
    return load_keras3_h5(w, plan)
