"""Model weights for the MI355X engine: canonical-name dicts (see ``plan.weight_shapes``).

``random_weights`` is the seeded stand-in used when a published checkpoint cannot be
shipped (He-uniform kernels, BN gamma~U[0.5,1.5], beta/mu~N(0,0.1),
var~U[0.5,1.5], seed 38341).  ``load_weights`` reads a model directory entry as produced by
``AvailableModels`` (utils/misc.py:346-392): a canonical ``.npz`` (written by
``save_npz``) is read directly; Keras-3 ``.weights.h5`` files are read with the built-in
:mod:`jaeger_amd.hdf5_lite` (layout per scripts/convert_legacy_classifier_checkpoint.py:29-175 of the reference:
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
    out: list[tuple[str, list[str]]] = []
    if plan.embedding_kind == "embedding":
        out.append(("embedding", ["embeddings"]))
    elif plan.embedding_kind == "onehot_dense":
        out.append(("embedding", ["kernel"]))              # Dense(E, use_bias=False) on the one-hot rows
    norm_names = {l.name for seq in (plan.rep,) for l in seq if isinstance(l, Norm)}

    def conv(c: Conv):
        out.append((c.name, ["kernel", "bias"] if c.use_bias else ["kernel"]))

    for seq in (plan.rep, plan.classifier, plan.reliability or []):
        if seq is plan.classifier and plan.nmd_merge_mode != "concat" and len(plan.nmd_dims) > 1:
            # NMDMerge(sum / mean / max / weighted) is called at the end of the representation block (builder.py:1176-1180):
            # its own variable (``layer_weights``, nmd.py:143-149) sits on the layer, the bias-free projections are the
            # sub-layers ``projections/<i>`` (nmd.py:133-141) - in a bundle the layer's path sorts in front of theirs
            if plan.nmd_merge_mode == "weighted":
                out.append(("rep/nmd_merge", ["layer_weights"]))
            for i in range(len(plan.nmd_dims)):
                out.append((f"rep/nmd_merge/proj_{i}", ["kernel"]))
        for layer in seq:
            if isinstance(layer, Conv):
                conv(layer)
            elif isinstance(layer, Norm):
                out.append((layer.name, norm_vars[layer.kind]))
            elif isinstance(layer, Nmd):
                if layer.name not in norm_names:           # (a return_nmd norm's tap shares the norm's own variables)
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


_BLOCK_SUBLAYERS = ("conv1", "conv2", "conv3", "bn1", "bn2", "bn3")


def _natural_key(path: tuple[str, ...]):
    """Keras de-duplicates auto names as ``name``, ``name_1``, ``name_2`` ... and a ResidualBlockStack names its
    blocks ``rep_residual_block_<i>_<j>`` (layers.py:2676-2692): every digit run compares as a number (``_10`` after
    ``_2``), a name without a suffix sorts in front of its ``_1``."""
    import re
    key = []
    for comp in path:
        toks = [(1, int(t), "") if t.isdigit() else (0, 0, t) for t in re.split(r"(\d+)", comp) if t != ""]
        key.append(tuple(toks))
    return key


def load_keras3_h5(path, plan: ModelPlan) -> dict[str, np.ndarray]:
    """Keras-3 ``.weights.h5`` -> canonical names.  The file holds ``layers/<name>/vars/<i>`` groups
    (containers add another ``layers/`` level; variables in ``add_weight`` order), with Keras-generated
    names that differ between export generations - the reference's own converter maps by position too
    (scripts/convert_legacy_classifier_checkpoint.py:76-175).  Mapping rules: residual blocks are paired
    in natural name order (``residual_block``, ``residual_block_1`` ...) and their sub-layers by their
    attribute names (``conv1 conv2 conv3 bn1 bn2 bn3``, layers.py:1839-1876); every other weighted
    layer takes the first unused group with its variable shapes in natural name order, which is
    creation order for same-type layers."""
    from .hdf5_lite import read_datasets
    shapes = weight_shapes(plan)
    by_group: dict[tuple[str, ...], dict[int, np.ndarray]] = {}
    for key, arr in read_datasets(path).items():
        parts = [c for c in key.strip("/").split("/")]
        if len(parts) >= 3 and parts[-2] == "vars" and parts[-1].isdigit():
            comps = tuple(c for c in parts[:-2] if c != "layers")
            by_group.setdefault(comps, {})[int(parts[-1])] = arr
    groups = {k: [g[i] for i in range(len(g))] for k, g in by_group.items() if g}
    block_paths = sorted({k[:-1] for k in groups if k[-1] in _BLOCK_SUBLAYERS and len(k) >= 2}, key=_natural_key)
    # loose layers: Keras numbers auto names per class GLOBALLY in creation order (``masked_batch_norm``,
    # ``masked_batch_norm_1`` ... wherever the layer sits), so the layer's own name decides first and its container only
    # breaks ties (names given per container - ``dense`` in ``functional_8`` and in ``functional_9`` - keep path order)
    # ... - but `save_weights` numbers the layers of EVERY container from scratch (``functional_8/{dense, dense_1}`` and
    # ``functional_9/{dense, dense_1}``): when a leaf name occurs in more than one container the names are per container,
    # and the containers' own order (path-major) is creation order
    loose = [k for k in groups if not (k[-1] in _BLOCK_SUBLAYERS and k[:-1] in block_paths)]
    if len({k[-1] for k in loose}) == len(loose):
        loose.sort(key=lambda k: (_natural_key(k[-1:]), _natural_key(k)))
    else:
        loose.sort(key=_natural_key)
    order = _layer_order(plan)
    plan_blocks: list[str] = []
    for prefix, _ in order:
        head, _, sub = prefix.rpartition("/")
        if sub in _BLOCK_SUBLAYERS and head not in plan_blocks:
            plan_blocks.append(head)

    def describe(keys):
        return "; ".join(f"{'/'.join(k)} {[tuple(a.shape) for a in groups[k]]}" for k in keys) or "none"

    if len(plan_blocks) != len(block_paths):
        raise ValueError(f"{path}: {len(block_paths)} residual blocks in the file ({', '.join('/'.join(b) for b in block_paths)}), "
                         f"the plan has {len(plan_blocks)}")
    out = {}
    used: set = set()
    missing: list[str] = []
    for prefix, leaves in order:
        want = [tuple(shapes[f"{prefix}/{v}"]) for v in leaves]
        head, _, sub = prefix.rpartition("/")
        if sub in _BLOCK_SUBLAYERS and head in plan_blocks:
            cands = [block_paths[plan_blocks.index(head)] + (sub,)]
        else:
            cands = [k for k in loose if k not in used]
        for k in cands:
            if k in groups and k not in used and [tuple(a.shape) for a in groups[k]] == want:
                used.add(k)
                for v, a in zip(leaves, groups[k]):
                    out[f"{prefix}/{v}"] = np.asarray(a, np.float32)
                break
        else:
            missing.append(f"{prefix} {want}")
    left = [k for k in sorted(groups, key=_natural_key) if k not in used]
    if missing or left:
        raise ValueError(f"{path}: the file does not line up with the layer plan - plan layers without a weight group: "
                         f"{'; '.join(missing) or 'none'} - weight groups of the file no plan layer took: {describe(left)}")
    return out


class BundleSchemeError(ValueError):
    """A variable bundle whose checkpoint keys follow no object-path scheme this loader knows (``_operations/<n>/...`` or
    ``layers/<n>/...``) or name variables no layer of the reference owns: the one bundle failure `load_weights` answers
    with the weights file.  A bundle that IS understood and disagrees with the plan (a missing layer, a shape) stays fatal."""


_KNOWN_VARIABLES = {"kernel", "bias", "embeddings", "gamma", "beta", "moving_mean", "moving_variance", "alpha",
                    "layer_weights"}


def assign_groups(groups: list[tuple[tuple[str, ...], dict[str, np.ndarray]]], order: list[tuple[str, list[str]]],
                  shapes: dict[str, tuple], source: str) -> dict[str, np.ndarray]:
    """Owner groups in creation order (``savedmodel_lite.bundle_layer_groups``) -> canonical names.  Nothing is guessed:
    a residual block's sub-layers are found by their attribute names under the block's own path (blocks paired in
    order), every other weighted layer takes the NEXT unused group, whose variable names and shapes must be exactly
    the layer's - the first disagreement is an error that names both sides."""
    for path, have in groups:
        if not (path[0] in ("_operations", "layers") and len(path) >= 2 and path[1].isdigit()):
            raise BundleSchemeError(f"{source}: object path {'/'.join(path)} follows no known scheme "
                                    f"(_operations/<n>/... or layers/<n>/...)")
        odd = sorted(set(have) - _KNOWN_VARIABLES)
        if odd:
            raise BundleSchemeError(f"{source}: {'/'.join(path)} holds variables {odd} no layer of the plan's families owns")
    block_paths = []
    for path, _ in groups:
        if path[-1] in _BLOCK_SUBLAYERS and path[:-1] not in block_paths:
            block_paths.append(path[:-1])
    by_path = dict(groups)
    loose = [path for path, _ in groups if not (path[-1] in _BLOCK_SUBLAYERS and path[:-1] in block_paths)]
    plan_blocks: list[str] = []
    for prefix, _ in order:
        head, _, sub = prefix.rpartition("/")
        if sub in _BLOCK_SUBLAYERS and head not in plan_blocks:
            plan_blocks.append(head)
    if len(plan_blocks) != len(block_paths):
        raise ValueError(f"{source}: {len(block_paths)} residual blocks in the bundle, the plan has {len(plan_blocks)}")
    if len(groups) != len(order):
        raise ValueError(f"{source}: {len(groups)} weighted layers in the bundle, the plan has {len(order)}")
    out: dict[str, np.ndarray] = {}
    nxt = 0
    for prefix, leaves in order:
        head, _, sub = prefix.rpartition("/")
        if sub in _BLOCK_SUBLAYERS and head in plan_blocks:
            path = block_paths[plan_blocks.index(head)] + (sub,)
            if path not in by_path:
                raise ValueError(f"{source}: residual block {'/'.join(path[:-1])} has no sub-layer {sub!r} (plan layer {prefix})")
        else:
            if nxt >= len(loose):
                raise ValueError(f"{source}: no variables left for plan layer {prefix}")
            path = loose[nxt]
            nxt += 1
        have = by_path[path]
        if sorted(have) != sorted(leaves):
            raise ValueError(f"{source}: {'/'.join(path)} holds {sorted(have)}, plan layer {prefix} needs {sorted(leaves)}")
        for leaf in leaves:
            want = tuple(shapes[f"{prefix}/{leaf}"])
            if tuple(have[leaf].shape) != want:
                raise ValueError(f"{source}: {'/'.join(path)}/{leaf} has shape {tuple(have[leaf].shape)}, "
                                 f"plan layer {prefix} needs {want}")
            out[f"{prefix}/{leaf}"] = np.asarray(have[leaf], np.float32)
    return out


def load_savedmodel_bundle(graph_dir, plan: ModelPlan) -> dict[str, np.ndarray]:
    """Weights out of ``<name>_graph/variables`` - the artefact the reference executes (nnlib/inference.py:307-325) -
    by object-graph order and variable attribute names (:func:`assign_groups`), no shape heuristics."""
    from .savedmodel_lite import bundle_layer_groups
    vdir = Path(graph_dir) / "variables"
    return assign_groups(bundle_layer_groups(vdir), _layer_order(plan), weight_shapes(plan), str(vdir))


def bundle_checkpoint_keys(plan: ModelPlan) -> dict[str, str]:
    """Canonical name -> the checkpoint key a Keras-3 export of this plan's model carries: ``_operations/<n>/...`` with n
    counting the model's operations in graph order (weighted or not - here only the weighted ones are known, so they are
    numbered densely from 1), ``_kernel`` / ``_embeddings`` for the kernels, the attribute name otherwise.  Used to
    write bundles in the reference's container (tests; exporting stand-in weights)."""
    keys = {}
    n = 0
    last_block = None
    merge_n = None
    for prefix, leaves in _layer_order(plan):
        head, _, sub = prefix.rpartition("/")
        if prefix == "rep/nmd_merge" or head == "rep/nmd_merge":
            # one operation: ``layer_weights`` on the layer itself, the projections under ``projections/<i>``
            if merge_n is None:
                n += 1
                merge_n = n
                last_block = None
            base = f"_operations/{merge_n}" if prefix == "rep/nmd_merge" else f"_operations/{merge_n}/projections/{sub[len('proj_'):]}"
        elif sub in _BLOCK_SUBLAYERS:
            if head != last_block:
                n += 1
                last_block = head
            base = f"_operations/{n}/{sub}"
        else:
            n += 1
            last_block = None
            base = f"_operations/{n}"
        for leaf in leaves:
            attr = "_" + leaf if leaf in ("kernel", "embeddings") else leaf
            keys[f"{prefix}/{leaf}"] = f"{base}/{attr}/.ATTRIBUTES/VARIABLE_VALUE"
    return keys


def load_weights(path_dict: dict, plan: ModelPlan, trust_project: bool = False) -> dict[str, np.ndarray]:
    """Weights of a model entry (AvailableModels keys, utils/misc.py:346-392).  The SavedModel's own variable bundle
    (``graph``) comes first - it is what the reference runs; then the weights file: the Keras-3 ``.weights.h5`` the
    reference's ``save_model`` writes beside the graph (nnlib/builder.py:1505-1509) before the canonical ``.npz`` this
    package derives.  A bundle under a key SCHEME this loader does not know (:class:`BundleSchemeError`) is not fatal when
    the entry also holds a weights file: that file is used, with a warning through the run log that names what failed.
    A bundle that is understood and disagrees with the plan - a missing layer, another shape - stays an error (a model
    directory whose file differs from the variables the reference executes must refuse, not predict something else);
    ``trust_project`` skips the bundle altogether."""
    graph = path_dict.get("graph")
    w = path_dict.get("weights") or path_dict.get("weights_npz")     # AvailableModels keys (predict.py)
    scheme_error = None
    if not trust_project and graph is not None and (Path(graph) / "variables" / "variables.index").exists():
        try:
            return load_savedmodel_bundle(graph, plan)
        except BundleSchemeError as e:
            if w is None:
                raise
            scheme_error = e
    if w is None:
        raise FileNotFoundError("model entry has no weights (a <name>_graph/variables bundle, *.weights.h5 or canonical *.npz)")
    w = Path(w)
    if scheme_error is not None:
        import logging
        import warnings
        msg = (f"{graph}/variables could not be mapped onto the layer plan ({scheme_error}); falling back to {w}.  The "
               f"SavedModel is what the reference executes - check the model with `jaeger_amd verify-model` before "
               f"trusting these weights.")
        logging.getLogger("Jaeger").warning(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    if w.suffix == ".npz":
        return load_npz(w)
    return load_keras3_h5(w, plan)
