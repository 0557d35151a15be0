"""Oracle: nucleotide (two-strand) input and the branched model built on it (TEST INFRASTRUCTURE, see
oracle/__init__.py).

**Parity unpinned** at the TensorFlow boundary, like :mod:`oracle.forward`: TensorFlow cannot be imported here and the
reference holds no numeric vectors for this path; its unit tests pin the two lookup tables only
(``tests/unit/test_seqops_encode.py:11-26``: ``A, G, C, T -> 0, 1, 2, 3``, complement with ``N`` for anything else),
and those known answers are asserted in ``tests/test_strands_oracle.py``.

Restated here:

* encoder  ``process_string_inference`` for ``input_type="nucleotide"`` - ``src/jaeger/seqops/encode.py:228-271``
  with ``_map_complement`` ``:28-33`` and ``_map_nucleotide`` ``:36-41``: the window's first ``crop_size`` bytes, their
  reverse complement, both mapped to ids (either case; anything else -1) and one-hot encoded at depth 4
  (-1 = the all-zero row); batches of ragged windows are zero padded (``commands/predict.py:159-183``);
* model    ``nnlib/builder.py:881-882`` (the one-hot strands are the network input), ``:1195-1266``
  ``_build_branched_block`` (ONE branch model - plain Keras ``Conv1D`` / ``Activation`` / ``Dense`` /
  ``GlobalMax/AveragePooling1D``, ``:280-302``, ``:1706-1707`` - applied to every strand), ``:563-590`` (branched
  classifier closed by a ``merge`` layer), ``:776-791`` (outputs: merged ``prediction``, ``embedding`` = Average of the
  strand vectors).

Weight names follow the position of a layer in the YAML (``rep/<i>/...`` inside ``representation_learner.branch``,
``classifier/<i>/...`` inside ``classifier.branch``), as in :mod:`oracle.forward`.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from .forward import activation, conv1d_nwc

_NUC = {"A": 0, "G": 1, "C": 2, "T": 3, "a": 0, "g": 1, "c": 2, "t": 3}                 # encode.py:36-41
_COMPLEMENT = {"A": "T", "T": "A", "G": "C", "C": "G", "a": "t", "t": "a", "g": "c", "c": "g"}   # encode.py:28-33
_ACTS = ("relu", "gelu", "sigmoid", "softmax", "tanh")


def encode_nucleotide_literal(window: str, crop_size: int) -> np.ndarray:
    """One window -> (2, n, 4) float32 one-hot, step by step as the TF ops run (encode.py:234, :256-271)."""
    fwd = list(window)[:crop_size]                                  # bytes_split(x[0])[:crop_size]
    rev = [_COMPLEMENT.get(b, "N") for b in fwd[::-1]]              # map_complement.lookup(forward[::-1]), default "N"
    rows = [[_NUC.get(b, -1) for b in strand] for strand in (fwd, rev)]   # (upper() first when masking is False: same ids)
    ids = np.asarray(rows, np.int64).reshape(2, len(fwd))
    out = np.zeros((2, len(fwd), 4), np.float32)                    # tf.one_hot(nuc, depth=4): -1 -> all zeros
    s, p = np.nonzero(ids >= 0)
    out[s, p, ids[s, p]] = 1.0
    return out


def encode_nucleotide(windows: list[bytes | str], crop_size: int, pad_to: int | None = None) -> np.ndarray:
    """Vectorised: list of windows -> (W, 2, Lmax) uint8 device ids (nucleotide id + 1; 0 = the all-zero one-hot row of any
    other byte and of the zero padding ``padded_batch`` adds)."""
    table = np.zeros(256, np.uint8)
    for ch, v in _NUC.items():
        table[ord(ch)] = v + 1
    comp = np.array([0, 4, 3, 2, 1], np.uint8)                      # A(1) <-> T(4), G(2) <-> C(3); 0 stays 0 ("N" -> -1)
    lens = [min(len(w), crop_size) for w in windows]
    lmax = max(lens + [pad_to or 0])
    out = np.zeros((len(windows), 2, lmax), np.uint8)
    for i, w in enumerate(windows):
        raw = np.frombuffer(w.encode() if isinstance(w, str) else w, np.uint8)[:crop_size]
        f = table[raw]
        out[i, 0, :f.size] = f
        out[i, 1, :f.size] = comp[f[::-1]]
    return out


def _branch_specs(prefix: str, layers: list[dict], cin: int, specs: dict) -> int:
    for i, layer in enumerate(layers):
        name = str(layer.get("name", "")).lower()
        cfg = dict(layer.get("config", {}) or {})
        if name == "conv1d":
            specs[f"{prefix}/{i}/kernel"] = (int(cfg["kernel_size"]), cin, int(cfg["filters"]))
            if cfg.get("use_bias", True):
                specs[f"{prefix}/{i}/bias"] = (int(cfg["filters"]),)
            cin = int(cfg["filters"])
        elif name == "dense":
            specs[f"{prefix}/{i}/kernel"] = (cin, int(cfg["units"]))
            if cfg.get("use_bias", True):
                specs[f"{prefix}/{i}/bias"] = (int(cfg["units"]),)
            cin = int(cfg["units"])
        elif name in _ACTS or name in ("activation", "dropout", "merge"):
            continue
        else:
            raise ValueError(f"oracle: unsupported branch layer {name!r}")
    return cin


def weight_specs(model_cfg: dict) -> dict[str, tuple]:
    specs: dict[str, tuple] = {}
    c = _branch_specs("rep", model_cfg["representation_learner"]["branch"].get("hidden_layers", []), 4, specs)
    _branch_specs("classifier", model_cfg["classifier"]["branch"].get("hidden_layers", []), c, specs)
    return specs


def random_weights(model_cfg: dict, seed: int = 38341) -> dict[str, np.ndarray]:
    """Seeded stand-in weights: He-uniform kernels, N(0, 0.1) biases (as :func:`oracle.forward.random_weights`)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in sorted(weight_specs(model_cfg).items()):
        if name.endswith("/kernel"):
            lim = math.sqrt(6.0 / int(np.prod(shp[:-1])))
            v = rng.uniform(-lim, lim, shp)
        else:
            v = rng.normal(0.0, 0.1, shp)
        out[name] = v.astype(np.float32)
    return out


def _run_branch(x, layers: list[dict], prefix: str, weights: dict, dtype, pooling=None):
    """``_build_block`` (builder.py:982-1193) over plain Keras layers; no masks anywhere."""
    for i, layer in enumerate(layers):
        name = str(layer.get("name", "")).lower()
        cfg = dict(layer.get("config", {}) or {})
        p = f"{prefix}/{i}"
        if name == "conv1d":                                  # tf.keras.layers.Conv1D: strides 1, "valid", dilation 1
            x = conv1d_nwc(x, torch.as_tensor(weights[f"{p}/kernel"]).to(dtype), int(cfg.get("strides", 1)),
                           str(cfg.get("padding", "valid")).upper(), int(cfg.get("dilation_rate", 1)))
            if cfg.get("use_bias", True):
                x = x + torch.as_tensor(weights[f"{p}/bias"]).to(dtype)
            x = activation(cfg.get("activation"), x)
        elif name == "dense":
            x = x @ torch.as_tensor(weights[f"{p}/kernel"]).to(dtype)
            if cfg.get("use_bias", True):
                x = x + torch.as_tensor(weights[f"{p}/bias"]).to(dtype)
            x = activation(cfg.get("activation"), x)
        elif name == "activation" or name in _ACTS:
            x = activation(name if name in _ACTS else cfg.get("activation"), x)
        elif name == "dropout":
            pass
        else:
            raise ValueError(f"oracle: unsupported branch layer {name!r}")
    if pooling is not None:
        pooling = pooling.lower()
        if pooling == "max1d":                                # GlobalMaxPooling1D: over the time axis, padding included
            x = x.max(dim=1).values
        elif pooling == "average1d":
            x = x.mean(dim=1)
        else:
            raise ValueError(f"oracle: unsupported branch pooling {pooling!r}")
    return x


def forward(model_cfg: dict, weights: dict, ids: np.ndarray, dtype=torch.float32) -> dict[str, np.ndarray]:
    """ids (W, 2, L) device ids (0 = all-zero one-hot row) -> {"prediction", "embedding"} (builder.py:776-791)."""
    idt = torch.as_tensor(np.asarray(ids).astype(np.int64))
    onehot = torch.nn.functional.one_hot(torch.clamp(idt - 1, min=0), 4).to(dtype) * (idt != 0).unsqueeze(-1)
    rep_cfg = model_cfg["representation_learner"]["branch"]
    hidden = list(model_cfg["classifier"]["branch"].get("hidden_layers", []))
    if not hidden or str(hidden[-1].get("name", "")).lower() != "merge":
        raise ValueError("Branched classifier must end with a 'merge' layer")       # builder.py:565-568
    method = str((hidden[-1].get("config") or {}).get("method", "average")).lower()
    reps, heads = [], []
    for b in range(onehot.shape[1]):                          # tf.split + squeeze on axis 1, shared-weight branch model
        r = _run_branch(onehot[:, b], rep_cfg.get("hidden_layers", []), "rep", weights, dtype, rep_cfg.get("pooling"))
        reps.append(r)
        heads.append(_run_branch(r, hidden[:-1], "classifier", weights, dtype))
    stack = torch.stack(heads)
    if method not in ("average", "sum", "max", "concat"):
        raise ValueError(f"Unknown merge method: {method}")                             # builder.py:1266
    pred = {"average": stack.mean(dim=0), "sum": stack.sum(dim=0), "max": stack.max(dim=0).values,
            "concat": torch.cat(heads, dim=-1)}[method]                                 # builder.py:1251-1265
    out = {"prediction": pred, "embedding": torch.stack(reps).mean(dim=0)}
    return {k: v.detach().to(torch.float32).numpy() for k, v in out.items()}
