"""Oracle: model forward pass on the CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

**Parity unpinned** at the TensorFlow boundary: TensorFlow/Keras cannot be
imported here and the reference's tests hold no numeric vectors for these ops.
This module restates, op by op in torch-CPU (f32 by default, f64 on request),
what the reference's SavedModel graph computes for the conv model family:

* graph assembly         ``nnlib/builder.py:442-838`` (``build_fragment_classifier``),
                         ``:844-894`` (``_build_embedding``), ``:982-1193`` (``_build_block``),
                         ``:1697-1714`` (``_get_pooler``)
* MaskedConv1D           ``nnlib/v2/layers.py:1128-1332``
* MaskedBatchNorm        ``nnlib/v2/layers.py:796-973`` (inference branch ``:918-938``)
* MaskedDYT / MaskedLN   ``nnlib/v2/layers.py:385-452`` / ``:293-382``
* GELU (tanh form)       ``nnlib/v2/layers.py:21-32``; Keras-3 ``Activation("gelu")``
* ResidualBlock / Stack  ``nnlib/v2/layers.py:1774-1973`` / ``:2648-2721``
* NMDLayer / NMDMerge    ``nnlib/v2/nmd.py:8-90`` / ``:93-192``
* masked pools           ``nnlib/v2/layers.py:455-538``
* OODSignalLayer         ``nnlib/v2/layers.py:1598-1667``

Weights are addressed by canonical names derived from the position of a layer
in the ``*_project.yaml`` (see :func:`weight_specs`), independent of the product's
``jaeger_amd.plan``.
"""

from __future__ import annotations

import math
from typing import Any

import numpy as np
import torch
import torch.nn.functional as F

_ACT_ALIASES = {"relu", "gelu", "sigmoid", "softmax", "tanh"}


# --------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------
def gelu_tanh(x):
    """tf.nn.gelu(approximate=True), layers.py:29."""
    if FAST:
        return F.gelu(x, approximate="tanh")
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x * x * x)))


def gelu_erf(x):
    """exact GELU 0.5*x*erfc(-x/sqrt(2)) (legacy v1 tower, nnlib/v1/layers.py:72-79)."""
    return 0.5 * x * torch.erfc(-x * 0.7071067811865476)


def activation(name: str | None, x):
    if name is None or name == "linear":
        return x
    name = name.lower()
    if name == "gelu":
        return gelu_tanh(x)
    if name == "relu":
        return torch.relu(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "softmax":
        return torch.softmax(x, dim=-1)
    raise ValueError(f"oracle: unsupported activation {name!r}")


def same_pad(length: int, k: int, stride: int, dilation: int) -> tuple[int, int, int]:
    """TF 'SAME': (L_out, pad_left, pad_right)."""
    l_out = -(-length // stride)
    total = max((l_out - 1) * stride + (k - 1) * dilation + 1 - length, 0)
    return l_out, total // 2, total - total // 2


#: False = the checker's default: every op spelled out (k shifted sgemm calls per conv, GELU / batch norm as the
#: reference writes them).  True = the same f32 forward on the fused CPU kernels a TensorFlow-CPU build would run
#: (oneDNN channels-last convolution, fused tanh-GELU, batch norm as one multiply-add with constant-folded
#: scale / shift); used by bench.py's cpu_baseline leg, agrees with the default to rounding
#: (tests/test_oracle_forward.py).
FAST = False


def conv1d_nwc(x, kernel, stride, padding, dilation):
    """tf.nn.conv1d on (N, L, Cin) with a (k, Cin, Cout) kernel: k shifted GEMMs
    (y[:, m] = sum_t x[:, m*s + t*d - pad_left] @ W[t]) or, with ``CONV_IMPL = "conv1d"``, the oneDNN
    convolution behind ``torch.nn.functional.conv1d``."""
    k, cin, cout = kernel.shape
    n, length, _ = x.shape
    if padding == "SAME":
        l_out, pl, pr = same_pad(length, k, stride, dilation)
        x = F.pad(x, (0, 0, pl, pr))
    else:
        span = dilation * (k - 1) + 1
        l_out = (length - span) // stride + 1 if length >= span else 0
    if FAST and l_out > 0 and cin > 1:
        # (N, L, C) contiguous IS the channels-last image of an (N, C, 1, L) tensor: no copies either way
        x4 = x.contiguous().permute(0, 2, 1).unsqueeze(2)
        w4 = kernel.permute(2, 1, 0).unsqueeze(2).contiguous(memory_format=torch.channels_last)
        y = F.conv2d(x4, w4, stride=(1, stride), dilation=(1, dilation))
        return y[:, :, 0, :l_out].permute(0, 2, 1).contiguous()
    y = None
    for t in range(k):
        start = t * dilation
        xs = x[:, start:start + (l_out - 1) * stride + 1:stride, :]
        term = xs.reshape(-1, cin) @ kernel[t]
        y = term if y is None else y + term
    return y.reshape(n, l_out, cout)


def masked_conv1d(x, mask, w: dict, *, kernel_size, strides=1, padding="valid",
                  dilation_rate=1, use_bias=True, act=None, use_masking=True,
                  mask_mode="any"):
    """layers.py:1217-1280.  x (W,6,L,Cin), mask (W,6,L) float 0/1 or None."""
    padding = padding.upper()
    W_, Fr, L, Cin = x.shape
    out_mask = None
    if use_masking and mask is not None:
        x = x * mask.unsqueeze(-1)
        mconv = conv1d_nwc(
            mask.reshape(-1, L, 1),
            torch.ones((kernel_size, 1, 1), dtype=mask.dtype),
            strides, padding, dilation_rate,
        )
        if mask_mode == "any":
            om = mconv > 0
        elif mask_mode == "majority":
            om = mconv >= (kernel_size + 1) // 2
        else:
            om = mconv == float(kernel_size)
        out_mask = om.squeeze(-1).reshape(W_, Fr, -1).to(x.dtype)
    y = conv1d_nwc(x.reshape(-1, L, Cin), w["kernel"], strides, padding, dilation_rate)
    if use_bias:
        y = y + w["bias"]
    y = activation(act, y)
    y = y.reshape(W_, Fr, y.shape[1], y.shape[2])
    return y, out_mask


def masked_batchnorm(x, w, eps=1e-5):
    """layers.py:918-938 inference branch: gamma*((x-mm)*rsqrt(mv+eps))+beta."""
    inv_std = torch.rsqrt(w["moving_variance"] + eps)
    if FAST:
        scale = w["gamma"] * inv_std
        return torch.addcmul(w["beta"] - w["moving_mean"] * scale, x, scale)
    return w["gamma"] * ((x - w["moving_mean"]) * inv_std) + w["beta"]


def masked_dyt(x, mask, w):
    """layers.py:431-444."""
    out = torch.tanh(w["alpha"] * x) * w["gamma"] + w["beta"]
    if mask is not None:
        out = out * mask.unsqueeze(-1)
    return out


def masked_layernorm(x, mask, w, eps=1e-3):
    """layers.py:335-367."""
    if mask is not None:
        x = x * mask.unsqueeze(-1)
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    y = (x - mean) / torch.sqrt(var + eps)
    y = y * w["gamma"] + w["beta"]
    if mask is not None:
        y = y * mask.unsqueeze(-1)
    return y


def nmd_vector(x, mask, moving_mean, eps=1e-5):
    """nmd.py:52-77 inference: masked per-example channel mean - moving_mean."""
    if mask is not None:
        m = mask.unsqueeze(-1)
        mean_channel = (x * m).sum(dim=(1, 2)) / (m.sum(dim=(1, 2)) + eps)
    else:
        mean_channel = x.mean(dim=(1, 2))
    return mean_channel - moving_mean


def masked_global_avg(x, mask):
    """layers.py:460-480."""
    if mask is None:
        return x.mean(dim=(1, 2))
    m = mask.unsqueeze(-1)
    s = (x * m).sum(dim=(1, 2))
    cnt = torch.clamp(m.sum(dim=(1, 2)), min=1e-7)
    return s / cnt


def masked_global_max(x, mask):
    """layers.py:517-529: sentinel -1e9, all-masked sample -> zeros."""
    if mask is None:
        return x.amax(dim=(1, 2))
    m = mask.unsqueeze(-1)
    sentinel = torch.tensor(-1.0e9, dtype=x.dtype)
    pooled = torch.where(m > 0, x, sentinel).amax(dim=(1, 2))
    has_valid = m.amax(dim=(1, 2))
    return torch.where(has_valid > 0, pooled, torch.zeros_like(pooled))


def sinusoidal_position_embedding(x, max_wavelength):
    """SinusoidalPositionEmbedding.call (nnlib/v2/layers.py:2155-2195) for an input [..., seq_length, hidden]: positions
    0 .. seq_length - 1, timescales min_freq ** (2 floor(i / 2) / hidden), sine on the even dims, cosine on the odd ones,
    broadcast over the leading axes (the Add layer that follows keeps the input's mask).  Evaluated in float32 with numpy's
    powf / sinf / cosf, operation by operation as the layer does in its compute dtype: at angles of hundreds of radians
    the LAST BIT of a float32 timescale moves the sine by 1e-5, so the rows are only defined up to the libm in use
    (TensorFlow's Eigen kernels are not available here: this part of the oracle is unpinned like the rest of it)."""
    import numpy as np
    f = np.float32
    seq, hidden = int(x.shape[-2]), int(x.shape[-1])
    positions = np.arange(seq).astype(f)
    min_freq = f(1.0) / f(max_wavelength)
    dims = np.arange(hidden).astype(f)
    even = np.floor(dims / f(2)) * f(2)
    timescales = np.power(min_freq, even / f(hidden), dtype=f)
    angles = np.expand_dims(positions, -1) * np.expand_dims(timescales, 0)
    sin_mask = (dims % f(2) == 0).astype(f)
    enc = np.sin(angles, dtype=f) * sin_mask + np.cos(angles, dtype=f) * (f(1.0) - sin_mask)
    return torch.as_tensor(enc).to(x.dtype).expand(x.shape)


def nmd_merge_dim(merge: dict | None, nmd_dims: list) -> int:
    """Width of NMDMerge's output (nmd.py:120-134): the vectors side by side, or ``target_dim`` (default: the common width)."""
    if len(nmd_dims) < 2 or not merge or merge.get("mode", "concat") == "concat":
        return sum(nmd_dims)
    if merge.get("target_dim") is not None:
        return int(merge["target_dim"])
    if len(set(nmd_dims)) != 1:
        raise ValueError(f"target_dim is required for merge mode '{merge.get('mode')}' when NMD channel dimensions differ.")
    return int(nmd_dims[0])


def nmd_merge(nmds: list, merge: dict | None, weights: dict, dtype):
    """NMDMerge.call (nmd.py:141-155): concat; or every vector through its own bias-free Dense, then add_n / mean / max /
    softmax(layer_weights)-weighted sum - in that order of operations.  builder.py:1176-1184: no merge config = Concatenate."""
    mode = (merge or {}).get("mode", "concat")
    if mode == "concat":
        return torch.cat(nmds, dim=-1)
    if mode not in ("sum", "mean", "max", "weighted"):
        raise ValueError(f"Unsupported NMD merge mode: {mode}")
    # Dense(target_dim, use_bias=False, **projection_kwargs) (nmd.py:133-141): the only graph-changing keyword is `activation`
    act = ((merge or {}).get("projection_kwargs") or {}).get("activation")
    projected = [activation(act, v @ torch.as_tensor(weights[f"rep/nmd_merge/proj_{i}/kernel"]).to(dtype)) for i, v in enumerate(nmds)]
    if mode == "sum":
        return sum(projected[1:], projected[0])
    if mode == "mean":
        return sum(projected[1:], projected[0]) / len(projected)
    stacked = torch.stack(projected, dim=0)
    if mode == "max":
        return stacked.max(dim=0).values
    w = torch.softmax(torch.as_tensor(weights["rep/nmd_merge/layer_weights"]).to(dtype), dim=0).reshape(-1, 1, 1)
    return (stacked * w).sum(dim=0)


def ood_signals(logits, nmd, signals, eps=1e-10):
    """layers.py:1632-1667."""
    probs = torch.softmax(logits, dim=-1)
    cols = []
    for s in signals:
        if s == "max_prob":
            cols.append(probs.amax(dim=-1, keepdim=True))
        elif s == "entropy":
            sp = torch.clamp(probs, min=eps)
            cols.append(-(sp * torch.log(sp)).sum(dim=-1, keepdim=True))
        elif s == "energy":
            cols.append(torch.logsumexp(logits, dim=-1, keepdim=True))
        elif s == "margin":
            top2 = torch.topk(probs, 2, dim=-1).values
            cols.append(top2[..., 0:1] - top2[..., 1:2])
        elif s == "nmd_norm":
            cols.append(torch.linalg.vector_norm(nmd, dim=-1, keepdim=True))
        else:
            raise ValueError(f"oracle: unsupported signal {s!r}")
    return torch.cat(cols, dim=-1)


# --------------------------------------------------------------------------
# weight naming + random initialisation (the stand-in for absent checkpoints)
# --------------------------------------------------------------------------
def _norm_vars(norm_type: str, c: int) -> dict[str, tuple]:
    if norm_type == "masked_batchnorm":
        return {"gamma": (c,), "beta": (c,), "moving_mean": (c,), "moving_variance": (c,)}
    if norm_type == "masked_dyt":
        return {"alpha": (1,), "gamma": (c,), "beta": (c,)}
    if norm_type == "masked_layernorm":
        return {"gamma": (c,), "beta": (c,)}
    raise ValueError(norm_type)


def onehot_depth(model_cfg: dict) -> int:
    """max(codon_id) + 1 (seqops/encode.py:297-302).  The id maps come from ``tests/golden/maps.json`` - the vectors
    ``tests/golden/make_golden.py`` dumped from the reference's ``seqops/maps.py`` - not from the product's ``jaeger_amd.maps``:
    the checker's one-hot depth must not depend on the code under test."""
    import json
    from pathlib import Path
    name = (model_cfg.get("string_processor", {}) or {}).get("codon_id", "CODON_ID")
    if name == "DICODON_ID":
        return 4096
    table = json.loads((Path(__file__).resolve().parents[1] / "tests" / "golden" / "maps.json").read_text())
    return max(table[name]) + 1


def vocab_size(model_cfg: dict) -> int:
    """len(codon_id)+1 (inference.py:449): the codon id maps have 64 entries, DICODON_ID 4 096."""
    name = (model_cfg.get("string_processor", {}) or {}).get("codon_id", "CODON_ID")
    return 4097 if name == "DICODON_ID" else 65


def _block_specs(prefix: str, layers: list[dict], cin: int, specs: dict,
                 nmd_dims: list | None = None) -> int:
    for i, layer in enumerate(layers):
        name = layer.get("name", "").lower()
        cfg = dict(layer.get("config", {}) or {})
        p = f"{prefix}/{i}"
        if name == "masked_conv1d":
            k, co = cfg["kernel_size"], cfg["filters"]
            specs[f"{p}/kernel"] = (k, cin, co)
            if cfg.get("use_bias", True):
                specs[f"{p}/bias"] = (co,)
            cin = co
        elif name in ("masked_batchnorm", "masked_dyt", "masked_layernorm"):
            for v, shp in _norm_vars(name, cin).items():
                specs[f"{p}/{v}"] = shp
            if cfg.get("return_nmd") and nmd_dims is not None:
                nmd_dims.append(cin)
        elif name == "nmd":
            specs[f"{p}/moving_mean"] = (cin,)
            if nmd_dims is not None:
                nmd_dims.append(cin)
        elif name == "residual_block":
            co = cfg["filters"]
            k = cfg.get("kernel_size", 3)
            stride = cfg.get("strides", 1)
            nt = cfg.get("norm_type", "masked_batchnorm").lower()
            bias = cfg.get("use_bias", True)
            for j in range(cfg.get("block_size", 1)):
                bp = f"{p}/block{j}"
                bypass = (cfg.get("use_1x1conv", False) and j == 0) or stride > 1
                specs[f"{bp}/conv1/kernel"] = (k, cin, co)
                specs[f"{bp}/conv2/kernel"] = (k, co, co)
                if bias:
                    specs[f"{bp}/conv1/bias"] = (co,)
                    specs[f"{bp}/conv2/bias"] = (co,)
                for bn in ("bn1", "bn2"):
                    for v, shp in _norm_vars(nt, co).items():
                        specs[f"{bp}/{bn}/{v}"] = shp
                if bypass:
                    specs[f"{bp}/conv3/kernel"] = (1, cin, co)
                    if bias:
                        specs[f"{bp}/conv3/bias"] = (co,)
                    for v, shp in _norm_vars(nt, co).items():
                        specs[f"{bp}/bn3/{v}"] = shp
                cin = co
            if cfg.get("return_nmd") and nmd_dims is not None:
                nmd_dims.append(co)
        elif name == "dense":
            specs[f"{p}/kernel"] = (cin, cfg["units"])
            if cfg.get("use_bias", True):
                specs[f"{p}/bias"] = (cfg["units"],)
            cin = cfg["units"]
        elif name in ("activation", "dropout") or name in _ACT_ALIASES:
            pass
        else:
            raise ValueError(f"oracle: unsupported layer {name!r}")
    return cin


def weight_specs(model_cfg: dict) -> dict[str, tuple]:
    """Canonical variable names -> shapes for a ``model:`` config section."""
    specs: dict[str, tuple] = {}
    emb = model_cfg["embedding"]
    e = emb.get("embedding_size", 4)
    if emb.get("use_embedding_layer", False):
        specs["embedding/embeddings"] = (vocab_size(model_cfg), e)
        cin = e
    else:                                   # one-hot rows -> Masking -> bias-free Dense(E), or straight through (E = 0)
        depth = onehot_depth(model_cfg)
        if e > 0:
            specs["embedding/kernel"] = (depth, e)
        cin = e if e > 0 else depth
    nmd_dims: list[int] = []
    rep_out = _block_specs("rep", model_cfg["representation_learner"]["hidden_layers"], cin,
                           specs, nmd_dims)
    if "classifier" in model_cfg:
        _block_specs("classifier", model_cfg["classifier"]["hidden_layers"], rep_out, specs)
    if "reliability_model" in model_cfg and nmd_dims:
        rel = model_cfg["reliability_model"]
        n_sig = 0
        if rel.get("mode", "nmd") == "nmd_plus_signals":
            n_sig = len(rel.get("signals", ["max_prob", "entropy", "energy", "margin", "nmd_norm"]))
        merge = rel.get("merge")
        merged = nmd_merge_dim(merge, nmd_dims)
        if len(nmd_dims) > 1 and merge and merge.get("mode", "concat") != "concat":
            for i, d in enumerate(nmd_dims):
                specs[f"rep/nmd_merge/proj_{i}/kernel"] = (d, merged)
            if merge["mode"] == "weighted":
                specs["rep/nmd_merge/layer_weights"] = (len(nmd_dims),)
        _block_specs("reliability", rel["hidden_layers"], merged + n_sig, specs)
    return specs


def random_weights(model_cfg: dict, seed: int = 38341) -> dict[str, np.ndarray]:
    """Seeded stand-in weights: He-uniform
    kernels, BN gamma~U[0.5,1.5], beta/mu~N(0,0.1), var~U[0.5,1.5]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for name, shp in sorted(weight_specs(model_cfg).items()):
        leaf = name.rsplit("/", 1)[1]
        if leaf == "kernel":
            fan_in = int(np.prod(shp[:-1]))
            lim = math.sqrt(6.0 / fan_in)
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


# --------------------------------------------------------------------------
# graph walk
# --------------------------------------------------------------------------
def _sub(weights: dict, prefix: str, dtype) -> dict:
    n = len(prefix) + 1
    return {k[n:]: torch.as_tensor(v).to(dtype) for k, v in weights.items()
            if k.startswith(prefix + "/") and "/" not in k[n:]}


def _norm(norm_type, x, mask, w, use_masking=True):
    if norm_type == "masked_batchnorm":
        return masked_batchnorm(x, w)          # mask unused at inference
    if norm_type == "masked_dyt":
        return masked_dyt(x, mask, w)
    if norm_type == "masked_layernorm":
        return masked_layernorm(x, mask, w)
    raise ValueError(norm_type)


def _residual_block(x, mask, weights, bp, cfg, first: bool, use_masking, dtype, nmd_out=None):
    """layers.py:1882-1915.  ``nmd_out`` (a list): bn2 runs with return_nmd=True and appends its side output."""
    k = cfg.get("kernel_size", 3)
    stride = cfg.get("strides", 1)
    d = cfg.get("dilation_rate", 1)
    pad = cfg.get("padding", "same")
    bias = cfg.get("use_bias", True)
    nt = cfg.get("norm_type", "masked_batchnorm").lower()
    act = cfg.get("activation", "gelu")
    conv = dict(kernel_size=k, padding=pad, dilation_rate=d, use_bias=bias,
                use_masking=use_masking)
    h, m1 = masked_conv1d(x, mask, _sub(weights, f"{bp}/conv1", dtype), strides=stride, **conv)
    m1 = m1 if use_masking else None
    h = activation(act, _norm(nt, h, m1, _sub(weights, f"{bp}/bn1", dtype)))
    h, m2 = masked_conv1d(h, m1, _sub(weights, f"{bp}/conv2", dtype), strides=1, **conv)
    m2 = m2 if use_masking else None
    if nmd_out is not None:                      # MaskedBatchNorm(return_nmd=True), layers.py:943-954
        nmd_out.append(nmd_vector(h, m2, _sub(weights, f"{bp}/bn2", dtype)["moving_mean"]))
    h = _norm(nt, h, m2, _sub(weights, f"{bp}/bn2", dtype))
    if (cfg.get("use_1x1conv", False) and first) or stride > 1:
        c3 = dict(conv, kernel_size=1)
        sc, m3 = masked_conv1d(x, mask, _sub(weights, f"{bp}/conv3", dtype), strides=stride, **c3)
        sc = _norm(nt, sc, m3 if use_masking else None, _sub(weights, f"{bp}/bn3", dtype))
    else:
        sc = x
    out = activation(act, h + sc)               # MaskedAdd = add_n, mask = mask[0]
    return out, m2


def _run_block(x, mask, layers, prefix, weights, model_cfg, dtype, pooling=None):
    """builder.py:982-1193 (+ pooling :1160-1182)."""
    use_masking_default = bool(model_cfg.get("use_masking", True))
    nmds = []
    for i, layer in enumerate(layers):
        name = layer.get("name", "").lower()
        cfg = dict(layer.get("config", {}) or {})
        p = f"{prefix}/{i}"
        if name in ("masked_conv1d", "masked_batchnorm", "residual_block"):
            cfg.setdefault("use_masking", use_masking_default)
        if name == "masked_conv1d":
            x, om = masked_conv1d(
                x, mask, _sub(weights, p, dtype),
                kernel_size=cfg["kernel_size"], strides=cfg.get("strides", 1),
                padding=cfg.get("padding", "valid"), dilation_rate=cfg.get("dilation_rate", 1),
                use_bias=cfg.get("use_bias", True), act=cfg.get("activation"),
                use_masking=cfg["use_masking"], mask_mode=cfg.get("mask_mode", "any"))
            # a masking-disabled conv forwards no mask (supports_masking False)
            mask = om if cfg["use_masking"] else None
        elif name in ("masked_batchnorm", "masked_dyt", "masked_layernorm"):
            if cfg.get("return_nmd"):
                if name != "masked_batchnorm":
                    raise ValueError("return_nmd=True is only defined for masked_batchnorm")   # layers.py:307, 398
                # layers.py:943-954: masked mean of the norm's input minus its moving_mean (eps = the norm's 1e-5)
                nmds.append(nmd_vector(x, mask, _sub(weights, p, dtype)["moving_mean"]))
            x = _norm(name, x, mask, _sub(weights, p, dtype))
            if name == "masked_batchnorm" and not cfg["use_masking"]:
                mask = None
        elif name == "nmd":
            nmds.append(nmd_vector(x, mask, _sub(weights, p, dtype)["moving_mean"]))
        elif name == "activation" or name in _ACT_ALIASES:
            x = activation(name if name in _ACT_ALIASES else cfg.get("activation"), x)
        elif name == "residual_block":
            um = cfg["use_masking"]
            nb = cfg.get("block_size", 1)
            for j in range(nb):                 # return_nmd only reaches the last block of a stack (layers.py:2682-2686)
                x, mask = _residual_block(x, mask if um else None, weights, f"{p}/block{j}", cfg, j == 0, um, dtype,
                                          nmd_out=nmds if (cfg.get("return_nmd") and j == nb - 1) else None)
        elif name == "dense":
            w = _sub(weights, p, dtype)
            x = x @ w["kernel"]
            if cfg.get("use_bias", True):
                x = x + w["bias"]
            x = activation(cfg.get("activation"), x)
        elif name == "dropout":
            pass
        else:
            raise ValueError(f"oracle: unsupported layer {name!r}")
    if pooling is not None:
        pooling = pooling.lower()
        if pooling in ("max", "masked_max"):
            x = masked_global_max(x, mask)
        elif pooling in ("average", "masked_average"):
            x = masked_global_avg(x, mask)
        else:
            raise ValueError(f"oracle: unsupported pooling {pooling!r}")
    return x, nmds


def forward(model_cfg: dict, weights: dict[str, Any], ids: np.ndarray,
            dtype=torch.float32) -> dict[str, np.ndarray]:
    """ids (W, 6, L) in 0..vocab-1 (0 = invalid / pad) -> dict of outputs keyed like
    the SavedModel (builder.py:796-836): prediction, embedding[, nmd, reliability]."""
    idt = torch.as_tensor(np.asarray(ids).astype(np.int64))
    emb_cfg = model_cfg["embedding"]
    if emb_cfg.get("use_embedding_layer", False):
        table = torch.as_tensor(weights["embedding/embeddings"]).to(dtype)
        x = table[idt]                                     # Embedding(mask_zero=True)
        mask = (idt != 0).to(dtype)                        # builder.py:858-867
    else:
        # seq_onehot=True (builder.py:844-880): the encoder emits one_hot(codon_id, depth) with an all-zero row for an
        # invalid codon (encode.py:297-302); device ids are codon_id + 1, so the row is one_hot(id - 1); Masking(0.0)
        # masks rows that are entirely zero; Dense(E, use_bias=False) (or nothing for embedding_size 0) follows
        depth = onehot_depth(model_cfg)
        onehot = torch.nn.functional.one_hot(torch.clamp(idt - 1, min=0), depth).to(dtype) * (idt != 0).unsqueeze(-1)
        mask = (onehot != 0).any(dim=-1).to(dtype)
        x = onehot @ torch.as_tensor(weights["embedding/kernel"]).to(dtype) if "embedding/kernel" in weights else onehot
    if model_cfg["embedding"].get("use_positional_embeddings", False):            # builder.py:886-892
        x = x + sinusoidal_position_embedding(x, model_cfg["embedding"].get("positional_embedding_length"))
    rep = model_cfg["representation_learner"]
    emb, nmds = _run_block(x, mask, rep["hidden_layers"], "rep", weights, model_cfg, dtype,
                           pooling=rep.get("pooling"))
    out = {"embedding": emb}
    logits, _ = _run_block(emb, None, model_cfg["classifier"]["hidden_layers"], "classifier",
                           weights, model_cfg, dtype)
    out["prediction"] = logits
    if nmds:
        nmd = nmds[0] if len(nmds) == 1 else nmd_merge(nmds, (model_cfg.get("reliability_model") or {}).get("merge"), weights, dtype)
        out["nmd"] = nmd
        rel_cfg = model_cfg.get("reliability_model")
        if rel_cfg is not None:
            rin = nmd
            if rel_cfg.get("mode", "nmd") == "nmd_plus_signals":
                sig = rel_cfg.get("signals", ["max_prob", "entropy", "energy", "margin", "nmd_norm"])
                rin = torch.cat([nmd, ood_signals(logits, nmd, sig)], dim=-1)
            rel, _ = _run_block(rin, None, rel_cfg["hidden_layers"], "reliability", weights,
                                model_cfg, dtype)
            out["reliability"] = rel
    return {k: v.detach().to(torch.float32).numpy() if dtype == torch.float32
            else v.detach().numpy() for k, v in out.items()}
