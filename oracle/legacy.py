"""TEST INFRASTRUCTURE - CPU restatement of the reference's legacy ``default`` model forward
(``nnlib/v1/layers.py:154-207`` ConvolutionalTower, ``:399-423`` WRes_model_embeddings) and of its
v1 window encoder (``preprocess/v1/convert.py:56-125``).  Only tests / smoke / the bench's CPU
baseline may import this.

Pinning: the encoder's id table is pinned against the reference's ``preprocess/v1/maps.py``
(tests/golden/maps.json: TRIMER_INT == AA_ID); the float forward is **parity unpinned** (no
TensorFlow here) but runs the reference's own shipped weights (``WRes_1024.h5``).

Semantics restated (Keras 2.5 graph recorded in the weight file's ``layer_names`` attribute):
Embedding(22, 4) - its mask is dropped by Conv1D (``supports_masking`` False) so id 0 contributes
its learned row; per frame, shared weights: Conv1D(k9, same) -> exact-erf GELU (``tf.nn.gelu``
default) -> BatchNormalization(eps 1e-3, Keras default) -> MaxPool(2); Conv1D(k5, d2) -> GELU ->
BN -> MaxPool(2); five blocks of [Conv1D(k5, d=3+i) -> GELU -> BN] x 2 followed by one more GELU
(``add_residual=False``); sum over the six frames; GlobalMaxPool1D; Dense(128, gelu) x 2
(the second is the ``embedding`` output); Dense(4) = ``output``.
"""

from __future__ import annotations

import numpy as np
import torch

from .forward import conv1d_nwc, gelu_erf

BN_EPS = 1e-3


def tower_layers():
    """(conv name, bn name, kernel, dilation, pool_after, extra_gelu_after) in graph order."""
    rows = [("block1_0", "bn_block1_1", 9, 1, True, False), ("block1_1", "bn_block1_2", 5, 2, True, False)]
    for n in range(5):
        rows.append((f"block2_{n}1", f"bn_block2_{n}1", 5, 3 + n, False, False))
        rows.append((f"block2_{n}2", f"bn_block2_{n}2", 5, 3 + n, False, True))
    return rows


def forward(weights: dict[str, np.ndarray], ids: np.ndarray, dtype=torch.float32) -> dict[str, np.ndarray]:
    """ids (W, 6, L) amino-acid ids 0..21 -> {"output": (W, 4), "embedding": (W, 128)}."""
    w = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in weights.items()}
    idt = torch.as_tensor(np.asarray(ids).astype(np.int64))
    W_, Fr, L = idt.shape
    x = w["aa/embeddings"][idt].reshape(W_ * Fr, L, -1)
    for conv, bn, k, d, pool, extra in tower_layers():
        x = conv1d_nwc(x, w[f"{conv}/kernel"], 1, "SAME", d) + w[f"{conv}/bias"]
        x = gelu_erf(x)
        x = (x - w[f"{bn}/moving_mean"]) * torch.rsqrt(w[f"{bn}/moving_variance"] + BN_EPS) * w[f"{bn}/gamma"] \
            + w[f"{bn}/beta"]
        if pool:
            n = x.shape[1] // 2
            x = torch.maximum(x[:, 0:2 * n:2, :], x[:, 1:2 * n:2, :])
        if extra:
            x = gelu_erf(x)
    x = x.reshape(W_, Fr, x.shape[1], x.shape[2]).sum(dim=1)          # Add() over the six frames
    x = x.max(dim=1).values                                            # GlobalMaxPool1D
    x = gelu_erf(x @ w["augdense-1/kernel"] + w["augdense-1/bias"])
    emb = gelu_erf(x @ w["augdense-2/kernel"] + w["augdense-2/bias"])
    out = emb @ w["outdense/kernel"] + w["outdense/bias"]
    return {"output": out.numpy().astype(np.float32), "embedding": emb.numpy().astype(np.float32)}
