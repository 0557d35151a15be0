"""Model configs / weights / inputs of the reference's layer known-answer tests, restated as whole (tiny) models so that
the same case runs through the CPU oracle (tests/test_oracle_forward.py) and through ``JaegerHipEngine`` on the GPU
(tests/test_gpu_reference_kats.py).  Data only - no reference code:

* ``tests/unit/test_masked_pooling.py:186-209``  Embedding(mask_zero) -> MaskedBatchNorm(moving mean 3, variance 2) ->
  masked max pool: a right-padded batch equals the truncated one (atol 1e-5)
* ``tests/unit/test_nnlib_v2_nmd.py:32-56``      NMDLayer == MaskedBatchNorm(return_nmd=True) side output under a random
  mask (two fully valid examples, two with 70 % valid positions), max diff < 1e-5
* ``tests/unit/test_ood_signal_layer.py:20-106`` the five OOD signal formulas on logits [[1,2,3],[.5,-1,1.5]] and
  nmd [[3,4],[0,-5]], rtol 1e-6
"""
import numpy as np

_SP = {"data_format": "numpy", "seq_onehot": False, "codon": "CODON", "codon_id": "CODON_ID", "crop_size": 100}


def _base(e: int, rep_layers: list, pooling: str, n_out: int, identity_head: bool = True) -> dict:
    return {
        "name": "kat", "classifier_out_dim": n_out,
        "class_label_map": [{"class": f"c{i}", "label": i} for i in range(n_out)],
        "embedding": {"use_embedding_layer": True, "input_type": "translated", "strands": 2, "frames": 6,
                      "input_shape": [6, None], "embedding_size": e},
        "string_processor": dict(_SP),
        "representation_learner": {"hidden_layers": rep_layers, "pooling": pooling},
        "classifier": {"input_shape": e, "hidden_layers": [
            {"name": "dense", "config": {"units": n_out, "activation": None, "use_bias": not identity_head}}]},
    }


# ---- Embedding -> BN -> masked max pool: padded == truncated ---------------------------------------------------------
def padded_pooling_case(seed: int = 1):
    dim, vocab, valid, length = 8, 33, 20, 32
    cfg = _base(dim, [{"name": "masked_batchnorm", "config": {}}], "max", dim)
    rng = np.random.default_rng(seed)
    w = {
        "embedding/embeddings": rng.uniform(-0.05, 0.05, (65, dim)).astype(np.float32),     # Keras "uniform" initialiser
        "rep/0/gamma": np.ones(dim, np.float32), "rep/0/beta": np.zeros(dim, np.float32),
        "rep/0/moving_mean": np.full(dim, 3.0, np.float32), "rep/0/moving_variance": np.full(dim, 2.0, np.float32),
        "classifier/0/kernel": np.eye(dim, dtype=np.float32),
    }
    w["embedding/embeddings"][0] = 0.06          # row 0 (the padding id) above every other row: it wins any unmasked maximum
    full = rng.integers(1, vocab, size=(2, 6, length))
    padded = full.copy()
    padded[:, :, valid:] = 0
    return cfg, w, padded, full[:, :, :valid]


# ---- NMDLayer vs MaskedBatchNorm(return_nmd) ---------------------------------------------------------------------
def nmd_vs_bn_case(seed: int = 42, zero_mean: bool = True):
    dim, length = 8, 32
    rng = np.random.default_rng(seed)
    table = rng.normal(size=(65, dim)).astype(np.float32)
    mm = np.zeros(dim, np.float32) if zero_mean else rng.normal(size=dim).astype(np.float32)
    bn = {"gamma": np.ones(dim, np.float32), "beta": np.zeros(dim, np.float32), "moving_mean": mm,
          "moving_variance": np.ones(dim, np.float32)}
    cfg_nmd = _base(dim, [{"name": "nmd", "config": {"epsilon": 1e-5}}, {"name": "masked_batchnorm", "config": {}}], "max", dim)
    cfg_bn = _base(dim, [{"name": "masked_batchnorm", "config": {"return_nmd": True, "epsilon": 1e-5}}], "max", dim)
    for c in (cfg_nmd, cfg_bn):          # an NMD output needs a reliability head to be exported (builder.py:589-613)
        c["reliability_model"] = {"mode": "nmd", "hidden_layers": [
            {"name": "dense", "config": {"units": 1, "activation": None}}]}
    head = {"classifier/0/kernel": np.eye(dim, dtype=np.float32), "embedding/embeddings": table,
            "reliability/0/kernel": rng.normal(size=(dim, 1)).astype(np.float32),
            "reliability/0/bias": np.zeros(1, np.float32)}
    w_nmd = dict(head, **{"rep/0/moving_mean": mm}, **{f"rep/1/{k}": v for k, v in bn.items()})
    w_bn = dict(head, **{f"rep/0/{k}": v for k, v in bn.items()})
    ids = rng.integers(1, 65, size=(4, 6, length))
    ids[2:][rng.random((2, 6, length)) >= 0.7] = 0          # two fully valid examples, two with ~70 % valid positions
    return (cfg_nmd, w_nmd), (cfg_bn, w_bn), ids


# ---- OODSignalLayer ---------------------------------------------------------------------------------------------------
SIGNALS = ["max_prob", "entropy", "energy", "margin", "nmd_norm"]
KAT_LOGITS = np.array([[1.0, 2.0, 3.0], [0.5, -1.0, 1.5]])
KAT_NMD = np.array([[3.0, 4.0], [0.0, -5.0]])


def ood_expected(logits, nmd, eps: float = 1e-10) -> np.ndarray:
    """The five formulas of OODSignalLayer.call (nnlib/v2/layers.py:1632-1667), written out in f64 numpy."""
    z = np.asarray(logits, np.float64)
    p = np.exp(z - z.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    sp = np.maximum(p, eps)
    top = np.sort(p, axis=-1)[:, ::-1]
    return np.stack([p.max(-1), -(sp * np.log(sp)).sum(-1),
                     z.max(-1) + np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1)),
                     top[:, 0] - top[:, 1], np.sqrt((np.asarray(nmd, np.float64) ** 2).sum(-1))], axis=-1)


def ood_signal_case():
    """A model whose logits are KAT_LOGITS and whose NMD vector is KAT_NMD (+ two zero channels: the engine's convs take
    channel counts in multiples of 4, and zeros change neither the norm nor the logits) for two constant-id windows, with
    an identity reliability head: ``reliability`` = [nmd (4) | the five signals]."""
    cfg = _base(4, [{"name": "nmd", "config": {}}], "max", 3, identity_head=False)
    cfg["reliability_model"] = {"mode": "nmd_plus_signals", "signals": list(SIGNALS), "hidden_layers": [
        {"name": "dense", "config": {"units": 9, "activation": None, "use_bias": False}}]}
    table = np.zeros((65, 4), np.float32)
    table[1, :2], table[2, :2] = KAT_NMD[0], KAT_NMD[1]      # window w is all id w + 1: its rows are constant
    # logits = pooled @ K + b with pooled = KAT_NMD rows: K = KAT_NMD^-1 @ KAT_LOGITS (f64 solve, rounded to f32)
    kernel = np.zeros((4, 3))
    kernel[:2] = np.linalg.solve(KAT_NMD, KAT_LOGITS)
    w = {"embedding/embeddings": table, "rep/0/moving_mean": np.zeros(4, np.float32),
         "classifier/0/kernel": kernel.astype(np.float32), "classifier/0/bias": np.zeros(3, np.float32),
         "reliability/0/kernel": np.eye(9, dtype=np.float32)}
    ids = np.stack([np.full((6, 24), 1), np.full((6, 24), 2)])
    return cfg, w, ids


# ---- MaskedConv1D mask_mode: the reference's boolean answers, read through a pooled indicator model -----------------
# tests/unit/test_mask_mode.py:41-98: a k = 5 valid conv over 20 positions with Ns at given places; the OUTPUT MASK per mode.
MASK_MODE_KATS = [
    # (name, N positions, mode, the 16 expected output-mask booleans)
    ("isolated_any", [10], "any", [True] * 16),
    ("isolated_strict", [10], "strict", [not 6 <= i <= 10 for i in range(16)]),
    ("isolated_majority", [10], "majority", [True] * 16),
    ("short_run_any", [9, 10, 11], "any", [True] * 16),
    ("long_run_any", [9, 10, 11, 12, 13], "any", [i != 9 for i in range(16)]),
    ("long_run_strict", [9, 10, 11, 12, 13], "strict", [not 5 <= i <= 13 for i in range(16)]),
    ("right_padding_any", list(range(10, 20)), "any", [i < 10 for i in range(16)]),
    ("right_padding_strict", list(range(10, 20)), "strict", [i < 6 for i in range(16)]),
]
MASK_PROBE_BIAS = 8.0


def mask_mode_case(n_pos: list, mode: str, deep: bool = False):
    """A model whose pooled output IS the conv's output mask.  Position p carries id p + 1 (0 where the reference puts an N)
    and the embedding row of id p + 1 is the unit vector e_p, so the conv sees WHERE it is.  Output channel c of the k = 5
    valid conv has weight -1 on every (tap t, input channel j != c + t) and 0 on the diagonal j = c + t, bias 8:

        y[i][c] = 8 - (number of valid inputs under window i) for i != c,      y[c][c] = 8.

    A window without a valid input is masked in every mode, so the masked max pool returns 8 in channel c exactly when output
    position c is valid (<= 7 when it is not, 0 when no output position is).  ``deep``: an identity k = 1 conv in front, so
    that the probed conv is not the model's first layer (the first conv on ids runs as a table lookup on the GPU)."""
    length, k, n_out, e = 20, 5, 16, 20
    layers = [{"name": "masked_conv1d", "config": {"filters": n_out, "kernel_size": k, "padding": "valid", "mask_mode": mode}}]
    if deep:
        layers.insert(0, {"name": "masked_conv1d", "config": {"filters": e, "kernel_size": 1, "padding": "valid",
                                                              "use_bias": False}})
    cfg = _base(e, layers, "max", n_out)
    cfg["classifier"]["input_shape"] = n_out
    table = np.zeros((65, e), np.float32)
    table[0] = 9.0                                          # the padding row: a conv that did not zero masked inputs would see it
    table[np.arange(1, length + 1), np.arange(length)] = 1.0
    kernel = -np.ones((k, e, n_out), np.float32)
    for c in range(n_out):
        for t in range(k):
            kernel[t, c + t, c] = 0.0
    w = {"embedding/embeddings": table, "classifier/0/kernel": np.eye(n_out, dtype=np.float32)}
    p = "rep/1" if deep else "rep/0"
    w[f"{p}/kernel"], w[f"{p}/bias"] = kernel, np.full(n_out, MASK_PROBE_BIAS, np.float32)
    if deep:
        w["rep/0/kernel"] = np.eye(e, dtype=np.float32)[None]
    row = np.arange(1, length + 1)
    row[n_pos] = 0
    ids = np.broadcast_to(row, (1, 6, length)).copy()
    return cfg, w, ids


def mask_from_pooled(pooled: np.ndarray) -> np.ndarray:
    """The 16 booleans the model of :func:`mask_mode_case` encodes; anything but 8 / <= 7 is a failure of the probe."""
    pooled = np.asarray(pooled, np.float64).reshape(-1)
    assert ((np.abs(pooled - MASK_PROBE_BIAS) < 1e-3) | (pooled < MASK_PROBE_BIAS - 0.999)).all(), pooled
    return np.abs(pooled - MASK_PROBE_BIAS) < 1e-3


# ---- MaskedGlobalAvgPooling ignores padding: [[2, 3], [5, 6]] ---------------------------------------------------------
def masked_average_case():
    """tests/unit/test_nnlib_v2_layers_short_fragment.py:111-127: rows [1,2],[3,4],pad -> [2,3]; [5,6],pad,pad -> [5,6]
    (rtol 1e-5).  Here the rows are embedding rows of ids 1..3 (two zero channels more: channel counts come in fours), the
    padding is id 0 whose row is made huge, and every one of the six frames repeats the row."""
    cfg = _base(4, [], "average", 4)
    table = np.zeros((65, 4), np.float32)
    table[0] = 1000.0
    table[1, :2], table[2, :2], table[3, :2] = (1.0, 2.0), (3.0, 4.0), (5.0, 6.0)
    w = {"embedding/embeddings": table, "classifier/0/kernel": np.eye(4, dtype=np.float32)}
    ids = np.stack([np.broadcast_to([1, 2, 0], (6, 3)), np.broadcast_to([3, 0, 0], (6, 3))])
    return cfg, w, ids, np.array([[2.0, 3.0], [5.0, 6.0]])


# ---- MaskedDYT re-zeroes masked positions -----------------------------------------------------------------------------
def dyt_zeroes_masked_case(seed: int = 5):
    """tests/unit/test_resblock_norm_type.py:74-81: MaskedDYT over 32 positions of which the last 16 are masked returns
    zeros there (atol 1e-5).  Read through a model: Embedding -> masked_dyt (beta != 0, so an un-zeroed masked position
    would hold tanh(alpha E[0]) gamma + beta) -> an identity k = 1 conv with use_masking False, which forwards NO mask
    (supports_masking False) -> the average pool, now a plain mean over all 32 positions: it equals the sum over the 16
    valid positions / 32 exactly when the masked positions hold zeros."""
    dim, length, valid = 8, 32, 16
    rng = np.random.default_rng(seed)
    cfg = _base(dim, [{"name": "masked_dyt", "config": {}},
                      {"name": "masked_conv1d", "config": {"filters": dim, "kernel_size": 1, "padding": "valid",
                                                           "use_bias": False, "use_masking": False}}], "average", dim)
    table = rng.normal(size=(65, dim)).astype(np.float32)
    table[0] = 3.0                                          # the padding row, far from zero
    alpha, gamma, beta = np.float32(0.5), rng.uniform(0.5, 1.5, dim).astype(np.float32), rng.uniform(0.3, 0.9, dim).astype(np.float32)
    w = {"embedding/embeddings": table, "rep/0/alpha": np.array([alpha]), "rep/0/gamma": gamma, "rep/0/beta": beta,
         "rep/1/kernel": np.eye(dim, dtype=np.float32)[None], "classifier/0/kernel": np.eye(dim, dtype=np.float32)}
    row = rng.integers(1, 65, length)
    row[valid:] = 0
    ids = np.broadcast_to(row, (1, 6, length)).copy()
    x = table.astype(np.float64)[row[:valid]]
    dyt = np.tanh(0.5 * x) * gamma.astype(np.float64) + beta.astype(np.float64)
    want = dyt.sum(axis=0) / length
    not_zeroed = want + (length - valid) * (np.tanh(0.5 * table[0].astype(np.float64)) * gamma + beta) / length
    return cfg, w, ids, want[None], not_zeroed[None]


# ---- ResidualBlock(norm_type=masked_layernorm): masked == truncated away from the boundary ---------------------------
def layernorm_block_masked_vs_truncated_case(pooling: str, seed: int = 11):
    """tests/unit/test_resblock_norm_type.py:84-94: a k = 5 LayerNorm residual block on 32 positions whose last 16 are
    masked equals the same block on the first 16 positions alone, on output positions 0..9 (atol 1e-4; nearer the boundary
    the windows legitimately differ).  The pooled twin: behind the block, first-tap-identity VALID convs in strict mode
    erode the valid region from the right only - two 6-tap ones take the masked run's [0, 20) (the 'any' rule grew the 16
    valid positions by 2 per conv) to [0, 10), one 7-tap one takes the truncated run's [0, 16) to [0, 10) - so that each
    model's own pool (mean or max over positions 0..9 and the six identical frames) reads exactly the positions the
    reference compares.  Returns (cfg, weights, ids) of the masked and of the truncated run."""
    c, length, valid = 16, 32, 16
    rng = np.random.default_rng(seed)
    block = {"name": "residual_block", "config": {"filters": c, "kernel_size": 5, "strides": 1, "activation": "gelu",
                                                  "norm_type": "masked_layernorm", "block_size": 1}}

    def selector(k):
        return {"name": "masked_conv1d", "config": {"filters": c, "kernel_size": k, "padding": "valid", "use_bias": False,
                                                    "mask_mode": "strict"}}

    def first_tap_identity(k):
        w = np.zeros((k, c, c), np.float32)
        w[0] = np.eye(c, dtype=np.float32)
        return w
    lim = np.sqrt(6.0 / (5 * c + 5 * c))                                    # glorot_uniform, the layer's default
    shared = {"embedding/embeddings": rng.normal(size=(65, c)).astype(np.float32),
              "classifier/0/kernel": np.eye(c, dtype=np.float32)}
    for conv in ("conv1", "conv2"):
        shared[f"rep/0/block0/{conv}/kernel"] = rng.uniform(-lim, lim, (5, c, c)).astype(np.float32)
        shared[f"rep/0/block0/{conv}/bias"] = np.zeros(c, np.float32)
    for bn in ("bn1", "bn2"):
        shared[f"rep/0/block0/{bn}/gamma"] = np.ones(c, np.float32)
        shared[f"rep/0/block0/{bn}/beta"] = np.zeros(c, np.float32)
    cfg_m = _base(c, [block, selector(6), selector(6)], pooling, c)
    w_m = dict(shared, **{"rep/1/kernel": first_tap_identity(6), "rep/2/kernel": first_tap_identity(6)})
    cfg_t = _base(c, [block, selector(7)], pooling, c)
    w_t = dict(shared, **{"rep/1/kernel": first_tap_identity(7)})
    row = rng.permutation(np.arange(1, 65))[:length]
    masked = row.copy()
    masked[valid:] = 0
    ids_m = np.broadcast_to(masked, (1, 6, length)).copy()
    ids_t = np.broadcast_to(row[:valid], (1, 6, valid)).copy()
    return (cfg_m, w_m, ids_m), (cfg_t, w_t, ids_t)


# ---- strided ResidualBlock on a half-padded batch, every norm type ----------------------------------------------------
def strided_block_partial_mask_case(norm_type: str, seed: int = 3):
    """tests/unit/test_resblock_norm_type.py:160-173 (the crash case): a strides = 2 block (1x1 bypass + bn3) on a batch
    of two windows of 32 positions whose last 16 are padding must give length 16 and finite values for all three norms.
    The pooled twin asserts finite outputs equal to the oracle's, and - through the output mask's 'any' rule - that the
    masked average changes when a position INSIDE the valid half changes and does not when a padded one does."""
    c, length, valid = 16, 32, 16
    rng = np.random.default_rng(seed)
    block = {"name": "residual_block", "config": {"filters": c, "kernel_size": 5, "strides": 2, "activation": "gelu",
                                                  "norm_type": norm_type, "block_size": 1}}
    cfg = _base(c, [block], "average", c)
    lim = np.sqrt(6.0 / (5 * c + 5 * c))
    w = {"embedding/embeddings": rng.normal(size=(65, c)).astype(np.float32),
         "classifier/0/kernel": np.eye(c, dtype=np.float32)}
    for conv, k in (("conv1", 5), ("conv2", 5), ("conv3", 1)):
        w[f"rep/0/block0/{conv}/kernel"] = rng.uniform(-lim, lim, (k, c, c)).astype(np.float32)
        w[f"rep/0/block0/{conv}/bias"] = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    for bn in ("bn1", "bn2", "bn3"):
        if norm_type == "masked_batchnorm":
            w[f"rep/0/block0/{bn}/gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
            w[f"rep/0/block0/{bn}/beta"] = rng.uniform(-0.2, 0.2, c).astype(np.float32)
            w[f"rep/0/block0/{bn}/moving_mean"] = rng.uniform(-0.2, 0.2, c).astype(np.float32)
            w[f"rep/0/block0/{bn}/moving_variance"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        elif norm_type == "masked_dyt":
            w[f"rep/0/block0/{bn}/alpha"] = np.array([0.5], np.float32)
            w[f"rep/0/block0/{bn}/gamma"] = np.ones(c, np.float32)
            w[f"rep/0/block0/{bn}/beta"] = np.zeros(c, np.float32)
        else:
            w[f"rep/0/block0/{bn}/gamma"] = np.ones(c, np.float32)
            w[f"rep/0/block0/{bn}/beta"] = np.zeros(c, np.float32)
    ids = rng.integers(1, 65, (2, 6, length))
    ids[:, :, valid:] = 0
    return cfg, w, ids


# ---- MaskedLayerNormalization: masked positions zero, unmasked ones normalised ---------------------------------------
def layernorm_zeroes_masked_case(seed: int = 21):
    """tests/unit/test_nnlib_v2_layers.py:94-107: MaskedLayerNormalization(epsilon 1e-3) over inputs of standard deviation 5
    with two masked positions: |masked| < 1e-5, mean of the unmasked values within 0.05 of 0, their standard deviation within
    0.05 of 1.  The model twin reads both through an unmasking identity conv and the plain mean pool (as the DyT case does):
    the pooled vector is the sum of LN(x) over the VALID positions / all positions - a non-zeroed masked position would add
    LN(E[0]) / L per channel -, and the per-position statistics are asserted on the numpy expectation the pooled vector must
    equal."""
    dim, length = 16, 8
    rng = np.random.default_rng(seed)
    cfg = _base(dim, [{"name": "masked_layernorm", "config": {"epsilon": 1e-3}},
                      {"name": "masked_conv1d", "config": {"filters": dim, "kernel_size": 1, "padding": "valid",
                                                           "use_bias": False, "use_masking": False}}], "average", dim)
    table = (5.0 * rng.normal(size=(65, dim))).astype(np.float32)
    w = {"embedding/embeddings": table, "rep/0/gamma": np.ones(dim, np.float32), "rep/0/beta": np.zeros(dim, np.float32),
         "rep/1/kernel": np.eye(dim, dtype=np.float32)[None], "classifier/0/kernel": np.eye(dim, dtype=np.float32)}
    ids = rng.integers(1, 65, (2, 6, length))
    ids[0, :, 0] = 0                                        # the reference masks [0, 0, 0] and [1, 2, 3]: one position per window
    ids[1, :, 3] = 0
    x = table.astype(np.float64)[ids]                                       # (2, 6, L, C)
    ln = (x - x.mean(-1, keepdims=True)) / np.sqrt(x.var(-1, keepdims=True) + 1e-3)
    valid = (ids != 0)[..., None]
    want = (ln * valid).sum(axis=(1, 2)) / (6 * length)
    not_zeroed = ln.sum(axis=(1, 2)) / (6 * length)
    return cfg, w, ids, want, not_zeroed, ln[np.broadcast_to(valid, ln.shape)]
