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
