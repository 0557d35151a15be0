"""The reference's layer known-answer tests through ``JaegerHipEngine`` on the GPU (the CPU twins on the oracle are in
tests/test_oracle_forward.py; the cases themselves in tests/kat_models.py):

* tests/unit/test_masked_pooling.py:186-209   padded batch == truncated batch through Embedding -> BN -> masked max pool
* tests/unit/test_nnlib_v2_nmd.py:32-56       NMDLayer == MaskedBatchNorm(return_nmd) under a random mask
* tests/unit/test_ood_signal_layer.py:20-106  the five OOD signal formulas, rtol 1e-6
* tests/unit/test_mask_mode.py:41-98          (round 6) the conv's OUTPUT MASK per mask_mode - eight boolean vectors - read
                                              through a pooled indicator model, the probed conv first and second in the model
* tests/unit/test_nnlib_v2_layers_short_fragment.py:111-127   masked average [[2, 3], [5, 6]], rtol 1e-5
* tests/unit/test_resblock_norm_type.py:74-81     MaskedDYT re-zeroes masked positions, atol 1e-5
* tests/unit/test_resblock_norm_type.py:84-94     LayerNorm residual block: masked == truncated on positions 0..9, atol 1e-4
* tests/unit/test_resblock_norm_type.py:160-173   strides = 2 block on a half-padded batch, all three norms: finite
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(cfg, w, ids, precision=None):
    from jaeger_amd.engine import JaegerHipEngine
    eng = JaegerHipEngine(model_cfg=cfg, weights=w, device_id=0, precision=precision)
    try:
        return eng.model.forward(np.ascontiguousarray(ids, np.uint8))
    finally:
        eng.close()


@pytest.mark.parametrize("precision", [None, "f32"])
def test_padded_batch_pooling_matches_truncated(precision):
    from kat_models import padded_pooling_case
    from oracle import forward as F
    cfg, w, padded, trunc = padded_pooling_case()
    a, b = _run(cfg, w, padded, precision), _run(cfg, w, trunc, precision)
    np.testing.assert_allclose(a["embedding"], b["embedding"], atol=1e-5)           # the reference's assertion
    np.testing.assert_allclose(a["prediction"], b["prediction"], atol=1e-5)
    ref = F.forward(cfg, w, padded)
    np.testing.assert_allclose(a["embedding"], ref["embedding"], atol=1e-5)         # and the oracle's values
    table = w["embedding/embeddings"].astype(np.float64)
    want = ((table[trunc] - 3.0) / np.sqrt(2.0 + 1e-5)).max(axis=(1, 2))
    np.testing.assert_allclose(a["embedding"], want, atol=1e-5)
    unmasked = ((table[padded] - 3.0) / np.sqrt(2.0 + 1e-5)).max(axis=(1, 2))
    assert (unmasked > a["embedding"] + 1e-3).all()                                 # the padding id would have won unmasked


@pytest.mark.parametrize("zero_mean", [True, False])
def test_nmd_layer_matches_masked_batchnorm_return_nmd(zero_mean):
    from kat_models import nmd_vs_bn_case
    from oracle import forward as F
    (cfg_a, w_a), (cfg_b, w_b), ids = nmd_vs_bn_case(zero_mean=zero_mean)
    a, b = _run(cfg_a, w_a, ids), _run(cfg_b, w_b, ids)
    assert a["nmd"].shape == b["nmd"].shape == (4, 8)
    assert float(np.abs(a["nmd"] - b["nmd"]).max()) < 1e-5                           # the reference's assertion
    ref = F.forward(cfg_a, w_a, ids)
    for k in ("nmd", "embedding", "prediction", "reliability"):
        np.testing.assert_allclose(a[k], ref[k], atol=1e-5, err_msg=k)
        np.testing.assert_allclose(b[k], ref[k], atol=1e-5, err_msg=k)
    x = w_a["embedding/embeddings"].astype(np.float64)[ids]
    m = (ids != 0)[..., None]
    want = (x * m).sum(axis=(1, 2)) / (m.sum(axis=(1, 2)) + 1e-5) - w_a["rep/0/moving_mean"]
    np.testing.assert_allclose(a["nmd"], want, atol=1e-5)


def test_ood_signal_formulas():
    from kat_models import KAT_LOGITS, KAT_NMD, ood_expected, ood_signal_case
    cfg, w, ids = ood_signal_case()
    out = _run(cfg, w, ids)
    np.testing.assert_allclose(out["prediction"], KAT_LOGITS, rtol=1e-6)
    nmd = out["nmd"][:, :4]
    np.testing.assert_allclose(nmd[:, :2], KAT_NMD, rtol=1e-6, atol=1e-6)
    assert not nmd[:, 2:].any()
    rel = out["reliability"]
    assert rel.shape == (2, 9)
    np.testing.assert_array_equal(rel[:, :4], nmd)                                   # identity head: [nmd | signals]
    # the reference's tolerance, against the formulas applied to the logits / NMD vector the engine itself returned
    np.testing.assert_allclose(rel[:, 4:], ood_expected(out["prediction"], nmd), rtol=1e-6)
    np.testing.assert_allclose(rel[:, 4:], ood_expected(KAT_LOGITS, KAT_NMD), rtol=1e-5)


# ---- round 6: the remaining reference-held layer assertions ------------------------------------------------------------
PRECISIONS = [None, "f32"]          # None = the engine's own choice (split-f16 where a kernel covers the op), "f32" = exact path


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("deep", [False, True])
def test_mask_mode_known_answers(deep, precision):
    """The reference's eight output masks (any / majority / strict; isolated N, short and long N runs, right padding) as
    the DEVICE computes them: first conv on ids (the table-lookup kernel's mask) and a conv behind another (mask_kernel)."""
    from kat_models import MASK_MODE_KATS, mask_from_pooled, mask_mode_case
    for name, n_pos, mode, expected in MASK_MODE_KATS:
        cfg, w, ids = mask_mode_case(n_pos, mode, deep)
        out = _run(cfg, w, ids, precision)
        got = mask_from_pooled(out["embedding"])
        np.testing.assert_array_equal(got, np.array(expected), err_msg=f"{name} deep={deep} precision={precision}")
        np.testing.assert_array_equal(mask_from_pooled(out["prediction"]), np.array(expected))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_masked_average_ignores_padding(precision):
    from kat_models import masked_average_case
    cfg, w, ids, want = masked_average_case()
    out = _run(cfg, w, ids, precision)
    np.testing.assert_allclose(out["embedding"][:, :2], want, rtol=1e-5)            # the reference's assertion
    assert not out["embedding"][:, 2:].any()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_masked_dyt_zeroes_masked_positions(precision):
    from kat_models import dyt_zeroes_masked_case
    cfg, w, ids, want, not_zeroed = dyt_zeroes_masked_case()
    got = _run(cfg, w, ids, precision)["embedding"]
    np.testing.assert_allclose(got, want, atol=1e-5)                                # zeros at the 16 masked positions
    assert np.abs(got - not_zeroed).min() > 0.3


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("pooling", ["average", "max"])
def test_layernorm_block_masked_equals_truncated(pooling, precision):
    from kat_models import layernorm_block_masked_vs_truncated_case
    from oracle import forward as F
    (cfg_m, w_m, ids_m), (cfg_t, w_t, ids_t) = layernorm_block_masked_vs_truncated_case(pooling)
    a, b = _run(cfg_m, w_m, ids_m, precision), _run(cfg_t, w_t, ids_t, precision)
    np.testing.assert_allclose(a["embedding"], b["embedding"], atol=1e-4)           # the reference's assertion, pooled over 0..9
    np.testing.assert_allclose(a["embedding"], F.forward(cfg_m, w_m, ids_m)["embedding"], atol=1e-4)
    np.testing.assert_allclose(b["embedding"], F.forward(cfg_t, w_t, ids_t)["embedding"], atol=1e-4)
    assert np.abs(a["embedding"]).max() > 0.1


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("norm_type", ["masked_batchnorm", "masked_layernorm", "masked_dyt"])
def test_strided_block_on_a_half_padded_batch(norm_type, precision):
    from kat_models import strided_block_partial_mask_case
    from oracle import forward as F
    cfg, w, ids = strided_block_partial_mask_case(norm_type)
    out = _run(cfg, w, ids, precision)
    assert np.isfinite(out["embedding"]).all() and np.isfinite(out["prediction"]).all()     # the reference's assertion
    ref = F.forward(cfg, w, ids)
    np.testing.assert_allclose(out["embedding"], ref["embedding"], atol=1e-4)
    # more padding behind the padding (40 instead of 32 positions; TF's SAME pad of a stride-2 conv is the same for both
    # even lengths) leaves the result as it is, a change in the valid half moves it
    longer = np.zeros((2, 6, 40), ids.dtype)
    longer[:, :, :32] = ids
    np.testing.assert_allclose(_run(cfg, w, longer, precision)["embedding"], out["embedding"], atol=1e-6)
    inside = ids.copy()
    inside[:, :, 3] = (inside[:, :, 3] % 64) + 1
    assert np.abs(_run(cfg, w, inside, precision)["embedding"] - out["embedding"]).max() > 1e-4


@pytest.mark.parametrize("precision", PRECISIONS)
def test_masked_layernorm_zeroes_masked_and_normalises_unmasked(precision):
    """tests/unit/test_nnlib_v2_layers.py:94-107: masked positions hold zeros behind MaskedLayerNormalization (|.| < 1e-5), the
    unmasked values are normalised (mean within 0.05 of 0, std within 0.05 of 1) - the pooled vector equals the numpy
    expectation that carries both."""
    from kat_models import layernorm_zeroes_masked_case
    cfg, w, ids, want, not_zeroed, unmasked = layernorm_zeroes_masked_case()
    assert abs(unmasked.mean()) < 0.05 and abs(unmasked.std() - 1.0) < 0.05
    got = _run(cfg, w, ids, precision)["embedding"]
    np.testing.assert_allclose(got, want, atol=1e-5)
    assert np.abs(got - not_zeroed).max() > 1e-2
