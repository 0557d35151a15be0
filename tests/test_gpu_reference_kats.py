"""The reference's layer known-answer tests through ``JaegerHipEngine`` on the GPU (the CPU twins on the oracle are in
tests/test_oracle_forward.py; the cases themselves in tests/kat_models.py):

* tests/unit/test_masked_pooling.py:186-209   padded batch == truncated batch through Embedding -> BN -> masked max pool
* tests/unit/test_nnlib_v2_nmd.py:32-56       NMDLayer == MaskedBatchNorm(return_nmd) under a random mask
* tests/unit/test_ood_signal_layer.py:20-106  the five OOD signal formulas, rtol 1e-6
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
