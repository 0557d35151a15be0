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


def test_padded_batch_pooling_matches_truncated():
    """tests/unit/test_masked_pooling.py:186-209 of the reference: Embedding(mask_zero) -> MaskedBatchNorm (moving mean 3,
    variance 2: padded zeros become non-zero constants) -> masked max pool gives the same representation for a
    right-padded batch as for the truncated input, atol 1e-5 - the two-pass short-contig situation."""
    from kat_models import padded_pooling_case
    from oracle import forward as F
    cfg, w, padded, trunc = padded_pooling_case()
    a, b = F.forward(cfg, w, padded), F.forward(cfg, w, trunc)
    np.testing.assert_allclose(a["embedding"], b["embedding"], atol=1e-5)
    np.testing.assert_allclose(a["prediction"], b["prediction"], atol=1e-5)
    # the pool really sees the BN output: (E - 3) / sqrt(2 + 1e-5), maximum over the valid positions of both strands' frames
    table = w["embedding/embeddings"].astype(np.float64)
    want = ((table[trunc] - 3.0) / np.sqrt(2.0 + 1e-5)).max(axis=(1, 2))
    np.testing.assert_allclose(a["embedding"], want, atol=1e-6)
    # and a padded position would have won without the mask: BN(E[0]) sits above every window's maximum
    unmasked = ((table[padded] - 3.0) / np.sqrt(2.0 + 1e-5)).max(axis=(1, 2))
    assert (unmasked > want + 1e-3).all()


def test_padded_equals_truncated_for_pooled_outputs():
    """Same property through a real architecture (baseline500: valid k7 conv, two k3 residual blocks, masked average
    pool): right padding with id 0 changes nothing but the positions whose receptive field touches the padding, which
    the 'any' mask rule keeps - so the padded logits equal the logits of the same window run alone only when the mask
    rule excludes them; here we assert the weaker, exact statement: padding AFTER the padding changes nothing."""
    from oracle import forward as F
    cfg = load_model_cfg("baseline500")
    w = F.random_weights(cfg)
    rng = np.random.default_rng(1)
    ids = rng.integers(1, 65, (1, 6, 120))
    p1 = np.zeros((1, 6, 150), np.int64)
    p2 = np.zeros((1, 6, 165), np.int64)
    p1[:, :, :120] = ids
    p2[:, :, :120] = ids
    a, b = F.forward(cfg, w, p1), F.forward(cfg, w, p2)
    assert a["prediction"].shape == b["prediction"].shape == (1, 3)
    np.testing.assert_allclose(a["prediction"], b["prediction"], atol=1e-5)
    np.testing.assert_allclose(a["embedding"], b["embedding"], atol=1e-5)


def test_nmd_layer_matches_masked_batchnorm_return_nmd():
    """tests/unit/test_nnlib_v2_nmd.py:32-56 of the reference: NMDLayer(eps 1e-5) and the side output of
    MaskedBatchNorm(return_nmd=True, eps 1e-5) agree under a random mask (max diff < 1e-5), in the layers' initial state
    (moving mean 0) and with a shared non-trivial moving mean."""
    from kat_models import nmd_vs_bn_case
    from oracle import forward as F
    for zero in (True, False):
        (cfg_a, w_a), (cfg_b, w_b), ids = nmd_vs_bn_case(zero_mean=zero)
        a, b = F.forward(cfg_a, w_a, ids), F.forward(cfg_b, w_b, ids)
        assert a["nmd"].shape == b["nmd"].shape == (4, 8)
        assert float(np.abs(a["nmd"] - b["nmd"]).max()) < 1e-5
        # against the formula itself (nmd.py:52-77): sum(x m) / (sum(m) + 1e-5) - moving_mean
        x = w_a["embedding/embeddings"].astype(np.float64)[ids]
        m = (ids != 0)[..., None]
        want = (x * m).sum(axis=(1, 2)) / (m.sum(axis=(1, 2)) + 1e-5) - w_a["rep/0/moving_mean"]
        np.testing.assert_allclose(a["nmd"], want, atol=1e-5)


def test_ood_signal_formulas():
    """tests/unit/test_ood_signal_layer.py:20-106 of the reference: max_prob, entropy, energy (logsumexp), margin and
    nmd_norm on logits [[1,2,3],[.5,-1,1.5]] / nmd [[3,4],[0,-5]], rtol 1e-6 against the formulas written out in numpy;
    then the same through a whole model whose reliability head is the identity."""
    from kat_models import KAT_LOGITS, KAT_NMD, SIGNALS, ood_expected, ood_signal_case
    from oracle import forward as F
    want = ood_expected(KAT_LOGITS, KAT_NMD)
    got = F.ood_signals(torch.tensor(KAT_LOGITS, dtype=torch.float32), torch.tensor(KAT_NMD, dtype=torch.float32), SIGNALS)
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-6)
    np.testing.assert_allclose(want[:, 4], [5.0, 5.0], rtol=1e-12)
    for j, s in enumerate(SIGNALS):                                  # one signal at a time, and a plain-logits call
        one = F.ood_signals(torch.tensor(KAT_LOGITS, dtype=torch.float32), torch.tensor(KAT_NMD, dtype=torch.float32), [s])
        np.testing.assert_allclose(one.numpy()[:, 0], want[:, j], rtol=1e-6)
    cfg, w, ids = ood_signal_case()
    out = F.forward(cfg, w, ids)
    np.testing.assert_allclose(out["prediction"], KAT_LOGITS, rtol=1e-6)
    np.testing.assert_allclose(out["nmd"][:, :2], KAT_NMD, rtol=1e-6, atol=1e-6)
    assert not out["nmd"][:, 2:4].any()
    np.testing.assert_allclose(out["reliability"][:, :4], out["nmd"][:, :4], rtol=0, atol=0)
    np.testing.assert_allclose(out["reliability"][:, 4:], ood_expected(out["prediction"], out["nmd"][:, :4]), rtol=1e-6)
    np.testing.assert_allclose(out["reliability"][:, 4:], want, rtol=1e-5)


def test_f32_oracle_close_to_f64():
    from oracle import forward as F
    cfg = load_model_cfg("brain")
    w = F.random_weights(cfg)
    ids = np.random.default_rng(2).integers(0, 65, (2, 6, 498))
    a = F.forward(cfg, w, ids)
    b = F.forward(cfg, w, ids, dtype=torch.float64)
    assert np.abs(a["prediction"] - b["prediction"]).max() < 1e-4
    assert a["nmd"].shape == (2, 512) and a["reliability"].shape == (2, 1) and a["embedding"].shape == (2, 128)


def test_mask_mode_known_answers_through_the_indicator_model():
    """tests/unit/test_mask_mode.py:41-98 again, this time through the pooled indicator model of tests/kat_models.py (the
    form the GPU twin runs, tests/test_gpu_reference_kats.py): the probed conv first in the model and behind an identity conv."""
    from kat_models import MASK_MODE_KATS, mask_from_pooled, mask_mode_case
    from oracle import forward as F
    for deep in (False, True):
        for name, n_pos, mode, expected in MASK_MODE_KATS:
            cfg, w, ids = mask_mode_case(n_pos, mode, deep)
            got = mask_from_pooled(F.forward(cfg, w, ids)["embedding"])
            np.testing.assert_array_equal(got, np.array(expected), err_msg=f"{name} deep={deep}")


def test_masked_average_ignores_padding():
    """tests/unit/test_nnlib_v2_layers_short_fragment.py:111-127: [[2, 3], [5, 6]], rtol 1e-5."""
    from kat_models import masked_average_case
    from oracle import forward as F
    cfg, w, ids, want = masked_average_case()
    out = F.forward(cfg, w, ids)
    np.testing.assert_allclose(out["embedding"][:, :2], want, rtol=1e-5)
    assert not out["embedding"][:, 2:].any()


def test_masked_dyt_zeroes_masked_positions():
    """tests/unit/test_resblock_norm_type.py:74-81, directly on the layer and through the model of kat_models."""
    from kat_models import dyt_zeroes_masked_case
    from oracle import forward as F
    x = torch.tensor(np.random.default_rng(0).normal(size=(1, 1, 32, 8)).astype(np.float32))
    m = torch.cat([torch.ones(1, 1, 16), torch.zeros(1, 1, 16)], dim=-1)
    out = F.masked_dyt(x, m, {"alpha": torch.tensor([0.5]), "gamma": torch.ones(8), "beta": torch.full((8,), 0.25)})
    np.testing.assert_allclose(out.numpy()[:, :, 16:], 0.0, atol=1e-5)
    assert np.abs(out.numpy()[:, :, :16]).min() > 0
    cfg, w, ids, want, not_zeroed = dyt_zeroes_masked_case()
    got = F.forward(cfg, w, ids)["embedding"]
    np.testing.assert_allclose(got, want, atol=1e-5)
    assert np.abs(not_zeroed - want).min() > 0.3                    # what an un-zeroed masked half would have added


def test_layernorm_block_masked_equals_truncated():
    """tests/unit/test_resblock_norm_type.py:84-94: positions 0..9 of the masked and the truncated run agree (1e-4) - on the
    block's per-position output, and through the eroding-selector models whose pools read exactly those positions."""
    from kat_models import layernorm_block_masked_vs_truncated_case
    from oracle import forward as F
    (cfg_m, w_m, ids_m), (cfg_t, w_t, ids_t) = layernorm_block_masked_vs_truncated_case("average")
    tw = {k: torch.tensor(v) for k, v in w_m.items()}
    bcfg = cfg_m["representation_learner"]["hidden_layers"][0]["config"]
    table = tw["embedding/embeddings"]
    xm, mm = table[torch.tensor(ids_m)], torch.tensor((ids_m != 0).astype(np.float32))
    xt, mt = table[torch.tensor(ids_t)], torch.tensor((ids_t != 0).astype(np.float32))
    ym, om = F._residual_block(xm, mm, tw, "rep/0/block0", bcfg, True, True, torch.float32)
    yt, _ = F._residual_block(xt, mt, tw, "rep/0/block0", bcfg, True, True, torch.float32)
    np.testing.assert_allclose(ym.numpy()[:, :, :10], yt.numpy()[:, :, :10], atol=1e-4)       # the reference's assertion
    assert np.abs(ym.numpy()[:, :, 12:16] - yt.numpy()[:, :, 12:16]).max() > 1e-3            # and the boundary does differ
    assert om[0, 0].numpy().astype(bool).tolist() == [i < 20 for i in range(32)]             # 'any': 16 + 2 + 2 valid
    for pooling in ("average", "max"):
        (cfg_m, w_m, ids_m), (cfg_t, w_t, ids_t) = layernorm_block_masked_vs_truncated_case(pooling)
        a, b = F.forward(cfg_m, w_m, ids_m), F.forward(cfg_t, w_t, ids_t)
        np.testing.assert_allclose(a["embedding"], b["embedding"], atol=1e-4)
        red = ym.numpy()[0, :, :10].mean(axis=(0, 1)) if pooling == "average" else ym.numpy()[0, :, :10].max(axis=(0, 1))
        np.testing.assert_allclose(a["embedding"][0], red, atol=1e-5)                        # the pool reads positions 0..9


def test_strided_block_on_a_half_padded_batch():
    """tests/unit/test_resblock_norm_type.py:160-173: length 16 and finite for all three norms."""
    from kat_models import strided_block_partial_mask_case
    from oracle import forward as F
    for nt in ("masked_batchnorm", "masked_layernorm", "masked_dyt"):
        cfg, w, ids = strided_block_partial_mask_case(nt)
        tw = {k: torch.tensor(v) for k, v in w.items()}
        x, m = tw["embedding/embeddings"][torch.tensor(ids)], torch.tensor((ids != 0).astype(np.float32))
        y, om = F._residual_block(x, m, tw, "rep/0/block0", cfg["representation_learner"]["hidden_layers"][0]["config"],
                                  True, True, torch.float32)
        assert y.shape[2] == 16 and om.shape[2] == 16 and np.isfinite(y.numpy()).all(), nt
        assert np.isfinite(F.forward(cfg, w, ids)["embedding"]).all()


def test_masked_layernorm_zeroes_masked_and_normalises_unmasked():
    """tests/unit/test_nnlib_v2_layers.py:94-107 on the layer and through the model of kat_models."""
    from kat_models import layernorm_zeroes_masked_case
    from oracle import forward as F
    rng = np.random.default_rng(0)
    x = torch.tensor((5.0 * rng.normal(size=(2, 4, 8, 16))).astype(np.float32))
    m = torch.ones(2, 4, 8)
    m[0, 0, 0] = 0
    m[1, 2, 3] = 0
    out = F.masked_layernorm(x, m, {"gamma": torch.ones(16), "beta": torch.zeros(16)}, eps=1e-3).numpy()
    mb = m.numpy().astype(bool)
    assert np.abs(out[~mb]).max() < 1e-5
    assert abs(out[mb].mean()) < 0.05 and abs(out[mb].std() - 1.0) < 0.05
    cfg, w, ids, want, not_zeroed, unmasked = layernorm_zeroes_masked_case()
    assert abs(unmasked.mean()) < 0.05 and abs(unmasked.std() - 1.0) < 0.05          # the reference's two bounds
    got = F.forward(cfg, w, ids)["embedding"]
    np.testing.assert_allclose(got, want, atol=1e-5)
    assert np.abs(not_zeroed - want).max() > 1e-2
