"""project.yaml -> layer plan -> op program (host logic, no GPU)."""
import copy

import numpy as np
import pytest

from conftest import GOLDEN, load_model_cfg


def test_plan_defaults_follow_reference_constructors():
    from jaeger_amd import plan as P
    pl = P.build_plan(load_model_cfg("brain"))
    conv0 = pl.rep[0]
    assert isinstance(conv0, P.Conv) and conv0.padding == "valid"        # MaskedConv1D default (layers.py:1156)
    assert conv0.mask_mode == "any" and conv0.kernel_size == 7 and conv0.cin == 128
    blocks = [l for l in pl.rep if isinstance(l, P.ResBlock)]
    assert len(blocks) == 6 and all(b.conv1.padding == "same" and b.conv1.dilation_rate == 3 for b in blocks)
    assert all(b.conv3 is None for b in blocks)
    assert pl.nmd_dims == [128] * 4 and pl.pooling == "max" and pl.n_classes == 6
    assert pl.vocab == 65 and pl.string_processor["seq_onehot"] is False
    assert pl.string_processor["input_type"] == "translated"
    n_params = sum(int(np.prod(s)) for s in P.weight_shapes(pl).values())
    assert n_params == 1121303


def test_crop_resolution_and_seq_onehot_inference():
    """nnlib/inference.py:452-482 (tests/unit/test_inference.py of the reference)."""
    from jaeger_amd.plan import resolve_string_processor
    cfg = load_model_cfg("baseline500")
    sp = resolve_string_processor(cfg)
    assert sp["crop_size_codons"] == 500 and sp["crop_size_nt"] == 1505     # crop_units defaults to codon
    cfg2 = copy.deepcopy(cfg)
    cfg2["string_processor"].pop("seq_onehot")
    cfg2["embedding"]["input_shape"] = [6, None, 64]
    assert resolve_string_processor(cfg2)["seq_onehot"] is True
    cfg2["embedding"]["input_shape"] = [6, None]
    assert resolve_string_processor(cfg2)["seq_onehot"] is False
    cfg3 = copy.deepcopy(cfg)
    cfg3["string_processor"]["crop_units"] = "nucleotide"
    cfg3["string_processor"]["crop_size"] = 1505
    assert resolve_string_processor(cfg3)["crop_size_codons"] == 500


def test_weight_names_agree_with_oracle():
    from jaeger_amd import plan as P
    from oracle import forward as F
    for name in ("brain", "zeus", "baseline500"):
        cfg = load_model_cfg(name)
        assert P.weight_shapes(P.build_plan(cfg)) == F.weight_specs(cfg)


def test_program_fuses_epilogues_and_fits_slots():
    from jaeger_amd import _lib as L
    from jaeger_amd import plan as P
    from jaeger_amd import program as G
    from oracle import forward as F
    cfg = load_model_cfg("brain")
    prog = G.compile_plan(P.build_plan(cfg), F.random_weights(cfg))
    kinds = [op.kind for op in prog.ops]
    assert kinds.count(L.OP_CONV) == 13 and kinds.count(L.OP_MASK) == 13
    assert kinds.count(L.OP_NMD_FINAL) == 4 and kinds.count(L.OP_POOL) == 1 and kinds.count(L.OP_DENSE) == 3
    convs = [op for op in prog.ops if op.kind == L.OP_CONV]
    st = lambda op: [op.stages[i].kind for i in range(op.n_stages)]
    assert st(convs[0]) == [L.ST_BIAS, L.ST_NMD, L.ST_BN, L.ST_ACT]
    assert st(convs[1]) == [L.ST_BIAS, L.ST_BN, L.ST_ACT]
    assert st(convs[2]) == [L.ST_BIAS, L.ST_BN, L.ST_ADD, L.ST_ACT]
    assert st(convs[4]) == [L.ST_BIAS, L.ST_BN, L.ST_ADD, L.ST_ACT, L.ST_NMD, L.ST_BN, L.ST_ACT]
    assert convs[0].in_buf == L.JG_BUF_IDS and convs[0].in_mask == L.JG_BUF_IDS
    for op in prog.ops:
        assert op.out_buf < L.JG_MAX_BUFS and op.out_mask < L.JG_MAX_BUFS
    assert prog.blob.size % 4 == 0 and prog.nmd_dim == 512 and prog.has_reliability


def test_use_masking_false_drops_masks():
    from jaeger_amd import _lib as L
    from jaeger_amd import plan as P
    from jaeger_amd import program as G
    from oracle import forward as F
    cfg = load_model_cfg("baseline500")
    cfg["use_masking"] = False                                     # legacy SavedModels (builder.py:259)
    prog = G.compile_plan(P.build_plan(cfg), F.random_weights(cfg))
    assert not any(op.kind == L.OP_MASK for op in prog.ops)
    pool = [op for op in prog.ops if op.kind == L.OP_POOL][0]
    assert pool.in_mask == L.JG_BUF_NONE


def test_unsupported_layers_fail_loudly():
    from jaeger_amd import plan as P
    cfg = load_model_cfg("brain")
    cfg["representation_learner"]["hidden_layers"].insert(1, {"name": "transformer_encoder", "config": {}})
    with pytest.raises(P.UnsupportedLayer):
        P.build_plan(cfg)
    cfg = load_model_cfg("brain")
    cfg["representation_learner"]["pooling"] = "gatedframe"
    with pytest.raises(P.UnsupportedLayer):
        P.build_plan(cfg)
    cfg = load_model_cfg("brain")
    cfg["reliability_model"]["input_shape"] = 100
    with pytest.raises(ValueError):
        P.build_plan(cfg)


def test_missing_or_misshaped_weights_rejected():
    from jaeger_amd import plan as P
    from jaeger_amd import program as G
    from oracle import forward as F
    cfg = load_model_cfg("baseline500")
    w = F.random_weights(cfg)
    bad = dict(w)
    bad.pop("rep/0/kernel")
    with pytest.raises(KeyError):
        G.compile_plan(P.build_plan(cfg), bad)
    bad = dict(w)
    bad["rep/0/kernel"] = bad["rep/0/kernel"][:, :, :16]
    with pytest.raises(ValueError):
        G.compile_plan(P.build_plan(cfg), bad)


def test_keras3_weights_h5_reader():
    """A Keras-3 style weights file (layers/<auto name>/vars/<i>, residual blocks as containers) written
    by HDF5's h5import from seeded weights maps back onto the plan's canonical names."""
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import load_keras3_h5, load_weights, random_weights
    plan = build_plan(load_model_cfg("baseline500"))
    want = random_weights(plan, seed=500)
    got = load_keras3_h5(GOLDEN / "baseline500_keras3.weights.h5", plan)
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    again = load_weights({"weights": GOLDEN / "baseline500_keras3.weights.h5"}, plan)
    assert all(np.array_equal(again[k], want[k]) for k in want)


@pytest.mark.parametrize("layout", ["flat", "nested"])
def test_keras3_weights_h5_both_export_generations(layout):
    """First-contact safety for a real checkpoint: the two ``.weights.h5`` layouts the reference's converter documents
    (scripts/convert_legacy_classifier_checkpoint.py:29-32,76-175) - custom ``ResidualBlockStack`` containers
    (``layers/residual_block_stack[_N]/blocks/residual_block[_k]/{conv1,..}/vars/<i>``, blocks numbered per container) and
    the older Functional sub-models (``layers/functional[_N]/layers/residual_block[_k]/...`` with block names counting on
    across the sub-models, the head under ``layers/functional_8/layers/dense[_k]``) - written by the HDF5 library's h5import
    (tests/golden/make_golden_h5.py) map onto the plan by type + order: two stacks, a 1x1 bypass, two stem convs and two
    batch norms of identical shapes that only their auto-name order tells apart."""
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import load_keras3_h5, random_weights
    plan = build_plan(load_model_cfg("stacks2"))
    want = random_weights(plan, seed=2026)
    got = load_keras3_h5(GOLDEN / f"stacks2_keras3_{layout}.weights.h5", plan)
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    assert want["rep/0/kernel"].shape == want["rep/3/kernel"].shape and not np.array_equal(got["rep/0/kernel"], got["rep/3/kernel"])


def test_keras3_weights_h5_mismatch_lists_what_is_left():
    """A file that does not line up with the plan fails with BOTH sides named: the plan layers that found no weight group
    and the groups of the file nobody took (here: the baseline500 file against the two-stack plan and the reverse)."""
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import load_keras3_h5
    with pytest.raises(ValueError, match="residual blocks in the file"):
        load_keras3_h5(GOLDEN / "baseline500_keras3.weights.h5", build_plan(load_model_cfg("stacks2")))
    cfg = copy.deepcopy(load_model_cfg("stacks2"))
    cfg["classifier"]["hidden_layers"][0]["config"]["units"] = 12          # the head's first Dense no longer fits its group
    with pytest.raises(ValueError) as ei:
        load_keras3_h5(GOLDEN / "stacks2_keras3_nested.weights.h5", build_plan(cfg))
    msg = str(ei.value)
    assert "plan layers without a weight group: classifier/0 [(16, 12), (12,)]; classifier/2 [(12, 3), (3,)]" in msg
    assert "functional_8/dense [(16, 8), (8,)]" in msg and "functional_8/dense_1 [(8, 3), (3,)]" in msg


def test_dicodon_model_compiles_to_an_embedding_op_in_front_of_the_first_conv():
    """``codon: DICODON`` / ``codon_id: DICODON_ID`` (nnlib/inference.py:430-451, commands/predict.py:224): vocabulary 4 097,
    6-grams; the 16-bit ids get an EMBED op of their own (rows + mask), and nothing else reads the id tensor."""
    from jaeger_amd import _lib as L
    from jaeger_amd import plan as P
    from jaeger_amd import program as G
    from oracle import forward as F
    cfg = copy.deepcopy(load_model_cfg("baseline500"))
    cfg["string_processor"]["codon"], cfg["string_processor"]["codon_id"] = "DICODON", "DICODON_ID"
    cfg["embedding"]["embedding_size"] = 16
    plan = P.build_plan(cfg)
    assert plan.vocab == 4097 and plan.string_processor["ngram_width"] == 6
    w = F.random_weights(cfg)
    assert w["embedding/embeddings"].shape == (4097, 16)
    prog = G.compile_plan(plan, w)
    assert prog.ops[0].kind == L.OP_EMBED and prog.ops[0].cout == 16 and prog.vocab == 4097
    assert not any(op.in_buf == L.JG_BUF_IDS or op.in_mask == L.JG_BUF_IDS for op in prog.ops[1:])
    cfg["string_processor"]["codon_id"] = "CODON_ID"                 # 6-grams with a 64-entry id map: no such encoding
    with pytest.raises(P.UnsupportedLayer):
        P.build_plan(cfg)


def test_layernorm_cuts_the_epilogue_into_an_elementwise_op():
    """MaskedLayerNormalization reduces over the channels: the conv keeps the stages in front of it, an element-wise
    op led by the LN stage runs the norm and everything behind it (shortcut add, activation, NMD tap, next norm)."""
    from jaeger_amd import _lib as L
    from jaeger_amd import plan as P
    from jaeger_amd import program as G
    from oracle import forward as F
    cfg = copy.deepcopy(load_model_cfg("brain"))
    for layer in cfg["representation_learner"]["hidden_layers"]:
        if layer["name"] == "residual_block":
            layer["config"]["norm_type"] = "masked_layernorm"
    prog = G.compile_plan(P.build_plan(cfg), F.random_weights(cfg))
    st = lambda op: [op.stages[i].kind for i in range(op.n_stages)]
    convs = [op for op in prog.ops if op.kind == L.OP_CONV]
    elts = [op for op in prog.ops if op.kind == L.OP_ELTWISE]
    assert len(convs) == 13 and len(elts) == 12
    assert st(convs[1]) == [L.ST_BIAS] and st(elts[0]) == [L.ST_LN, L.ST_ACT]                       # block conv1
    assert st(convs[2]) == [L.ST_BIAS] and st(elts[1]) == [L.ST_LN, L.ST_ADD, L.ST_ACT]             # block conv2
    assert st(elts[3]) == [L.ST_LN, L.ST_ADD, L.ST_ACT, L.ST_NMD, L.ST_BN, L.ST_ACT]                # end of a stack
    assert all(e.in_buf == e.out_buf and st(e)[0] == L.ST_LN and L.ST_LN not in st(e)[1:] for e in elts)


def _return_nmd_variant():
    """brain with its four `nmd` layers replaced by return_nmd=True on the first batch norm and on the three
    residual stacks (train_config/nn_config.yaml:205 uses the block form)."""
    import copy
    cfg = copy.deepcopy(load_model_cfg("brain"))
    hl = cfg["representation_learner"]["hidden_layers"]
    new, first_bn = [], True
    for layer in hl:
        if layer["name"] == "nmd":
            continue
        layer = copy.deepcopy(layer)
        layer.setdefault("config", {})
        if layer["name"] == "masked_batchnorm" and first_bn:
            layer["config"]["return_nmd"] = True
            first_bn = False
        if layer["name"] == "residual_block":
            layer["config"]["return_nmd"] = True
        new.append(layer)
    cfg["representation_learner"]["hidden_layers"] = new
    return cfg


def test_return_nmd_norms_and_blocks_compile():
    """MaskedBatchNorm(return_nmd=True) = an NMD tap on the norm's input sharing its moving_mean (layers.py:943-954);
    ResidualBlockStack(return_nmd=True) = that tap on the LAST block's bn2 (layers.py:1896-1899, 2682-2686)."""
    from jaeger_amd import _lib as L
    from jaeger_amd.plan import Nmd, ResBlock, build_plan, weight_shapes
    from jaeger_amd.program import compile_plan
    from jaeger_amd.weights import random_weights
    cfg = _return_nmd_variant()
    plan = build_plan(cfg)
    assert plan.nmd_dims == [128, 128, 128, 128]
    blocks = [l for l in plan.rep if isinstance(l, ResBlock)]
    assert [b.nmd is not None for b in blocks] == [False, True] * 3          # only the last block of each stack
    assert isinstance(plan.rep[1], Nmd) and plan.rep[1].name == plan.rep[2].name   # shares the norm's moving_mean
    assert not any(k.endswith("/moving_mean") and "nmd" in k for k in weight_shapes(plan))
    prog = compile_plan(plan, random_weights(plan))
    taps = [op for op in prog.ops if op.kind == L.OP_CONV and any(op.stages[j].kind == L.ST_NMD for j in range(op.n_stages))]
    finals = [op for op in prog.ops if op.kind == L.OP_NMD_FINAL]
    assert len(taps) == 4 and len(finals) == 4 and [f.vec_off for f in finals] == [0, 128, 256, 384]
    # the tap sits in front of the norm and of the residual add
    st = [taps[1].stages[j].kind for j in range(taps[1].n_stages)]
    assert st[:5] == [L.ST_BIAS, L.ST_NMD, L.ST_BN, L.ST_ADD, L.ST_ACT]
    # oracle: same outputs as explicit nmd layers carrying the norms' moving means
    import numpy as np
    from oracle import forward as ofwd
    w = ofwd.random_weights(cfg, seed=3)
    ids = np.random.default_rng(0).integers(0, 65, (3, 6, 60)).astype(np.uint8)
    out = ofwd.forward(cfg, w, ids)
    assert out["nmd"].shape == (3, 512) and out["reliability"].shape == (3, 1)
    with pytest.raises(Exception):
        bad = _return_nmd_variant()
        bad["representation_learner"]["hidden_layers"][3]["config"]["norm_type"] = "masked_dyt"
        build_plan(bad)


def _onehot_variant(embedding_size):
    import copy
    cfg = copy.deepcopy(load_model_cfg("baseline500"))
    cfg["embedding"].update(use_embedding_layer=False, embedding_size=embedding_size, input_shape=[6, None, 64])
    cfg["string_processor"]["seq_onehot"] = True
    return cfg


@pytest.mark.parametrize("embedding_size", [64, 0])
def test_onehot_translated_input_compiles_to_the_same_gather(embedding_size):
    """seq_onehot=True (one-hot rows -> Masking -> bias-free Dense, builder.py:869-880) is an id gather with a zero
    row for invalid codons; embedding_size 0 feeds the one-hot rows straight into the first conv."""
    from jaeger_amd.plan import build_plan, weight_shapes
    from jaeger_amd.program import compile_plan
    from jaeger_amd.weights import random_weights
    from oracle import forward as ofwd
    cfg = _onehot_variant(embedding_size)
    plan = build_plan(cfg)
    assert plan.embedding_kind == ("onehot_dense" if embedding_size else "onehot") and plan.vocab == 65
    assert plan.embedding_dim == (embedding_size or 64) and plan.string_processor["seq_onehot"] is True
    shapes = weight_shapes(plan)
    assert ("embedding/kernel" in shapes) == bool(embedding_size) and "embedding/embeddings" not in shapes
    w = random_weights(plan)
    prog = compile_plan(plan, w)
    assert prog.ops[1].cin == (embedding_size or 64)
    assert set(ofwd.weight_specs(cfg)) == set(shapes)
    # oracle: identical to the Embedding form whose table is [0; kernel]
    ids = np.random.default_rng(2).integers(0, 65, (3, 6, 40)).astype(np.uint8)
    got = ofwd.forward(cfg, w, ids)
    emb_cfg = copy.deepcopy(load_model_cfg("baseline500"))
    emb_cfg["embedding"]["embedding_size"] = embedding_size or 64
    rows = w["embedding/kernel"] if embedding_size else np.eye(64, dtype=np.float32)
    w2 = dict(w, **{"embedding/embeddings": np.concatenate([np.zeros((1, rows.shape[1]), np.float32), rows])})
    w2.pop("embedding/kernel", None)
    ref = ofwd.forward(emb_cfg, w2, ids)
    for k in ref:
        np.testing.assert_allclose(got[k], ref[k], atol=1e-6)
    with pytest.raises(Exception):
        bad = _onehot_variant(64)
        bad["string_processor"]["seq_onehot"] = False
        build_plan(bad)


def test_layers_directly_on_the_embedding_get_an_identity_conv():
    """Embedding -> MaskedBatchNorm -> masked max pool (the reference's tests/unit/test_masked_pooling.py:186-209 model)
    and Embedding -> nmd: a one-tap identity conv that does not mask its input carries the table rows, the Embedding's
    mask becomes a slot of its own, the norm / NMD tap fuse into the conv's epilogue."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).parent))
    from kat_models import nmd_vs_bn_case, padded_pooling_case
    from jaeger_amd import _lib as L
    from jaeger_amd.plan import build_plan
    from jaeger_amd.program import compile_plan
    cfg, w, _, _ = padded_pooling_case()
    prog = compile_plan(build_plan(cfg), w)
    kinds = [op.kind for op in prog.ops]
    assert kinds == [L.OP_MASK, L.OP_CONV, L.OP_POOL, L.OP_DENSE]
    mask, conv, pool = prog.ops[0], prog.ops[1], prog.ops[2]
    assert mask.in_mask == L.JG_BUF_IDS and mask.k == 1 and conv.in_buf == L.JG_BUF_IDS and conv.in_mask == L.JG_BUF_NONE
    assert conv.k == 1 and conv.cin == conv.cout == 8 and conv.out_mask == mask.out_mask == pool.in_mask
    assert [conv.stages[j].kind for j in range(conv.n_stages)] == [L.ST_BN]
    kern = prog.blob[conv.w_off:conv.w_off + 8 * 32].reshape(8, 32)
    np.testing.assert_array_equal(kern[:, :8], np.eye(8, dtype=np.float32))
    (cfg_a, w_a), (cfg_b, w_b), _ = nmd_vs_bn_case()
    for c, ww in ((cfg_a, w_a), (cfg_b, w_b)):
        prog = compile_plan(build_plan(c), ww)
        conv = next(op for op in prog.ops if op.kind == L.OP_CONV)
        assert [conv.stages[j].kind for j in range(conv.n_stages)] == [L.ST_NMD, L.ST_BN]
        assert prog.nmd_dim == 8


@pytest.mark.parametrize("mode", ["sum", "mean", "max", "weighted"])
def test_nmd_merge_modes_compile_to_one_dense_layer_or_a_block_diagonal_one_and_a_maximum(mode):
    """NMDMerge (nnlib/v2/nmd.py:93-170) over the two NMD taps of the nmdmerge500 family: every vector through its own
    bias-free projection, then added / averaged / softmax-weighted - ONE dense op over the vectors side by side, its kernel
    the projections stacked and scaled - or maximised (a block-diagonal dense op + JG_OP_VECMAX).  The taps write a scratch
    vector, the model's nmd output is the merged one (target_dim wide) and feeds the reliability head."""
    import copy

    from jaeger_amd import _lib as L
    from jaeger_amd.plan import UnsupportedLayer, build_plan, weight_shapes
    from jaeger_amd.program import compile_plan
    from oracle import forward as ofwd
    cfg = copy.deepcopy(load_model_cfg("nmdmerge500"))
    rel = cfg["reliability_model"]
    rel["merge"] = {"mode": mode, "target_dim": 24}
    rel["input_shape"] = 24
    plan = build_plan(cfg)
    assert plan.nmd_merge_mode == mode and plan.nmd_dim == 24 and plan.nmd_raw_dim == 64 and len(plan.nmd_dims) == 2
    shapes = weight_shapes(plan)
    assert shapes["rep/nmd_merge/proj_0/kernel"] == (plan.nmd_dims[0], 24)
    assert ("rep/nmd_merge/layer_weights" in shapes) == (mode == "weighted")
    assert shapes == ofwd.weight_specs(cfg)
    weights = ofwd.random_weights(cfg, seed=5)
    prog = compile_plan(plan, weights)
    kinds = [op.kind for op in prog.ops]
    finals = [op for op in prog.ops if op.kind == L.OP_NMD_FINAL]
    assert len(finals) == 2 and all(op.out_vec == L.JG_MAX_VECS - 1 for op in finals)
    merged = [op for op in prog.ops if op.out_vec == L.VEC_NMD]
    if mode == "max":
        assert [op.kind for op in merged] == [L.OP_VECMAX] and merged[0].k == 2 and merged[0].cout == 24
        dense = prog.ops[kinds.index(L.OP_VECMAX) - 1]
        assert dense.kind == L.OP_DENSE and dense.cin == 64 and dense.cout == 48 and dense.b_off < 0
    else:
        assert [op.kind for op in merged] == [L.OP_DENSE] and merged[0].cin == 64 and merged[0].cout == 24 and merged[0].b_off < 0
    assert prog.nmd_dim == 24
    # round 6: projections with an activation (projection_kwargs, nmd.py:133-141) = the block-diagonal dense op with the
    # activation + a linear dense op over the blocks (scaled identities stacked), or + JG_OP_VECMAX; against numpy on the oracle
    rel["merge"] = {"mode": mode, "target_dim": 24, "projection_kwargs": {"activation": "relu", "kernel_regularizer": None}}
    plan_a = build_plan(cfg)
    assert plan_a.nmd_merge_act == "relu"
    prog_a = compile_plan(plan_a, weights)
    merged = [op for op in prog_a.ops if op.out_vec == L.VEC_NMD]
    blocks = [op for op in prog_a.ops if op.out_vec == L.JG_MAX_VECS - 2]
    assert len(blocks) == 1 and blocks[0].kind == L.OP_DENSE and (blocks[0].cin, blocks[0].cout) == (64, 48)
    assert blocks[0].arg == __import__("jaeger_amd.program", fromlist=["act_code"]).act_code("relu")
    if mode == "max":
        assert [op.kind for op in merged] == [L.OP_VECMAX]
    else:
        assert [op.kind for op in merged] == [L.OP_DENSE] and (merged[0].cin, merged[0].cout) == (48, 24)
    ids = np.random.default_rng(3).integers(1, 65, (5, 6, 165))
    out = ofwd.forward(cfg, weights, ids)
    lin = copy.deepcopy(cfg)
    lin["reliability_model"]["merge"] = {"mode": "concat"}
    lin["reliability_model"]["input_shape"] = 64
    lin_w = dict(weights)                                   # (the head reads 64 values there: its weights from the concat model)
    lin_w.update({k: v for k, v in ofwd.random_weights(lin, seed=5).items() if k.startswith("reliability")})
    raw = ofwd.forward(lin, lin_w, ids)["nmd"].astype(np.float64)
    pr = [np.maximum(raw[:, :32] @ weights["rep/nmd_merge/proj_0/kernel"].astype(np.float64), 0),
          np.maximum(raw[:, 32:] @ weights["rep/nmd_merge/proj_1/kernel"].astype(np.float64), 0)]
    if mode == "weighted":
        lw = weights["rep/nmd_merge/layer_weights"].astype(np.float64)
        sm = np.exp(lw - lw.max()) / np.exp(lw - lw.max()).sum()
        want = sm[0] * pr[0] + sm[1] * pr[1]
    else:
        want = {"sum": pr[0] + pr[1], "mean": (pr[0] + pr[1]) / 2, "max": np.maximum(pr[0], pr[1])}[mode]
    np.testing.assert_allclose(out["nmd"], want, atol=1e-4, rtol=1e-4)
    # what stays rejected, loudly: a projection keyword the reference's Dense call would not survive / this engine does not build
    rel["merge"] = {"mode": mode, "target_dim": 24, "projection_kwargs": {"use_bias": True}}
    with pytest.raises(UnsupportedLayer):
        build_plan(cfg)
    rel["merge"] = {"mode": mode, "target_dim": 24, "projection_kwargs": {"activation": "swish"}}
    with pytest.raises(UnsupportedLayer):
        build_plan(cfg)
    rel["merge"] = {"mode": "bogus"}
    with pytest.raises(ValueError):
        build_plan(cfg)


def test_positional_embeddings_compile_to_an_embedding_op_with_a_position_table():
    """use_positional_embeddings (builder.py:886-892): the embedded input + SinusoidalPositionEmbedding rows.  The sum
    depends on the position, so the lookup leaves the first conv's id gather: JG_OP_EMBED opens the program (one-byte ids -
    the vocabulary is the codons'), its position table holds program.POSITION_ROWS rows equal to the oracle's restatement
    of the layer; without the wavelength the plan fails like the reference's 1 / None."""
    import copy

    import torch

    from jaeger_amd import _lib as L
    from jaeger_amd.plan import build_plan
    from jaeger_amd.program import POSITION_ROWS, compile_plan, sinusoidal_position_rows
    from oracle import forward as ofwd
    cfg = copy.deepcopy(load_model_cfg("baseline500"))
    cfg["embedding"]["use_positional_embeddings"] = True
    with pytest.raises(ValueError):
        build_plan(cfg)
    cfg["embedding"]["positional_embedding_length"] = 1000
    plan = build_plan(cfg)
    assert plan.positional_wavelength == 1000.0 and plan.vocab <= 256
    prog = compile_plan(plan, ofwd.random_weights(cfg, seed=3))
    first = prog.ops[0]
    assert first.kind == L.OP_EMBED and first.k == POSITION_ROWS and first.w_off >= 0 and first.cout == plan.embedding_dim
    assert all(op.in_buf != L.JG_BUF_IDS and op.in_mask != L.JG_BUF_IDS for op in prog.ops[1:])
    # the reference's own assertion (tests/unit/test_nnlib_v2_layers.py:187-192): the layer's output has its input's shape
    assert tuple(ofwd.sinusoidal_position_embedding(torch.zeros(2, 6, 32, 8), 1000).shape) == (2, 6, 32, 8)
    rows = sinusoidal_position_rows(700, plan.embedding_dim, 1000.0)
    ref = ofwd.sinusoidal_position_embedding(torch.zeros(2, 6, 700, plan.embedding_dim), 1000.0)[0, 0].numpy()
    assert rows.dtype == np.float32 and np.abs(rows - ref).max() <= 2e-6          # (same libm: equal; see the oracle's note)
    assert np.allclose(rows[0, 0::2], 0.0) and np.allclose(rows[0, 1::2], 1.0)
    table = np.asarray(prog.blob[first.w_off:first.w_off + 700 * plan.embedding_dim]).reshape(700, -1)
    np.testing.assert_array_equal(table, rows)


def _nmd_merge_builder_config():
    """The model of the reference's ``tests/integration/test_builder_nmd_merge.py:13-90`` (data only): two convs with an
    ``nmd`` layer each (16 and 8 channels), max pool, a three-class head."""
    return {
        "name": "test_nmd_merge", "classifier_out_dim": 3, "reliability_out_dim": 0,
        "class_label_map": [{"class": "chromosome", "label": 0}, {"class": "virus", "label": 1}, {"class": "plasmid", "label": 2}],
        "embedding": {"use_embedding_layer": True, "input_type": "translated", "strands": 2, "frames": 6,
                      "input_shape": [6, None], "embedding_size": 64},
        "string_processor": {"data_format": "numpy", "seq_onehot": False, "codon": "CODON", "codon_id": "CODON_ID", "crop_size": 100},
        "representation_learner": {"hidden_layers": [
            {"name": "masked_conv1d", "config": {"filters": 16, "kernel_size": 3}}, {"name": "nmd", "config": {}},
            {"name": "activation", "config": {"activation": "gelu"}},
            {"name": "masked_conv1d", "config": {"filters": 8, "kernel_size": 3}}, {"name": "nmd", "config": {}},
            {"name": "activation", "config": {"activation": "gelu"}}], "pooling": "max"},
        "classifier": {"input_shape": 8, "hidden_layers": [{"name": "dense", "config": {"units": 3, "activation": None}}]},
    }


def test_reference_builder_kats_for_nmd_merge():
    """The reference's builder-level assertions for several NMD layers and their merge
    (``tests/integration/test_builder_nmd_merge.py:93-185``, ``tests/unit/test_nnlib_v2_nmd.py:66-97``) on the plan and the
    oracle's forward: concat -> a 24-wide nmd output and reliability input; sum / mean / max / weighted with target_dim 8 ->
    8 wide; a reliability ``input_shape`` that disagrees raises "does not match"; no NMD tensor under a configured
    reliability head raises "no NMD tensor"; an unknown mode raises ValueError."""
    import copy

    from jaeger_amd.plan import build_plan
    from oracle import forward as ofwd
    head = [{"name": "dense", "config": {"units": 1, "activation": None}}]
    ids = np.random.default_rng(1).integers(1, 60, (4, 6, 40))
    for merge, width in (({"mode": "concat"}, 24), ({"mode": "sum", "target_dim": 8}, 8), ({"mode": "mean", "target_dim": 8}, 8),
                         ({"mode": "max", "target_dim": 8}, 8), ({"mode": "weighted", "target_dim": 8}, 8)):
        cfg = _nmd_merge_builder_config()
        cfg["reliability_model"] = {"merge": merge, "input_shape": width, "hidden_layers": copy.deepcopy(head)}
        plan = build_plan(cfg)
        assert plan.nmd_dims == [16, 8] and plan.nmd_dim == width and plan.reliability[0].cin == width
        out = ofwd.forward(cfg, ofwd.random_weights(cfg, seed=2), ids)
        assert out["nmd"].shape == (4, width) and out["reliability"].shape == (4, 1)
    cfg = _nmd_merge_builder_config()
    cfg["reliability_model"] = {"merge": {"mode": "sum", "target_dim": 8}, "input_shape": 999, "hidden_layers": head}
    with pytest.raises(ValueError, match="does not match"):
        build_plan(cfg)
    cfg = _nmd_merge_builder_config()
    cfg["representation_learner"]["hidden_layers"] = [{"name": "masked_conv1d", "config": {"filters": 16, "kernel_size": 3}},
                                                      {"name": "activation", "config": {"activation": "gelu"}}]
    cfg["classifier"]["input_shape"] = 16
    cfg["reliability_model"] = {"merge": {"mode": "concat"}, "input_shape": 16, "hidden_layers": head}
    with pytest.raises(ValueError, match="no NMD tensor"):
        build_plan(cfg)
    cfg = _nmd_merge_builder_config()
    cfg["reliability_model"] = {"merge": {"mode": "unsupported"}, "hidden_layers": head}
    with pytest.raises(ValueError):
        build_plan(cfg)
    cfg = _nmd_merge_builder_config()                      # differing widths need a target_dim (nmd.py:128-132)
    cfg["reliability_model"] = {"merge": {"mode": "mean"}, "hidden_layers": head}
    with pytest.raises(ValueError, match="target_dim is required"):
        build_plan(cfg)
