"""Nucleotide (two-strand) input: the oracle's encoder against the reference's known answers and its own literal form,
the plan / program of the branched model (``train_config/nn_config_500bp_dvf.yaml`` = tests/golden/dvf500_project.yaml)."""
import numpy as np
import pytest
from conftest import load_model_cfg

from oracle import strands as ost


def test_reference_lookup_known_answers():
    """tests/unit/test_seqops_encode.py:15-21 in the reference: A,T,G,C complement to T,A,C,G; A,G,C,T map to 0,1,2,3."""
    assert [ost._COMPLEMENT[b] for b in "ATGC"] == list("TACG")
    assert [ost._NUC[b] for b in "AGCT"] == [0, 1, 2, 3]
    ids = ost.encode_nucleotide([b"AGCT"], 4)
    assert ids[0, 0].tolist() == [1, 2, 3, 4]                    # id + 1
    assert ids[0, 1].tolist() == [1, 2, 3, 4]                    # AGCT is its own reverse complement
    assert ost.encode_nucleotide([b"AAGN"], 4)[0].tolist() == [[1, 1, 2, 0], [0, 3, 4, 4]]     # N -> "N" -> -1 -> 0


def test_vectorised_encoder_equals_literal():
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgtNnRY-", np.uint8)
    windows = [alphabet[rng.integers(0, alphabet.size, n)].tobytes() for n in (1, 7, 10, 33, 500, 501, 640)]
    for crop in (10, 500):
        ids = ost.encode_nucleotide(windows, crop)
        assert ids.shape == (len(windows), 2, min(max(len(w) for w in windows), crop))
        for w, row in zip(windows, ids):
            lit = ost.encode_nucleotide_literal(w.decode(), crop)                 # (2, n, 4) one-hot
            n = lit.shape[1]
            want = np.where(lit.any(-1), lit.argmax(-1) + 1, 0)
            np.testing.assert_array_equal(row[:, :n], want)
            assert not row[:, n:].any()                                           # zero padding
    # case never matters for nucleotide ids (both cases are keys of the lookup, encode.py:36-41)
    np.testing.assert_array_equal(ost.encode_nucleotide([b"acgtn"], 5), ost.encode_nucleotide([b"ACGTN"], 5))


def test_plan_and_program_of_the_branched_model():
    from jaeger_amd import _lib as L
    from jaeger_amd.plan import UnsupportedLayer, build_plan, weight_shapes
    from jaeger_amd.program import compile_plan
    cfg = load_model_cfg("dvf500")
    plan = build_plan(cfg)
    assert (plan.strands, plan.merge, plan.pooling, plan.vocab, plan.embedding_dim) == (2, "average", "max1d", 5, 4)
    sp = plan.string_processor
    assert sp["input_type"] == "nucleotide" and sp["crop_size_nt"] == 500 and "input_type_note" in sp
    assert weight_shapes(plan) == ost.weight_specs(cfg)                           # product and checker name the same tensors
    prog = compile_plan(plan, ost.random_weights(cfg))
    kinds = [op.kind for op in prog.ops]
    assert kinds == [L.OP_CONV, L.OP_POOL, L.OP_DENSE, L.OP_DENSE, L.OP_STRANDS]
    conv, pool, last = prog.ops[0], prog.ops[1], prog.ops[-1]
    assert (conv.k, conv.cin, conv.cout, conv.in_mask, conv.padding) == (10, 4, 500, L.JG_BUF_NONE, L.PAD_VALID)
    assert pool.in_mask == L.JG_BUF_NONE and pool.arg == L.POOL_MAX
    assert (last.k, last.arg) == (2, L.MERGE_AVERAGE) and prog.strands == 2
    typed = dict(cfg, embedding=dict(cfg["embedding"], type="nucleotide"))        # what nnlib/inference.py:443-444 reads
    assert "input_type_note" not in build_plan(typed).string_processor
    # what stays outside: a branched section on translated input, masked layers inside a strand
    with pytest.raises(UnsupportedLayer):
        build_plan(dict(cfg, embedding=dict(cfg["embedding"], input_type="translated")))
    bad = dict(cfg, representation_learner={"branch": {"hidden_layers": [
        {"name": "masked_conv1d", "config": {"filters": 8, "kernel_size": 3}}], "pooling": "max1d"}})
    with pytest.raises(UnsupportedLayer):
        build_plan(bad)


def test_oracle_forward_shapes_and_merge_methods():
    cfg = load_model_cfg("dvf500")
    w = ost.random_weights(cfg, 3)
    rng = np.random.default_rng(2)
    ids = rng.integers(0, 5, (6, 2, 64)).astype(np.uint8)
    out = ost.forward(cfg, w, ids)
    assert out["prediction"].shape == (6, 3) and out["embedding"].shape == (6, 500)
    # swapping the strands changes nothing under a symmetric merge; "sum" is twice "average"
    np.testing.assert_allclose(ost.forward(cfg, w, ids[:, ::-1])["prediction"], out["prediction"], atol=1e-6)
    import copy
    cs = copy.deepcopy(cfg)
    cs["classifier"]["branch"]["hidden_layers"][-1]["config"]["method"] = "sum"
    np.testing.assert_allclose(ost.forward(cs, w, ids)["prediction"], 2 * out["prediction"], rtol=1e-6, atol=1e-6)
    # "concat" (builder.py:1262-1265, round 6): the two strands' head outputs side by side - their mean is the "average" merge
    cs["classifier"]["branch"]["hidden_layers"][-1]["config"]["method"] = "concat"
    cat = ost.forward(cs, w, ids)["prediction"]
    assert cat.shape == (6, 6)
    np.testing.assert_allclose((cat[:, :3] + cat[:, 3:]) / 2, out["prediction"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ost.forward(cs, w, ids[:, ::-1])["prediction"], np.concatenate([cat[:, 3:], cat[:, :3]], axis=1), atol=1e-6)
    from jaeger_amd import _lib as L
    from jaeger_amd.plan import build_plan
    from jaeger_amd.program import compile_plan
    from jaeger_amd.weights import random_weights
    plan = build_plan(cs)
    assert plan.merge == "concat" and compile_plan(plan, random_weights(plan, 1)).ops[-1].arg == L.MERGE_CONCAT
    cs["classifier"]["branch"]["hidden_layers"][-1]["config"]["method"] = "median"
    with pytest.raises(ValueError, match="Unknown merge method"):
        build_plan(cs)
    with pytest.raises(ValueError, match="Unknown merge method"):
        ost.forward(cs, w, ids)


def test_branched_model_with_an_nmd_merge_is_refused():
    """the reference hands ``nmd_merge`` only to the non-branched representation learner (builder.py:486-502; its
    parallel_branches path builds the blocks with ``nmd_merge=None``, :1121) - a branched plan with a reliability head / merge
    config must not compile to some other graph: it is refused."""
    import copy

    import pytest
    from conftest import load_model_cfg
    from jaeger_amd.plan import UnsupportedLayer, build_plan
    cfg = copy.deepcopy(load_model_cfg("dvf500"))
    build_plan(cfg)                                                    # as it ships: fine
    cfg["reliability_model"] = {"merge": {"mode": "sum", "target_dim": 8},
                                "hidden_layers": [{"name": "dense", "config": {"units": 1, "activation": None}}]}
    with pytest.raises(UnsupportedLayer, match="reliability head on a branched model"):
        build_plan(cfg)
