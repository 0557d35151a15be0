"""`--crf` window decoding: the native decoder (jg_viterbi_decode, host code of libjaeger_hip.so) and the
numpy oracle against paths decoded by the reference itself (tests/golden/make_golden_crf.py) and the known
answers of the reference's tests/unit/test_viterbi_decode.py:28-117; transition costs against the
reference's matrices; pred_to_dict + write_output under --crf byte-identical to the reference's TSV."""
import json

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN

SIX = ["bacteria", "phage", "eukarya", "archaea", "plasmid", "virus"]
CASES = json.loads((GOLDEN / "crf_cases.json").read_text())


def _decoders():
    from jaeger_amd.postprocess import viterbi_decode as native
    from oracle.crf import viterbi_decode as oracle
    return {"native": native, "oracle": oracle}


def _logits(dominant, margin=10.0, n_classes=3):
    z = np.zeros((len(dominant), n_classes))
    z[np.arange(len(dominant)), dominant] = margin
    return z


@pytest.mark.parametrize("which", ["native", "oracle"])
@pytest.mark.parametrize("i", range(len(CASES)))
def test_paths_match_reference(which, i):
    case, dec = CASES[i], _decoders()[which]
    z = np.asarray(case["logits"], np.float32)
    if case["kind"] == "binary":
        got = dec(np.concatenate([np.zeros_like(z), z], axis=-1), case["switch_cost"])
    else:
        got = dec(z, case["switch_cost"], np.asarray(case["costs"]))
    np.testing.assert_array_equal(got, case["path"])


@pytest.mark.parametrize("i", [k for k, c in enumerate(CASES) if c["kind"] == "softmax"])
def test_transition_costs_match_reference(i):
    from jaeger_amd.postprocess import build_transition_costs
    case = CASES[i]
    costs = build_transition_costs(case["names"], case["switch_cost"], case["prior"], case["user_matrix"])
    np.testing.assert_array_equal(costs, np.asarray(case["costs"]))


@pytest.mark.parametrize("which", ["native", "oracle"])
def test_reference_known_answers(which):
    dec = _decoders()[which]
    rng = np.random.default_rng(42)
    z = rng.normal(size=(25, 6))
    np.testing.assert_array_equal(dec(z, 0.0), np.argmax(z, axis=-1))                      # zero cost = argmax
    flip = [0, 0, 0, 1, 0, 0, 0]
    np.testing.assert_array_equal(dec(_logits(flip), 6.0), np.zeros(7, dtype=int))         # singleton suppressed
    np.testing.assert_array_equal(dec(_logits(flip), 4.0), flip)                           # ... kept below threshold
    run = [0, 0, 1, 1, 1, 0, 0]
    np.testing.assert_array_equal(dec(_logits(run), 6.0), run)                             # sustained run preserved
    np.testing.assert_array_equal(dec(np.array([[1.0, 5.0, 2.0]]), 2.0), [1])              # single window
    out = dec(rng.normal(size=(13, 4)), 2.0)
    assert out.shape == (13,) and np.issubdtype(out.dtype, np.integer)
    zb = np.array([[-1.5], [-1.5], [1.5], [-1.5], [-1.5]])                                 # binary head, [0, z] stacking
    stacked = np.concatenate([np.zeros_like(zb), zb], axis=-1)
    np.testing.assert_array_equal(dec(stacked, 0.0), (zb[:, 0] > 0).astype(int))
    np.testing.assert_array_equal(dec(stacked, 5.0), np.zeros(5, dtype=int))


def test_chains_decode_independently():
    from jaeger_amd.postprocess import viterbi_decode, viterbi_decode_chains
    rng = np.random.default_rng(3)
    lens = [1, 7, 0, 30, 2]
    z = rng.normal(0, 2, (sum(lens), 6)).astype(np.float32)
    first = np.concatenate(([0], np.cumsum(lens)))
    got = viterbi_decode_chains(z, first, 2.0)
    for a, b in zip(first[:-1], first[1:]):
        if b > a:
            np.testing.assert_array_equal(got[a:b], viterbi_decode(z[a:b], 2.0))
    with pytest.raises(Exception):
        viterbi_decode_chains(z, np.array([0, 5, 3, len(z)]), 2.0)                         # malformed chain table


def test_crf_tsv_identical_to_reference(tmp_path):
    from jaeger_amd import postprocess as P
    zf = np.load(GOLDEN / "postprocess_input.npz")
    y = {k: zf[k] for k in zf.files}
    rep = pd.read_csv(GOLDEN / "postprocess_repeats.csv")
    cm = {"num_classes": 6, "class": SIX, "index": list(range(6))}
    data, _ = P.pred_to_dict(y, class_map=cm, fsize=1500, term_repeats=rep, crf_switch_cost=2.0, crf_prior="biological")
    out, out_ph = tmp_path / "o.tsv", tmp_path / "o_phages.tsv"
    P.write_output(data, labels=SIX, indices=list(range(6)), output_table_path=out, output_phage_table_path=out_ph,
                   reliability_cutoff=0.1, phage_score=3)
    assert out.read_text() == (GOLDEN / "postprocess_crf.tsv").read_text()
    assert out.read_text() != (GOLDEN / "postprocess_rel.tsv").read_text()                # the decode changed calls
    assert out_ph.exists() == (GOLDEN / "postprocess_crf_phages.tsv").exists()
