"""pred_to_dict / write_output vs TSVs written by the reference itself on the same seeded
logits (tests/golden/make_golden.py; postprocess/collect.py:247-608)."""
import json

import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN

CLASSES = ["bacteria", "phage", "eukarya", "archaea", "plasmid", "virus"]


def _inputs(with_rel):
    z = np.load(GOLDEN / "postprocess_input.npz")
    y = {k: z[k] for k in z.files}
    if not with_rel:
        y.pop("reliability")
    return y, pd.read_csv(GOLDEN / "postprocess_repeats.csv")


@pytest.mark.parametrize("tag", ["rel", "norel"])
def test_tsv_identical_to_reference(tag, tmp_path):
    from jaeger_amd import postprocess as P
    y, rep = _inputs(tag == "rel")
    data, full = P.pred_to_dict(y, class_map={"num_classes": 6}, fsize=1500, term_repeats=rep)
    out, out_ph = tmp_path / "o.tsv", tmp_path / "o_phages.tsv"
    n = P.write_output(data, labels=CLASSES, indices=list(range(6)), output_table_path=out,
                       output_phage_table_path=out_ph, reliability_cutoff=0.1, phage_score=3)
    g = json.loads((GOLDEN / f"postprocess_{tag}.json").read_text())
    assert n == g["n_written"]
    assert out.read_text() == (GOLDEN / f"postprocess_{tag}.tsv").read_text()
    ref_ph = GOLDEN / f"postprocess_{tag}_phages.tsv"
    assert out_ph.exists() == ref_ph.exists()
    if ref_ph.exists():
        assert out_ph.read_text() == ref_ph.read_text()
    np.testing.assert_array_equal(np.asarray(data["consensus"]), g["consensus"])
    np.testing.assert_allclose(np.asarray(data["entropy"], np.float64), g["entropy"])
    np.testing.assert_allclose(np.asarray(data["energy"], np.float64), g["energy"])
    assert len(full["predictions"]) == 5


def test_frac_above_threshold_known_answers():
    """tests/unit/test_postprocess_collect.py:12-21."""
    from jaeger_amd.postprocess import frac_above_threshold
    assert frac_above_threshold(None) == "-"
    assert frac_above_threshold(np.array([])) == "0.00"
    assert frac_above_threshold(np.array([[0.6, 0.3], [0.2, 0.8]]), threshold=0.5) == "0.50"


def test_window_summary_and_unavailable_reliability():
    from jaeger_amd import postprocess as P
    cm = {0: "bacteria", 1: "phage", 2: "virus"}
    assert P.get_window_summary(np.array([0, 0, 1, 1, 1, 2]), cm, ["virus", "phage"]) == "2b3P1V"
    y, rep = _inputs(False)
    data, _ = P.pred_to_dict(y, class_map={"num_classes": 6}, fsize=1500, term_repeats=rep)
    df = P.generate_summary(data, labels=CLASSES, indices=list(range(6)))
    assert (df["reliability_score"] == "unavailable").all()
    assert df["contig_id"].iloc[0] == "contig_0,x"           # '___' -> ',' restored
