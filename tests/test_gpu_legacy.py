"""BASELINE config #1 on the MI355X: the reference's legacy ``default`` model with its own shipped
weights on its bundled ``test_contigs.fasta`` (9 contigs, fsize 2000 / stride 1500 -> 135 windows)."""
import json

import numpy as np
import pandas as pd
import pytest
from click.testing import CliRunner
from conftest import GOLDEN, oracle_term_repeats

pytestmark = pytest.mark.gpu
LEGACY = GOLDEN / "legacy_data"
H5 = LEGACY / "models" / "default" / "WRes_1024.h5"


def _windows(fsize=2000, stride=1500):
    from jaeger_amd.fragment import read_fasta
    from oracle import fragmenter as ofr
    records = [(n, s.decode()) for n, s in read_fasta(str(GOLDEN / "test_contigs.fasta"))]
    return records, [r.split(",") for r in ofr.fragment_strings(records, fsize, stride)]


def _oracle_ids(wins, fsize):
    from jaeger_amd.maps import V1_TRIMER_INT
    from oracle import encoder as oenc
    return oenc.encode_windows(wins, fsize, codon_id=[v - 1 for v in V1_TRIMER_INT], masking=True)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_legacy_forward_parity_real_weights(precision):
    from jaeger_amd import legacy
    from jaeger_amd.fragment import concat_records
    from oracle import legacy as ol
    w = legacy.load_legacy_h5(H5)
    records, rows = _windows()
    assert len(rows) == 135
    wins = [r[0] for r in rows]
    ids = _oracle_ids(wins, 2000)
    assert ids.shape == (135, 6, 665) and ids.max() == 21
    ref = ol.forward(w, ids)
    eng = legacy.LegacyHipEngine(w, precision=precision)
    assert eng.model.precision == precision
    got_ids = eng.forward_ids(ids)
    # fused path: window table on the raw contigs
    bases, offsets = concat_records([s.encode() for _, s in records])
    starts, lens = [], []
    for ci, (_, s) in enumerate(records):
        for st in range(0, len(s) - 1999, 1500):
            starts.append(offsets[ci] + st)
            lens.append(2000)
    got = eng.predict_windows(bases, np.array(starts), np.array(lens), 2000)
    again = eng.forward_ids(ids)
    eng.close()
    for k in ("output", "embedding"):
        np.testing.assert_array_equal(again[k], got_ids[k])          # repeatable bit for bit
        err = float(np.abs(got[k] - ref[k]).max())
        print(k, f"{err:.2e}", float(np.abs(ref[k]).max()))
        if k == "output":
            assert err <= 1e-4, (k, err)
        else:                                                          # 1e-4 absolute where |ref| <= 8, 1.25e-5 relative above
            g64, r64 = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64)
            small = np.abs(r64) <= 8.0
            assert not small.any() or np.abs(g64 - r64)[small].max() <= 1e-4, k
            assert small.all() or (np.abs(g64 - r64)[~small] / np.abs(r64)[~small]).max() <= 1.25e-5, k
        np.testing.assert_array_equal(got[k], got_ids[k])


def test_legacy_lowercase_breaks_codons():
    """The v1 string processor does not upper-case (preprocess/v1/convert.py:84-99): a soft-masked
    base voids the codons it touches."""
    from jaeger_amd import legacy
    rng = np.random.Generator(np.random.PCG64(3))
    seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), 2000).copy()
    seq[300:340] |= 0x20
    eng = legacy.LegacyHipEngine(legacy.random_weights(3))
    ids, _ = eng.device.encode(seq, np.array([0]), np.array([2000]), 2000, eng.lut, flags=eng.encode_flags | 1)
    eng.close()
    ref = _oracle_ids([seq.tobytes()], 2000)
    np.testing.assert_array_equal(ids, ref)
    assert (ref[0, 0, 100:113] == 0).all() and ref[0, 0, 98] != 0


def test_cli_config1_default_model(tmp_path):
    """`jaeger predict -i test_contigs.fasta -o out -m default` end to end vs the oracle pipeline."""
    import joblib

    from jaeger_amd import legacy
    from jaeger_amd.cli import main
    from jaeger_amd.postprocess_legacy import pred_to_dict_legacy, write_output_legacy
    from oracle import legacy as ol
    r = CliRunner().invoke(main, ["predict", "-i", str(GOLDEN / "test_contigs.fasta"), "-o", str(tmp_path / "out"),
                                  "--legacy-data", str(LEGACY), "--no-dustmask", "--window-scores"])
    assert r.exit_code == 0, r.output
    out = tmp_path / "out" / "default"
    w = legacy.load_legacy_h5(H5)
    records, rows = _windows()
    res = ol.forward(w, _oracle_ids([r_[0] for r_ in rows], 2000))
    meta = [np.array([r_[1] for r_ in rows])] + [np.array([int(r_[j]) for r_ in rows]) for j in range(2, 10)] + \
           [np.array([float(r_[10]) for r_ in rows])]
    conf = json.loads((LEGACY / "config.json").read_text())["default"]
    conf["model"] = "default"
    conf["labels"] = [v for _, v in conf["default_labels"].items()]
    ood = {"type": "sklearn", "model": joblib.load(LEGACY / "models/default/LR_ood_4_class_default.pkl"),
           "batch_mean": np.load(LEGACY / "models/default/batch_means.npy"),
           "batch_std": np.load(LEGACY / "models/default/batch_std.npy")}
    data, _ = pred_to_dict_legacy(conf, {"y_hat": res, "meta": meta}, model="default", fsize=2000, ood_params=ood,
                                  term_repeats=oracle_term_repeats(records, 2000))
    write_output_legacy(conf, data, output_table_path=tmp_path / "exp.tsv",
                        output_phage_table_path=tmp_path / "exp_ph.tsv", reliability_cutoff=0.1, phage_score=3)
    got, exp = pd.read_csv(out / "test_contigs_jaeger.tsv", sep="\t"), pd.read_csv(tmp_path / "exp.tsv", sep="\t")
    assert list(got.columns) == list(exp.columns) and len(got) == 9
    for col in exp.columns:
        if exp[col].dtype.kind == "f":
            np.testing.assert_allclose(got[col].to_numpy(float), exp[col].to_numpy(float), rtol=2e-3, atol=2e-3,
                                       equal_nan=True, err_msg=col)
        else:
            assert got[col].astype(str).tolist() == exp[col].astype(str).tolist(), col
    print(got[["contig_id", "length", "prediction", "phage_score", "reliability_score", "window_summary"]])
    assert (out / "test_contigs_phages_jaeger.tsv").exists()
    assert (out / "test_contigs_default_window_scores.npz").exists()


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_legacy_gpu_vs_reference_savedmodel(precision):
    """The HIP legacy path against the reference's OWN graph + weights: logits and embeddings of the bundled
    ``jaeger_fragment_graph`` SavedModel, executed TF-free (tests/golden/make_golden_savedmodel.py), on the 135
    windows of test_contigs.fasta.  Gate: 1e-4 absolute against the graph's exact (f64) value."""
    from jaeger_amd import legacy
    gold = np.load(GOLDEN / "legacy_savedmodel_logits.npz")
    eng = legacy.LegacyHipEngine(H5, precision=precision)
    got = eng.forward_ids(gold["ids"])
    eng.close()
    e_out = float(np.abs(got["output"] - gold["output_f64"]).max())
    e_emb = float(np.abs(got["embedding"] - gold["embedding_f64"]).max())
    print(f"{precision}: logits {e_out:.2e} (|max| {np.abs(gold['output_f64']).max():.1f}), embedding {e_emb:.2e}")
    assert e_out < 1e-4 and e_emb < 1e-4
