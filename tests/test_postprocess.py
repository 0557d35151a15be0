"""pred_to_dict / write_output vs TSVs written by the reference itself on the same seeded
logits (tests/golden/make_golden.py; postprocess/collect.py:247-608)."""
import json
from pathlib import Path

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


def _random_case(rng, n_contigs, n_cls, with_rel, big=False):
    t = rng.integers(1, 40, n_contigs)
    t[rng.integers(0, n_contigs, max(1, n_contigs // 10))] = rng.integers(100, 700 if big else 300, max(1, n_contigs // 10))
    t[: min(3, n_contigs)] = [1, 8, 129][: min(3, n_contigs)]            # pairwise-summation block edges
    n = int(t.sum())
    y = {"prediction": rng.normal(0, 3, (n, n_cls)).astype(np.float32),
         "meta_0": np.array(sum([[f"ctg_{i}___z"] * int(k) for i, k in enumerate(t)], [])),
         "meta_2": np.array(sum([[0] * (int(k) - 1) + [1] for k in t], [])),
         "meta_4": np.array(sum([[1500 * int(k) + 11] * int(k) for k in t], [])),
         "meta_9": np.round(rng.normal(0, 0.1, n), 2)}
    for k in ("meta_5", "meta_6", "meta_7", "meta_8"):
        y[k] = rng.integers(250, 380, n)
    y["meta_5"][: int(t[0])] = 5                                           # first contig mostly N
    if with_rel:
        y["reliability"] = rng.normal(0, 2, (n, 1)).astype(np.float32)
    rep = pd.DataFrame({"contig_id": [f"ctg_{i}___z" for i in range(n_contigs)],
                        "terminal_repeats": [("DTR" if i % 7 == 0 else None) for i in range(n_contigs)],
                        "repeat_length": [(40.0 + i if i % 7 == 0 else np.nan) for i in range(n_contigs)]})
    return y, rep


@pytest.mark.parametrize("n_cls,with_rel,crf", [(6, True, None), (6, False, 2.0), (3, True, None),
                                               (1, True, None), (1, False, 1.5), (4, True, 0.7)])
def test_vectorised_aggregation_equals_per_contig_restatement(n_cls, with_rel, crf, tmp_path):
    """The product computes every statistic for all contigs at once; the oracle loops over contigs like the
    reference.  Same TSV bytes, same per-contig arrays - including fp16 rounding and numpy's summation order."""
    from jaeger_amd import postprocess as P
    from oracle import postprocess as O
    rng = np.random.default_rng(100 * n_cls + (7 if with_rel else 0))
    names = (CLASSES + ["x", "y"])[:max(n_cls, 2)]
    idx = list(range(len(names)))
    for n_contigs in (1, 2, 57, 400):
        y, rep = _random_case(rng, n_contigs, n_cls, with_rel, big=n_contigs == 57)
        kw = dict(class_map={"num_classes": len(names), "class": names, "index": idx}, fsize=1500, term_repeats=rep)
        if crf is not None:
            kw.update(crf_switch_cost=crf, crf_prior="biological")
        outs = {}
        for tag, mod in (("p", P), ("o", O)):
            data, full = mod.pred_to_dict(y, **kw)
            out, out_ph = tmp_path / f"{tag}.tsv", tmp_path / f"{tag}_ph.tsv"
            out_ph.unlink(missing_ok=True)
            if n_cls == 1:
                # the reference's write_output needs a "<viral>_score" column and raises for one-logit heads:
                # compare the summary table through each side's own TSV formatter instead
                df = mod.generate_summary(data, labels=names, indices=idx)
                n = len(df)
                if mod is P:
                    P._to_tsv(df, out)
                else:
                    df.to_csv(out, sep="\t", index=False, float_format="%.3f")
            else:
                n = mod.write_output(data, labels=names, indices=idx, output_table_path=out,
                                     output_phage_table_path=out_ph, reliability_cutoff=0.1, phage_score=1)
            outs[tag] = (data, full, n, out.read_bytes(), out_ph.read_bytes() if out_ph.exists() else None)
        dp, fp, n_p, tsv_p, ph_p = outs["p"]
        do, fo, n_o, tsv_o, ph_o = outs["o"]
        assert n_p == n_o and tsv_p == tsv_o and ph_p == ph_o
        for key in ("pred_sum", "pred_var", "entropy", "energy", "consensus", "host_contam", "prophage_contam", "length"):
            a, b = np.asarray(dp[key]), np.asarray(do[key])
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True), key
        if with_rel:
            assert np.array_equal(np.asarray(dp["ood"]), np.asarray(do["ood"]))
        assert [np.asarray(x).ravel().tolist() for x in dp["frag_pred"]] == [np.asarray(x).ravel().tolist() for x in do["frag_pred"]]
        assert len(fp["predictions"]) == len(fo["predictions"]) == n_contigs
        assert all(np.array_equal(a, b) for a, b in zip(fp["gcs"], fo["gcs"]))


@pytest.mark.parametrize("n_cls,with_rel,crf", [(6, True, None), (3, False, 2.0), (4, True, 0.7)])
def test_batchwise_aggregation_and_table_writer_equal_one_call(n_cls, with_rel, crf, tmp_path):
    """run_core aggregates whole contigs in batches beside the forward and appends their rows to the tables as it goes
    (``merge_data``, ``TableWriter``): same per-contig arrays and the same TSV / phage-TSV bytes as ONE ``pred_to_dict`` +
    ``write_output`` call over everything - also when a batch contributes no row, and with run-length strings prebuilt."""
    from jaeger_amd import postprocess as P
    rng = np.random.default_rng(5 * n_cls + (3 if with_rel else 0))
    names = (CLASSES + ["x", "y"])[:max(n_cls, 2)]
    idx = list(range(len(names)))
    cm = {"num_classes": len(names), "class": names, "index": idx}
    y, rep = _random_case(rng, 211, n_cls, with_rel)
    kw = dict(class_map=cm, fsize=1500, term_repeats=None)
    if crf is not None:
        kw.update(crf_switch_cost=crf, crf_prior="biological")
    whole, full_whole = P.pred_to_dict(y, **kw)
    whole["repeats"] = rep
    one, one_ph = tmp_path / "one.tsv", tmp_path / "one_ph.tsv"
    n_one = P.write_output(whole, labels=names, indices=idx, output_table_path=one, output_phage_table_path=one_ph,
                           reliability_cutoff=0.1, phage_score=1)
    ends = np.nonzero(np.asarray(y["meta_2"]) == 1)[0] + 1            # contig boundaries in window units
    cuts = [0, int(ends[0]), int(ends[0]), int(ends[17]), int(ends[100]), int(ends[-1])]     # a 1-contig batch (all N: no row), an empty one
    w = P.TableWriter(names, idx, tmp_path / "b.tsv", tmp_path / "b_ph.tsv", reliability_cutoff=0.1, phage_score=1)
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        if a == b:
            continue
        data, full = P.pred_to_dict({k: v[a:b] for k, v in y.items()}, **kw)
        data["frag_pred"] = P._Summaries(data["frag_pred"].summaries(P.window_letters(cm)))
        data["repeats"] = rep
        parts.append((data, full))
        w.append(data)
        # rows land in <path>.partial; nothing exists at the final path until close() (a failing run leaves no truncated table)
        assert not (tmp_path / "b.tsv").exists() and (tmp_path / "b.tsv.partial").exists()
    assert w.close() == n_one
    assert not (tmp_path / "b.tsv.partial").exists() and not (tmp_path / "b_ph.tsv.partial").exists()
    w2 = P.TableWriter(names, idx, tmp_path / "c.tsv", tmp_path / "c_ph.tsv", reliability_cutoff=0.1, phage_score=1)
    w2.append(parts[-1][0])
    w2.abort()
    assert not list(tmp_path.glob("c*"))
    assert (tmp_path / "b.tsv").read_bytes() == one.read_bytes()
    assert (tmp_path / "b_ph.tsv").exists() == one_ph.exists()
    if one_ph.exists():
        assert (tmp_path / "b_ph.tsv").read_bytes() == one_ph.read_bytes()
    merged, merged_full = P.merge_data([p[0] for p in parts]), P.merge_data([p[1] for p in parts])
    for key in ("headers", "length", "pred_sum", "pred_var", "entropy", "energy", "consensus", "host_contam",
                "prophage_contam", "per_class_counts"):
        a, b = np.asarray(merged[key]), np.asarray(whole[key])
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")), key
    assert merged["frag_pred"].texts == whole["frag_pred"].summaries(P.window_letters(cm))
    assert len(merged_full["predictions"]) == len(full_whole["predictions"]) == 211
    assert all(np.array_equal(a, b) for a, b in zip(merged_full["gcs"], full_whole["gcs"]))


def test_write_fasta_from_results(tmp_path):
    """--getsequences (collect.py:613-639): records named in the phage table, as read, 70 bases per line."""
    from jaeger_amd.postprocess import write_fasta_from_results
    fasta = tmp_path / "in.fasta"
    fasta.write_text(">a desc\nACGTacgtNN\nGG\n>b\n" + "ACGT" * 40 + "\n>c,x\nTTTT\n")
    (tmp_path / "ph.tsv").write_text("contig_id\tlength\nb\t160\na\t12\nc___x\t4\n")
    n = write_fasta_from_results(fasta, tmp_path / "ph.tsv", tmp_path / "out.fasta")
    assert n == 2
    seq_b = "ACGT" * 40
    assert (tmp_path / "out.fasta").read_text() == ">a\nACGTacgtNNGG\n>b\n" + seq_b[:70] + "\n" + seq_b[70:140] + "\n" + seq_b[140:] + "\n"


def test_native_table_text_equals_pandas():
    """jg_table_format (the library's TSV renderer) against pandas' own to_csv(float_format="%.3f"): floats incl. the
    x.xxx5 ties, NaN, infinities, negative zero; integers; booleans; strings with NaN holes and non-ASCII names; and
    the fall-back for fields the csv writer quotes."""
    from jaeger_amd.postprocess import _tsv_bytes
    rng = np.random.default_rng(5)
    n = 20000
    f = np.concatenate([rng.normal(size=n - 20) * 10.0 ** rng.integers(-6, 9, n - 20),
                        [0.0, -0.0, np.nan, np.inf, -np.inf, 1e15, -3.9e12, 4.1e12, 0.0625, 0.1875, 0.0005, 0.0015, 0.0025,
                         -0.0005, 1e-300, 5e-324, 0.9995, 0.99949999, 2.5e-4, 4e-4]])
    ties = (rng.integers(-10 ** 7, 10 ** 7, n) * 2 + 1) / 2000.0
    names = np.array([f"contig_{i}" if i % 7 else f"cöntig {i}|x=1" for i in range(n)], dtype=object)
    holes = names.copy()
    holes[::3] = np.nan
    holes[1::3] = None
    df = pd.DataFrame({"contig_id": names, "length": rng.integers(-5, 10 ** 12, n), "f": f, "ties": ties,
                       "up": np.nextafter(ties, np.inf), "down": np.nextafter(ties, -np.inf),
                       "f32": rng.normal(size=n).astype(np.float32), "flag": rng.random(n) < 0.5,
                       "small": rng.integers(0, 200, n).astype(np.int16), "terminal_repeats": holes})
    want = df.to_csv(None, sep="\t", index=False, float_format="%.3f").encode("utf-8")
    assert _tsv_bytes(df) == want
    assert _tsv_bytes(df, header=False) == want.split(b"\n", 1)[1]
    assert _tsv_bytes(df.iloc[:0]) == df.iloc[:0].to_csv(None, sep="\t", index=False, float_format="%.3f").encode()
    for bad in ('a"b', "a\tb", "a\nb"):                       # quoted by the csv writer: pandas writes these tables
        q = df.iloc[:50].copy()
        q.iloc[7, 0] = bad
        assert _tsv_bytes(q) == q.to_csv(None, sep="\t", index=False, float_format="%.3f").encode("utf-8")
    mixed = pd.DataFrame({"a": np.array(["x", 3, 2.5], dtype=object), "b": [1.0, 2.0, 3.0]})
    assert _tsv_bytes(mixed) == mixed.to_csv(None, sep="\t", index=False, float_format="%.3f").encode("utf-8")
    ext = pd.DataFrame({"a": pd.array(["x", pd.NA, "z"], dtype="string"), "b": [1.0, 2.0, np.nan],
                        "c": pd.array([1, pd.NA, 3], dtype="Int64")})          # extension dtypes: pandas writes them
    assert _tsv_bytes(ext) == ext.to_csv(None, sep="\t", index=False, float_format="%.3f").encode("utf-8")


def test_native_run_summaries_equal_python_form():
    """jg_run_summaries against the per-run Python strings: ragged contigs, long and one-window runs, classes without a
    letter, an empty letter map; a multi-character letter falls back to the Python form."""
    from jaeger_amd.postprocess import _Runs, _Segments
    rng = np.random.default_rng(3)
    for trial in range(6):
        counts = rng.integers(1, 60, int(rng.integers(1, 2500)))
        calls = rng.integers(0, 4, int(counts.sum()))
        if trial % 2:
            calls = np.repeat(rng.integers(0, 4, calls.size // 7 + 1), 7)[:calls.size]
        runs = _Runs(calls, _Segments(np.cumsum(counts)[:-1], calls.size))
        for letter in ({0: "b", 1: "V", 2: "e", 3: "a"}, {0: "b", 1: "V"}, {}, {0: "xx", 1: "V"}):
            got = runs.summaries(letter)
            assert got == runs.summaries_py(letter) and len(got) == len(counts)
    one = _Runs(np.array([2, 2, 2, 1]), _Segments(np.array([], np.int64), 4))
    assert one.summaries({1: "V", 2: "p"}) == ["3p1V"]


def test_native_segment_reductions_equal_numpy_bit_for_bit():
    """``jg_segment_mean_var`` / ``jg_segment_mean_1d`` (csrc/jg_segments.hip) restate numpy's summation order: the means and
    variances of every contig must equal BOTH the grouped numpy forms they replace and the per-contig calls the reference
    makes (``np.mean(x[a:b], axis=0)`` etc., collect.py:332-356,393-395) bit for bit - every slice length from 1 to 1 100
    (the pairwise scheme's three regimes), one to six columns, f32 and f64."""
    from jaeger_amd.postprocess import _Segments
    rng = np.random.default_rng(11)
    counts = np.concatenate([np.arange(1, 1101), rng.integers(1, 40, 3000), [5000, 8193, 20001]])
    rng.shuffle(counts)
    n = int(counts.sum())
    seg = _Segments(np.cumsum(counts)[:-1], n)
    for dtype in (np.float32, np.float64):
        v = (rng.standard_normal(n) * np.exp(rng.uniform(-6, 6, n))).astype(dtype)
        got = seg.mean_1d(v)
        assert got.dtype == dtype
        np.testing.assert_array_equal(got, seg.mean_1d_numpy(v))
        per = np.array([np.mean(v[a:a + c]) for a, c in zip(seg.first, seg.count)], dtype)
        np.testing.assert_array_equal(got, per)
    for c in (1, 2, 3, 4, 6):
        m = (rng.standard_normal((n, c)) * 7).astype(np.float32)
        mean, var = seg.mean_var_rows(m)
        mean_np, var_np = seg.mean_var_rows_numpy(m)
        np.testing.assert_array_equal(mean, mean_np)
        np.testing.assert_array_equal(var, var_np)
        for a, cnt in list(zip(seg.first, seg.count))[:400]:
            np.testing.assert_array_equal(mean[np.searchsorted(seg.first, a)], np.mean(m[a:a + cnt], axis=0))
            np.testing.assert_array_equal(var[np.searchsorted(seg.first, a)], np.var(m[a:a + cnt], axis=0))
        np.testing.assert_array_equal(seg.mean_flat(m), seg.mean_flat_numpy(m))
    m64 = rng.standard_normal((n, 2))
    np.testing.assert_array_equal(seg.mean_flat(m64), seg.mean_flat_numpy(m64))


@pytest.mark.parametrize("n_cls,with_rel", [(6, True), (3, False), (4, True)])
def test_table_writer_column_path_equals_the_dataframe_path(n_cls, with_rel, tmp_path):
    """``TableWriter`` renders a batch from its column arrays (join by row number or one hash lookup per name, filters as
    masks, text by ``jg_table_format`` for the surviving rows) - byte for byte what the DataFrame path
    (``generate_summary`` -> ``query`` -> ``to_csv``) writes: with and without a repeat table, a repeat table that misses
    contigs or has integer lengths, contigs dropped by the ``N%`` filter, commas in names, NaN scores, the library's
    summary text handed through, ``repeat_rows`` given or not; a repeat table with a repeated name takes the DataFrame."""
    from jaeger_amd import postprocess as P
    rng = np.random.default_rng(31 * n_cls + with_rel)
    names = (CLASSES + ["x", "y"])[:max(n_cls, 2)]
    idx = list(range(len(names)))
    cm = {"num_classes": len(names), "class": names, "index": idx}
    n_contigs = 300
    y, rep = _random_case(rng, n_contigs, n_cls, with_rel)
    hdr = np.asarray(y["meta_0"]).astype(object)
    plain = np.array([h.replace("___z", "") for h in hdr.tolist()], dtype=object)            # names without commas
    ends = np.nonzero(np.asarray(y["meta_2"]) == 1)[0] + 1
    # a few windows of contig 5 mostly N: dropped by the filter in the middle of a batch
    first5 = int(ends[4])
    y["meta_5"][first5:int(ends[5])] = 3
    y["meta_6"][first5:int(ends[5])] = 3
    if n_cls > 1:                                     # (a binary head has no class for a NaN mean: both paths raise)
        y["prediction"][int(ends[9]):int(ends[10])] = np.nan                                  # NaN scores print as nothing
    variants = {
        "none": (hdr, None, False),
        "full": (hdr, rep, False),
        "rows": (hdr, rep, True),
        "holes": (plain, pd.DataFrame({"contig_id": [f"ctg_{i}" for i in range(0, n_contigs, 3)],
                                       "terminal_repeats": ["ITR"] * len(range(0, n_contigs, 3)),
                                       "repeat_length": np.arange(len(range(0, n_contigs, 3)), dtype=np.int64) + 13}), False),
        "repeated": (plain, pd.DataFrame({"contig_id": ["ctg_1", "ctg_1", "ctg_2"], "terminal_repeats": ["DTR", "ITR", None],
                                          "repeat_length": [20.0, 30.0, np.nan]}), False),
    }
    from jaeger_amd.termini import RepeatColumns
    res = np.full((n_contigs, 10), 0, np.int64)
    res[::7, 1] = 40 + np.arange(0, n_contigs, 7)                       # a direct repeat on every seventh contig
    res[::7, 0] = 2 * res[::7, 1]
    res[5::11, 6] = 300                                               # an inverted one on some others
    res[5::11, 5] = 600
    res[3::50, 0] = -1                                                # not scanned
    variants["columns"] = (hdr, RepeatColumns(res, [f"ctg_{i}___z" for i in range(n_contigs)], np.full(n_contigs, 9000)), True)
    for tag, (headers, repeats, by_row) in variants.items():
        y["meta_0"] = headers
        outs = []
        # (columns path, names as bytes): round 6 - the pipeline hands the contigs' names over as spans of the FASTA parser's
        # name buffer (SpanColumn), the class labels and repeat kinds go to the library as codes (EnumColumn)
        for columns_path, as_bytes in ((True, False), (False, False), (True, True)):
            base = tmp_path / f"{tag}_{int(columns_path)}{int(as_bytes)}"
            w = P.TableWriter(names, idx, f"{base}.tsv", f"{base}_ph.tsv", reliability_cutoff=0.1, phage_score=0.5,
                              columns_path=columns_path)
            for a, b in ((0, int(ends[120])), (int(ends[120]), int(ends[121])), (int(ends[121]), int(ends[-1]))):
                extra = {}
                if as_bytes:
                    firsts = np.concatenate(([a], ends[(ends > a) & (ends < b)]))
                    extra["headers"] = P.SpanColumn.from_strings(headers[firsts])
                data, _ = P.pred_to_dict({k: (v[a:b] if not (as_bytes and k == "meta_0") else None) for k, v in y.items()},
                                         class_map=cm, fsize=1500, term_repeats=None, **extra)
                assert isinstance(data["headers"], P.SpanColumn) == as_bytes
                if columns_path:                       # as the pipeline hands it over: the library's text, rows of the join
                    blob = data["frag_pred"].summaries_blob(P.window_letters(cm))
                    assert blob is not None
                    data["frag_pred"] = P._Summaries(blob=blob, n=len(data["headers"]))
                data["repeats"] = repeats
                if by_row and columns_path:
                    frame = repeats.frame() if hasattr(repeats, "frame") else repeats
                    lookup = {c: i for i, c in enumerate(frame["contig_id"])}
                    data["repeat_rows"] = np.array([lookup.get(h, -1) for h in data["headers"].tolist()], dtype=np.int64)
                    data["names_unique"] = True
                w.append(data)
            n = w.close()
            outs.append((n, Path(f"{base}.tsv").read_bytes(),
                         Path(f"{base}_ph.tsv").read_bytes() if Path(f"{base}_ph.tsv").exists() else None))
        assert outs[0][0] == outs[1][0] == outs[2][0] and 0 < outs[0][0] < n_contigs, tag
        assert outs[0][1] == outs[1][1] == outs[2][1], tag
        assert outs[0][2] == outs[1][2] == outs[2][2] and outs[0][2] is not None, tag


def test_table_format_span_columns_and_the_file_writer(tmp_path):
    """``jg_table_format`` with JG_COL_SPANS columns (strings as [begin, end) spans of a byte buffer: record names out of the
    parser's name buffer, labels out of a label blob) prints what the same strings print as a JG_COL_STRING column, for all
    rows and for a row selection; ``jg_table_write`` puts the same bytes into an open file."""
    import ctypes as C
    import os

    from jaeger_amd import _lib
    from jaeger_amd import postprocess as P
    rng = np.random.default_rng(8)
    n = 5000
    names = [f"contig_{i}_" + "x" * int(rng.integers(0, 9)) for i in range(n)]
    names[17] = ""                                                    # an empty name
    names[18] = "caf\u00e9_\u00e0"                                  # UTF-8
    codes = rng.integers(0, 4, n)
    labels = [None, "DTR", "ITR", "LTR_DTR"]
    vals = rng.normal(size=n)
    obj_names = np.array(names, dtype=object)
    obj_kinds = np.array([np.nan if labels[c] is None else labels[c] for c in codes], dtype=object)
    span_names = P.SpanColumn.from_strings(names)
    # spans of a LARGER buffer in another order (as fragment.Names.spans(records) hands them out)
    order = rng.permutation(n)
    inv = np.argsort(order)
    shuffled = P.SpanColumn.from_strings([names[i] for i in order.tolist()])
    picked = P.SpanColumn(shuffled.buf, shuffled.begin[inv], shuffled.end[inv])
    assert picked.tolist() == names and span_names.tolist() == names and np.asarray(picked).tolist() == names
    cols = ["contig_id", "score", "terminal_repeats", "n"]
    rows = np.sort(rng.choice(n, 700, replace=False))
    for sel in (None, rows):
        ref = P._columns_text(cols, [obj_names, vals, obj_kinds, np.arange(n)], sel, header=True)
        for nm in (span_names, picked):
            got = P._columns_text(cols, [nm, vals, P.EnumColumn(codes, labels), np.arange(n)], sel, header=True)
            assert got == ref and got is not None
    assert (P.EnumColumn(codes, labels) == "ITR").tolist() == (codes == 2).tolist()
    assert not (P.EnumColumn(codes, labels) == "phage").any()
    # a name the csv writer would quote: the library is not asked (the caller goes through pandas)
    assert P._columns_text(cols, [P.SpanColumn.from_strings(["a\tb"] + names[1:]), vals, obj_kinds, np.arange(n)], None, True) is None
    # the file writer: same bytes
    lib = _lib.load()
    st = span_names.spans()
    v64 = np.ascontiguousarray(vals, np.float64)
    kinds = np.array([_lib.JG_COL_SPANS, _lib.JG_COL_FLOAT], np.int32)
    col_ptrs = (C.c_void_p * 2)(span_names.buf.ctypes.data, v64.ctypes.data)
    start_ptrs = (C.c_void_p * 2)(st.ctypes.data, None)
    text, size = C.c_void_p(), C.c_int64()
    _lib.check(lib.jg_table_format(2, kinds.ctypes.data, col_ptrs, start_ptrs, None, n, 3, C.byref(text), C.byref(size)))
    body = C.string_at(text, size.value)
    lib.jg_table_free(text)
    fd = os.open(tmp_path / "t.tsv", os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        os.write(fd, b"head\n")
        written = C.c_int64()
        _lib.check(lib.jg_table_write(2, kinds.ctypes.data, col_ptrs, start_ptrs, None, n, 3, fd, C.byref(written)))
    finally:
        os.close(fd)
    assert written.value == len(body) and (tmp_path / "t.tsv").read_bytes() == b"head\n" + body
    assert body.split(b"\n")[17] == b"\t%.3f" % vals[17] and body.count(b"\n") == n
    assert lib.jg_table_write(2, kinds.ctypes.data, col_ptrs, start_ptrs, None, n, 3, -1, C.byref(written)) != 0   # a bad descriptor


def test_byte_columns_print_what_object_columns_print_property():
    """Property test (hypothesis): for arbitrary lists of names - unicode, empty, spaces, commas, triple underscores - a
    ``SpanColumn`` / ``EnumColumn`` table renders byte for byte what the same strings render as object arrays, for every
    row selection; names the csv writer would quote make BOTH forms decline (None: the caller goes through pandas)."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from jaeger_amd import postprocess as P
    alphabet = st.characters(blacklist_categories=("Cs",), blacklist_characters="\x00")
    names_st = st.lists(st.text(alphabet, max_size=12), min_size=1, max_size=40)

    @settings(max_examples=120, deadline=None)
    @given(names_st, st.data())
    def check(names, data):
        n = len(names)
        codes = np.array(data.draw(st.lists(st.integers(0, 3), min_size=n, max_size=n)))
        vals = np.array(data.draw(st.lists(st.one_of(st.floats(-1e6, 1e6, allow_nan=False, width=32), st.just(float("nan"))),
                                           min_size=n, max_size=n)), np.float64)
        rows = sorted(set(data.draw(st.lists(st.integers(0, n - 1), max_size=n))))
        labels = [None, "DTR", "ITR", "LTR_DTR"]
        obj_names = np.array(names, dtype=object)
        obj_kinds = np.array([np.nan if labels[c] is None else labels[c] for c in codes], dtype=object)
        cols = ["contig_id", "score", "terminal_repeats"]
        span = P.SpanColumn.from_strings(names)
        assert span.tolist() == names and len(span) == n
        for sel in (None, np.array(rows, np.int64) if rows else None):
            ref = P._columns_text(cols, [obj_names, vals, obj_kinds], sel, header=True)
            got = P._columns_text(cols, [span, vals, P.EnumColumn(codes, labels)], sel, header=True)
            quoted = any(c in s for s in names for c in '\t"\n\r')
            if quoted:
                assert got is None and ref is None
            else:
                assert got == ref and got is not None

    check()
