"""Prophage-segmentation inputs (logits_to_df_v2) against frames produced by the reference's own function
(tests/golden/make_golden_prophage.py): softmax, host call, width-4 box tracks, gc and scaled gc-skew."""
import numpy as np

from conftest import GOLDEN


def test_logits_to_df_v2_matches_reference():
    from jaeger_amd.prophage_inputs import logits_to_df_v2
    g = np.load(GOLDEN / "prophage_inputs.npz", allow_pickle=False)
    classes = ["bacteria", "phage", "eukarya", "archaea", "plasmid", "virus"]
    n = len(g["n_win"])
    out = logits_to_df_v2({"class": classes, "index": list(range(6))}, {"lc": 500_000, "stride": 1500, "fsize": 2000},
                          g["headers"], [g[f"pred_{i}"].copy() for i in range(n)], g["lengths"],
                          [g[f"gc_skew_{i}"].copy() for i in range(n)], [g[f"gc_{i}"].copy() for i in range(n)])
    assert list(out.keys()) == g["kept"].tolist() == ["ctg1", "ctg3"]            # length >= --lc only
    for key, (df, host, length) in out.items():
        assert list(df.columns) == g[f"cols_{key}"].tolist()
        assert host == str(g[f"host_{key}"])
        np.testing.assert_array_equal(df.to_numpy(dtype=np.float64), g[f"df_{key}"])
        assert df["gc_skew"].min() == -1.0 and abs(df["gc_skew"].max() - 1.0) < 1e-12
    assert out["ctg1"][1] in classes and out["ctg1"][2] == 600_123


def test_flat_columns_equal_a_per_contig_evaluation():
    """The flat-column form against a contig-by-contig evaluation written out here (softmax without a shift, width-4 box SUM of
    every class track, width-10 mean of the gc-skew track min-max scaled to [-1, 1], window starts clamped to the contig) on
    random contigs incl. ones with fewer windows than either filter is wide, several kept / dropped by ``lc``, a permuted class
    index: bit-identical columns, same hosts."""
    from jaeger_amd.prophage_inputs import logits_to_df_v2
    rng = np.random.default_rng(12)
    classes = ["bacteria", "phage", "eukarya", "archaea"]
    index = [2, 0, 3, 1]                                   # class -> logit column
    ns = [1, 2, 3, 5, 9, 10, 11, 40, 333, 7]
    lens = [int(n * 1500 + rng.integers(0, 1400)) for n in ns]
    lens[4] = 100                                          # below lc: dropped
    hdr = [f"c{i}" for i in range(len(ns))]
    preds = [rng.normal(scale=3.0, size=(n, 4)) for n in ns]
    skews = [rng.normal(size=n) for n in ns]
    gcs = [rng.random(n) for n in ns]
    out = logits_to_df_v2({"class": classes, "index": index}, {"lc": 1000, "stride": 1500, "fsize": 2000}, hdr,
                          [p.copy() for p in preds], lens, [s.copy() for s in skews], [g.copy() for g in gcs])
    assert list(out) == [h for h, ln in zip(hdr, lens) if ln >= 1000] and "c4" not in out
    for i, h in enumerate(hdr):
        if h not in out:
            continue
        df, host, length = out[h]
        n = ns[i]
        v = np.exp(preds[i]) / np.sum(np.exp(preds[i]), axis=1).reshape(-1, 1)
        assert list(df.columns) == classes + ["length", "gc", "gc_skew"] and len(df) == n and length == lens[i]
        assert host == classes[index.index(int(np.argmax(np.mean(v, axis=0))))]
        for name, col in zip(classes, index):
            np.testing.assert_array_equal(df[name].to_numpy(), np.convolve(v[:, col], np.ones(4), mode="same")[:n])
        assert df["length"].tolist() == [min(j * 1500, lens[i]) for j in range(n)]
        np.testing.assert_array_equal(df["gc"].to_numpy(), gcs[i])
        s = np.convolve(skews[i], np.ones(10) / 10, mode="same")[:n].copy()
        s += -np.min(s)
        with np.errstate(invalid="ignore", divide="ignore"):
            s /= np.max(s) / 2
        s += -1
        np.testing.assert_array_equal(df["gc_skew"].to_numpy(), s)
