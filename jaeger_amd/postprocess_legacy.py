"""Result tables of the legacy ``default`` model: what ``postprocess/collect.py:23-232`` (with ``helpers.py:43-70,
476-564``) produces, computed for ALL contigs at once on :mod:`jaeger_amd.postprocess`'s segment aggregator.

The reference cuts every per-window array into one piece per contig and loops; here the windows stay in their flat
arrays, a contig is a ``(first, count)`` run (``_Segments``), and every statistic is one segment reduction that keeps
numpy's own summation order per contig - the table bytes depend on it (fp16 rounding, ``"%.3f"``).  Behaviour pinned
to the reference by ``tests/golden/postprocess_legacy*.tsv`` (written by the reference's own code).

Input: ``{"y_hat": {"output": (N, 4), "embedding": (N, 128)}, "meta": [meta_0 .. meta_9]}`` as
``JaegerModel.predict`` returns it (``nnlib/inference.py:69-75``); ``meta_2`` flags a contig's last window.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

from .postprocess import _Means, _Runs, _Segments, _to_tsv, softmax_entropy


def window_reliability(embedding: np.ndarray, params: dict) -> np.ndarray:
    """Per-window in-distribution probability from the embedding (``helpers.py:476-492,530-564``): every step is
    row-wise, so all windows go through at once.  ``params``: the closed-form logistic regression on per-row
    standardised features, or a fitted sklearn model on batch-standardised, L2-normalised ones."""
    x = np.asarray(embedding)
    kind = params["type"]
    if kind == "params":
        z = (x - x.mean(axis=1, keepdims=True)) / x.std(axis=1, keepdims=True)
        return (1.0 / (1.0 + np.exp(-(z @ params["coeff"].reshape(-1, 1) + params["intercept"])))).ravel()
    if kind == "sklearn":
        z = (x - params["batch_mean"]) / params["batch_std"]
        z = z / np.linalg.norm(z, 2, axis=1, keepdims=True)
        return params["model"].predict_proba(z)[:, 0]
    raise ValueError(f"unknown ood parameter type {kind!r}")


def pred_to_dict_legacy(config, y_pred, **kwargs):
    """Per-contig statistics of a legacy run.  ``kwargs``: ``fsize``, ``ood_params``, ``term_repeats``;
    ``want_full=False`` skips the per-contig window lists only the ``--window-scores`` writer reads."""
    meta = y_pred["meta"]
    logits = np.asarray(y_pred["y_hat"]["output"])
    n_win = logits.shape[0]
    cuts = np.flatnonzero(np.asarray(meta[2], dtype=np.int32) == 1) + 1
    if cuts.size and cuts[-1] == n_win:
        cuts = cuts[:-1]
    seg = _Segments(cuts, n_win)

    mean, var = seg.mean_var_rows(logits)
    score, spread = mean.astype(np.float16), var.astype(np.float16)
    best = np.argmax(score, axis=1)
    calls = np.argmax(logits, axis=-1)
    n_cls = int(config["num_classes"])
    contig_of = np.repeat(np.arange(seg.n), seg.count)
    counts = np.bincount(contig_of * n_cls + calls, minlength=seg.n * n_cls).reshape(seg.n, n_cls)
    # the phage column's mean under its own variance marks a mixed contig (collect.py:76-77)
    mixed = score[:, 1] < spread[:, 1]

    fsize = kwargs.get("fsize")
    c, g, a, t = (np.asarray(meta[i]).astype(float) for i in (5, 6, 7, 8))
    gc_w = (g + c) / fsize
    n_w = (fsize - (a + t + g + c)) / fsize
    vindex = config.get("vindex", 1)
    data = {
        "headers": np.asarray(meta[0], dtype=np.str_)[seg.first],
        "length": np.asarray(meta[4], dtype=np.int32)[seg.first],
        "consensus": best, "per_class_counts": counts, "pred_sum": score, "pred_var": spread,
        # windows called phage or not; the reference letters a run by comparing the run's BOOLEAN with the phage index
        # (helpers.py:43-70): True only equals index 1, False only index 0
        "frag_pred": _Runs((calls == vindex).astype(np.int32), seg),
        "ood": _Means(seg.mean_1d(window_reliability(y_pred["y_hat"]["embedding"], kwargs.get("ood_params")))),
        "entropy": seg.mean_1d(softmax_entropy(logits)).astype(np.float16),
        "host_contam": mixed & (best == 1), "prophage_contam": mixed & (best == 0),
        "repeats": kwargs.get("term_repeats"), "gc": _Means(seg.mean_1d(gc_w)), "ns": _Means(seg.mean_1d(n_w)),
    }
    full = {"headers": data["headers"], "lengths": data["length"]}
    if kwargs.get("want_full", True):
        full.update(predictions=np.split(logits, cuts, axis=0), gcs=np.split(gc_w, cuts),
                    gc_skews=np.split(np.asarray(meta[9]).astype(float), cuts))
    return data, full


#: second-choice code from the two top-scoring classes (ascending argsort columns 2, 3): (runner-up, winner) pairs the
#: reference recognises (collect.py:141-147); any other pair is code 0
_SECOND_CHOICE = {(2, 1): 1, (3, 1): 2, (0, 1): 3}


def generate_summary_legacy(config, data) -> pd.DataFrame:
    """The result frame; column order is part of the surface (collect.py:104-185)."""
    names = config["labels"]
    n = len(data["headers"])
    cols: dict = {"contig_id": data["headers"], "length": data["length"],
                  "prediction": np.asarray(names, dtype=object)[data["consensus"]], "entropy": data["entropy"]}
    cols["reliability_score"] = data["ood"].means if data.get("has_reliability", True) else np.full(n, "unavailable", object)
    cols["host_contam"] = data["host_contam"]
    cols["prophage_contam"] = data["prophage_contam"]
    if config["model"] == "default":
        cols["G+C"] = data["gc"].means
        cols["N%"] = data["ns"].means
        top2 = np.argsort(data["pred_sum"], axis=1)[:, 2:4]
        code = np.zeros(n, dtype=np.int64)
        for pair, value in _SECOND_CHOICE.items():
            code[(top2 == np.asarray(pair)).all(axis=1)] = value
        second = {int(k): v for k, v in config["second"].items()}
        cols["prediction_2"] = np.asarray([second[i] for i in range(max(second) + 1)], dtype=object)[code]
    for i, label in sorted((int(k), v) for k, v in config["all_labels"].items()):
        cols[f"#_{label}_windows"] = data["per_class_counts"][:, i]
        cols[f"{label}_score"] = data["pred_sum"][:, i]
        cols[f"{label}_var"] = data["pred_var"][:, i]
    vindex = config["vindex"]
    cols["window_summary"] = data["frag_pred"].summaries({1: "V" if vindex == 1 else "n", 0: "V" if vindex == 0 else "n"})
    df = pd.DataFrame(cols).set_index("contig_id")
    repeats = data.get("repeats")
    if repeats is None:
        repeats = pd.DataFrame({"contig_id": pd.Series([], dtype=str), "terminal_repeats": [], "repeat_length": []})
    df = df.join(repeats.set_index("contig_id")[["terminal_repeats", "repeat_length"]], how="left")
    df = df.reset_index(names="contig_id")
    df["contig_id"] = df["contig_id"].str.replace("___", ",")
    return df


def write_output_legacy(config, data: dict, reliability_cutoff: float = 0.5, phage_score: int = 3, **kwargs):
    """Both tables are always written, the phage one possibly header-only (collect.py:188-232)."""
    df = generate_summary_legacy(config, data)
    _to_tsv(df, kwargs.get("output_table_path"))
    keep = (df["prediction"] == "phage") & (df["phage_score"] > phage_score)
    if data.get("has_reliability", True):
        keep &= df["reliability_score"] > reliability_cutoff
    _to_tsv(df[keep], kwargs.get("output_phage_table_path"))
    return len(df)
