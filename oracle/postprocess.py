"""TEST INFRASTRUCTURE - per-contig restatement of the reference's aggregation and TSV writers.

``pred_to_dict`` / ``generate_summary`` / ``write_output`` follow ``postprocess/collect.py:233-608`` and
``postprocess/helpers.py:8-219`` call by call: one numpy call per contig, pandas ``to_csv`` with
``float_format="%.3f"`` - exactly how the reference computes them.  Pinned: the TSVs the reference itself
wrote for seeded inputs (``tests/golden/postprocess_*.tsv``, ``postprocess_crf.tsv``).  The product
(``jaeger_amd/postprocess.py``) computes the same tables for all contigs at once; tests compare the two
byte for byte.  Only tests import this module.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

from .crf import viterbi_decode

# ---- helpers (postprocess/helpers.py) ------------------------------------------------
def find_runs(x):
    """Run-length encode a 1-D array -> (values, lengths, starts) (helpers.py:8-40)."""
    x = np.asanyarray(x)
    if x.ndim != 1:
        raise ValueError("Only 1D arrays are supported")
    n = x.shape[0]
    if n == 0:
        return np.array([], dtype=x.dtype), np.array([], dtype=int), np.array([], dtype=int)
    change = np.concatenate(([True], x[1:] != x[:-1]))
    starts = np.nonzero(change)[0]
    return x[starts], np.diff(np.append(starts, n)), starts


def get_window_summary(x, class_map: dict[int, str], classes: list[str]) -> str:
    """Run-length string of window calls, e.g. ``3b12P2b``; upper case for the classes in
    ``classes`` (helpers.py:73-108)."""
    letter = {k: (v[0].upper() if v.lower() in classes else v[0].lower()) for k, v in class_map.items()}
    values, lengths, _ = find_runs(np.asarray(x).flatten())
    return "".join(f"{n}{letter.get(int(v), '')}" for v, n in zip(values, lengths))


def update_dict(x, num_classes: int = 4) -> dict:
    """Class-count dict with zeros for absent classes (helpers.py:111-127)."""
    return {i: 0 for i in range(num_classes)} | dict(zip(x[0], x[1]))


def softmax_entropy(p, axis=-1, eps=1e-12):
    """helpers.py:175-177 - applied to *raw logits* by the caller (reference quirk)."""
    p = np.clip(p, eps, 1.0)
    return -np.sum(p * np.log2(p), axis=axis)


def binary_entropy(p, eps=1e-12):
    p = np.clip(p, eps, 1 - eps)
    return -(p * np.log2(p) + (1 - p) * np.log2(1 - p))


def logsumexp(x: np.ndarray, axis: int = -1) -> np.ndarray:
    xmax = np.max(x, axis=axis, keepdims=True)
    return xmax.squeeze(axis=axis) + np.log(np.sum(np.exp(x - xmax), axis=axis))


def energy(x: np.ndarray, axis: int = -1) -> np.ndarray:
    """helpers.py:189-219.  Note: only 2-class logits take the softmax branch; any other
    width is treated element-wise as binary logits (-log(1 + e^z)), which is what the
    reference's per-contig ``energy`` column averages for multi-class models."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 0:
        return -logsumexp(np.array([x, 0.0]), axis=-1)
    if x.shape[-1] == 2:
        return -logsumexp(x, axis=axis)
    squeezed = x.squeeze(axis=-1) if x.shape[-1] == 1 else x
    return -logsumexp(np.stack([squeezed, np.zeros_like(squeezed)], axis=-1), axis=-1)


def sigmoid(x):
    return 1 / (1 + np.exp(-x))


def frac_above_threshold(pairs, threshold: float = 0.5, fmt: str = "{:.2f}", none_str: str = "-") -> str:
    """collect.py:233-244."""
    if pairs is None:
        return none_str
    arr = np.asarray(pairs, dtype=float)
    if arr.size == 0:
        return fmt.format(0.0)
    return fmt.format((arr > threshold).mean())


# ---- aggregation (postprocess/collect.py:247-435) --------------------------------------
def pred_to_dict(y_pred: dict, **kwargs) -> tuple[dict, dict]:
    """Window outputs + metadata -> per-contig statistics.

    ``y_pred`` keys as returned by the engine: ``prediction`` (N, C), optional
    ``reliability`` (N, 1), ``meta_0`` header, ``meta_2`` is-last flag, ``meta_4`` contig
    length, ``meta_5..8`` base counts, ``meta_9`` gc skew.  kwargs: ``fsize``,
    ``class_map`` ({"num_classes": ...}), ``term_repeats`` (DataFrame).
    """
    crf_switch_cost = kwargs.get("crf_switch_cost")
    split_flags = np.array(y_pred["meta_2"], dtype=np.int32)
    split_indices = np.where(split_flags == 1)[0] + 1
    classifier_type = "binary" if y_pred["prediction"].shape[-1] == 1 else "softmax"
    if y_pred["prediction"].shape[0] == split_indices[-1]:
        split_indices = split_indices[:-1]

    predictions = np.split(y_pred["prediction"], split_indices, axis=0)
    has_reliability = "reliability" in y_pred
    ood = np.split(y_pred["reliability"], split_indices, axis=0) if has_reliability else None

    headers = np.array([h[0] for h in np.split(np.array(y_pred["meta_0"], dtype=str), split_indices)])
    lengths = np.array([b[0] for b in np.split(np.array(y_pred["meta_4"], dtype=np.int32), split_indices)])
    gc_skews = np.split(np.asarray(y_pred["meta_9"]).astype(float), split_indices)

    # nucleotide content; the reference labels the columns a,t,g,c = meta_7,8,6,5 and only
    # uses their sums (collect.py:319-324)
    a, t, g, c = (np.asarray(y_pred[k]).astype(float) for k in ("meta_7", "meta_8", "meta_6", "meta_5"))
    fsize = kwargs["fsize"]
    ns = np.split((fsize - (a + t + g + c)) / fsize, split_indices)
    gcs = np.split((g + c) / fsize, split_indices)

    pred_sum = np.array([np.squeeze(np.mean(p, axis=0)) for p in predictions], dtype=np.float16)
    pred_var = np.array([np.squeeze(np.var(p, axis=0)) for p in predictions], dtype=np.float16)
    num_classes = kwargs.get("class_map").get("num_classes")
    energy_pred = [energy(p) for p in predictions]
    if classifier_type == "softmax":
        entropy_pred = [softmax_entropy(p) for p in predictions]
        consensus = np.argmax(pred_sum, axis=1)
        if crf_switch_cost is not None:
            # joint MAP decoding of each contig's windows instead of independent argmax (collect.py:269-289,343-346)
            cm = kwargs.get("class_map")
            names = [name for _, name in sorted(zip(cm.get("index"), cm.get("class")), key=lambda t: int(t[0]))]
            from jaeger_amd.postprocess import build_transition_costs     # host logic, pinned in tests/test_crf.py
            costs = build_transition_costs(names, switch_cost=crf_switch_cost, prior=kwargs.get("crf_prior", "biological"),
                                           user_matrix=kwargs.get("crf_transition_matrix"))
            frag_pred = [viterbi_decode(p, crf_switch_cost, costs) for p in predictions]
        else:
            frag_pred = [np.argmax(p, axis=-1) for p in predictions]
        per_class_counts = [update_dict(np.unique(fp, return_counts=True), num_classes) for fp in frag_pred]
        prophage_contam = (pred_sum[:, 1] < pred_var[:, 1]) & (consensus == 0)
        host_contam = (pred_sum[:, 1] < pred_var[:, 1]) & (consensus == 1)
    else:
        entropy_pred = [binary_entropy(p) for p in predictions]
        consensus = np.array([sigmoid(p) for p in pred_sum])
        consensus[consensus > 0.5] = 1.0
        consensus[consensus <= 0.5] = 0.0
        if crf_switch_cost is not None:
            # two-class CRF on stacked [0, z] logits, uniform switch cost (collect.py:365-372)
            frag_pred = [viterbi_decode(np.concatenate([np.zeros_like(p), p], axis=-1), crf_switch_cost)
                         for p in predictions]
        else:
            frag_pred = [(sigmoid(p) > 0.5).astype(int) for p in predictions]
        per_class_counts = [update_dict(np.unique(fp, return_counts=True), num_classes) for fp in frag_pred]
        prophage_contam = (pred_sum < pred_var) & (consensus == 0)
        host_contam = (pred_sum < pred_var) & (consensus == 1)

    if ood is not None:
        ood = np.array([frac_above_threshold(sigmoid(p)) for p in ood], dtype=np.float16)
    entropy_mean = np.array([np.squeeze(np.mean(e)) for e in entropy_pred], dtype=np.float16)
    energy_mean = np.array([np.squeeze(np.mean(e)) for e in energy_pred], dtype=np.float16)

    data = {
        "headers": headers, "length": lengths, "consensus": consensus,
        "per_class_counts": per_class_counts, "pred_sum": pred_sum, "pred_var": pred_var,
        "frag_pred": frag_pred, "ood": ood, "has_reliability": has_reliability,
        "entropy": entropy_mean, "energy": energy_mean, "host_contam": host_contam,
        "prophage_contam": prophage_contam, "repeats": kwargs.get("term_repeats"),
        "gc": gcs, "ns": ns,
    }
    data_full = {"predictions": predictions, "headers": headers, "lengths": lengths,
                 "gc_skews": gc_skews, "gcs": gcs}
    return data, data_full


def generate_summary(data, **kwargs) -> pd.DataFrame:
    """Per-contig summary table (collect.py:438-558); column order is part of the surface."""
    classes_, indices_ = kwargs.get("labels"), kwargs.get("indices")
    class_map = {int(k): v for k, v in zip(indices_, classes_)}
    reliability = data["ood"] if data.get("has_reliability", True) else ["unavailable"] * len(data["headers"])
    columns = {
        "contig_id": data["headers"],
        "length": data["length"],
        "prediction": [class_map[x] for x in data["consensus"]],
        "entropy": data["entropy"],
        "energy": data["energy"],
        "reliability_score": reliability,
        "host_contam": data["host_contam"],
        "prophage_contam": data["prophage_contam"],
        "G+C": [np.mean(x) for x in data["gc"]],
        "N%": [np.mean(x) for x in data["ns"]],
    }
    for i, label in class_map.items():
        columns[f"#_{label}_windows"] = [x[i] for x in data["per_class_counts"]]
    if len(class_map) > 2:
        for i, label in class_map.items():
            columns[f"{label}_score"] = [x[i] for x in data["pred_sum"]]
            columns[f"{label}_var"] = [x[i] for x in data["pred_var"]]
    else:
        columns["score"] = data["pred_sum"]
        columns["var"] = data["pred_var"]
    columns["window_summary"] = [get_window_summary(x, class_map=class_map, classes=["virus", "phage"])
                                 for x in data["frag_pred"]]
    df = pd.DataFrame(columns)
    repeats = data.get("repeats")
    if repeats is None:
        repeats = pd.DataFrame({"contig_id": [], "terminal_repeats": [], "repeat_length": []})
    df = pd.merge(left=df, right=repeats[["contig_id", "terminal_repeats", "repeat_length"]],
                  on="contig_id", how="left")
    refined = kwargs.get("refined_contig")
    if refined is not None:
        df = pd.merge(left=df, right=refined[["contig_id", "contig_call", "contig_top_logit", "contig_margin",
                                              "n_windows_used", "n_merged_windows"]], on="contig_id", how="left")
    df["contig_id"] = df["contig_id"].str.replace("___", ",")
    return df


def write_output(data: dict, reliability_cutoff: float = 0.5, phage_score=1, **kwargs) -> int:
    """Write ``<base>.tsv`` and (if non-empty) ``<base>_phages.tsv`` (collect.py:561-608)."""
    df = generate_summary(data, **kwargs).query("`N%` < 0.3")
    df.to_csv(kwargs.get("output_table_path"), sep="\t", index=False, float_format="%.3f")
    classes = kwargs.get("labels", [])
    lower = [label.lower() for label in classes]
    viral_label = "phage"
    if "phage" in lower:
        viral_label = classes[lower.index("phage")]
    elif "virus" in lower:
        viral_label = classes[lower.index("virus")]
    clause = f" and (reliability_score > {reliability_cutoff})" if data.get("has_reliability", True) else ""
    phage_df = df.query(f'(prediction == "{viral_label}") and ({viral_label}_score > {phage_score}){clause}')
    if not phage_df.empty:
        phage_df.to_csv(kwargs.get("output_phage_table_path"), sep="\t", index=False, float_format="%.3f")
    return len(df)
