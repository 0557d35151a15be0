"""Per-contig aggregation and TSV writers of the legacy ``default`` model
(``postprocess/collect.py:23-232``, helpers ``postprocess/helpers.py:43-70,495-564``).

``y_pred`` is keyed like ``JaegerModel.predict`` (``nnlib/inference.py:69-75``):
``{"y_hat": {"output": (N, 4), "embedding": (N, 128)}, "meta": [meta_0 .. meta_9]}``.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

from .postprocess import find_runs, softmax_entropy, update_dict


def get_window_summary_legacy(x, phage_pos: int) -> str:
    """``3n12V2n``-style run-length string (helpers.py:43-70)."""
    x = np.asarray(x).flatten()
    items, run_length, _ = find_runs(x == phage_pos)
    return "".join(f"{n}{'V' if it == phage_pos else 'n'}" for it, n in zip(items, run_length))


def normalize(x):
    """helpers.py:476-492: per-row standardisation."""
    x = np.asarray(x)
    return (x - x.mean(axis=1).reshape(-1, 1)) / x.std(axis=1).reshape(-1, 1)


def ood_predict_default(x_features, params):
    """helpers.py:530-564."""
    if params["type"] == "params":
        feats = normalize(x_features)
        logits = np.dot(feats, params["coeff"].reshape(-1, 1)) + params["intercept"]
        return (1 / (1 + np.exp(-logits))).flatten(), logits
    if params["type"] == "sklearn":
        feats = (x_features - params["batch_mean"]) / params["batch_std"]
        feats = feats / np.linalg.norm(feats, 2, axis=1).reshape(-1, 1)
        return params["model"].predict_proba(feats)[:, 0], 0
    raise ValueError(f"unknown ood parameter type {params['type']!r}")


def pred_to_dict_legacy(config, y_pred, **kwargs):
    """collect.py:23-101."""
    meta = y_pred["meta"]
    split_indices = np.where(np.array(meta[2], dtype=np.int32) == 1)[0] + 1
    output = y_pred["y_hat"]["output"]
    if output.shape[0] == split_indices[-1]:
        split_indices = split_indices[:-1]
    predictions = np.split(output, split_indices, axis=0)
    ood = np.split(y_pred["y_hat"]["embedding"], split_indices, axis=0)
    ood = [ood_predict_default(x, kwargs.get("ood_params"))[0] for x in ood]
    headers = np.split(np.array(meta[0], dtype=np.str_), split_indices, axis=0)
    lengths = np.split(np.array(meta[4], dtype=np.int32), split_indices, axis=0)
    gc_skews = np.split(np.asarray(meta[-1]).astype(float), split_indices, axis=0)
    g, c = np.asarray(meta[-4]).astype(float), np.asarray(meta[-5]).astype(float)
    a, t = np.asarray(meta[-3]).astype(float), np.asarray(meta[-2]).astype(float)
    fsize = kwargs.get("fsize")
    ns = np.split((fsize - (a + t + g + c)) / fsize, split_indices, axis=0)
    gcs = np.split((g + c) / fsize, split_indices, axis=0)
    lengths = np.array([x[0] for x in lengths])
    headers = np.array([x[0] for x in headers])
    pred_sum = np.array([np.mean(x, axis=0) for x in predictions], np.float16)
    pred_var = np.array([np.var(x, axis=0) for x in predictions], np.float16)
    consensus = np.argmax(pred_sum, axis=1)
    frag_pred = [np.argmax(x, axis=-1) for x in predictions]
    per_class_counts = [update_dict(np.unique(x, return_counts=True), config["num_classes"]) for x in frag_pred]
    entropy_pred = [softmax_entropy(x) for x in predictions]
    entropy_mean = np.array([np.mean(x, axis=0) for x in entropy_pred], np.float16)
    prophage_contam = (pred_sum[:, 1] < pred_var[:, 1]) * (consensus == 0)
    host_contam = (pred_sum[:, 1] < pred_var[:, 1]) * (consensus == 1)
    data = {"headers": headers, "length": lengths, "consensus": consensus, "per_class_counts": per_class_counts,
            "pred_sum": pred_sum, "pred_var": pred_var, "frag_pred": frag_pred, "ood": ood,
            "entropy": entropy_mean, "host_contam": host_contam, "prophage_contam": prophage_contam,
            "repeats": kwargs.get("term_repeats"), "gc": gcs, "ns": ns}
    data_full = {"predictions": predictions, "headers": headers, "lengths": lengths, "gc_skews": gc_skews,
                 "gcs": gcs}
    return data, data_full


def generate_summary_legacy(config, data) -> pd.DataFrame:
    """collect.py:104-185; column order is part of the surface."""
    class_map = config["labels"]
    lab = {int(k): v for k, v in config["all_labels"].items()}
    if data.get("has_reliability", True):
        reliability = [np.mean(x) for x in data["ood"]]
    else:
        reliability = ["unavailable"] * len(data["headers"])
    columns = {
        "contig_id": data["headers"], "length": data["length"],
        "prediction": [class_map[x] for x in data["consensus"]], "entropy": data["entropy"],
        "reliability_score": reliability, "host_contam": data["host_contam"],
        "prophage_contam": data["prophage_contam"],
    }
    if config["model"] == "default":
        columns["G+C"] = [np.mean(x) for x in data["gc"]]
        columns["N%"] = [np.mean(x) for x in data["ns"]]
        order = np.argsort(data["pred_sum"], axis=1)[:, 2:4]
        ev = np.prod(order == np.array([2, 1]), axis=1)
        av = np.prod(order == np.array([3, 1]), axis=1) * 2
        bv = np.prod(order == np.array([0, 1]), axis=1) * 3
        class_map2 = {int(k): v for k, v in config["second"].items()}
        columns["prediction_2"] = [class_map2[x] for x in (ev + av + bv)]
    for i, label in lab.items():
        columns[f"#_{label}_windows"] = [x[i] for x in data["per_class_counts"]]
        columns[f"{label}_score"] = [x[i] for x in data["pred_sum"]]
        columns[f"{label}_var"] = [x[i] for x in data["pred_var"]]
    columns["window_summary"] = [get_window_summary_legacy(x, config["vindex"]) for x in data["frag_pred"]]
    df = pd.DataFrame(columns).set_index("contig_id")
    repeats = data.get("repeats")
    if repeats is None:
        repeats = pd.DataFrame({"contig_id": pd.Series([], dtype=str), "terminal_repeats": [], "repeat_length": []})
    df = df.join(repeats.set_index("contig_id")[["terminal_repeats", "repeat_length"]],
                 how="left").reset_index(names="contig_id")
    df["contig_id"] = df["contig_id"].str.replace("___", ",")
    return df


def write_output_legacy(config, data: dict, reliability_cutoff: float = 0.5, phage_score: int = 3, **kwargs):
    """collect.py:188-232 (both tables are always written, the phage one possibly header-only)."""
    df = generate_summary_legacy(config, data)
    df.to_csv(kwargs.get("output_table_path"), sep="\t", index=False, float_format="%.3f")
    clause = f" and (reliability_score > {reliability_cutoff})" if data.get("has_reliability", True) else ""
    df.query(f'(prediction == "phage") and (phage_score > {phage_score}){clause}').to_csv(
        kwargs.get("output_phage_table_path"), sep="\t", index=False, float_format="%.3f")
    return len(df)
