"""Inputs of the prophage segmentation (``postprocess/prophages.py:99-153`` ``logits_to_df_v2``).

For every contig of at least ``lc`` bases (CLI ``--lc``, default 500 000): per-window softmax, the host class
(arg-max of the mean probabilities), the class tracks smoothed with a width-4 box (``np.convolve(..., "same")``),
the per-window G+C fraction and the gc-skew track (width-10 mean, min-max scaled to [-1, 1]).  The change-point
segmentation and the plots that consume these frames (``ruptures`` / ``kneed`` / ``pycirclize``) are not part of
the MI355X predict path; the frames are what ``--prophage`` would hand to them.
"""

from __future__ import annotations

import numpy as np
import pandas as pd


def scale_range(x: np.ndarray, min: float, max: float) -> np.ndarray:      # noqa: A002 - reference's argument names
    """Min-max scaling, in place (``postprocess/helpers.py:656-675``)."""
    x += -(np.min(x))
    x /= np.max(x) / (max - min)
    x += min
    return x


def _fit(track: np.ndarray, n: int) -> np.ndarray:
    if len(track) > n:
        return track[:n]
    if len(track) < n:
        return np.pad(track, (0, n - len(track)), mode="edge")
    return track


def logits_to_df_v2(class_map: dict, cmdline_kwargs: dict, headers, predictions, lengths, gc_skews, gcs) -> dict:
    """contig id -> [DataFrame (one row per window: smoothed class tracks, ``length`` = window start clamped to the
    contig, ``gc``, ``gc_skew``), host label, contig length] for contigs with ``length >= lc``."""
    lab = {int(i): c for i, c in zip(class_map.get("index", []), class_map.get("class", []))}
    out = {}
    for key, value, length, gc_skew, gc in zip(headers, predictions, lengths, gc_skews, gcs):
        if length < cmdline_kwargs.get("lc", 500_000):
            continue
        value = np.exp(value) / np.sum(np.exp(value), axis=1).reshape(-1, 1)
        host = lab.get(np.argmax(np.mean(value, axis=0)), "unknown")
        t = pd.DataFrame(value, columns=list(lab.values()))
        stride = cmdline_kwargs.get("stride") or cmdline_kwargs.get("fsize", 2000)
        t = t.assign(length=[min(i * stride, length) for i in range(len(t))])
        for k, v in lab.items():
            t[v] = _fit(np.convolve(value[:, k], np.ones(4), mode="same"), len(t))
        t["gc"] = gc[: len(t)] if len(gc) > len(t) else gc
        t["gc_skew"] = scale_range(_fit(np.convolve(np.array(gc_skew), np.ones(10) / 10, mode="same"), len(t)),
                                   min=-1, max=1)
        out[f"{key}"] = [t, host, length]
    return out
