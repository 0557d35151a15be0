"""Inputs of the prophage segmentation: what ``--prophage`` hands to the change-point code of the reference
(``postprocess/prophages.py:99-153`` builds them contig by contig; ``postprocess/helpers.py:656-675`` is the scaling).

For every contig of at least ``lc`` bases (CLI ``--lc``, default 500 000) a frame with one row per window: the class
probabilities smoothed with a width-4 box, the window's start position clamped to the contig, its G+C fraction and the
gc-skew track (width-10 mean, min-max scaled to [-1, 1]); beside it the host call (arg-max of the contig's mean
probabilities) and the contig length.  The segmentation and the plots that consume the frames (``ruptures`` / ``kneed`` /
``pycirclize``) are not part of the MI355X predict path.

Layout here: the kept contigs' windows are ONE row block - one softmax over all of them, one column block per output
column, the contig only present as a segment ``[first, first + n)`` of those columns; the two box filters run per segment
into the shared columns (numpy's own ``convolve``, so that the sums are taken in its order: the frames are bit-identical
to the reference's, ``tests/golden/prophage_inputs.npz``), and the frames are cut out of the finished columns at the end.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

BOX_CLASS = np.ones(4)                 # class tracks: width-4 box SUM
BOX_SKEW = np.ones(10) / 10            # gc-skew: width-10 mean


def scale_range(x: np.ndarray, min: float, max: float) -> np.ndarray:      # noqa: A002 - the reference's argument names
    """Min-max scaling to [min, max], in place; returns ``x`` (``postprocess/helpers.py:656-675``)."""
    np.subtract(x, x.min(), out=x)
    np.divide(x, x.max() / (max - min), out=x)
    np.add(x, min, out=x)
    return x


def _same_box(track: np.ndarray, box: np.ndarray, out: np.ndarray) -> None:
    """``out[:] = np.convolve(track, box, "same")`` brought to ``len(out)`` rows: a filter longer than the track (or a
    track longer than the segment) is cut, a shorter result repeats its last value."""
    n = len(out)
    res = np.convolve(track, box, mode="same")
    m = min(len(res), n)
    out[:m] = res[:m]
    if m < n:
        out[m:] = res[-1]


def logits_to_df_v2(class_map: dict, cmdline_kwargs: dict, headers, predictions, lengths, gc_skews, gcs) -> dict:
    """contig id -> [frame, host label, contig length] for the contigs with ``length >= lc``."""
    labels = [int(i) for i in class_map.get("index", [])]
    names = list(class_map.get("class", []))[:len(labels)]
    labels = labels[:len(names)]
    lc = cmdline_kwargs.get("lc", 500_000)
    stride = cmdline_kwargs.get("stride") or cmdline_kwargs.get("fsize", 2000)
    contig_len = np.asarray(lengths)
    kept = [c for c in range(len(headers)) if contig_len[c] >= lc]
    if not kept:
        return {}
    n_win = np.array([len(predictions[c]) for c in kept], np.int64)
    first = np.concatenate(([0], np.cumsum(n_win)))
    total = int(first[-1])
    logits = np.concatenate([np.asarray(predictions[c]) for c in kept], axis=0) if total else np.zeros((0, len(names)))
    if logits.shape[1] != len(names):
        raise ValueError(f"Shape of passed values is {logits.shape}, indices imply ({total}, {len(names)})")
    e = np.exp(logits)
    prob = e / np.sum(e, axis=1).reshape(-1, 1)                       # the softmax as the reference writes it (no shift)
    # window start positions, clamped to the contig: i * stride within each segment
    seg = np.repeat(np.arange(len(kept)), n_win)
    start = np.minimum((np.arange(total) - first[seg]) * stride, np.repeat(contig_len[kept], n_win))
    tracks = np.empty((len(labels), total))                           # one column per class label, then gc, gc_skew
    skew = np.empty(total)
    gc_col = None
    hosts = []
    for j, c in enumerate(kept):
        lo, hi = int(first[j]), int(first[j + 1])
        p = prob[lo:hi]
        best = int(np.argmax(np.mean(p, axis=0))) if hi > lo else 0
        hosts.append(names[labels.index(best)] if best in labels else "unknown")
        for col, k in enumerate(labels):
            _same_box(p[:, k], BOX_CLASS, tracks[col, lo:hi])
        gc = np.asarray(gcs[c])
        if len(gc) < hi - lo:
            raise ValueError(f"Length of values ({len(gc)}) does not match length of index ({hi - lo})")
        if gc_col is None:
            gc_col = np.empty(total, gc.dtype if gc.dtype.kind == "f" else np.float64)
        gc_col[lo:hi] = gc[:hi - lo]
        _same_box(np.asarray(gc_skews[c]), BOX_SKEW, skew[lo:hi])
        if hi > lo:
            scale_range(skew[lo:hi], min=-1, max=1)
    out = {}
    for j, c in enumerate(kept):
        lo, hi = int(first[j]), int(first[j + 1])
        cols = {name: tracks[col, lo:hi] for col, name in enumerate(names)}
        cols["length"] = start[lo:hi]
        cols["gc"] = gc_col[lo:hi]
        cols["gc_skew"] = skew[lo:hi]
        out[f"{headers[c]}"] = [pd.DataFrame(cols), hosts[j], lengths[c]]
    return out
