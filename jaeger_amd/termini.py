"""Terminal-repeat scan on the GPU (``utils/termini.py:88-189`` ``scan_for_terminal_repeats``).

``jg_terminal_repeats`` aligns the two ends of every contig (direct and reverse-complemented) with the
reference's scoring and returns score / alignment length / query gaps per alignment; the decision rule
(``termini.py:137-154``, ``:58-63``) and the DataFrame the summary merges (``postprocess/collect.py:527-532``:
``contig_id``, ``terminal_repeats``, ``repeat_length``) are assembled here.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pandas as pd

from . import _lib as L


def terminal_repeat_table(device, fa, fsize: int) -> np.ndarray:
    """(n_records, 10) int32 from ``jg_terminal_repeats``: per record DTR then ITR (score, alignment length, query gaps,
    end in the query, end in the reference); -1 rows for records shorter than ``fsize``."""
    n = len(fa)
    res = np.full((max(n, 1), 10), -1, np.int32)
    bases = np.ascontiguousarray(fa.bases, np.uint8)
    offsets = np.ascontiguousarray(fa.offsets, np.int64)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L.check(device.lib.jg_terminal_repeats(device.handle, ptr(bases), bases.size, L.JG_PTR_HOST, ptr(offsets), n,
                                           int(fsize), ptr(res)), "jg_terminal_repeats")
    return res[:n]


def repeats_frame(res: np.ndarray, names: list[str], lengths: np.ndarray) -> pd.DataFrame:
    """Decision rule (termini.py:137-154, :58-63) over a :func:`terminal_repeat_table`; one row per scanned record."""
    keep = np.nonzero(res[:, 0] >= 0)[0] if len(res) else np.zeros(0, np.int64)
    d_score, d_len, d_fg = res[keep, 0], res[keep, 1], res[keep, 2]
    i_score, i_len = res[keep, 5], res[keep, 6]
    found = (i_len > 12) | (d_len > 12)
    is_itr = found & (i_score > d_score)
    is_dtr = found & ~is_itr
    kind = np.full(keep.size, None, dtype=object)
    kind[is_itr] = "ITR"
    kind[is_dtr] = "DTR"
    kind[is_dtr & ((d_len - d_fg) >= 250)] = "LTR_DTR"
    length = np.where(is_itr, i_len, d_len).astype(np.float64)
    length[~found] = np.nan
    score = np.where(is_itr, i_score, d_score).astype(np.float64)
    score[~found] = np.nan
    lengths = np.asarray(lengths)
    from .fragment import normalise_headers
    ids = normalise_headers(names)[keep] if keep.size * 2 > len(names) else \
        np.array([names[i].strip().replace(",", "___") for i in keep.tolist()], dtype=object)
    df = pd.DataFrame({
        "contig_id": ids,
        "repeat_length": length, "score": score, "terminal_repeats": kind,
        "seq_len": lengths[keep] if keep.size else np.array([], np.int64),
    })
    # what the table writer's join needs, made here (the scan runs beside the forward, the join behind it): the name index
    # with its uniqueness known, and the row of every record (-1: not scanned) for a join by record number
    index = df.attrs["_contig_index"] = pd.Index(ids)
    _ = index.is_unique
    row_of = np.full(len(names), -1, dtype=np.int64)
    row_of[keep] = np.arange(keep.size)
    df.attrs["_row_of_record"] = row_of
    return df


def scan_for_terminal_repeats(device, fa, fsize: int) -> pd.DataFrame:
    """``device``: :class:`~jaeger_amd.engine.HipDevice`; ``fa``: FastaBatch.  One row per record with
    ``len >= fsize`` (the reference's filter, termini.py:164-168)."""
    return repeats_frame(terminal_repeat_table(device, fa, fsize), fa.names, fa.lengths)
