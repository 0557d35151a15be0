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


REPORT_MIN_COLUMNS = 13       # the decision rule reads alignments of MORE than 12 columns (termini.py:137-154)


def terminal_repeat_table(device, fa, fsize: int, report_min: int = 0) -> np.ndarray:
    """(n_records, 10) int32 from ``jg_terminal_repeats``: per record DTR then ITR (score, alignment length, query gaps,
    end in the query, end in the reference); -1 rows for records shorter than ``fsize``.  ``report_min`` (0, or 2 .. 15):
    alignments of fewer columns may come back as none (``JG_OPT_TERMINI_REPORT_MIN``) - ``REPORT_MIN_COLUMNS`` leaves
    every row of :class:`RepeatColumns` as it is and skips the dynamic programme for most records."""
    n = len(fa)
    L.check(device.lib.jg_engine_set_option(device.handle, L.JG_OPT_TERMINI_REPORT_MIN, int(report_min)),
            "jg_engine_set_option")
    res = np.full((max(n, 1), 10), -1, np.int32)
    bases = np.ascontiguousarray(fa.bases, np.uint8)
    offsets = np.ascontiguousarray(fa.offsets, np.int64)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    try:
        L.check(device.lib.jg_terminal_repeats(device.handle, ptr(bases), bases.size, L.JG_PTR_HOST, ptr(offsets), n,
                                               int(fsize), ptr(res)), "jg_terminal_repeats")
    finally:
        if report_min:                       # the option belongs to this call: a later jg_terminal_repeats on the same
            device.lib.jg_engine_set_option(device.handle, L.JG_OPT_TERMINI_REPORT_MIN, 0)   # engine scores every alignment
    return res[:n]


class RepeatColumns:
    """The decision rule's columns (termini.py:137-154, :58-63) for the scanned records, as arrays: what the table writer's
    join by record number reads (``kind``, ``length``, ``row_of_record``).  The DataFrame the reference passes around
    (:meth:`frame`: names, a pandas index over them) is only built when something asks for it - on a million records it
    costs more than the scan's kernels."""

    KIND_LABELS = [None, "DTR", "ITR", "LTR_DTR"]

    def __init__(self, res: np.ndarray, names: list[str], lengths: np.ndarray):
        keep = np.nonzero(res[:, 0] >= 0)[0] if len(res) else np.zeros(0, np.int64)
        d_score, d_len, d_fg = res[keep, 0], res[keep, 1], res[keep, 2]
        i_score, i_len = res[keep, 5], res[keep, 6]
        found = (i_len > 12) | (d_len > 12)
        is_itr = found & (i_score > d_score)
        is_dtr = found & ~is_itr
        kind_code = np.zeros(keep.size, np.int8)                  # index into KIND_LABELS: what the table writer prints from
        kind_code[is_itr] = 2
        kind_code[is_dtr] = 1
        kind_code[is_dtr & ((d_len - d_fg) >= 250)] = 3
        labels = np.empty(4, dtype=object)
        labels[:] = self.KIND_LABELS
        kind = labels[kind_code]
        self.kind_code = kind_code
        length = np.where(is_itr, i_len, d_len).astype(np.float64)
        length[~found] = np.nan
        score = np.where(is_itr, i_score, d_score).astype(np.float64)
        score[~found] = np.nan
        lengths = np.asarray(lengths)
        self.keep, self.kind, self.length, self.score, self.n_found = keep, kind, length, score, int(found.sum())
        self.seq_len = lengths[keep] if keep.size else np.array([], np.int64)
        self.row_of_record = np.full(len(names), -1, dtype=np.int64)      # row of every record (-1: not scanned)
        self.row_of_record[keep] = np.arange(keep.size)
        self._names, self._frame = names, None

    def __len__(self) -> int:
        return int(self.keep.size)

    def frame(self) -> pd.DataFrame:
        """One row per scanned record: contig_id, repeat_length, score, terminal_repeats, seq_len."""
        if self._frame is None:
            from .fragment import normalise_headers
            names, keep = self._names, self.keep
            ids = normalise_headers(names)[keep] if keep.size * 2 > len(names) else \
                np.array([names[i].strip().replace(",", "___") for i in keep.tolist()], dtype=object)
            df = pd.DataFrame({"contig_id": ids, "repeat_length": self.length, "score": self.score,
                               "terminal_repeats": self.kind, "seq_len": self.seq_len})
            # what the table writer's joins need: the name index with its uniqueness known, the row of every record
            index = df.attrs["_contig_index"] = pd.Index(ids)
            _ = index.is_unique
            df.attrs["_row_of_record"] = self.row_of_record
            self._frame = df
        return self._frame


def repeats_frame(res: np.ndarray, names: list[str], lengths: np.ndarray) -> pd.DataFrame:
    """Decision rule (termini.py:137-154, :58-63) over a :func:`terminal_repeat_table`; one row per scanned record."""
    return RepeatColumns(res, names, lengths).frame()


def scan_for_terminal_repeats(device, fa, fsize: int) -> pd.DataFrame:
    """``device``: :class:`~jaeger_amd.engine.HipDevice`; ``fa``: FastaBatch.  One row per record with
    ``len >= fsize`` (the reference's filter, termini.py:164-168)."""
    return repeats_frame(terminal_repeat_table(device, fa, fsize), fa.names, fa.lengths)
