"""Per-contig aggregation and TSV writers of ``jaeger predict`` (host side, numpy/pandas).

Keeps the output-table surface of the reference so the MI355X engine drops in under
it: ``pred_to_dict`` / ``generate_summary`` / ``write_output`` follow
``postprocess/collect.py:233-608`` and the helpers they use follow
``postprocess/helpers.py:8-219`` - including the quirks a TSV diff would notice
(entropy on clipped raw logits, fp16 storage of the per-contig statistics, the
element-wise ``energy`` of multi-class logits, G+C and N% divided by ``fsize``).
``--crf`` window decoding (``helpers.py:273-449``) runs natively for all contigs at once
(``jg_viterbi_decode``).  Not implemented: ``--refine``, prophage extraction.
"""

from __future__ import annotations

import numpy as np
import pandas as pd


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
    # -logsumexp([z, 0]): of the two exponentials one is exp(0) = 1 exactly and the other exp(-|z|), so the
    # pair sum is 1 + exp(-|z|) bit for bit (IEEE addition commutes) - one exp and no stacked temporary
    return -(np.maximum(squeezed, 0.0) + np.log(1.0 + np.exp(-np.abs(squeezed))))


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


# ---- CRF window decoding (postprocess/helpers.py:273-449) --------------------------------
#: co-occurrence tiers of the biological transition prior, lower-cased class names
#: (helpers.py:283-313): 0.5 = plausible on one contig, 3.0 = implausible, others 1.0
_CRF_PRIOR_TIERS = (
    (0.5, (("bacteria", "phage"), ("bacteria", "plasmid"), ("archaea", "phage"), ("archaea", "plasmid"),
           ("phage", "plasmid"), ("eukarya", "virus"))),
    (3.0, (("bacteria", "eukarya"), ("archaea", "eukarya"), ("bacteria", "archaea"), ("eukarya", "phage"),
           ("eukarya", "plasmid"))),
)


def _pair_matrix(names: list[str], pairs: dict) -> tuple[np.ndarray, np.ndarray]:
    """(values, given) of a symmetric pair table over ``names``: ``pairs`` maps frozenset({a, b}) -> value; pairs that
    name an absent class are skipped.  Pure array code: one comparison grid per listed pair."""
    idx = {n: i for i, n in enumerate(names)}
    n = len(names)
    rows = np.array([(idx[a], idx[b], v) for (a, b), v in ((tuple(k), v) for k, v in pairs.items()) if a in idx and b in idx],
                    dtype=np.float64).reshape(-1, 3)
    values, given = np.zeros((n, n)), np.zeros((n, n), bool)
    i, j = rows[:, 0].astype(int), rows[:, 1].astype(int)
    values[i, j] = values[j, i] = rows[:, 2]
    given[i, j] = given[j, i] = True
    return values, given


def default_transition_prior(class_names: list[str]) -> np.ndarray:
    """Symmetric prior matrix P of the CRF (helpers.py:316-344): the listed co-occurrence tiers, 1.0 for every other
    pair of distinct classes, 0 on the diagonal."""
    names = [str(n).lower() for n in class_names]
    tiers = {frozenset(p): v for v, ps in _CRF_PRIOR_TIERS for p in ps}
    values, given = _pair_matrix(names, {tuple(sorted(k)): v for k, v in tiers.items()})
    return np.where(np.eye(len(names), dtype=bool), 0.0, np.where(given, values, 1.0))


def build_transition_costs(class_names: list[str], switch_cost: float, prior: str = "biological",
                           user_matrix: dict | None = None) -> np.ndarray:
    """``lambda * P`` (helpers.py:347-395): ``user_matrix`` ({"bacteria": {"phage": 0.5}}, applied symmetrically in the
    order given, unknown names ignored) overrides ``prior`` ("biological" | "uniform")."""
    names = [str(n).lower() for n in class_names]
    off_diag = ~np.eye(len(names), dtype=bool)
    if user_matrix:
        p = off_diag.astype(np.float64)
        entries = [(str(a).lower(), str(b).lower(), float(v)) for a, row in user_matrix.items() if isinstance(row, dict)
                   for b, v in row.items()]
        for a, b, v in entries:                      # later entries win, as when the reference assigns them in turn
            if a in names and b in names:
                p[names.index(a), names.index(b)] = p[names.index(b), names.index(a)] = v
        p = np.where(off_diag, p, 0.0)
    elif prior == "uniform":
        p = off_diag.astype(np.float64)
    else:
        p = default_transition_prior(names)
    return float(switch_cost) * p


def viterbi_decode_chains(logits: np.ndarray, first: np.ndarray, switch_cost: float = 2.0,
                          transition_costs: np.ndarray | None = None) -> np.ndarray:
    """MAP class path of every chain (contig) of a window-logit matrix in one native call
    (``jg_viterbi_decode``): chain c = rows ``first[c]:first[c+1]``.  Semantics of
    helpers.py:398-449 per chain (f64 log-softmax emissions, uniform ``switch_cost`` off the
    diagonal when no matrix is given, ties to the lowest class index)."""
    import ctypes as C

    from . import _lib as L
    z = np.ascontiguousarray(np.asarray(logits, dtype=np.float32))
    if z.ndim == 1:
        z = z.reshape(1, -1)
    n, c = z.shape
    if transition_costs is None:
        costs = np.full((c, c), float(switch_cost), dtype=np.float64)
        np.fill_diagonal(costs, 0.0)
    else:
        costs = np.ascontiguousarray(np.asarray(transition_costs, dtype=np.float64))
        if costs.shape != (c, c):
            raise ValueError(f"transition costs {costs.shape} do not match {c} classes")
    first = np.ascontiguousarray(np.asarray(first, dtype=np.int64))
    path = np.zeros(n, dtype=np.int32)
    L.check(L.load().jg_viterbi_decode(z.ctypes.data_as(C.c_void_p), n, c, first.ctypes.data_as(C.c_void_p),
                                       len(first) - 1, costs.ctypes.data_as(C.c_void_p),
                                       path.ctypes.data_as(C.c_void_p)), "jg_viterbi_decode")
    return path.astype(np.int64)


def viterbi_decode(logits: np.ndarray, switch_cost: float = 2.0,
                   transition_costs: np.ndarray | None = None) -> np.ndarray:
    """One contig's window sequence (T, C) -> (T,) class indices (helpers.py:398-449).  The reference
    takes f64 logits; the engine's are f32, and that is what the native decoder reads."""
    z = np.asarray(logits)
    if z.ndim == 1:
        z = z.reshape(1, -1)
    return viterbi_decode_chains(z, np.array([0, z.shape[0]], np.int64), switch_cost, transition_costs)


# ---- aggregation (postprocess/collect.py:247-435) --------------------------------------
# The reference loops over contigs in Python (np.split + one numpy call per contig and statistic): 78 s for a
# million single-window fragments.  Here every statistic is computed for all contigs at once.  Results must
# not move by a bit (they are rounded to fp16 and printed with three decimals), so reductions keep numpy's own
# summation order: contigs are grouped by window count T, a group is gathered into an (n, T, ...) block and
# reduced along T with the same ufunc loop the per-contig call would run (sequential over rows for the (T, C)
# statistics, numpy's pairwise scheme for the 1-D means).  tests/test_postprocess.py compares byte for byte with
# the per-contig restatement (oracle/postprocess.py) and with TSVs written by the reference itself.
class _Segments:
    """Contigs as runs of consecutive windows: ``first[i] .. first[i] + count[i]``; ``groups`` yields
    (T, contig indices with T windows, (n, T) window-index block)."""

    def __init__(self, split_indices: np.ndarray, n_windows: int):
        self.first = np.concatenate(([0], split_indices)).astype(np.int64)
        self.count = np.diff(np.concatenate((self.first, [n_windows]))).astype(np.int64)
        self.n = len(self.first)
        order = np.argsort(self.count, kind="stable")
        ts, starts = np.unique(self.count[order], return_index=True)
        bounds = np.append(starts, len(order))
        self._groups = [(int(t), order[a:b]) for t, a, b in zip(ts, bounds[:-1], bounds[1:])]

    def groups(self):
        for t, idx in self._groups:
            yield t, idx, self.first[idx][:, None] + np.arange(t, dtype=np.int64)[None, :]

    # The three reductions below run natively (``jg_segment_mean_var`` / ``jg_segment_mean_1d``, csrc/jg_segments.hip: numpy's
    # own summation order restated, every core) - a Python-level loop over the distinct window counts cost 0.15 s per
    # 10 000 contigs, most of what ran beside (and behind) a short forward.  ``*_numpy`` are the grouped numpy forms they
    # replace; tests/test_postprocess.py holds the two equal bit for bit.
    def mean_1d(self, v: np.ndarray) -> np.ndarray:
        """[np.mean(v[a:b]) for every contig] for a 1-D per-window array."""
        v = np.asarray(v)
        if v.dtype == np.float32 or v.dtype == np.float64:
            return _native_mean_1d(np.ascontiguousarray(v), self.first, self.count)
        return self.mean_1d_numpy(v)

    def mean_flat(self, m: np.ndarray) -> np.ndarray:
        """[np.mean(m[a:b]) for every contig] for an (N, C) array (mean over windows AND columns)."""
        m = np.asarray(m)
        if (m.dtype == np.float32 or m.dtype == np.float64) and m.ndim == 2:
            c = m.shape[1]
            return _native_mean_1d(np.ascontiguousarray(m).reshape(-1), self.first * c, self.count * c)
        return self.mean_flat_numpy(m)

    def mean_var_rows(self, m: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """[np.mean(m[a:b], axis=0)], [np.var(m[a:b], axis=0)] for an (N, C) array."""
        m = np.asarray(m)
        if m.dtype == np.float32 and m.ndim == 2 and m.shape[1] >= 1:
            import ctypes as C

            from . import _lib
            m = np.ascontiguousarray(m)
            mean = np.empty((self.n, m.shape[1]), np.float32)
            var = np.empty_like(mean)
            first, count = np.ascontiguousarray(self.first, np.int64), np.ascontiguousarray(self.count, np.int64)
            _lib.check(_lib.load().jg_segment_mean_var(m.ctypes.data_as(C.c_void_p), m.shape[0], m.shape[1],
                                                       first.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p),
                                                       self.n, mean.ctypes.data_as(C.c_void_p),
                                                       var.ctypes.data_as(C.c_void_p), 0), "jg_segment_mean_var")
            return mean, var
        return self.mean_var_rows_numpy(m)

    def mean_1d_numpy(self, v: np.ndarray) -> np.ndarray:
        out = np.empty(self.n, dtype=np.result_type(v.dtype, np.float32) if v.dtype.kind != "f" else v.dtype)
        for t, idx, rows in self.groups():
            out[idx] = np.mean(v[rows], axis=1)
        return out

    def mean_flat_numpy(self, m: np.ndarray) -> np.ndarray:
        out = np.empty(self.n, dtype=m.dtype)
        for t, idx, rows in self.groups():
            out[idx] = np.mean(m[rows].reshape(len(idx), -1), axis=1)
        return out

    def mean_var_rows_numpy(self, m: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        mean = np.empty((self.n, m.shape[1]), dtype=m.dtype)
        var = np.empty_like(mean)
        for t, idx, rows in self.groups():
            block = m[rows]                                   # (n, T, C)
            mean[idx] = np.mean(block, axis=1)
            var[idx] = np.var(block, axis=1)
        return mean, var


def _native_mean_1d(v: np.ndarray, first: np.ndarray, count: np.ndarray) -> np.ndarray:
    import ctypes as C

    from . import _lib
    first, count = np.ascontiguousarray(first, np.int64), np.ascontiguousarray(count, np.int64)
    out = np.empty(len(first), v.dtype)
    _lib.check(_lib.load().jg_segment_mean_1d(v.ctypes.data_as(C.c_void_p), int(v.dtype == np.float64), v.size,
                                              first.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p), len(first),
                                              out.ctypes.data_as(C.c_void_p), 0), "jg_segment_mean_1d")
    return out


def _frac_strings(flags: np.ndarray, seg: _Segments) -> np.ndarray:
    """frac_above_threshold per contig -> the fp16 array the reference builds from the formatted strings
    (collect.py:233-244, :402-405): "{:.2f}".format(mean(flags)) parsed back.  ``flags`` is the (N, W) boolean
    block of all windows; one format per distinct (count above, elements) pair."""
    width = flags.shape[1]
    k = np.add.reduceat(flags.sum(axis=1).astype(np.int64), seg.first)
    total = seg.count * width
    base = int(total.max()) + 1
    uniq, inv = np.unique(k * base + total, return_inverse=True)
    table = np.empty(len(uniq), dtype=np.float16)
    for j, u in enumerate(uniq.tolist()):
        kk, tt = divmod(u, base)
        f = np.zeros(tt, dtype=bool)
        f[:kk] = True
        table[j] = np.float16("{:.2f}".format(f.mean()))
    return table[inv]


def pred_to_dict(y_pred: dict, **kwargs) -> tuple[dict, dict]:
    """Window outputs + metadata -> per-contig statistics.

    ``y_pred`` keys as returned by the engine: ``prediction`` (N, C), optional
    ``reliability`` (N, 1), ``meta_0`` header, ``meta_2`` is-last flag, ``meta_4`` contig
    length, ``meta_5..8`` base counts, ``meta_9`` gc skew.  kwargs: ``fsize``,
    ``class_map`` ({"num_classes": ...}), ``term_repeats`` (DataFrame), ``crf_*``;
    ``want_full=False`` skips the per-contig lists of ``data_full`` (only the ``--window-scores`` writer
    reads them).
    """
    crf_switch_cost = kwargs.get("crf_switch_cost")
    pred = np.asarray(y_pred["prediction"])
    n_win = pred.shape[0]
    split_flags = np.array(y_pred["meta_2"], dtype=np.int32)
    split_indices = np.where(split_flags == 1)[0] + 1
    classifier_type = "binary" if pred.shape[-1] == 1 else "softmax"
    if n_win == split_indices[-1]:
        split_indices = split_indices[:-1]
    seg = _Segments(split_indices, n_win)
    chain_first = np.append(seg.first, n_win)
    has_reliability = "reliability" in y_pred

    headers = kwargs.get("headers")                                    # a SpanColumn of the contigs' names (the pipeline's)
    if headers is None:
        headers = np.asarray(y_pred["meta_0"])[seg.first].astype(str)  # (one string per contig, not per window)
    elif len(headers) != seg.n:
        raise ValueError(f"pred_to_dict: {len(headers)} headers for {seg.n} contigs")
    lengths = np.array(y_pred["meta_4"], dtype=np.int32)[seg.first]
    # nucleotide content; the reference labels the columns a,t,g,c = meta_7,8,6,5 and only
    # uses their sums (collect.py:319-324)
    a, t, g, c = (np.asarray(y_pred[k]).astype(float) for k in ("meta_7", "meta_8", "meta_6", "meta_5"))
    fsize = kwargs["fsize"]
    ns_w = (fsize - (a + t + g + c)) / fsize
    gcs_w = (g + c) / fsize

    mean, var = seg.mean_var_rows(pred)
    pred_sum = np.array(mean[:, 0] if pred.shape[1] == 1 else mean, dtype=np.float16)
    pred_var = np.array(var[:, 0] if pred.shape[1] == 1 else var, dtype=np.float16)
    num_classes = kwargs.get("class_map").get("num_classes")
    energy_w = energy(pred)                                   # element-wise for every head width but 2
    energy_mean_ = seg.mean_1d(energy_w) if energy_w.ndim == 1 else seg.mean_flat(energy_w)
    if classifier_type == "softmax":
        entropy_mean_ = seg.mean_1d(softmax_entropy(pred))
        consensus = np.argmax(pred_sum, axis=1)
        if crf_switch_cost is not None:
            # joint MAP decoding of each contig's windows instead of independent argmax (collect.py:269-289,343-346)
            cm = kwargs.get("class_map")
            names = [name for _, name in sorted(zip(cm.get("index"), cm.get("class")), key=lambda t: int(t[0]))]
            costs = build_transition_costs(names, switch_cost=crf_switch_cost, prior=kwargs.get("crf_prior", "biological"),
                                           user_matrix=kwargs.get("crf_transition_matrix"))
            calls = viterbi_decode_chains(pred, chain_first, crf_switch_cost, costs)
        else:
            calls = np.argmax(pred, axis=-1)
        prophage_contam = (pred_sum[:, 1] < pred_var[:, 1]) & (consensus == 0)
        host_contam = (pred_sum[:, 1] < pred_var[:, 1]) & (consensus == 1)
    else:
        entropy_mean_ = seg.mean_flat(binary_entropy(pred))
        consensus = np.array(sigmoid(pred_sum))
        consensus[consensus > 0.5] = 1.0
        consensus[consensus <= 0.5] = 0.0
        if crf_switch_cost is not None:
            # two-class CRF on stacked [0, z] logits, uniform switch cost (collect.py:365-372)
            z = np.asarray(pred, np.float32).reshape(-1, 1)
            calls = viterbi_decode_chains(np.concatenate([np.zeros_like(z), z], axis=-1), chain_first, crf_switch_cost)
        else:
            calls = (sigmoid(pred) > 0.5).astype(int)[:, 0]
        prophage_contam = (pred_sum < pred_var) & (consensus == 0)
        host_contam = (pred_sum < pred_var) & (consensus == 1)
    # windows per class and contig (update_dict(np.unique(...)), helpers.py:111-127): row i = counts of contig i
    width = max(int(num_classes), int(calls.max()) + 1 if calls.size else 0)
    cid = np.repeat(np.arange(seg.n), seg.count)
    per_class_counts = np.bincount(cid * width + calls, minlength=seg.n * width).reshape(seg.n, width)

    ood = None
    if has_reliability:
        ood = _frac_strings((sigmoid(np.asarray(y_pred["reliability"])) > 0.5).reshape(n_win, -1), seg)
    data = {
        "headers": headers, "length": lengths, "consensus": consensus,
        "per_class_counts": per_class_counts, "pred_sum": pred_sum, "pred_var": pred_var,
        "frag_pred": _Runs(calls, seg), "ood": ood, "has_reliability": has_reliability,
        "entropy": np.array(entropy_mean_, dtype=np.float16), "energy": np.array(energy_mean_, dtype=np.float16),
        "host_contam": host_contam, "prophage_contam": prophage_contam, "repeats": kwargs.get("term_repeats"),
        "gc": _Means(seg.mean_1d(gcs_w)), "ns": _Means(seg.mean_1d(ns_w)),
    }
    data_full = {"headers": headers, "lengths": lengths}
    if kwargs.get("want_full", True):
        data_full.update(predictions=np.split(pred, split_indices, axis=0),
                         gc_skews=np.split(np.asarray(y_pred["meta_9"]).astype(float), split_indices),
                         gcs=np.split(gcs_w, split_indices))
    return data, data_full


class _SegView:
    """The part of :class:`_Segments` that :class:`_Runs` reads, for segmentations merged from several batches."""

    def __init__(self, first: np.ndarray, count: np.ndarray):
        self.first, self.count, self.n = first, count, len(first)


def merge_data(parts: list[dict]) -> dict:
    """Concatenate the ``data`` (or ``data_full``) dicts :func:`pred_to_dict` returned for consecutive batches of
    whole contigs into the dict one call over all of them returns (per-contig statistics do not depend on the batch)."""
    parts = [p for p in parts if p]
    if len(parts) == 1:
        return parts[0]
    if not parts:
        return {}
    out = {}
    for key, v0 in parts[0].items():
        vals = [p[key] for p in parts]
        if any(isinstance(v, SpanColumn) for v in vals):                      # (names as bytes in some batches, strings in others)
            out[key] = SpanColumn.concat(vals)
        elif isinstance(v0, _Summaries):
            out[key] = _Summaries([t for v in vals for t in v.texts])
        elif isinstance(v0, _Runs):
            offs = np.cumsum([0] + [len(v.calls) for v in vals[:-1]])
            out[key] = _Runs(np.concatenate([v.calls for v in vals]),
                             _SegView(np.concatenate([v.seg.first + o for v, o in zip(vals, offs)]),
                                      np.concatenate([v.seg.count for v in vals])))
        elif isinstance(v0, _Means):
            out[key] = _Means(np.concatenate([v.means for v in vals]))

        elif isinstance(v0, np.ndarray):
            if v0.ndim == 2 and len({v.shape[1] for v in vals}) > 1:          # per-class counts of differing widths
                width = max(v.shape[1] for v in vals)
                vals = [np.pad(v, ((0, 0), (0, width - v.shape[1]))) for v in vals]
            out[key] = np.concatenate(vals, axis=0)
        elif isinstance(v0, list):
            out[key] = [x for v in vals for x in v]
        else:                                                                 # has_reliability, repeats, ood = None
            out[key] = v0
    return out


class SpanColumn:
    """A column of strings kept as BYTES: string r = ``buf[begin[r]:end[r]]`` (UTF-8).  Record names come this way straight out
    of the FASTA parser's name buffer (``fragment.Names.spans``) and go this way into ``jg_table_format`` (JG_COL_SPANS): a
    million rows never become a million Python strings.  ``to_objects`` / ``tolist`` make them where a DataFrame is asked for."""

    def __init__(self, buf: np.ndarray, begin: np.ndarray, end: np.ndarray, facts: dict | None = None):
        self.buf = np.ascontiguousarray(buf, np.uint8)
        self.begin, self.end = np.ascontiguousarray(begin, np.int64), np.ascontiguousarray(end, np.int64)
        self.facts = facts if facts is not None else {}      # what is known about ``buf`` (shared by the columns cut from it)

    def __len__(self) -> int:
        return len(self.begin)

    def tolist(self) -> list[str]:
        raw = self.buf.tobytes()
        return [raw[a:b].decode() for a, b in zip(self.begin.tolist(), self.end.tolist())]

    def to_objects(self) -> np.ndarray:
        out = np.empty(len(self), dtype=object)
        out[:] = self.tolist()
        return out

    def __array__(self, dtype=None, copy=None):
        out = self.to_objects()
        return out if dtype is None else out.astype(dtype)

    def spans(self) -> np.ndarray:
        """[begin_0, end_0, begin_1, end_1, ...] as ``jg_table_format`` reads a JG_COL_SPANS column."""
        st = np.empty(2 * len(self), np.int64)
        st[0::2], st[1::2] = self.begin, self.end
        return st

    def contains(self, sub: bytes) -> bool:
        """Conservative: ``sub`` occurs somewhere in the buffer (possibly outside this column's spans)."""
        key = ("contains", sub)
        if key not in self.facts:
            self.facts[key] = sub in self.buf.tobytes()
        return self.facts[key]

    def csv_safe(self) -> bool:
        if "csv_safe" not in self.facts:
            raw = self.buf.tobytes()
            self.facts["csv_safe"] = not any(c in raw for c in _CSV_SPECIAL)
        return self.facts["csv_safe"]

    @staticmethod
    def from_strings(values) -> "SpanColumn":
        raw = [str(v).encode("utf-8") for v in (values.tolist() if isinstance(values, np.ndarray) else values)]
        ln = np.array([len(r) for r in raw], np.int64)
        end = np.cumsum(ln)
        return SpanColumn(np.frombuffer(b"".join(raw) or b"\0", np.uint8), end - ln, end)

    @staticmethod
    def concat(cols: list) -> "SpanColumn":
        cols = [c if isinstance(c, SpanColumn) else SpanColumn.from_strings(c) for c in cols]
        if all(c.buf is cols[0].buf for c in cols):
            return SpanColumn(cols[0].buf, np.concatenate([c.begin for c in cols]), np.concatenate([c.end for c in cols]),
                              cols[0].facts)
        shift = np.cumsum([0] + [c.buf.size for c in cols[:-1]])
        return SpanColumn(np.concatenate([c.buf for c in cols]), np.concatenate([c.begin + o for c, o in zip(cols, shift)]),
                          np.concatenate([c.end + o for c, o in zip(cols, shift)]))


def header_strings(headers) -> np.ndarray:
    """Contig names as the fixed-width string array ``pred_to_dict`` used to hand out (the npz writers store it)."""
    return np.asarray(headers.tolist() if isinstance(headers, SpanColumn) else headers).astype(str)


class EnumColumn:
    """A column with a handful of distinct strings: integer ``codes`` into ``labels`` (None = missing: printed as nothing,
    NaN in the object form).  The class label of every contig, the kind of its terminal repeat."""

    def __init__(self, codes: np.ndarray, labels: list):
        self.codes, self.labels = np.asarray(codes), list(labels)

    def __len__(self) -> int:
        return len(self.codes)

    def to_objects(self) -> np.ndarray:
        table = np.empty(len(self.labels), dtype=object)
        table[:] = [np.nan if v is None else v for v in self.labels]
        return table[self.codes]

    def __array__(self, dtype=None, copy=None):
        out = self.to_objects()
        return out if dtype is None else out.astype(dtype)

    def __eq__(self, other):                          # (the phage query: prediction == "phage")
        if isinstance(other, str):
            hits = [i for i, v in enumerate(self.labels) if v == other]
            return np.isin(self.codes, hits) if hits else np.zeros(len(self), dtype=bool)
        return NotImplemented

    __hash__ = None

    def as_spans(self) -> SpanColumn:
        raw = [b"" if v is None else str(v).encode("utf-8") for v in self.labels]
        ends = np.cumsum([len(r) for r in raw]).astype(np.int64)
        begins = ends - np.array([len(r) for r in raw], np.int64)
        codes = self.codes.astype(np.int64)
        return SpanColumn(np.frombuffer(b"".join(raw) or b"\0", np.uint8), begins[codes], ends[codes])


class _Summaries:
    """The contigs' run-length strings, already built (``_Runs.summaries`` per aggregation batch, beside the forward): a list
    of strings, or the library's text as it came (``blob`` = the strings back to back, one NUL behind each) - the table writer
    hands that straight back to ``jg_table_format``, and Python strings are only made when something asks for ``texts``."""

    def __init__(self, texts: list[str] | None = None, blob: bytes | None = None, n: int = 0):
        self._texts, self.blob, self._n = texts, blob, (len(texts) if texts is not None else n)

    @property
    def texts(self) -> list[str]:
        if self._texts is None:
            self._texts = self.blob.decode("ascii").split("\0")[:-1]
        return self._texts

    def __len__(self):
        return self._n


def window_letters(class_map: dict) -> dict:
    """Letter of every class in ``window_summary``: first character, upper case only for virus / phage (helpers.py:73-108)."""
    cm = {int(k): v for k, v in zip(class_map.get("index"), class_map.get("class"))}
    return {k: (v[0].upper() if v.lower() in ("virus", "phage") else v[0].lower()) for k, v in cm.items()}


class _Means:
    """Per-contig means of a per-window quantity, already reduced (generate_summary takes the mean of
    ``data["gc"]`` / ``data["ns"]`` entries, collect.py:470-471)."""

    def __init__(self, means: np.ndarray):
        self.means = means


class _Runs:
    """Per-window calls of all contigs + the contig segmentation; ``summaries`` builds every contig's
    run-length string (helpers.py:8-40,73-108) from one pass over the runs."""

    def __init__(self, calls: np.ndarray, seg: _Segments):
        self.calls, self.seg = np.asarray(calls), seg

    def __len__(self):
        return self.seg.n

    def __getitem__(self, i):
        a = int(self.seg.first[i])
        return self.calls[a:a + int(self.seg.count[i])]

    def summaries(self, letter: dict) -> list[str]:
        """Every contig's run-length string; rendered by the library (``jg_run_summaries``) when the letters are single
        ASCII characters - a Python string per RUN was the largest single cost of the per-contig aggregation - else by
        :meth:`summaries_py` (same strings, tests/test_postprocess.py)."""
        blob = self.summaries_blob(letter)
        if blob is not None:
            return blob.decode("ascii").split("\0")[:-1]
        return self.summaries_py(letter) if self.calls.size else []

    def summaries_blob(self, letter: dict) -> bytes | None:
        """The library's text itself (every contig's string + a NUL), or None where :meth:`summaries_py` has to do it."""
        calls, seg = self.calls, self.seg
        if calls.size == 0:
            return None
        ids = [int(k) for k in letter]
        if all(len(v) == 1 and v.isascii() for v in letter.values()) and (not ids or (min(ids) >= 0 and max(ids) < 256)):
            import ctypes as C

            from . import _lib
            lib = _lib.load()
            table = np.zeros(max(ids) + 1 if ids else 1, dtype=np.uint8)
            for k, v in letter.items():
                table[int(k)] = ord(v)
            c32 = np.ascontiguousarray(calls, dtype=np.int32)
            first = np.ascontiguousarray(seg.first, dtype=np.int64)
            count = np.ascontiguousarray(seg.count, dtype=np.int64)
            text, size = C.c_void_p(), C.c_int64()
            _lib.check(lib.jg_run_summaries(c32.ctypes.data, c32.size, first.ctypes.data, count.ctypes.data, seg.n,
                                            table.ctypes.data, table.size, 0, C.byref(text), C.byref(size)), "jg_run_summaries")
            try:
                return C.string_at(text, size.value)
            finally:
                lib.jg_table_free(text)
        return None

    def summaries_py(self, letter: dict) -> list[str]:
        calls, seg = self.calls, self.seg
        if calls.size == 0:
            return []
        start = np.ones(calls.size, dtype=bool)
        start[1:] = calls[1:] != calls[:-1]
        start[seg.first] = True
        run_start = np.nonzero(start)[0]
        run_len = np.diff(np.append(run_start, calls.size))
        run_val = calls[run_start]
        pieces = np.array([f"{n}{letter.get(int(v), '')}" for v, n in zip(run_val.tolist(), run_len.tolist())], dtype=object)
        first_run = np.searchsorted(run_start, seg.first)
        return np.add.reduceat(pieces, first_run).tolist()


def _left_join(df: pd.DataFrame, right: pd.DataFrame, columns: list) -> pd.DataFrame:
    """``pd.merge(left=df, right=right[["contig_id", *columns]], on="contig_id", how="left")`` (collect.py:527-532).  With
    unique names on the right - every FASTA without repeated record names - a left join is one hash lookup per row; the
    lookup table is built once per ``right`` frame and kept with it, where pandas factorises the whole right table again
    for every batch of rows (1.4 s of 5.4 s on a million short contigs).  Repeated names take pandas' merge."""
    index = right.attrs.get("_contig_index")
    if index is None:
        index = right.attrs["_contig_index"] = pd.Index(right["contig_id"].to_numpy(dtype=object))
    if not index.is_unique or df["contig_id"].dtype != object:
        return pd.merge(left=df, right=right[["contig_id", *columns]], on="contig_id", how="left")
    at = index.get_indexer(df["contig_id"].to_numpy(dtype=object))
    hit = at >= 0
    out = df.copy(deep=False)
    out.index = pd.RangeIndex(len(out))                       # what merge returns
    for col in columns:
        src = right[col].to_numpy()
        if src.dtype.kind in "iub" and not hit.all():         # merge widens integers and booleans that get holes
            src = src.astype(np.float64) if src.dtype.kind != "b" else src.astype(object)
        val = np.empty(len(df), dtype=src.dtype)
        val[hit] = src[at[hit]]
        if not hit.all():
            val[~hit] = np.nan
        out[col] = val
    return out


def _summary_columns(data, **kwargs) -> dict:
    """The columns of the per-contig summary table in output order, before the repeat table is joined on
    (collect.py:438-526); values are arrays, lists or a :class:`_Summaries`."""
    classes_, indices_ = kwargs.get("labels"), kwargs.get("indices")
    class_map = {int(k): v for k, v in zip(indices_, classes_)}
    n = len(data["headers"])
    reliability = data["ood"] if data.get("has_reliability", True) else ["unavailable"] * n
    mean_of = lambda x: x.means if isinstance(x, _Means) else [np.mean(v) for v in x]          # noqa: E731
    counts = data["per_class_counts"]
    consensus = np.asarray(data["consensus"])
    if n and consensus.dtype.kind in "iu" and consensus.min() >= 0 and all(k in class_map for k in range(int(consensus.max()) + 1)):
        prediction = EnumColumn(consensus, [class_map[k] for k in range(int(consensus.max()) + 1)])
    else:
        prediction = [class_map[x] for x in consensus.tolist()]
    columns = {
        "contig_id": data["headers"],
        "length": data["length"],
        "prediction": prediction,
        "entropy": data["entropy"],
        "energy": data["energy"],
        "reliability_score": reliability,
        "host_contam": data["host_contam"],
        "prophage_contam": data["prophage_contam"],
        "G+C": mean_of(data["gc"]),
        "N%": mean_of(data["ns"]),
    }
    for i, label in class_map.items():
        columns[f"#_{label}_windows"] = counts[:, i] if isinstance(counts, np.ndarray) else [x[i] for x in counts]
    if len(class_map) > 2:
        for i, label in class_map.items():
            columns[f"{label}_score"] = data["pred_sum"][:, i]
            columns[f"{label}_var"] = data["pred_var"][:, i]
    else:
        columns["score"] = data["pred_sum"]
        columns["var"] = data["pred_var"]
    frag = data["frag_pred"]
    if isinstance(frag, _Summaries):
        columns["window_summary"] = frag
    elif isinstance(frag, _Runs):
        letter = {k: (v[0].upper() if v.lower() in ("virus", "phage") else v[0].lower()) for k, v in class_map.items()}
        columns["window_summary"] = frag.summaries(letter)
    else:
        columns["window_summary"] = [get_window_summary(x, class_map=class_map, classes=["virus", "phage"]) for x in frag]
    return columns


def generate_summary(data, **kwargs) -> pd.DataFrame:
    """Per-contig summary table (collect.py:438-558); column order is part of the surface."""
    columns = _summary_columns(data, **kwargs)
    if isinstance(columns["window_summary"], _Summaries):
        columns["window_summary"] = columns["window_summary"].texts
    for key, v in columns.items():
        if isinstance(v, (SpanColumn, EnumColumn)):
            columns[key] = v.to_objects()
    df = pd.DataFrame(columns)
    repeats = data.get("repeats")
    if repeats is not None and hasattr(repeats, "frame"):            # termini.RepeatColumns
        repeats = repeats.frame()
    if repeats is None:
        repeats = pd.DataFrame({"contig_id": [], "terminal_repeats": [], "repeat_length": []})
    df = _left_join(df, repeats, ["terminal_repeats", "repeat_length"])
    refined = kwargs.get("refined_contig")
    if refined is not None:
        df = pd.merge(left=df, right=refined[["contig_id", "contig_call", "contig_top_logit", "contig_margin",
                                              "n_windows_used", "n_merged_windows"]], on="contig_id", how="left")
    ids = df["contig_id"].tolist()
    if "___" in "\0".join(ids):
        df["contig_id"] = [x.replace("___", ",") for x in ids]
    return df


def _tsv_text(df: pd.DataFrame, header: bool = True) -> str:
    """``df.to_csv(sep="\t", index=False, float_format="%.3f")`` as text, with the float columns formatted by one
    vectorised ``%`` per column instead of pandas' per-value Python formatter (same text: ``"%.3f" % float(v)``, empty
    for NaN).  The general path: ``_tsv_bytes`` renders the usual tables natively and falls back to this one."""
    out = {}
    for col in df.columns:
        v = df[col]
        if v.dtype.kind == "f":
            arr = v.to_numpy().astype(np.float64)
            txt = np.char.mod("%.3f", arr).astype(object)
            txt[np.isnan(arr)] = ""
            out[col] = txt
        else:
            out[col] = v
    return pd.DataFrame(out, columns=df.columns).to_csv(None, sep="\t", index=False, header=header)


_CSV_SPECIAL = (b"\t", b'"', b"\n", b"\r")          # what makes csv.QUOTE_MINIMAL quote a field (pandas' default)


def _string_column(values):
    """A column of strings (NaN / None = empty, as ``na_rep=""``) laid end to end for ``jg_table_format``: (bytes, starts),
    or None when pandas has to write it - other objects, an embedded NUL, or a field the csv writer would quote."""
    vals = values.tolist()
    try:
        blob = "\0".join(vals)
    except TypeError:
        try:
            vals = ["" if v is None or v != v else v for v in vals]
        except (TypeError, ValueError):                       # pd.NA and friends: truth value undefined
            return None
        if not all(isinstance(v, str) for v in vals):
            return None
        blob = "\0".join(vals)
    blob = (blob + "\0").encode("utf-8") if vals else b""
    if any(c in blob for c in _CSV_SPECIAL):
        return None
    seps = np.flatnonzero(np.frombuffer(blob, dtype=np.uint8) == 0)
    if seps.size != len(vals):
        return None
    starts = np.zeros(len(vals) + 1, dtype=np.int64)
    starts[1:] = seps + 1
    return blob, starts


def _blob_column(blob: bytes, n: int):
    """(bytes, starts) of ``n`` strings laid end to end with a NUL behind each, or None when the csv writer would quote one."""
    if any(c in blob for c in _CSV_SPECIAL):
        return None
    seps = np.flatnonzero(np.frombuffer(blob, dtype=np.uint8) == 0)
    if seps.size != n:
        return None
    starts = np.zeros(n + 1, dtype=np.int64)
    starts[1:] = seps + 1
    return blob, starts


class _Prepared:
    """Columns of a table in the form ``jg_table_format`` reads them (:func:`_prepare_columns`)."""

    def __init__(self, names, n, kinds, ptrs, starts, keep):
        self.names, self.n, self.kinds, self.ptrs, self.starts, self.keep = names, n, kinds, ptrs, starts, keep

    def render(self, rows: np.ndarray | None, header: bool, fd: int | None = None):
        """The text of rows ``rows`` (None: all) - returned as bytes, or written to the open file descriptor ``fd``
        (``jg_table_write``: the library's buffers go straight to the file; returns the number of bytes)."""
        import ctypes as C
        import os

        from . import _lib
        lib = _lib.load()
        nc = len(self.names)
        kind_arr = np.asarray(self.kinds, dtype=np.int32)
        col_ptrs = (C.c_void_p * nc)(*self.ptrs)
        start_ptrs = (C.c_void_p * nc)(*[None if st is None else st.ctypes.data for st in self.starts])
        if rows is not None:
            rows = np.ascontiguousarray(rows, dtype=np.int64)
        n_out = self.n if rows is None else len(rows)
        head = ("\t".join(self.names) + "\n").encode("utf-8") if header else b""
        rows_ptr = None if rows is None else rows.ctypes.data
        if fd is not None:
            done = 0
            while done < len(head):
                done += os.write(fd, head[done:])
            size = C.c_int64(0)
            if n_out:
                _lib.check(lib.jg_table_write(nc, kind_arr.ctypes.data, col_ptrs, start_ptrs, rows_ptr, n_out, 0, int(fd),
                                              C.byref(size)), "jg_table_write")
            return len(head) + size.value
        body = b""
        if n_out:
            text, size = C.c_void_p(), C.c_int64()
            _lib.check(lib.jg_table_format(nc, kind_arr.ctypes.data, col_ptrs, start_ptrs, rows_ptr, n_out, 0, C.byref(text),
                                           C.byref(size)), "jg_table_format")
            try:
                body = C.string_at(text, size.value)
            finally:
                lib.jg_table_free(text)
        return head + body


def _prepare_columns(names: list, values: list) -> _Prepared | None:
    """A table given as columns - arrays, lists of strings, :class:`_Summaries`, :class:`SpanColumn`, :class:`EnumColumn` - in
    the form the library's ``jg_table_format`` prints as ``to_csv(sep="\t", index=False, float_format="%.3f")`` would print the
    frame made of them; None when a column is of a kind the library does not print (the caller goes through pandas then)."""
    from . import _lib
    n = len(values[0]) if values else 0
    if not (len(names) > 1 and n > 0 and all(isinstance(c, str) and not any(ch in c for ch in '\t"\n\r') for c in names)):
        return None
    kinds, keep, ptrs, starts = [], [], [], []
    for v in values:
        st = None
        if isinstance(v, EnumColumn):
            v = v.as_spans()
        if isinstance(v, SpanColumn):
            if not v.csv_safe():
                return None
            st = v.spans()
            if len(v) != n:
                return None
            starts.append(st)
            kinds.append(_lib.JG_COL_SPANS)
            keep.append(v)
            ptrs.append(v.buf.ctypes.data)
            continue
        if isinstance(v, _Summaries) and v.blob is not None:
            got = _blob_column(v.blob, len(v))
        elif isinstance(v, _Summaries):
            got = _string_column(np.asarray(v.texts, dtype=object))
        else:
            v = np.asarray(v) if not isinstance(v, np.ndarray) else v
            kind = v.dtype.kind
            got = None
            if kind == "f":
                arr, k = np.ascontiguousarray(v, dtype=np.float64), _lib.JG_COL_FLOAT
            elif kind in "iu" and v.dtype != np.uint64:
                arr, k = np.ascontiguousarray(v, dtype=np.int64), _lib.JG_COL_INT
            elif kind == "b":
                arr, k = np.ascontiguousarray(v).view(np.uint8), _lib.JG_COL_BOOL
            elif kind in "OU":
                got = _string_column(v)
                if got is None:
                    return None
            else:
                return None
        if got is not None or isinstance(v, _Summaries):
            if got is None:
                return None
            blob, st = got
            arr, k = np.frombuffer(blob, dtype=np.uint8), _lib.JG_COL_STRING
            keep.append(blob)
        if (len(st) - 1 if st is not None else len(arr)) != n:
            return None
        starts.append(st)
        kinds.append(k)
        keep.append(arr)
        ptrs.append(arr.ctypes.data)
    return _Prepared(list(names), n, kinds, ptrs, starts, keep)


def _columns_text(names: list, values: list, rows: np.ndarray | None, header: bool) -> bytes | None:
    """Rows ``rows`` (None: all) of a table given as columns, rendered by the library; None when a column is of a kind the
    library does not print."""
    prep = _prepare_columns(names, values)
    return None if prep is None else prep.render(rows, header)


def _tsv_bytes(df: pd.DataFrame, header: bool = True) -> bytes:
    """The same text as UTF-8 bytes, rendered by the library's ``jg_table_format`` (include/jaeger_hip.h) on every usable
    core: floats by an exact ``"%.3f"``, integers, booleans and strings as pandas prints them.  A table with a column
    it does not know, or with a string the csv writer would quote, goes through ``_tsv_text``."""
    cols = list(df.columns)
    text = _columns_text(cols, [df[c].to_numpy() for c in cols], None, header) if len(df) else None
    return text if text is not None else _tsv_text(df, header).encode("utf-8")


def _to_tsv(df: pd.DataFrame, path) -> None:
    with open(path, "wb") as fh:
        fh.write(_tsv_bytes(df))


class TableWriter:
    """``write_output`` (collect.py:561-608) in pieces: batches of whole contigs are appended to ``<base>.tsv`` and - the rows
    that pass the phage filters - to ``<base>_phages.tsv`` as they are aggregated, so that only the last batch is formatted
    behind the forward.  Every step of ``write_output`` is row-wise (left merge with the repeat table, the ``N% < 0.3``
    filter, the phage query, the column formats), so the files equal the ones a single call writes, byte for byte
    (tests/test_postprocess.py); the phage file only comes into being with its first row, as in the reference.  Rows go to
    ``<path>.partial`` and ``close()`` renames the files into place: a run that fails later (the forward, the repeat scan,
    a later batch) leaves no truncated table at the final path - ``abort()`` closes and removes the partial files."""

    def __init__(self, labels, indices, output_table_path, output_phage_table_path, reliability_cutoff=0.5, phage_score=1,
                 refined_contig=None, columns_path: bool = True):
        self.kw = dict(labels=labels, indices=indices, refined_contig=refined_contig)
        self.columns_path = columns_path          # False: every batch through the DataFrame (what the tests compare with)
        self.table_path, self.phage_path = output_table_path, output_phage_table_path
        self.rc, self.pc = reliability_cutoff, phage_score
        lower = [label.lower() for label in labels]
        self.viral = labels[lower.index("phage")] if "phage" in lower else \
            (labels[lower.index("virus")] if "virus" in lower else "phage")
        self.fh = self.fh_phage = None
        self.rows = 0
        self.header_written = False

    def append(self, data: dict) -> None:
        if self.columns_path and self.kw.get("refined_contig") is None and self._append_columns(data):
            return
        df = generate_summary(data, **self.kw).query("`N%` < 0.3")
        self._write(_tsv_bytes(df, header=not self.header_written), len(df))
        clause = f" and (reliability_score > {self.rc})" if data.get("has_reliability", True) else ""
        phage_df = df.query(f'(prediction == "{self.viral}") and ({self.viral}_score > {self.pc}){clause}')
        if not phage_df.empty:
            self._write_phage(_tsv_bytes(phage_df, header=self.fh_phage is None))

    # (unbuffered files: the library writes a batch's rows through the descriptor itself, ``jg_table_write``)
    def _open(self) -> None:
        if self.fh is None:
            self.fh = open(f"{self.table_path}.partial", "wb", buffering=0)

    def _open_phage(self) -> None:
        if self.fh_phage is None:
            self.fh_phage = open(f"{self.phage_path}.partial", "wb", buffering=0)

    @staticmethod
    def _write_all(fh, text: bytes) -> None:
        view = memoryview(text)
        while len(view):
            view = view[fh.write(view):]

    def _write(self, text: bytes, n_rows: int) -> None:
        self._open()
        self._write_all(self.fh, text)
        self.header_written = True
        self.rows += n_rows

    def _write_phage(self, text: bytes) -> None:
        self._open_phage()
        self._write_all(self.fh_phage, text)

    def _append_columns(self, data: dict) -> bool:
        """The same rows without a DataFrame: the columns as ``_summary_columns`` makes them, the repeat table joined on by
        row number (``data["repeat_rows"]``, set by the pipeline: row of every contig in the repeat table or -1) or by one
        hash lookup per name, the ``N% < 0.3`` filter and the phage query as masks over the column arrays, the text of the
        surviving rows straight from ``jg_table_format``.  False (nothing written) where that is not the frame pandas would
        build: repeated names in the repeat table, a column the library does not print."""
        n = len(data["headers"])
        if n == 0:
            return False
        cols = _summary_columns(data, **self.kw)
        repeats = data.get("repeats")
        at = data.get("repeat_rows")
        kinds_col = None                                              # (an object array only where a DataFrame's column is joined)
        length_col = np.full(n, np.nan, dtype=np.float64)
        if repeats is not None and hasattr(repeats, "frame"):        # termini.RepeatColumns: arrays, no DataFrame yet
            if at is not None and data.get("names_unique") and len(repeats):
                at = np.asarray(at, dtype=np.int64)
                hit = at >= 0
                codes = np.zeros(n, np.int8)
                codes[hit] = repeats.kind_code[at[hit]]
                kinds_col = EnumColumn(codes, repeats.KIND_LABELS)
                length_col[hit] = repeats.length[at[hit]]
                repeats = None
            else:
                repeats = repeats.frame()
        if kinds_col is None:
            kinds_col = np.full(n, np.nan, dtype=object)
        if repeats is not None and len(repeats):
            index = repeats.attrs.get("_contig_index")
            if index is None:
                index = repeats.attrs["_contig_index"] = pd.Index(repeats["contig_id"].to_numpy(dtype=object))
            if not index.is_unique:
                return False
            if at is None:
                at = index.get_indexer(np.asarray(cols["contig_id"], dtype=object))
            at = np.asarray(at, dtype=np.int64)
            hit = at >= 0
            rk, rl = repeats["terminal_repeats"].to_numpy(dtype=object), repeats["repeat_length"].to_numpy()
            if rl.dtype.kind not in "fiu":
                return False
            kinds_col[hit] = rk[at[hit]]
            length_col[hit] = rl[at[hit]]
            if rl.dtype.kind in "iu" and hit.all():
                length_col = rl[at]                                     # (merge keeps integers that get no holes)
        elif repeats is not None and not {"terminal_repeats", "repeat_length"} <= set(repeats.columns):
            return False
        cols["terminal_repeats"], cols["repeat_length"] = kinds_col, length_col
        if isinstance(cols["contig_id"], SpanColumn):
            if cols["contig_id"].contains(b"___"):                    # (rare: the object path below turns ___ back into commas)
                cols["contig_id"] = cols["contig_id"].to_objects()
        if not isinstance(cols["contig_id"], SpanColumn):
            ids = np.asarray(cols["contig_id"]).tolist()
            if "___" in "\0".join(ids):                             # (io.py:109 wrote commas as ___; collect.py:556 turns them back)
                cols["contig_id"] = np.array([x.replace("___", ",") for x in ids], dtype=object)
        with np.errstate(invalid="ignore"):
            keep = np.asarray(cols["N%"]) < 0.3
        rows = None if keep.all() else np.flatnonzero(keep)
        names = list(cols)
        score_name = f"{self.viral}_score"
        if score_name not in cols or not isinstance(cols["prediction"], (np.ndarray, EnumColumn)):
            return False
        prep = _prepare_columns(names, [cols[c] for c in names])
        if prep is None:
            return False
        with np.errstate(invalid="ignore"):
            phage = keep & (cols["prediction"] == self.viral) & (np.asarray(cols[score_name]) > self.pc)
            if data.get("has_reliability", True):
                phage &= np.asarray(cols["reliability_score"]) > self.rc
        self._open()
        prep.render(rows, header=not self.header_written, fd=self.fh.fileno())
        self.header_written = True
        self.rows += int(keep.sum())
        if phage.any():
            first = self.fh_phage is None
            self._open_phage()
            prep.render(np.flatnonzero(phage), header=first, fd=self.fh_phage.fileno())
        return True

    def close(self) -> int:
        import os
        for fh, path in ((self.fh, self.table_path), (self.fh_phage, self.phage_path)):
            if fh is not None:
                fh.close()
                os.replace(f"{path}.partial", path)
        self.fh = self.fh_phage = None
        return self.rows

    def abort(self) -> None:
        """Drop what has been written so far (the run failed): close the handles, remove the ``.partial`` files."""
        import os
        for fh, path in ((self.fh, self.table_path), (self.fh_phage, self.phage_path)):
            if fh is not None:
                fh.close()
                try:
                    os.unlink(f"{path}.partial")
                except OSError:
                    pass
        self.fh = self.fh_phage = None


def write_output(data: dict, reliability_cutoff: float = 0.5, phage_score=1, **kwargs) -> int:
    """Write ``<base>.tsv`` and (if non-empty) ``<base>_phages.tsv`` (collect.py:561-608)."""
    w = TableWriter(kwargs.get("labels", []), kwargs.get("indices"), kwargs.get("output_table_path"),
                    kwargs.get("output_phage_table_path"), reliability_cutoff, phage_score, kwargs.get("refined_contig"))
    w.append(data)
    return w.close()


def write_fasta_from_results(input_fasta, output_tsv, output_fasta, width: int = 70) -> int:
    """``--getsequences`` (postprocess/collect.py:613-639, commands/predict.py:444-456): the records of the input FASTA
    whose name is a ``contig_id`` of the phage table, as read from the file (original case), ``width`` bases per line.
    Returns the number of records written."""
    import os

    from .fragment import load_fasta
    if not os.path.exists(str(output_tsv)):              # no contig passed the phage filters: an empty FASTA
        open(str(output_fasta), "wb").close()
        return 0
    phages = set(pd.read_table(str(output_tsv))["contig_id"].to_list())
    fa = load_fasta(str(input_fasta))
    n = 0
    with open(str(output_fasta), "wb") as out:
        for i, name in enumerate(fa.names):
            if name in phages:
                out.write(b">" + name.encode() + b"\n")
                seq = fa.sequence(i)
                out.write(b"".join(seq[j:j + width] + b"\n" for j in range(0, len(seq), width)))
                n += 1
    return n
