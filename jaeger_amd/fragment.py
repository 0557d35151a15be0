"""Host side of the fragmenter: FASTA records -> window table + metadata.

The reference's ``fragment_generator`` (``seqops/io.py:74-147``) materialises one
Python string per window (slice + 4x ``str.count`` + f-string).  Here the host
only computes *where* the windows are (``_window_indices``, ``io.py:38-71``,
vectorised over contigs); slicing, base counting and encoding happen on the GPU
(``jg_encode`` / ``jg_predict_windows``), and the ten metadata fields the
post-processing needs (``postprocess/collect.py:259-327``) are produced as numpy
arrays in the reference's field order.
"""

from __future__ import annotations

import gzip
import math
from dataclasses import dataclass
from typing import Iterable, Iterator

import numpy as np


def read_fasta(path: str) -> Iterator[tuple[str, bytes]]:
    """(name, sequence bytes); name = header up to the first whitespace
    (``pyfastx.Fasta(build_index=False)`` semantics, io.py:98-103)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    name, chunks = None, []
    with opener(path, "rb") as fh:
        for line in fh:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    yield name, b"".join(chunks)
                fields = line[1:].split()
                name = fields[0].decode() if fields else ""
                chunks = []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        yield name, b"".join(chunks)


@dataclass
class FastaBatch:
    """All records of a FASTA file in the layout the encoder scans."""
    names: "list[str] | Names"     # (the native ingest hands out Names: the strings are made on demand)
    bases: np.ndarray          # uint8, all sequences back to back
    offsets: np.ndarray        # int64, len = n_records + 1

    def __len__(self) -> int:
        return len(self.names)

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets)

    def sequence(self, i: int) -> bytes:
        return self.bases[self.offsets[i]:self.offsets[i + 1]].tobytes()


_NOT_PLAIN = bytes(range(33)) + b",\x85\xa0"


class Names:
    """Record names as the native FASTA parser hands them out: ONE byte buffer + offsets (``name i`` = ``buf[off[i]:off[i+1]]``,
    UTF-8).  Behaves like the ``list[str]`` it stands for - ``len``, indexing, iteration, ``==`` with a list - but the
    Python strings are only made when something asks for them: a million short records' names cost 0.3 s as objects, and the
    table writer takes them as bytes (:meth:`take_bytes`), the repeat join by record number needs only :meth:`is_unique`."""

    def __init__(self, buf: np.ndarray, off: np.ndarray):
        self.buf = np.ascontiguousarray(buf, np.uint8)
        self.off = np.ascontiguousarray(off, np.int64)
        self._list: list[str] | None = None
        self._plain: bool | None = None
        self._unique: bool | None = None
        self.facts: dict = {}              # what the table writer has found out about ``buf`` (postprocess.SpanColumn.facts)

    def __len__(self) -> int:
        return max(len(self.off) - 1, 0)

    def tolist(self) -> list[str]:
        if self._list is None:
            raw, no = self.buf.tobytes(), self.off.tolist()
            self._list = [raw[no[i]:no[i + 1]].decode() for i in range(len(no) - 1)]
        return self._list

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            if self._list is not None:
                return self._list[i]
            n = len(self)
            j = int(i) + n if i < 0 else int(i)
            if not 0 <= j < n:
                raise IndexError(i)
            return self.buf[self.off[j]:self.off[j + 1]].tobytes().decode()
        return self.tolist()[i]

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        if isinstance(other, Names):
            return np.array_equal(self.off - self.off[:1], other.off - other.off[:1]) and \
                np.array_equal(self.buf[self.off[0]:self.off[-1]], other.buf[other.off[0]:other.off[-1]])
        try:
            return self.tolist() == list(other)
        except TypeError:
            return NotImplemented

    def __repr__(self) -> str:
        return f"Names({len(self)} records, {int(self.off[-1] - self.off[0]) if len(self.off) else 0} bytes)"

    def plain(self) -> bool:
        """No comma, no white space, no NUL in any name: ``name.strip().replace(",", "___")`` (io.py:109) leaves them as they
        are, and NUL can separate them in a blob."""
        if self._plain is None:
            b = self.buf[self.off[0]:self.off[-1]] if len(self) else self.buf[:0]
            # one pass (bytes.translate): comma, NUL / controls / space, and the bytes 0x85 / 0xA0 (the tails of the two-byte
            # white-space characters str.strip() also removes; conservative: other characters end in them too)
            raw = b.tobytes()
            self._plain = len(raw.translate(None, _NOT_PLAIN)) == len(raw)
        return self._plain

    def spans(self, idx=None) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(buf, begin, end): the name of record ``idx[r]`` (None: every record) is ``buf[begin[r]:end[r]]`` - the form the
        table writer hands to ``jg_table_format`` as a JG_COL_SPANS column (no copy of the names, no Python strings)."""
        if idx is None:
            return self.buf, self.off[:-1], self.off[1:]
        idx = np.asarray(idx, np.int64)
        return self.buf, self.off[idx], self.off[idx + 1]

    def is_unique(self) -> bool:
        """No name twice (``jg_names_unique``: 64-bit hashes on every core, one open-addressing pass, equal hashes settled by
        comparing the bytes)."""
        if self._unique is None:
            import ctypes as C

            from . import _lib as L
            flag = C.c_int32(1)
            L.check(L.load().jg_names_unique(C.c_void_p(self.buf.ctypes.data), C.c_void_p(self.off.ctypes.data), len(self), 0,
                                             C.byref(flag)), "jg_names_unique")
            self._unique = bool(flag.value)
        return self._unique


def _decode_names(names_buf: np.ndarray, name_off: np.ndarray) -> Names:
    return Names(names_buf, name_off)


def _scan_fill(text: np.ndarray, threads: int = 0, want_bases: bool = True, want_rec_off: bool = False):
    """``jg_fasta_scan`` + ``jg_fasta_fill`` over a file image: (names, bases or None, offsets, rec_off or None)."""
    import ctypes as C

    from . import _lib as L
    lib = L.load()
    ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731   (read-only memory maps have no writable data_as)
    h = C.c_void_p()
    n_rec, n_bases, name_bytes = C.c_int64(), C.c_int64(), C.c_int64()
    L.check(lib.jg_fasta_scan(ptr(text), text.size, int(threads), C.byref(h), C.byref(n_rec), C.byref(n_bases),
                              C.byref(name_bytes)), "jg_fasta_scan")
    try:
        n = n_rec.value
        bases = np.empty(n_bases.value, np.uint8) if want_bases else None
        offsets, name_off = np.empty(n + 1, np.int64), np.empty(n + 1, np.int64)
        names_buf = np.empty(max(name_bytes.value, 1), np.uint8)
        rec_off = np.empty(n + 1, np.int64) if want_rec_off else None
        L.check(lib.jg_fasta_fill(h, ptr(bases) if want_bases else None, ptr(offsets), ptr(names_buf), ptr(name_off),
                                  ptr(rec_off) if want_rec_off else None), "jg_fasta_fill")
    finally:
        lib.jg_fasta_scan_free(h)
    return _decode_names(names_buf[:name_bytes.value], name_off), bases, offsets, rec_off


def load_fasta(path, threads: int = 0) -> FastaBatch:
    """Native ingest on every core of the CPU quota (``jg_fasta_scan`` / ``jg_fasta_fill`` over a read-only memory map
    of the file); same record rules as :func:`read_fasta`."""
    names, bases, offsets, _ = _scan_fill(_read_text(path), threads)
    return FastaBatch(names, bases, offsets)


def _read_text(path) -> np.ndarray:
    """File image as uint8: a read-only memory map for plain files (pages are touched only where a rank reads its
    own records), the decompressed stream for ``.gz`` (not seekable: every rank inflates it)."""
    if str(path).endswith(".gz"):
        with gzip.open(path, "rb") as fh:
            return np.frombuffer(fh.read(), np.uint8)
    import os
    if os.path.getsize(path) == 0:
        return np.zeros(0, np.uint8)
    return np.memmap(path, dtype=np.uint8, mode="r")


@dataclass
class FastaIndex:
    """Where the records of a FASTA file are, without their sequences (``jg_fasta_index``)."""
    names: list[str]
    rec_off: np.ndarray        # int64, len = n_records + 1: byte offset of each header line, last = file size
    lengths: np.ndarray        # int64 sequence lengths

    def __len__(self) -> int:
        return len(self.names)


def index_fasta(path, threads: int = 0) -> FastaIndex:
    names, _, offsets, rec_off = _scan_fill(_read_text(path), threads, want_bases=False, want_rec_off=True)
    return FastaIndex(names, rec_off, np.diff(offsets))


def load_fasta_records(path, rec_off: np.ndarray, records) -> FastaBatch:
    """Only the listed records (ascending indices into ``rec_off``) of a FASTA file: their byte ranges are read
    (adjacent ones merged) and parsed; nothing else of the file is touched."""
    import ctypes as C

    from . import _lib as L
    lib = L.load()
    records = np.asarray(records, np.int64)
    if records.size == 0:
        return FastaBatch([], np.zeros(0, np.uint8), np.zeros(1, np.int64))
    text = _read_text(path)
    a, b = rec_off[records], rec_off[records + 1]
    cut = np.nonzero(a[1:] != b[:-1])[0] + 1                  # runs of adjacent records -> one read each
    run_a, run_b = a[np.concatenate(([0], cut))], b[np.concatenate((cut - 1, [records.size - 1]))]
    buf = np.empty(int((run_b - run_a).sum()), np.uint8)
    o = 0
    for x, y in zip(run_a.tolist(), run_b.tolist()):
        buf[o:o + (y - x)] = text[x:y]
        o += y - x
    ptr = lambda arr: arr.ctypes.data_as(C.c_void_p)  # noqa: E731
    n = records.size
    offsets, name_off = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
    # names: jg_fasta_parse copies every header's first token with no capacity argument, so the buffer is sized from
    # the exact upper bound jg_fasta_count computes over these very bytes (the sum of the header-line lengths)
    cnt, name_bytes = C.c_int64(), C.c_int64()
    L.check(lib.jg_fasta_count(ptr(buf), buf.size, C.byref(cnt), C.byref(name_bytes)), "jg_fasta_count")
    if cnt.value != n:
        raise ValueError(f"{path}: expected {n} records in the selected byte ranges, found {cnt.value}")
    names_buf = np.zeros(max(name_bytes.value, 1), np.uint8)
    got, nb = C.c_int64(), C.c_int64()
    L.check(lib.jg_fasta_parse(ptr(buf), buf.size, n, ptr(buf), ptr(offsets), ptr(names_buf), ptr(name_off),
                               C.byref(got), C.byref(nb)), "jg_fasta_parse")
    if got.value != n:
        raise ValueError(f"{path}: expected {n} records in the selected byte ranges, parsed {got.value}")
    raw, no = names_buf.tobytes(), name_off.tolist()
    return FastaBatch([raw[no[i]:no[i + 1]].decode() for i in range(n)], buf[:nb.value], offsets)


def dust_mask(fa: FastaBatch, window: int = 64, threshold: int = 20, threads: int = 0) -> int:
    """Soft-mask low-complexity intervals of every record in place (``jg_dust_mask``): all bases
    upper-cased, DUST intervals lower-cased - what ``fragment_generator`` does per contig with
    ``pydustmasker`` (io.py:104-108).  Returns the number of masked bases.  Afterwards pass
    ``pre_cased=True`` to the encoder."""
    import ctypes as C

    from . import _lib as L
    lib = L.load()
    n = C.c_int64()
    if not fa.bases.flags.writeable or not fa.bases.flags.c_contiguous:
        fa.bases = np.ascontiguousarray(fa.bases).copy()
    offsets = np.ascontiguousarray(fa.offsets, np.int64)
    L.check(lib.jg_dust_mask(fa.bases.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), len(fa),
                             int(window), int(threshold), int(threads), C.byref(n)), "jg_dust_mask")
    return int(n.value)


def window_indices(seqlen: int, fragsize: int, stride: int | None, dynamic_stride: bool = False,
                   dynamic_stride_threshold: float = 10.0) -> list[int]:
    """Window starts of one contig (io.py:38-71)."""
    if dynamic_stride and seqlen < dynamic_stride_threshold * fragsize:
        n = max(1, math.ceil(seqlen / fragsize))
        if n == 1:
            return [0]
        raw = (seqlen - fragsize) / (n - 1)
        starts = [int(round(i * raw)) for i in range(n)]
        starts[-1] = seqlen - fragsize
        return list(dict.fromkeys(starts))
    step = fragsize if stride is None else stride
    return list(range(0, seqlen - (fragsize - 1), step))


@dataclass
class WindowTable:
    """Windows of a set of contigs, in the reference's emission order."""
    contig: np.ndarray      # (W,) index into the record list
    start: np.ndarray       # (W,) offset inside the contig            (meta_1 "index")
    length: np.ndarray      # (W,) window length (< fsize: whole-contig window)
    is_last: np.ndarray     # (W,) 1 on the last window of a contig    (meta_2)
    ordinal: np.ndarray     # (W,) window number inside the contig     (meta_3)
    seqlen: np.ndarray      # (W,) contig length                       (meta_4)

    def __len__(self) -> int:
        return int(self.contig.size)


def build_window_table(lengths: Iterable[int], fragsize: int, stride: int | None = None,
                       dynamic_stride: bool = False, dynamic_stride_threshold: float = 10.0,
                       min_len: int | None = None, max_len: int | None = None) -> WindowTable:
    """Vectorised ``fragment_generator`` control flow (io.py:110-145) over contig lengths."""
    lengths = np.asarray(list(lengths) if not isinstance(lengths, np.ndarray) else lengths, np.int64)
    if min_len is None:
        min_len = fragsize
    step = fragsize if stride is None else stride
    keep = np.ones(lengths.size, bool)
    if max_len is not None:
        keep &= lengths <= max_len                                   # io.py:110-111
    long_ = keep & (lengths >= fragsize)
    short = keep & (lengths < fragsize) & (lengths >= min_len)       # io.py:134
    n_win = np.zeros(lengths.size, np.int64)
    n_win[long_] = (lengths[long_] - fragsize) // step + 1
    n_win[short] = 1
    dyn = np.zeros(lengths.size, bool)
    dyn_starts: dict[int, list[int]] = {}
    if dynamic_stride:
        dyn = long_ & (lengths < dynamic_stride_threshold * fragsize)
        for ci in np.nonzero(dyn)[0]:
            s = window_indices(int(lengths[ci]), fragsize, stride, True, dynamic_stride_threshold)
            dyn_starts[int(ci)] = s
            n_win[ci] = len(s)
    total = int(n_win.sum())
    contig = np.repeat(np.arange(lengths.size, dtype=np.int64), n_win)
    first = np.cumsum(n_win) - n_win
    ordinal = np.arange(total, dtype=np.int64) - np.repeat(first, n_win)
    start = ordinal * step
    for ci, s in dyn_starts.items():
        start[first[ci]:first[ci] + n_win[ci]] = s
    seqlen = lengths[contig]
    length = np.minimum(seqlen, fragsize).astype(np.int32)
    start[short[contig]] = 0
    is_last = (ordinal == n_win[contig] - 1).astype(np.int32)
    return WindowTable(contig, start, length, is_last, ordinal, seqlen)


def concat_records(seqs: list[bytes]) -> tuple[np.ndarray, np.ndarray]:
    """One contiguous base buffer + per-contig offsets (the layout ``jg_encode`` scans)."""
    lengths = np.fromiter((len(s) for s in seqs), np.int64, len(seqs))
    offsets = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    buf = np.empty(int(offsets[-1]), np.uint8)
    for s, o in zip(seqs, offsets[:-1]):
        buf[o:o + len(s)] = np.frombuffer(s, np.uint8)
    return buf, offsets


def safe_divide(numerator, denominator):
    """utils/misc.py:117-123 on arrays: round(n/d, 2), 0 where d == 0."""
    n = np.asarray(numerator, np.float64)
    d = np.asarray(denominator, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(d != 0, np.round(n / np.where(d != 0, d, 1.0), 2), 0.0)
    return out


def normalise_headers(headers: list[str]) -> np.ndarray:
    """io.py:109: ``name.strip().replace(",", "___")`` for every record, as an object array.  Names that hold neither a comma
    nor white space (every name the FASTA parser hands out: a header up to its first white space) are taken as they are -
    one scan of the joined names instead of two string calls per record (0.5 s per million records)."""
    if isinstance(headers, Names) and headers.plain():
        out = np.empty(len(headers), dtype=object)
        out[:] = headers.tolist()
        return out
    if len(headers) > 64:
        blob = "\x00".join(headers)
        if "," not in blob and len(blob.split()) <= 1:
            out = np.empty(len(headers), dtype=object)
            out[:] = headers
            return out
    return np.array([h.strip().replace(",", "___") for h in headers], dtype=object)


def window_metadata(table: WindowTable, headers, counts: np.ndarray, normalised: bool = False) -> dict[str, np.ndarray]:
    """meta_0..meta_9 as ``InferModel.predict`` returns them (inference.py:365-367):
    header, index, contig_end, i, seqlen, g, c, a, t, gc_skew (io.py:128-133).  ``meta_0`` is an object array that
    shares the per-record strings (a fixed-width copy per window would cost more than every other field together);
    ``normalised``: ``headers`` already is :func:`normalise_headers`' array."""
    hdr = headers if (normalised or headers is None) else normalise_headers(headers)
    g, c, a, t = (counts[:, i].astype(np.int64) for i in range(4))
    skew = safe_divide(g - c, g + c)
    return {
        # (``headers`` None: the caller carries the contigs' names as bytes - no per-window header array is built)
        "meta_0": None if hdr is None else (hdr[table.contig] if len(table) else np.array([], dtype=object)),
        "meta_1": table.start.astype(np.int64),
        "meta_2": table.is_last.astype(np.int32),
        "meta_3": table.ordinal.astype(np.int64),
        "meta_4": table.seqlen.astype(np.int64),
        "meta_5": g, "meta_6": c, "meta_7": a, "meta_8": t,
        "meta_9": skew,
    }
