"""Oracle: sliding-window fragmenter (TEST INFRASTRUCTURE, see oracle/__init__.py).

Literal restatement of ``src/jaeger/seqops/io.py:38-147`` (``_window_indices``,
``fragment_generator``) and the two helpers it uses,
``src/jaeger/utils/misc.py:117-123`` (``safe_divide``) and ``:137-144``
(``signal_l``).  FASTA iteration replaces ``pyfastx.Fasta(build_index=False)``:
record name = header up to the first whitespace, sequence = concatenated lines
(pinned by ``tests/unit/test_seqops_io.py:66-101`` of the reference).
DUST masking (``pydustmasker``, io.py:105-108) is NOT restated: the oracle only
covers ``dustmask=False``; an optional ``soft_mask`` callable can inject one.
"""

from __future__ import annotations

import math
from typing import Callable, Iterator


def window_indices(seqlen, fragsize, stride, dynamic_stride, dynamic_stride_threshold):
    """io.py:38-71: window start offsets of one contig."""
    if dynamic_stride and seqlen < dynamic_stride_threshold * fragsize:
        # short contig: spread ceil(len/fsize) windows so the last one ends at
        # the contig end (io.py:56-71)
        n = max(1, math.ceil(seqlen / fragsize))
        if n == 1:
            return [0]
        raw = (seqlen - fragsize) / (n - 1)
        starts = [int(round(i * raw)) for i in range(n)]
        starts[-1] = seqlen - fragsize
        return list(dict.fromkeys(starts))  # order-preserving de-duplication
    step = fragsize if stride is None else stride
    return list(range(0, seqlen - (fragsize - 1), step))  # io.py:52-54


def safe_divide(numerator, denominator):
    """utils/misc.py:117-123: 2-decimal ratio, 0 on division by zero."""
    if denominator == 0:
        return 0
    return round(numerator / denominator, 2)


def signal_l(it):
    """utils/misc.py:137-144: pair every element with an is-last flag."""
    items = list(it)
    for j, v in enumerate(items):
        yield (1 if j == len(items) - 1 else 0), v


def read_fasta(path: str) -> Iterator[tuple[str, str]]:
    """(name, sequence) records like ``pyfastx.Fasta(path, build_index=False)``."""
    name = None
    chunks: list[str] = []
    with open(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks)
                fields = line[1:].split()
                name = fields[0] if fields else ""
                chunks = []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        yield name, "".join(chunks)


def fragment_strings(
    records,
    fragsize: int,
    stride: int | None = None,
    dynamic_stride: bool = False,
    dynamic_stride_threshold: float = 10.0,
    min_len: int | None = None,
    max_len: int | None = None,
    soft_mask: Callable[[str], str] | None = None,
) -> Iterator[str]:
    """io.py:74-147 over an iterable of (name, sequence) records.

    Yields ``"seq,header,index,contig_end,i,seqlen,g,c,a,t,gc_skew"``.
    """
    if min_len is None:
        min_len = fragsize
    for name, seq in records:
        seqlen = len(seq)
        sequence = seq.strip().upper()
        if soft_mask is not None:
            sequence = soft_mask(sequence)
        header = name.strip().replace(",", "___")
        if max_len is not None and seqlen > max_len:
            continue
        if seqlen >= fragsize:
            indices = window_indices(
                seqlen, fragsize, stride, dynamic_stride, dynamic_stride_threshold
            )
            for i, (b, index) in enumerate(signal_l(indices)):
                win = sequence[index : index + fragsize]
                g = win.count("G")
                c = win.count("C")
                a = win.count("A")
                t = win.count("T")
                gc_skew = safe_divide((g - c), (g + c))
                yield (
                    f"{win},{header},{index},{b},{i},{seqlen},{g},{c},{a},{t},"
                    f"{gc_skew: .3f}"
                )
        elif seqlen >= min_len:
            g = sequence.count("G")
            c = sequence.count("C")
            a = sequence.count("A")
            t = sequence.count("T")
            gc_skew = safe_divide((g - c), (g + c))
            yield (
                f"{sequence},{header},0,1,0,{seqlen},{g},{c},{a},{t},"
                f"{gc_skew: .3f}"
            )
