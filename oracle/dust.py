"""TEST INFRASTRUCTURE - symmetric DUST by its definition (Morgulis, Gertz, Schaffer, Agarwala 2006,
"A fast and symmetric DUST implementation to mask low-complexity DNA sequences", J Comput Biol 13).

A sequence of triplets x with l > 1 triplets scores S(x) = sum_t c_t (c_t - 1) / 2 / (l - 1), c_t =
occurrences of triplet t in x.  x is *perfect* if S(x) > T / 10 and no sub-interval scores higher.
SDUST masks the union of all perfect intervals that fit a window of W bases (l <= W - 2).  This file
evaluates that definition directly (O(n W) dynamic programme over all intervals), independently of
the streaming formulation in ``jaeger_amd/csrc/jg_dust.hip``.  The reference calls
``pydustmasker.DustMasker(seq, window_size=64, score_threshold=20).mask()`` (seqops/io.py:104-108);
pydustmasker itself is not installable here, so parity with it is unpinned.
"""

from __future__ import annotations

import numpy as np

_CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3, ord("a"): 0, ord("c"): 1, ord("g"): 2, ord("t"): 3}


def sdust_intervals(seq: bytes, window: int = 64, threshold: int = 20) -> list[tuple[int, int]]:
    """Merged masked intervals [start, end) in base coordinates."""
    n = len(seq)
    code = np.array([_CODE.get(b, 4) for b in seq], np.int64)
    mask = np.zeros(n, bool)
    lmax = window - 2
    # maximal runs of unambiguous bases
    i = 0
    while i < n:
        if code[i] > 3:
            i += 1
            continue
        j = i
        while j < n and code[j] < 4:
            j += 1
        run = code[i:j]
        m = len(run) - 2                                   # triplets in the run
        if m >= 2:
            tri = run[:-2] * 16 + run[1:-1] * 4 + run[2:]
            # S[l][s]: score of the interval of l triplets starting at triplet s
            S = np.full((lmax + 1, m), -1.0)
            M = np.full((lmax + 1, m), -1.0)                # max score over all sub-intervals (l >= 2)
            r = np.zeros(m, np.int64)                       # running repeat count per start
            counts = np.zeros((m, 64), np.int64)
            for l in range(1, min(lmax, m) + 1):
                starts = np.arange(0, m - l + 1)
                t = tri[starts + l - 1]
                r[starts] += counts[starts, t]
                counts[starts, t] += 1
                if l >= 2:
                    S[l, starts] = r[starts] / (l - 1)
                    sub = np.maximum(M[l - 1, starts], M[l - 1, starts + 1]) if l >= 3 else np.full(len(starts), -1.0)
                    M[l, starts] = np.maximum(S[l, starts], sub)
                    perfect = (S[l, starts] * 10 > threshold) & (S[l, starts] >= sub)
                    for s in starts[perfect]:
                        mask[i + s:i + s + l + 2] = True
        i = j
    out = []
    p = 0
    while p < n:
        if mask[p]:
            q = p
            while q < n and mask[q]:
                q += 1
            out.append((p, q))
            p = q
        else:
            p += 1
    return out


def soft_mask(seq: bytes, window: int = 64, threshold: int = 20) -> bytes:
    """Upper-case, then lower-case the masked intervals (what ``DustMasker.mask()`` returns)."""
    s = bytearray(seq.upper())
    for a, b in sdust_intervals(seq, window, threshold):
        s[a:b] = bytes(s[a:b]).lower()
    return bytes(s)
