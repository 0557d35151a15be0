"""TEST INFRASTRUCTURE - terminal-repeat scan restated on the CPU (``utils/termini.py:88-189``):
Smith-Waterman local alignment with affine gaps and a full traceback, plain Python loops (small
cases only), plus the reference's decision rule.

Scoring as ``parasail.matrix_create("ACGT", 2, -100)`` with gap open 100 / extend 5 (a gap of k columns
costs 100 + 5 (k - 1)); letters compare case-insensitively and only A/C/G/T can match.  parasail is not
installable here: which of several co-optimal alignments its traceback reports is unpinned; this
restatement prefers the diagonal, then a gap in the query row, then a gap in the reference row, prefers
gap extension over gap opening, and ends at the first maximum in (reference, query) order.
"""

from __future__ import annotations

MATCH, MISMATCH, OPEN, EXT = 2, -100, 100, 5
NEG = -10 ** 9
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def reverse_complement(seq: str) -> str:
    """seqops/transform.py:11-35 restricted to what can ever match: ACGT (any case) complemented to
    upper case, everything else to a non-matching letter."""
    return "".join(_COMP.get(b.upper(), "N") for b in reversed(seq))


def smith_waterman(query: str, ref: str) -> dict:
    """Row-vectorised (numpy) evaluation of the same recurrences as :func:`smith_waterman_loops`, same
    traceback; used for end-sized inputs.  E needs no in-row recursion: a gap opened from a cell that was
    itself reached through a gap never beats extending that gap (open >= extend), so
    E[i][j] = max_{k<j}(H'[i][k] - open - ext (j-1-k)) with H' = max(0, diagonal, F), a prefix maximum."""
    import numpy as np
    q = np.frombuffer(query.upper().encode(), np.uint8)
    r = np.frombuffer(ref.upper().encode(), np.uint8)
    n, m = len(q), len(r)
    acgt = np.isin(r, np.frombuffer(b"ACGT", np.uint8))
    H = np.zeros((n + 1, m + 1), np.int64)
    E = np.full((n + 1, m + 1), NEG, np.int64)
    F = np.full((n + 1, m + 1), NEG, np.int64)
    jj = np.arange(m + 1)
    for i in range(1, n + 1):
        sub = np.where((r == q[i - 1]) & acgt, MATCH, MISMATCH)
        F[i, 1:] = np.maximum(F[i - 1, 1:] - EXT, H[i - 1, 1:] - OPEN)
        hp = np.zeros(m + 1, np.int64)
        hp[1:] = np.maximum(0, np.maximum(H[i - 1, :-1] + sub, F[i, 1:]))
        a = hp + EXT * jj                                   # H'[k] + ext k   (column 0: H = 0)
        best_prev = np.maximum.accumulate(a)[:-1]           # max over k <= j-1
        E[i, 1:] = best_prev - OPEN - EXT * (jj[1:] - 1)
        H[i, 1:] = np.maximum(hp[1:], E[i, 1:])
    best = int(H.max())
    if best == 0:
        return {"score": 0, "length": 0, "fgaps": 0, "rgaps": 0, "end_query": -1, "end_ref": -1}
    cols = np.nonzero((H == best).any(axis=0))[0]
    bj = int(cols[0])                                       # first maximum in (reference, query) order
    bi = int(np.nonzero(H[:, bj] == best)[0][0])
    i, j, state = bi, bj, "H"
    length = fgaps = rgaps = 0
    qs, rs = query.upper(), ref.upper()
    while True:
        if state == "H":
            if H[i, j] == 0:
                break
            sub = MATCH if (qs[i - 1] == rs[j - 1] and qs[i - 1] in "ACGT") else MISMATCH
            if H[i, j] == H[i - 1, j - 1] + sub:
                i, j, length = i - 1, j - 1, length + 1
            elif H[i, j] == E[i, j]:
                state = "E"
            else:
                state = "F"
        elif state == "E":
            length, fgaps = length + 1, fgaps + 1
            if E[i, j] != E[i, j - 1] - EXT:
                state = "H"
            j -= 1
        else:
            length, rgaps = length + 1, rgaps + 1
            if F[i, j] != F[i - 1, j] - EXT:
                state = "H"
            i -= 1
    return {"score": best, "length": length, "fgaps": fgaps, "rgaps": rgaps, "end_query": bi - 1, "end_ref": bj - 1}


def smith_waterman_loops(query: str, ref: str) -> dict:
    """-> score, alignment length (traceback columns), gaps in the query row, end positions."""
    q, r = query.upper(), ref.upper()
    n, m = len(q), len(r)
    H = [[0] * (m + 1) for _ in range(n + 1)]
    E = [[NEG] * (m + 1) for _ in range(n + 1)]
    F = [[NEG] * (m + 1) for _ in range(n + 1)]
    best, bi, bj = 0, 0, 0
    for j in range(1, m + 1):                 # column-major so that "first maximum" is (ref, query) order
        for i in range(1, n + 1):
            E[i][j] = max(E[i][j - 1] - EXT, H[i][j - 1] - OPEN)
            F[i][j] = max(F[i - 1][j] - EXT, H[i - 1][j] - OPEN)
            sub = MATCH if (q[i - 1] == r[j - 1] and q[i - 1] in "ACGT") else MISMATCH
            H[i][j] = max(0, H[i - 1][j - 1] + sub, E[i][j], F[i][j])
            if H[i][j] > best:
                best, bi, bj = H[i][j], i, j
    # traceback with the documented preferences
    i, j, state = bi, bj, "H"
    length = fgaps = rgaps = 0
    while True:
        if state == "H":
            if H[i][j] == 0:
                break
            sub = MATCH if (q[i - 1] == r[j - 1] and q[i - 1] in "ACGT") else MISMATCH
            if H[i][j] == H[i - 1][j - 1] + sub:
                i, j, length = i - 1, j - 1, length + 1
            elif H[i][j] == E[i][j]:
                state = "E"
            else:
                state = "F"
        elif state == "E":                    # column: ref base against a gap in the query row
            length, fgaps = length + 1, fgaps + 1
            if E[i][j] != E[i][j - 1] - EXT:
                state = "H"
            j -= 1
        else:
            length, rgaps = length + 1, rgaps + 1
            if F[i][j] != F[i - 1][j] - EXT:
                state = "H"
            i -= 1
    return {"score": best, "length": length, "fgaps": fgaps, "rgaps": rgaps, "end_query": bi - 1, "end_ref": bj - 1}


def scan_length(seq_len: int) -> int:
    return min(max(int(seq_len * 0.04), 400), 4000)


def classify(dtr: dict, itr: dict):
    """termini.py:137-154 + :58-63: -> (terminal_repeats, repeat_length) or (None, None)."""
    if itr["length"] > 12 or dtr["length"] > 12:
        if itr["score"] > dtr["score"]:
            return "ITR", itr["length"]
        kind = "DTR"
        if dtr["length"] - dtr["fgaps"] >= 250:
            kind = "LTR_DTR"
        return kind, dtr["length"]
    return None, None


def scan_record(seq: str, scan: int | None = None):
    scan = scan_length(len(seq)) if scan is None else scan
    front, rear = seq[:scan], seq[-scan:]
    return classify(smith_waterman(front, rear), smith_waterman(front, reverse_complement(rear)))
