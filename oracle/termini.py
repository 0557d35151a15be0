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
