"""Oracle: 6-frame codon encoder (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``process_string_inference(...)->p`` of
``src/jaeger/seqops/encode.py:203-318`` (with its lookup tables ``_map_codon``
``:20-25`` and ``_map_complement`` ``:28-33``) for ``input_type="translated"``,
``ngram_width=3`` (codons) or ``6`` (``codon: DICODON``, nnlib/inference.py:430-451: the 4 096 codon pairs of
``seqops/maps.py:544-546``), ``mutate=False``, ``shuffle=False``:

* :func:`encode_window_literal` is a step-by-step pure-Python restatement that
  mirrors the TF string ops (bytes_split / complement lookup / upper / ngrams /
  strided slices / hash lookup with default -1) - use on small cases;
* :func:`encode_windows` is the same arithmetic vectorised with numpy (used for
  larger parity inputs and as the CPU baseline's encoder leg).

Codon table order follows ``src/jaeger/seqops/maps.py:3-68``; it is generated
here from the NCBI standard-code string (TCAG order) and pinned against the
reference dump in ``tests/golden/maps.json``.
"""

from __future__ import annotations

import numpy as np

# NCBI translation table 1 in TCAG^3 order (first base slowest).
_NCBI_AAS = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
_B = "TCAG"
_NCBI_CODONS = [a + b + c for a in _B for b in _B for c in _B]
_AA_OF = dict(zip(_NCBI_CODONS, _NCBI_AAS))

#: reference codon order (maps.py:3-68): second base slowest, third fastest
CODONS = [b0 + b1 + b2 for b1 in _B for b0 in _B for b2 in _B]
CODON_ID = list(range(64))

_COMPLEMENT = {"A": "T", "T": "A", "G": "C", "C": "G", "a": "t", "t": "a", "g": "c", "c": "g"}
_OFFSET_LUT = (-2, -1, 0)  # encode.py:235 / :241


def frame_length(nucleotides: int) -> int:
    """Codons per frame for a crop of ``nucleotides`` (seqops/crop.py:44-61)."""
    nt = int(nucleotides)
    if nt < 3:
        return 0
    usable = (nt - 2) - (3 - _OFFSET_LUT[nt % 3])
    return 0 if usable <= 0 else -(-usable // 3)


#: the reference's dicodon maps (maps.py:544-546): every ordered pair of codons, numbered in that order
DICODONS = [a + b for a in CODONS for b in CODONS]
DICODON_ID = list(range(len(DICODONS)))


def dicodon_frame_length(nucleotides: int, crop_size: int | None = None) -> int:
    """Entries per frame with 6-grams: ``ngrams`` leaves n - 5 of them and every frame keeps each sixth one below the same
    stop the codon frames use (encode.py:272-284: ``tri[j : -3 + j + offset : ngram_width]``): ceil((n - 8 + off) / 6)."""
    n = int(nucleotides)
    off = _OFFSET_LUT[(n if crop_size is None else int(crop_size)) % 3]
    usable = n - 8 + off
    return 0 if usable <= 0 else -(-usable // 6)


def encode_window_literal(window: str, crop_size: int, codons=CODONS, codon_id=CODON_ID,
                          masking: bool = False, seq_onehot: bool = False, ngram_width: int | None = None):
    """One window -> (6, L) float32 ids (or (6, L, D) one-hot), encode.py:228-302.  ``ngram_width``: 3 for codons, 6 for
    dicodons (default: the length of the map's first entry - ``int(log4(len(codon)))`` at nnlib/inference.py:450)."""
    table = dict(zip(codons, codon_id))
    depth = max(codon_id) + 1
    if ngram_width is None:
        ngram_width = len(codons[0])
    offset = _OFFSET_LUT[crop_size % 3]                       # :232-236
    fwd = list(window)[:crop_size]                              # :234 bytes_split + crop
    rev = [_COMPLEMENT.get(b, "N") for b in fwd[::-1]]          # :256
    if masking is False:                                        # :259-261
        fwd = [b.upper() for b in fwd]
        rev = [b.upper() for b in rev]
    w = ngram_width
    tri_f = ["".join(fwd[p:p + w]) for p in range(len(fwd) - w + 1)]   # :272-277 ngrams
    tri_r = ["".join(rev[p:p + w]) for p in range(len(rev) - w + 1)]
    rows = []
    for tri in (tri_f, tri_r):                                  # :279-284
        for j in range(3):
            sl = tri[j:-3 + j + offset:w]
            rows.append([table.get(t, -1) for t in sl])
    seq = np.asarray(rows, dtype=np.int64)                      # :295 stack axis 0
    if seq_onehot:                                              # :297-300
        out = np.zeros(seq.shape + (depth,), np.float32)
        f, l = np.nonzero(seq >= 0)
        out[f, l, seq[f, l]] = 1.0
        return out
    return (seq + 1).astype(np.float32)                         # :302


def _codon_lut(codon_id) -> np.ndarray:
    """64-entry table indexed by 16*b0+4*b1+b2 over TCAG=0123 -> codon_id+1."""
    lut = np.zeros(64, np.int64)
    ref_index = {c: i for i, c in enumerate(CODONS)}
    for i0, a in enumerate(_B):
        for i1, b in enumerate(_B):
            for i2, c in enumerate(_B):
                lut[16 * i0 + 4 * i1 + i2] = codon_id[ref_index[a + b + c]] + 1
    return lut


def encode_windows(windows: list[bytes | str], crop_size: int, codon_id=CODON_ID,
                   masking: bool = False, pad_to: int | None = None) -> np.ndarray:
    """Vectorised ids-mode encoder: list of windows -> (W, 6, Lmax) uint8.

    Row order f1,f2,f3,r1,r2,r3; 0 = invalid codon or right padding (the value
    ``tf.data.padded_batch`` pads with, commands/predict.py:159-183).  Every
    window yields ``ceil((n-5+off)/3)`` codons per frame with ``off`` taken from
    ``crop_size`` (encode.py:232-236), ``n = min(len(window), crop_size)``.
    """
    lut = _codon_lut(codon_id)
    offset = _OFFSET_LUT[crop_size % 3]
    code = np.full(256, 4, np.int64)           # 4 = not an (upper-case) ACGT base
    for i, ch in enumerate(_B):
        code[ord(ch)] = i
        if not masking:
            code[ord(ch.lower())] = i          # upper-cased before lookup
    comp = np.array([2, 3, 0, 1, 4])           # T<->A, C<->G on TCAG codes
    lens = [max(0, -(-(min(len(w), crop_size) - 5 + offset) // 3)) for w in windows]
    lmax = max(lens) if lens else 0
    if pad_to is not None:
        lmax = max(lmax, pad_to)
    out = np.zeros((len(windows), 6, lmax), np.uint8)
    for wi, w in enumerate(windows):
        raw = np.frombuffer(w.encode() if isinstance(w, str) else w, np.uint8)[:crop_size]
        n, lw = raw.size, lens[wi]
        if lw <= 0:
            continue
        f = code[raw]
        if masking:
            # lower-case survives -> complement keeps case -> codon lookup misses
            pass
        r = comp[f[::-1]]
        for s, strand in enumerate((f, r)):
            for j in range(3):
                p = j + 3 * np.arange(lw)
                b0, b1, b2 = strand[p], strand[p + 1], strand[p + 2]
                bad = (b0 > 3) | (b1 > 3) | (b2 > 3)
                ids = lut[np.where(bad, 0, 16 * b0 + 4 * b1 + b2)]
                out[wi, 3 * s + j, :lw] = np.where(bad, 0, ids)
    return out


def encode_windows_dicodon(windows: list[bytes | str], crop_size: int, masking: bool = False,
                           pad_to: int | None = None) -> np.ndarray:
    """``codon: DICODON`` / ``codon_id: DICODON_ID`` ids-mode encoder, vectorised: (W, 6, Lmax) uint16, rows
    f1,f2,f3,r1,r2,r3, entry i of frame j = the 6-gram at base j + 6 i of the strand -> 64 * index(first codon) +
    index(second codon) + 1 in the reference's codon order (maps.py:544-546), 0 = a base outside ACGT (or lower case when
    ``masking``) / right padding."""
    idx = _codon_lut(CODON_ID) - 1             # 16 b0 + 4 b1 + b2 over TCAG -> position in CODONS
    offset = _OFFSET_LUT[crop_size % 3]
    code = np.full(256, 4, np.int64)
    for i, ch in enumerate(_B):
        code[ord(ch)] = i
        if not masking:
            code[ord(ch.lower())] = i
    comp = np.array([2, 3, 0, 1, 4])
    lens = [max(0, -(-(min(len(w), crop_size) - 8 + offset) // 6)) for w in windows]
    lmax = max(lens) if lens else 0
    if pad_to is not None:
        lmax = max(lmax, pad_to)
    out = np.zeros((len(windows), 6, lmax), np.uint16)
    for wi, w in enumerate(windows):
        raw = np.frombuffer(w.encode() if isinstance(w, str) else w, np.uint8)[:crop_size]
        lw = lens[wi]
        if lw <= 0:
            continue
        f = code[raw]
        r = comp[f[::-1]]
        for s, strand in enumerate((f, r)):
            for j in range(3):
                p = j + 6 * np.arange(lw)
                b = [strand[p + q] for q in range(6)]
                bad = np.zeros(lw, bool)
                for q in range(6):
                    bad |= b[q] > 3
                a_ = idx[np.where(bad, 0, 16 * b[0] + 4 * b[1] + b[2])]
                b_ = idx[np.where(bad, 0, 16 * b[3] + 4 * b[4] + b[5])]
                out[wi, 3 * s + j, :lw] = np.where(bad, 0, 64 * a_ + b_ + 1)
    return out


def window_counts(window: str | bytes) -> tuple[int, int, int, int]:
    """Upper-case-only G, C, A, T counts of a window (seqops/io.py:124-127)."""
    s = window.decode() if isinstance(window, bytes) else window
    return s.count("G"), s.count("C"), s.count("A"), s.count("T")


def amino_acid(codon: str) -> str:
    return _AA_OF[codon]
