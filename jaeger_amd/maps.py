"""Codon / amino-acid id tables used by the 6-frame encoder.

The reference keeps these as literal lists (``seqops/maps.py:3-68`` CODONS,
``:137`` AA_ID, ``:408`` MURPHY10_ID, ``:475`` PC5_ID, ``:542`` CODON_ID).  Here
they are *generated* from the standard genetic code and the published reduced
alphabets; ``tests/test_maps.py`` pins them against golden vectors dumped from
the reference (``tests/golden/maps.json``).

Codon order of the reference table: index = 16*i(b1) + 4*i(b0) + i(b2) over the
alphabet ``TCAG`` (second base slowest, third base fastest).
"""

from __future__ import annotations

_ALPHA = "TCAG"

#: 64 codons in the reference order (seqops/maps.py:3-68).
CODONS: list[str] = [b0 + b1 + b2 for b1 in _ALPHA for b0 in _ALPHA for b2 in _ALPHA]

#: identity id map (seqops/maps.py:542)
CODON_ID: list[int] = list(range(64))

# Standard genetic code (NCBI table 1), '*' = stop.
_GENETIC_CODE = {
    "TTT": "F", "TTC": "F", "TTA": "L", "TTG": "L",
    "CTT": "L", "CTC": "L", "CTA": "L", "CTG": "L",
    "ATT": "I", "ATC": "I", "ATA": "I", "ATG": "M",
    "GTT": "V", "GTC": "V", "GTA": "V", "GTG": "V",
    "TCT": "S", "TCC": "S", "TCA": "S", "TCG": "S",
    "CCT": "P", "CCC": "P", "CCA": "P", "CCG": "P",
    "ACT": "T", "ACC": "T", "ACA": "T", "ACG": "T",
    "GCT": "A", "GCC": "A", "GCA": "A", "GCG": "A",
    "TAT": "Y", "TAC": "Y", "TAA": "*", "TAG": "*",
    "CAT": "H", "CAC": "H", "CAA": "Q", "CAG": "Q",
    "AAT": "N", "AAC": "N", "AAA": "K", "AAG": "K",
    "GAT": "D", "GAC": "D", "GAA": "E", "GAG": "E",
    "TGT": "C", "TGC": "C", "TGA": "*", "TGG": "W",
    "CGT": "R", "CGC": "R", "CGA": "R", "CGG": "R",
    "AGT": "S", "AGC": "S", "AGA": "R", "AGG": "R",
    "GGT": "G", "GGC": "G", "GGA": "G", "GGG": "G",
}

#: amino acid of each codon, reference order
AA: list[str] = [_GENETIC_CODE[c] for c in CODONS]


def _ids_by_first_appearance(letters: list[str]) -> list[int]:
    order: dict[str, int] = {"*": 0}
    out = []
    for a in letters:
        if a not in order:
            order[a] = len(order)
        out.append(order[a])
    return out


#: amino-acid ids 1..20 by first appearance in codon order, stop = 0
#: (seqops/maps.py:137)
AA_ID: list[int] = _ids_by_first_appearance(AA)



def _v1_ids(letters: list[str]) -> list[int]:
    order: dict[str, int] = {}
    return [order.setdefault(a, len(order) + 1) for a in letters]


#: legacy (v1) amino-acid ids 1..21 by first appearance, the stop codons share id 11; 0 is reserved
#: for unknown trimers (preprocess/v1/maps.py TRIMER_INT, used by preprocess/v1/convert.py:18-21)
V1_TRIMER_INT: list[int] = _v1_ids(AA)

# Murphy-10 reduced alphabet, ids by first appearance of the group in codon
# order (seqops/maps.py:408): {F,Y,W} {L,I,M,V} {S,T} {P} {A} {H} {Q,N,D,E}
# {K,R} {C} {G}.
_MURPHY10_GROUP = {
    "F": "FYW", "Y": "FYW", "W": "FYW",
    "L": "LIMV", "I": "LIMV", "M": "LIMV", "V": "LIMV",
    "S": "ST", "T": "ST", "P": "P", "A": "A", "H": "H",
    "Q": "QNDE", "N": "QNDE", "D": "QNDE", "E": "QNDE",
    "K": "KR", "R": "KR", "C": "C", "G": "G", "*": "*",
}
MURPHY10_ID: list[int] = _ids_by_first_appearance([_MURPHY10_GROUP[a] for a in AA])

# 5-class physico-chemical alphabet (seqops/maps.py:475): aromatic {F,Y,W,H},
# aliphatic {L,I,V}, {M,P,T,Q,N}, small {S,A,C,G}, charged {K,D,E,R}.
_PC5_GROUP = {
    "F": "FYWH", "Y": "FYWH", "W": "FYWH", "H": "FYWH",
    "L": "LIV", "I": "LIV", "V": "LIV",
    "M": "MPTQN", "P": "MPTQN", "T": "MPTQN", "Q": "MPTQN", "N": "MPTQN",
    "S": "SACG", "A": "SACG", "C": "SACG", "G": "SACG",
    "K": "KDER", "D": "KDER", "E": "KDER", "R": "KDER", "*": "*",
}
PC5_ID: list[int] = _ids_by_first_appearance([_PC5_GROUP[a] for a in AA])

#: the 4 096 ordered codon pairs of ``codon: DICODON`` and their ids (seqops/maps.py:544-546): 6-grams, numbered
#: 64 * index(first codon) + index(second codon)
DICODONS = [a + b for a in CODONS for b in CODONS]
DICODON_ID = list(range(len(DICODONS)))

#: names a ``*_project.yaml`` may use for ``string_processor.codon`` /
#: ``codon_id`` (nnlib/inference.py:424-432).
NAMED_MAPS: dict[str, list | None] = {
    "CODON": CODONS,
    "CODON_ID": CODON_ID,
    "AA_ID": AA_ID,
    "MURPHY10_ID": MURPHY10_ID,
    "PC5_ID": PC5_ID,
    "DICODON": DICODONS,
    "DICODON_ID": DICODON_ID,
}


def codon_index(codon: str) -> int:
    """Index of an upper-case ACGT codon in :data:`CODONS`, ``-1`` otherwise."""
    try:
        i0, i1, i2 = (_ALPHA.index(ch) for ch in codon)
    except ValueError:
        return -1
    return 16 * i1 + 4 * i0 + i2
