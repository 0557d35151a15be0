"""Codon tables and the nt <-> codon-frame contract vs vectors dumped from the reference
(tests/golden/make_golden.py; seqops/maps.py, seqops/crop.py)."""
import json

from conftest import GOLDEN


def test_product_maps_match_reference():
    from jaeger_amd import maps
    g = json.loads((GOLDEN / "maps.json").read_text())
    for name in ("CODONS", "CODON_ID", "AA_ID", "MURPHY10_ID", "PC5_ID"):
        assert getattr(maps, name) == g[name], name


def test_oracle_tables_match_reference():
    from oracle import encoder
    g = json.loads((GOLDEN / "maps.json").read_text())
    assert encoder.CODONS == g["CODONS"] and encoder.CODON_ID == g["CODON_ID"]
    # amino-acid ids by first appearance reproduce AA_ID (stop = 0)
    order, ids = {"*": 0}, []
    for c in encoder.CODONS:
        a = encoder.amino_acid(c)
        order.setdefault(a, len(order))
        ids.append(order[a])
    assert ids == g["AA_ID"]


def test_frame_length_contract():
    """tests/unit/test_crop.py:6-49 / test_inference_crop.py:41-88 known answers + the dump."""
    from jaeger_amd.engine import frame_length
    from oracle.encoder import frame_length as oracle_fl
    g = json.loads((GOLDEN / "crop.json").read_text())
    for nt, n in g["tf_frame_length"].items():
        assert frame_length(int(nt)) == n == oracle_fl(int(nt)), nt
    assert frame_length(2000) == 665 and frame_length(1505) == 500 and frame_length(1500) == 498
    assert frame_length(500) == 165


def test_codon_lut_layout():
    import numpy as np
    from jaeger_amd.engine import codon_lut
    from jaeger_amd.maps import AA_ID, CODON_ID, CODONS
    lut = codon_lut(CODON_ID)
    alpha = "TCAG"
    for i, c in enumerate(CODONS):
        k = 16 * alpha.index(c[0]) + 4 * alpha.index(c[1]) + alpha.index(c[2])
        assert lut[k] == i + 1
    assert sorted(set(codon_lut(AA_ID)[:64].tolist())) == list(range(1, 22))
    assert lut.dtype == np.uint8 and lut.size == 65
