"""Encoder oracle vs the reference's numba statement of the same encoder
(dataops/convert.py:_process_batch_numba, golden in tests/golden/encoder_ids.json)."""
import hashlib
import json

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.mark.parametrize("tag", ["1500_1500", "2000_1500"])
def test_oracle_ids_match_reference(tag):
    from oracle import encoder as E
    from oracle import fragmenter as F
    g = json.loads((GOLDEN / "encoder_ids.json").read_text())[tag]
    fsize = g["fsize"]
    stride = int(tag.split("_")[1])
    wins = []
    for frag in F.fragment_strings(F.read_fasta(str(GOLDEN / "test_contigs.fasta")), fsize, stride):
        wins.append(frag.split(",", 1)[0])
        if len(wins) == len(g["windows_sha1"]) - 1:
            break
    wins.append(g["edge_window"])
    assert [hashlib.sha1(w.encode()).hexdigest() for w in wins] == g["windows_sha1"]
    ids = E.encode_windows(wins, fsize)
    assert ids.shape == (len(wins), 6, g["n_frames"])
    assert hashlib.sha256(ids.tobytes()).hexdigest() == g["ids_sha256"]
    np.testing.assert_array_equal(ids[0], np.array(g["first_window_ids"], np.uint8))
    np.testing.assert_array_equal(ids[-1], np.array(g["edge_window_ids"], np.uint8))
    # the literal string-op restatement agrees with the vectorised one
    for w in (wins[0], wins[-1]):
        lit = E.encode_window_literal(w, fsize)
        np.testing.assert_array_equal(lit.astype(np.uint8), E.encode_windows([w], fsize)[0])


def test_lookup_defaults_and_complement():
    """tests/unit/test_seqops_encode.py:11-26: unknown codon -> -1 (id 0 after the +1), complement."""
    from oracle import encoder as E
    out = E.encode_window_literal("BBBACGTACGTAC", 13)
    assert out[0, 0] == 0                       # 'BBB' misses the table -> -1 -> 0
    assert (E.encode_window_literal("acgtacgtacgtac", 14, masking=True) == 0).all()
    up = E.encode_window_literal("acgtacgtacgtac", 14, masking=False)
    np.testing.assert_array_equal(up, E.encode_window_literal("ACGTACGTACGTAC", 14))
    oh = E.encode_window_literal("ACGTACGTACGTAC", 14, seq_onehot=True)
    assert oh.shape[-1] == 64 and (oh.sum(-1) == 1).all()


def test_short_window_uses_crop_offset():
    """off comes from crop_size even for a shorter string (encode.py:232-236)."""
    from oracle import encoder as E
    w = "ACGTTGCAAC" * 10          # 100 nt
    for crop in (1500, 1501, 1502):
        lit = E.encode_window_literal(w, crop)
        vec = E.encode_windows([w], crop)
        assert vec.shape[2] == lit.shape[1]
        np.testing.assert_array_equal(vec[0], lit.astype(np.uint8))


def test_dicodon_maps_are_the_references():
    """``codon: DICODON`` (nnlib/inference.py:430-451): the 4 096 codon pairs and their identity ids, pinned to the digest /
    ends dumped from the reference's seqops/maps.py:544-546 (tests/golden/make_golden.py)."""
    import hashlib
    import json

    from conftest import GOLDEN
    from jaeger_amd import maps
    from oracle import encoder as oenc
    g = json.loads((GOLDEN / "maps.json").read_text())
    for table in (maps.DICODONS, oenc.DICODONS):
        assert len(table) == g["DICODONS_LEN"] == 4096
        assert table[:5] == g["DICODONS_HEAD"] and table[-3:] == g["DICODONS_TAIL"]
        assert hashlib.sha256(",".join(table).encode()).hexdigest() == g["DICODONS_SHA256"]
    assert g["DICODON_ID_IS_IDENTITY"] and maps.DICODON_ID == oenc.DICODON_ID == list(range(4096))


def test_dicodon_encoder_literal_equals_vectorised():
    """The step-by-step restatement of encode.py:228-302 at ``ngram_width = 6`` (all 6-grams, frames ``tri[j : -3 + j + off : 6]``,
    hash lookup with default -1) against the vectorised form the parity tests use: every crop residue, short and cropped
    windows, N / lower case with and without ``masking``."""
    from jaeger_amd.engine import dicodon_frame_length
    from oracle import encoder as oenc
    rng = np.random.default_rng(6)
    for n, crop in ((60, 60), (61, 61), (62, 62), (100, 150), (151, 150), (7, 60), (8, 60), (14, 62), (500, 500), (1500, 1500)):
        w = "".join(rng.choice(list("ACGTNacgt"), p=[.22, .22, .22, .22, .04, .02, .02, .02, .02], size=n))
        for masking in (False, True):
            lit = oenc.encode_window_literal(w, crop, codons=oenc.DICODONS, codon_id=oenc.DICODON_ID, masking=masking)
            vec = oenc.encode_windows_dicodon([w], crop, masking=masking)[0]
            assert lit.shape == (6, oenc.dicodon_frame_length(min(n, crop), crop))
            np.testing.assert_array_equal(vec, lit.astype(np.uint16))
    assert dicodon_frame_length(1500) == oenc.dicodon_frame_length(1500) == 249
    assert dicodon_frame_length(2000) == oenc.dicodon_frame_length(2000) == 332
    assert oenc.encode_windows_dicodon(["ACGTTT", "ACGTTTA" * 3], 60).shape == (2, 6, 2)     # (21 - 8 - 2) / 6 -> 2; a 6-base window yields no entry
