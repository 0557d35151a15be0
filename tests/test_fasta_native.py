"""Native FASTA ingest (jg_fasta_parse) against the pure-Python reader (pyfastx record rules)."""
import gzip

import numpy as np
import pytest
from conftest import GOLDEN

from jaeger_amd import fragment as frag


def _same(path):
    ref = list(frag.read_fasta(str(path)))
    fa = frag.load_fasta(str(path))
    assert fa.names == [n for n, _ in ref]
    assert fa.lengths.tolist() == [len(s) for _, s in ref]
    assert fa.bases.tobytes() == b"".join(s for _, s in ref)
    for i in range(len(fa)):
        assert fa.sequence(i) == ref[i][1]
    return fa


@pytest.mark.parametrize("name", ["test_contigs.fasta", "test_short.fasta", "test_empty.fasta"])
def test_bundled_files(name):
    fa = _same(GOLDEN / name)
    if name == "test_contigs.fasta":
        assert len(fa) == 9 and fa.lengths[0] == 44776 and fa.names[0] == "NODE_94_length_44776_cov_27.159388"


def test_edge_cases(tmp_path):
    text = (b"junk before the first header\nACGT\n"
            b">r1 description with spaces\tand tabs\r\nACGTN\r\n  acgt  \r\n\r\n"
            b">\n"                                # empty name, empty sequence
            b">r3\tx\nAC GT\n\n\n"                 # inner blank kept (strip is per line end only)
            b">r4,with,commas  trailing\nNNNN\nA")     # no final newline
    p = tmp_path / "edge.fa"
    p.write_bytes(text)
    fa = _same(p)
    assert fa.names == ["r1", "", "r3", "r4,with,commas"]
    assert fa.sequence(0) == b"ACGTNacgt" and fa.sequence(2) == b"AC GT" and fa.sequence(3) == b"NNNNA"
    gz = tmp_path / "edge.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(text)
    _same(gz)


def test_large_random_roundtrip(tmp_path):
    rng = np.random.Generator(np.random.PCG64(9))
    p = tmp_path / "big.fa"
    seqs = []
    with open(p, "wb") as fh:
        for i in range(300):
            n = int(rng.integers(1, 30000))
            s = np.frombuffer(b"ACGTNacgtn", np.uint8)[rng.integers(0, 10, n)].tobytes()
            seqs.append(s)
            fh.write(b">c%d len=%d\n" % (i, n))
            w = int(rng.integers(20, 200))
            for j in range(0, n, w):
                fh.write(s[j:j + w] + (b"\r\n" if i % 7 == 0 else b"\n"))
    fa = _same(p)
    assert [fa.sequence(i) for i in range(300)] == seqs


def test_selected_records_with_very_long_header_tokens(tmp_path):
    """load_fasta_records (the per-rank ingest under torchrun) sizes its name buffer from jg_fasta_count over the
    selected bytes: header tokens far beyond 4 KB must neither overflow it nor be truncated."""
    rng = np.random.Generator(np.random.PCG64(10))
    names = ["short", "x" * 5000, "y" * 70000 + "|tail", "z"]
    seqs = [np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].tobytes() for n in (50, 7, 300, 1)]
    p = tmp_path / "long.fa"
    with open(p, "wb") as fh:
        for nm, s in zip(names, seqs):
            fh.write(b">" + nm.encode() + b" some description\n" + s + b"\n")
    idx = frag.index_fasta(str(p))
    assert idx.names == names
    for sel in ([1, 2], [0, 2, 3], [2], [0, 1, 2, 3]):
        fb = frag.load_fasta_records(str(p), idx.rec_off, sel)
        assert fb.names == [names[i] for i in sel]
        assert [fb.sequence(i) for i in range(len(sel))] == [seqs[i] for i in sel]
