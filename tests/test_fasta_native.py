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


def _serial_parse(text: np.ndarray):
    """jg_fasta_count + jg_fasta_parse (the single-thread ingest) -> (names, bases, offsets)."""
    import ctypes as C
    from jaeger_amd import _lib as L
    lib = L.load()
    text = text.copy()
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    n_rec, name_bytes = C.c_int64(), C.c_int64()
    L.check(lib.jg_fasta_count(ptr(text), text.size, C.byref(n_rec), C.byref(name_bytes)))
    out = np.zeros(max(text.size, 1), np.uint8)
    offsets, name_off = np.zeros(n_rec.value + 1, np.int64), np.zeros(n_rec.value + 1, np.int64)
    names_buf = np.zeros(max(name_bytes.value, 1), np.uint8)
    got, nb = C.c_int64(), C.c_int64()
    L.check(lib.jg_fasta_parse(ptr(text), text.size, n_rec.value, ptr(out), ptr(offsets), ptr(names_buf), ptr(name_off),
                               C.byref(got), C.byref(nb)))
    raw, no = names_buf.tobytes(), name_off.tolist()
    return [raw[no[i]:no[i + 1]].decode() for i in range(got.value)], out[:nb.value], offsets[:got.value + 1]


@pytest.mark.parametrize("seed", range(6))
def test_parallel_ingest_equals_the_serial_parser(seed):
    """jg_fasta_scan / jg_fasta_fill with 1 - 13 slices vs jg_fasta_parse on random files: junk in front of the first
    header, CRLF, blank and whitespace-only lines, indented lines, spaces inside a line, empty records, a header as the
    last line, no trailing newline - slices that start inside any of these."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []
    if seed % 2:
        parts.append(b"junk line\nACGT\n\n")
    for r in range(int(rng.integers(1, 40))):
        eol = b"\r\n" if rng.random() < 0.3 else b"\n"
        parts.append(b">" + (b"rec%d" % r if rng.random() < 0.9 else b"") + (b" desc x" if rng.random() < 0.5 else b"") + eol)
        for _ in range(int(rng.integers(0, 12))):
            u = rng.random()
            if u < 0.1:
                parts.append(eol)
            elif u < 0.15:
                parts.append(b"  \t " + eol)
            else:
                line = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.integers(0, 9, int(rng.integers(1, 90)))].tobytes()
                if u < 0.25:
                    line = b" " + line[:len(line) // 2] + b" " + line[len(line) // 2:] + b"\t"
                parts.append(line + eol)
    data = b"".join(parts)
    if seed % 3 == 0:
        data = data.rstrip(b"\r\n")
    text = np.frombuffer(data, np.uint8)
    want = _serial_parse(text)
    for threads in (1, 2, 3, 5, 8, 13):
        names, bases, offsets, rec_off = frag._scan_fill(text, threads, want_rec_off=True)
        assert names == want[0], threads
        np.testing.assert_array_equal(bases, want[1])
        np.testing.assert_array_equal(offsets, want[2])
        assert all(data[o:o + 1] == b">" for o in rec_off[:-1].tolist()) and rec_off[-1] == len(data)
    names, bases, offsets, _ = frag._scan_fill(text, 0)          # automatic thread count
    assert names == want[0] and np.array_equal(bases, want[1]) and np.array_equal(offsets, want[2])


def test_names_container_stands_for_the_list_of_names(tmp_path):
    """``load_fasta`` hands the record names out as ``fragment.Names`` (one byte buffer + offsets; round 6): it compares, indexes
    and iterates like the list of strings it stands for, knows natively whether a name occurs twice (``jg_names_unique``) and
    whether io.py:109's normalisation would change anything, and gives the table writer spans instead of strings."""
    from jaeger_amd import fragment as frag
    names = ["r1", "", "r3", "r4,with,commas", "caf\u00e9", "r1x", "z" * 300]
    p = tmp_path / "n.fa"
    p.write_text("".join((f">{n} some description\nACGT\n" if n else ">\nACGT\n") for n in names))
    got = frag.load_fasta(str(p)).names
    assert isinstance(got, frag.Names) and got == names and list(got) == names and len(got) == 7
    assert got[3] == "r4,with,commas" and got[-1] == "z" * 300 and got[1:3] == ["", "r3"]
    with pytest.raises(IndexError):
        got[7]
    assert got.is_unique() and not got.plain()                       # (a comma: normalisation rewrites it)
    buf, b, e = got.spans([4, 0, 6])
    assert [buf[x:y].tobytes().decode() for x, y in zip(b.tolist(), e.tolist())] == ["caf\u00e9", "r1", "z" * 300]
    assert frag.normalise_headers(got).tolist() == [n.strip().replace(",", "___") for n in names]
    p.write_text("".join(f">{n}\nACGT\n" for n in ["a", "b", "c", "b"]))
    dup = frag.load_fasta(str(p)).names
    assert dup.plain() and not dup.is_unique() and frag.normalise_headers(dup).tolist() == ["a", "b", "c", "b"]
    # a few hundred thousand names: hashes on every core, still exact
    n = 300_000
    raw = "".join(f"r{i:07d}" for i in range(n)).encode()
    big = frag.Names(np.frombuffer(raw, np.uint8), np.arange(n + 1, dtype=np.int64) * 8)
    assert big.is_unique() and big.plain() and big[n - 1] == f"r{n - 1:07d}"
    raw2 = bytearray(raw)
    raw2[8 * 777:8 * 778] = raw[8 * 299_000:8 * 299_001]
    assert not frag.Names(np.frombuffer(bytes(raw2), np.uint8), np.arange(n + 1, dtype=np.int64) * 8).is_unique()


def test_names_container_property():
    """Property test (hypothesis): ``fragment.Names`` built from arbitrary name lists compares, indexes and slices like the list,
    finds duplicates exactly, and ``plain()`` is true exactly when io.py:109's normalisation is the identity on every name."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    alphabet = st.characters(blacklist_categories=("Cs",), blacklist_characters="\x00")

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.text(alphabet, max_size=10), max_size=30), st.booleans())
    def check(names, duplicate):
        if duplicate and names:
            names = names + [names[len(names) // 2]]
        raw = [n.encode() for n in names]
        off = np.concatenate(([0], np.cumsum([len(r) for r in raw]))).astype(np.int64)
        got = frag.Names(np.frombuffer(b"".join(raw) or b"\0", np.uint8)[:int(off[-1])] if raw else np.zeros(0, np.uint8), off)
        assert got == names and list(got) == names and len(got) == len(names)
        for i in range(len(names)):
            assert got[i] == names[i] and got[i - len(names)] == names[i]
        assert got[1:3] == names[1:3]
        assert got.is_unique() == (len(set(names)) == len(names))
        if got.plain():                         # (conservative the other way: a name it calls not plain may still be unchanged)
            assert all(n.strip().replace(",", "___") == n for n in names)
        assert frag.normalise_headers(got).tolist() == [n.strip().replace(",", "___") for n in names]
        buf, b, e = got.spans()
        assert [buf[x:y].tobytes().decode() for x, y in zip(b.tolist(), e.tolist())] == names

    check()
