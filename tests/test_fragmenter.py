"""Fragmenter: oracle vs the reference's golden fragment fields, the reference's known answers
(tests/unit/test_seqops_io.py:6-165), and the product's vectorised window table vs the oracle."""
import hashlib
import json

import numpy as np
import pytest

from conftest import GOLDEN

SETTINGS = {
    "1500_1500": dict(fragsize=1500, stride=1500),
    "2000_1500": dict(fragsize=2000, stride=1500),
    "2000_2000_dyn": dict(fragsize=2000, stride=2000, dynamic_stride=True, dynamic_stride_threshold=10.0),
    "4000_4000_min1000": dict(fragsize=40000, stride=40000, min_len=9000),
}


@pytest.mark.parametrize("tag", list(SETTINGS))
def test_oracle_matches_reference_fragments(tag):
    from oracle import fragmenter as F
    want = json.loads((GOLDEN / f"fragments_{tag}.json").read_text())
    got = list(F.fragment_strings(F.read_fasta(str(GOLDEN / "test_contigs.fasta")), **SETTINGS[tag]))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        f = g.split(",")
        assert hashlib.sha1(f[0].encode()).hexdigest() == w["sha1"] and len(f[0]) == w["len"]
        assert f[1:] == w["fields"]


def test_window_indices_known_answers():
    from jaeger_amd.fragment import window_indices as prod
    from oracle.fragmenter import window_indices as orc
    for fn in (prod, orc):
        assert fn(3400, 2000, 2000, False, 2.0) == [0]
        assert fn(3400, 2000, 2000, True, 2.0) == [0, 1400]
        assert fn(4000, 2000, 2000, True, 2.0) == [0, 2000]
        assert fn(6000, 2000, 2000, True, 2.0) == [0, 2000, 4000]
        assert fn(3999, 2000, 2000, True, 2.0) == [0, 1999]


def test_oracle_short_contig_and_header_rules(tmp_path):
    from oracle import fragmenter as F
    fa = tmp_path / "mixed.fa"
    fa.write_text(f">long len=3000\n{'A' * 3000}\n>sh,ort len=1500\n{'C' * 1500}\n>tiny len=900\n{'G' * 900}\n")
    recs = list(F.read_fasta(str(fa)))
    assert [r[0] for r in recs] == ["long", "sh,ort", "tiny"]          # name up to first whitespace
    long_only = list(F.fragment_strings(recs, 2000, 2000, min_len=2000))
    assert len(long_only) == 1 and long_only[0].split(",")[1] == "long"
    short = list(F.fragment_strings(recs, 2000, 2000, min_len=1000, max_len=1999))
    assert len(short) == 1
    p = short[0].split(",")
    assert p[1] == "sh___ort" and p[2] == "0" and p[3] == "1" and p[5] == "1500"      # ',' -> '___'
    assert p[-1] == " 0.000" or p[-1].startswith("-")                               # '{: .3f}'


@pytest.mark.parametrize("tag", list(SETTINGS))
def test_product_window_table_matches_reference(tag):
    from jaeger_amd import fragment as P
    want = json.loads((GOLDEN / f"fragments_{tag}.json").read_text())
    recs = list(P.read_fasta(str(GOLDEN / "test_contigs.fasta")))
    seqs = [s for _, s in recs]
    kw = dict(SETTINGS[tag])
    fsize = kw.pop("fragsize")
    table = P.build_window_table([len(s) for s in seqs], fsize, **kw)
    assert len(table) == len(want)
    bases, offsets = P.concat_records(seqs)
    counts = np.zeros((len(table), 4), np.int32)
    for i in range(len(table)):
        w = bases[offsets[table.contig[i]] + table.start[i]:][:table.length[i]].tobytes().upper()
        assert hashlib.sha1(w).hexdigest() == want[i]["sha1"]
        counts[i] = [w.count(b"G"), w.count(b"C"), w.count(b"A"), w.count(b"T")]
    meta = P.window_metadata(table, [n for n, _ in recs], counts)
    for i, w in enumerate(want):
        f = w["fields"]
        assert meta["meta_0"][i] == f[0]
        assert [int(meta[f"meta_{k}"][i]) for k in (1, 2, 3, 4, 5, 6, 7, 8)] == [int(x) for x in f[1:9]]
        assert f"{meta['meta_9'][i]: .3f}" == f[9]


def test_product_window_table_random_vs_oracle():
    from jaeger_amd import fragment as P
    from oracle import fragmenter as F
    rng = np.random.default_rng(5)
    lengths = rng.integers(10, 9000, 300)
    for kw in (dict(stride=700), dict(stride=1000, dynamic_stride=True, dynamic_stride_threshold=3.0),
               dict(stride=None, min_len=400), dict(stride=1000, min_len=300, max_len=999)):
        table = P.build_window_table(lengths, 1000, **kw)
        recs = [(f"c{i}", "A" * int(n)) for i, n in enumerate(lengths)]
        want = [s.split(",") for s in F.fragment_strings(recs, 1000, **kw)]
        assert len(table) == len(want)
        assert [int(x) for x in table.start] == [int(w[2]) for w in want]
        assert [int(x) for x in table.is_last] == [int(w[3]) for w in want]
        assert [int(x) for x in table.ordinal] == [int(w[4]) for w in want]
        assert [int(x) for x in table.seqlen] == [int(w[5]) for w in want]
        assert [f"c{int(c)}" for c in table.contig] == [w[1] for w in want]


def test_empty_and_short_fasta():
    from jaeger_amd import fragment as P
    assert list(P.read_fasta(str(GOLDEN / "test_empty.fasta"))) == []
    recs = list(P.read_fasta(str(GOLDEN / "test_short.fasta")))
    assert len(recs) == 1 and len(recs[0][1]) == 137
    assert len(P.build_window_table([137], 2000, 1500)) == 0
