"""Terminal-repeat scan kernel (jg_terminal_repeats) vs the CPU Smith-Waterman restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand(rng, n):
    return "".join(rng.choice(list("ACGT"), n))


def _cases():
    rng = np.random.Generator(np.random.PCG64(42))
    from oracle.termini import reverse_complement
    out = {}
    body = _rand(rng, 3000)
    rep = _rand(rng, 60)
    out["dtr_exact"] = rep + body + rep
    out["itr_exact"] = rep + body + reverse_complement(rep)
    out["none"] = _rand(rng, 2500)
    long_rep = _rand(rng, 300)
    out["ltr_dtr"] = long_rep + body + long_rep
    mm = list(_rand(rng, 140))
    mm2 = mm.copy()
    mm2[70] = "A" if mm[70] != "A" else "C"                    # one mismatch with > 50 matches on both sides
    out["dtr_mismatch"] = "".join(mm) + body + "".join(mm2)
    g = _rand(rng, 160)
    out["dtr_gap_in_rear"] = g + body + g[:80] + g[83:]          # 3 bases missing at the rear end: gap in the ref row
    out["dtr_gap_in_front"] = g[:80] + g[83:] + body + g          # gap in the query row
    out["inner_offset"] = _rand(rng, 37) + rep + body + _rand(rng, 11) + rep + _rand(rng, 90)
    out["with_n"] = rep[:30] + "N" + rep[31:] + body + rep
    out["lower_case"] = rep.lower() + body + rep
    out["short_1200"] = rep + _rand(rng, 1080) + rep              # scan 400 overlapping regions
    out["big_60kb"] = long_rep + _rand(rng, 60000) + long_rep      # scan = 2412
    return out


@pytest.mark.parametrize("exact", [0, 1])
def test_kernel_matches_smith_waterman(exact):
    """exact = 0: the packed score-only pass + the exact kernel for alignments scoring above 100 (the default);
    exact = 1: every alignment through the exact kernel (JG_OPT_TERMINI_EXACT)."""
    from jaeger_amd import fragment as frag
    from jaeger_amd import _lib as L
    from jaeger_amd.engine import HipDevice
    from jaeger_amd.termini import scan_for_terminal_repeats
    from oracle import termini as ot
    import ctypes as C
    cases = _cases()
    names = list(cases)
    bases, offsets = frag.concat_records([cases[k].encode() for k in names])
    fa = frag.FastaBatch(names, bases, offsets)
    dev = HipDevice(0)
    L.check(dev.lib.jg_engine_set_option(dev.handle, L.JG_OPT_TERMINI_EXACT, exact))
    res = np.full((len(names), 10), -1, np.int32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L.check(dev.lib.jg_terminal_repeats(dev.handle, ptr(bases), bases.size, L.JG_PTR_HOST, ptr(offsets), len(names),
                                        1000, ptr(res)))
    df = scan_for_terminal_repeats(dev, fa, 1000)
    dev.close()
    for k, name in enumerate(names):
        seq = cases[name]
        scan = ot.scan_length(len(seq))
        if scan > 1300:                      # keep the pure-Python oracle fast: check the big case by its plant
            assert res[k, 0] == 600 and res[k, 1] == 300 and res[k, 2] == 0, res[k]
            continue
        front, rear = seq[:scan], seq[-scan:]
        for col, ref in ((0, rear), (5, ot.reverse_complement(rear))):
            want = ot.smith_waterman(front, ref)
            got = dict(score=int(res[k, col]), length=int(res[k, col + 1]), fgaps=int(res[k, col + 2]),
                       end_query=int(res[k, col + 3]), end_ref=int(res[k, col + 4]))
            assert got["score"] == want["score"], (name, col, got, want)
            assert (got["length"], got["fgaps"]) == (want["length"], want["fgaps"]), (name, col, got, want)
            if want["score"] > 0:
                assert (got["end_query"], got["end_ref"]) == (want["end_query"], want["end_ref"]), (name, col)
        kind, length = ot.scan_record(seq)
        row = df[df.contig_id == name].iloc[0]
        assert (row.terminal_repeats if isinstance(row.terminal_repeats, str) else None) == kind, (name, row)
        assert (np.isnan(row.repeat_length) and length is None) or row.repeat_length == length, (name, row)
    d = dict(zip(df.contig_id, df.terminal_repeats))
    assert d["dtr_exact"] == "DTR" and d["itr_exact"] == "ITR" and d["ltr_dtr"] == "LTR_DTR" and d["big_60kb"] == "LTR_DTR"
    assert d["dtr_mismatch"] == "DTR" and df[df.contig_id == "dtr_mismatch"].repeat_length.iloc[0] == 140
    assert df[df.contig_id == "dtr_gap_in_rear"].repeat_length.iloc[0] == 160


def test_short_records_are_skipped():
    from jaeger_amd import fragment as frag
    from jaeger_amd.engine import HipDevice
    from jaeger_amd.termini import scan_for_terminal_repeats
    rng = np.random.Generator(np.random.PCG64(1))
    seqs = [_rand(rng, 900).encode(), _rand(rng, 2100).encode()]
    bases, offsets = frag.concat_records(seqs)
    dev = HipDevice(0)
    df = scan_for_terminal_repeats(dev, frag.FastaBatch(["a,b", "c"], bases, offsets), 2000)
    dev.close()
    assert list(df.contig_id) == ["c"]


def test_fast_pass_equals_exact_kernel_on_random_records():
    """Both passes on 3 000 records (scan lengths 400 - 4 000, every strip-height class of the packed kernel, planted
    repeats of 8 - 45 bases that stay below the score-100 line, N runs, lower case, a few real repeats): the same
    (n, 10) table, end cells included; and the 500-bp record shape (scan 400 on a 500-base record: overlapping ends)."""
    from jaeger_amd import fragment as frag
    from jaeger_amd import _lib as L
    from jaeger_amd.termini import terminal_repeat_table
    from jaeger_amd.engine import HipDevice
    from oracle.termini import reverse_complement
    rng = np.random.Generator(np.random.PCG64(7))
    seqs = []
    for r in range(3000):
        n = int(np.exp(rng.uniform(np.log(500), np.log(160_000))))
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        u = rng.random()
        if u < 0.5 and n > 1200:                       # a short repeat between the two scanned ends (direct or inverted)
            k = int(rng.integers(8, 46))
            a, b = int(rng.integers(0, 350 - k)), n - int(rng.integers(k, 350))
            piece = bytes(s[a:a + k])
            s[b:b + k] = np.frombuffer(piece if u < 0.25 else reverse_complement(piece.decode()).encode(), np.uint8)
        elif u < 0.53 and n > 3000:                    # a real one
            k = int(rng.integers(60, 300))
            s[n - k:] = s[:k]
        if rng.random() < 0.2:
            p = int(rng.integers(0, n - 20))
            s[p:p + int(rng.integers(1, 20))] = ord("N")
        if rng.random() < 0.2:
            p = int(rng.integers(0, n - 50))
            s[p:p + 50] |= 0x20
        seqs.append(bytes(s))
    bases, offsets = frag.concat_records(seqs)
    fa = frag.FastaBatch([f"r{i}" for i in range(len(seqs))], bases, offsets)
    dev = HipDevice(0)
    fast = terminal_repeat_table(dev, fa, 500)
    L.check(dev.lib.jg_engine_set_option(dev.handle, L.JG_OPT_TERMINI_EXACT, 1))
    exact = terminal_repeat_table(dev, fa, 500)
    dev.close()
    assert (exact[:, 0] > 100).sum() > 30 and ((exact[:, 0] > 24) & (exact[:, 0] <= 100)).sum() > 300
    bad = np.nonzero((fast != exact).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], fast[bad[:5]], exact[bad[:5]])


def _one_run_line_records(seed: int = 2025, n_records: int = 1500):
    """Records built to sit on the line between 'one exact run' and everything else (see the test below)."""
    from oracle.termini import reverse_complement
    rng = np.random.Generator(np.random.PCG64(seed))
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rand(n):
        return acgt[rng.integers(0, 4, n)].copy()

    def other(b):
        return acgt[(int(np.nonzero(acgt == b)[0][0]) + int(rng.integers(1, 4))) % 4]

    seqs = []
    for r in range(n_records):
        kind = r % 10
        if kind < 3:                                   # overlapping ends, 500 - 790 bases
            n = int(rng.integers(500, 790))
            s = rand(n)
            if kind == 1:
                p = int(rng.integers(0, n - 30))
                s[p:p + int(rng.integers(1, 30))] = ord("N")
            if kind == 2:
                p, k = int(rng.integers(0, n - 120)), int(rng.integers(10, 120))
                s[p:p + k] = np.tile(rand(int(rng.integers(1, 4))), k)[:k]
                s[int(rng.integers(0, n - 50)):][:50] |= 0x20
        else:
            n = int(rng.integers(1500, 9000))
            s = rand(n)
            k = int(rng.integers(51, 200))
            a, b = int(rng.integers(0, 400 - k)), n - int(rng.integers(k, 400))
            piece = s[a:a + k].copy()
            inverted = kind == 9
            plant = (lambda x: np.frombuffer(reverse_complement(bytes(x).decode()).encode(), np.uint8)) if inverted \
                else (lambda x: x)
            s[b:b + k] = plant(piece)
            if kind == 4:                              # a second repeat elsewhere between the two ends
                k2 = int(rng.integers(51, 120))
                a2, b2 = int(rng.integers(0, 400 - k2)), n - int(rng.integers(k2, 400))
                s[b2:b2 + k2] = s[a2:a2 + k2]
            if kind in (5, 6) and a + k + 60 < 400 and b + k + 60 < n:        # k matches, one mismatch, m matches
                m = int(rng.choice([49, 50, 51]))
                s[a + k] = other(s[a + k - 1]) if s[a + k] == s[b + k] else s[a + k]
                s[b + k] = other(s[a + k])
                s[b + k + 1:b + k + 1 + m] = s[a + k + 1:a + k + 1 + m]
                if b + k + 1 + m < n and a + k + 1 + m < n:
                    s[b + k + 1 + m] = other(s[a + k + 1 + m])
            if kind in (7, 8) and a > 60 and b > 460:                          # m matches, a mismatch or a gap, k matches
                m = int(rng.choice([49, 50, 51]))
                gap = int(rng.integers(0, 3)) if kind == 8 else 0
                s[b - 1] = other(s[a - 1])
                s[b - 1 - gap - m:b - 1 - gap] = s[a - 1 - m:a - 1]
                s[b - 2 - gap - m] = other(s[a - 2 - m])
        if r % 97 == 0:
            s[:] = ord("A")                            # every diagonal of the matrix is one long run
        if r % 89 == 0:
            s[:] = np.tile(np.frombuffer(b"AC", np.uint8), n)[:n]
        seqs.append(bytes(s))
    return seqs


def test_single_run_shortcut_equals_exact_kernel():
    """Alignments scoring above 100 that are one exact run are settled without the length / gap carrying kernel
    (termini_single_kernel); everything else still goes through it.  Shapes built to sit on the line between the two:
    overlapping ends (records shorter than twice their scan length - with N runs, lower case and low-complexity stretches
    inside the overlap), one planted repeat, two planted repeats, a repeat followed / preceded by exactly 50 (or 49, 51)
    matching bases behind one mismatch or a short gap, direct and inverted, homopolymer and dinucleotide ends."""
    from jaeger_amd import fragment as frag
    from jaeger_amd import _lib as L
    from jaeger_amd.termini import terminal_repeat_table
    from jaeger_amd.engine import HipDevice
    from oracle.termini import reverse_complement
    seqs = _one_run_line_records()
    bases, offsets = frag.concat_records(seqs)
    fa = frag.FastaBatch([f"r{i}" for i in range(len(seqs))], bases, offsets)
    dev = HipDevice(0)
    fast = terminal_repeat_table(dev, fa, 500)
    L.check(dev.lib.jg_engine_set_option(dev.handle, L.JG_OPT_TERMINI_EXACT, 1))
    exact = terminal_repeat_table(dev, fa, 500)
    dev.close()
    assert (exact[:, 0] > 100).sum() > 1200 and (exact[:, 2] > 0).sum() > 3         # real repeats, some with gaps
    bad = np.nonzero((fast != exact).any(axis=1))[0]
    assert bad.size == 0, (bad[:5], fast[bad[:5]], exact[bad[:5]])


def test_report_min_leaves_the_decision_rule_as_it_is():
    """JG_OPT_TERMINI_REPORT_MIN = 13 (what run_core sets): records whose ends share no 13 matching bases skip the dynamic
    programme, one-run repeats are settled by the 32-mer check alone - the decision rule's columns (kind, length, score of
    every record: termini.RepeatColumns) equal the exact scan's, and every alignment of at least 13 columns comes back with
    the same five numbers.  Random records of every strip-height class, planted repeats of 8 - 45 bases (both sides of the
    13-column line), real repeats, overlapping ends, N runs, the adversarial shapes of the one-run test."""
    from jaeger_amd import fragment as frag
    from jaeger_amd.termini import REPORT_MIN_COLUMNS, RepeatColumns, terminal_repeat_table
    from jaeger_amd.engine import HipDevice
    from oracle.termini import reverse_complement
    rng = np.random.Generator(np.random.PCG64(99))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    seqs = []
    for r in range(2500):
        n = int(np.exp(rng.uniform(np.log(500), np.log(160_000))))
        s = acgt[rng.integers(0, 4, n)].copy()
        u = rng.random()
        if u < 0.5 and n > 1200:                       # a short repeat between the two scanned ends (direct or inverted)
            k = int(rng.integers(8, 46))
            a, b = int(rng.integers(0, 350 - k)), n - int(rng.integers(k, 350))
            piece = bytes(s[a:a + k])
            s[b:b + k] = np.frombuffer(piece if u < 0.25 else reverse_complement(piece.decode()).encode(), np.uint8)
        elif u < 0.56 and n > 3000:                    # a real one, now and then with a mismatch inside
            k = int(rng.integers(60, 300))
            s[n - k:] = s[:k]
            if rng.random() < 0.4:
                s[n - k + k // 2] = acgt[(int(np.nonzero(acgt == s[k // 2])[0][0]) + 1) % 4]
        if rng.random() < 0.2:
            p = int(rng.integers(0, n - 20))
            s[p:p + int(rng.integers(1, 20))] = ord("N")
        if rng.random() < 0.2:
            p = int(rng.integers(0, n - 50))
            s[p:p + 50] |= 0x20
        if r % 211 == 0:
            s[:] = ord("A")
        seqs.append(bytes(s))
    bases, offsets = frag.concat_records(seqs)
    names = [f"r{i}" for i in range(len(seqs))]
    fa = frag.FastaBatch(names, bases, offsets)
    dev = HipDevice(0)
    exact = terminal_repeat_table(dev, fa, 500)
    quick = terminal_repeat_table(dev, fa, 500, report_min=REPORT_MIN_COLUMNS)
    again = terminal_repeat_table(dev, fa, 500)            # (the option is set per call: back to the exact table)
    dev.close()
    np.testing.assert_array_equal(again, exact)
    skipped = int(((quick[:, 1] == 0) & (quick[:, 6] == 0)).sum())
    assert skipped > 1000, skipped                        # most records never reach the dynamic programme
    for col in (0, 5):                                    # every alignment of >= 13 columns: the same five numbers
        long_enough = exact[:, col + 1] >= REPORT_MIN_COLUMNS
        assert long_enough.sum() > 300
        np.testing.assert_array_equal(quick[long_enough, col:col + 5], exact[long_enough, col:col + 5])
        assert (quick[~long_enough, col + 1] < REPORT_MIN_COLUMNS).all()
    a, b = RepeatColumns(exact, names, fa.lengths), RepeatColumns(quick, names, fa.lengths)
    np.testing.assert_array_equal(a.keep, b.keep)
    assert a.n_found == b.n_found > 300
    assert (a.kind == b.kind).all()
    np.testing.assert_array_equal(a.length, b.length)
    np.testing.assert_array_equal(a.score, b.score)


def test_report_min_on_the_one_run_line_equals_the_exact_kernel():
    """with JG_OPT_TERMINI_REPORT_MIN = 13 (run_core's setting) an alignment settled by the one-run check is
    final without the packed pass's score beside it - so the seed path is held to the exact kernel (JG_OPT_TERMINI_EXACT)
    on the shapes that sit on the line: the one-run test's records, plus runs of exactly 50 / 51 / 52 bases (the 'more than
    50' bound of the argument), two runs of 50 and 51 in one record, a run that touches the edge of the scanned end, N
    inside a run, a run of 12 / 13 / 14 bases (the report bound itself), mismatches where the sampled 32-mers sit.  Every
    alignment of >= 13 columns: the same five numbers; the decision rule's columns: equal."""
    from jaeger_amd import _lib as L
    from jaeger_amd import fragment as frag
    from jaeger_amd.engine import HipDevice
    from jaeger_amd.termini import REPORT_MIN_COLUMNS, RepeatColumns, terminal_repeat_table
    from oracle.termini import reverse_complement
    rng = np.random.Generator(np.random.PCG64(606))
    acgt = np.frombuffer(b"ACGT", np.uint8)

    def rand(n):
        return acgt[rng.integers(0, 4, n)].copy()

    def plant(s, a, b, k, inverted=False):
        piece = bytes(s[a:a + k])
        s[b:b + k] = np.frombuffer(reverse_complement(piece.decode()).encode() if inverted else piece, np.uint8)

    seqs = _one_run_line_records(seed=77, n_records=600)
    for r in range(900):
        n = int(rng.integers(1500, 12000))
        s = rand(n)
        kind = r % 9
        inv = bool(rng.random() < 0.3)
        if kind == 0:                                   # one run of exactly 50 / 51 / 52 bases
            k = int(rng.choice([50, 51, 52]))
            plant(s, int(rng.integers(0, 300)), n - int(rng.integers(k, 350)), k, inv)
        elif kind == 1:                                 # two runs, 50 and 51 bases, apart
            plant(s, 10, n - 390, 50, inv)
            plant(s, 200, n - 200, 51, inv)
        elif kind == 2:                                 # a run that ends with the record (the scan's far edge) ...
            k = int(rng.integers(51, 180))
            plant(s, int(rng.integers(0, 200)), n - k, k, inv)
        elif kind == 3:                                 # ... or starts with it
            k = int(rng.integers(51, 180))
            plant(s, 0, n - int(rng.integers(k, 380)), k, inv)
        elif kind == 4:                                 # N inside the planted run (one copy / both copies)
            k = int(rng.integers(80, 200))
            a, b = int(rng.integers(0, 150)), n - int(rng.integers(k, 380))
            plant(s, a, b, k, inv)
            q = int(rng.integers(5, k - 5))
            s[b + q:b + q + int(rng.integers(1, 4))] = ord("N")
            if rng.random() < 0.5:
                s[a + q] = ord("N")
        elif kind == 5:                                 # the report bound: 12, 13, 14 shared bases
            k = int(rng.choice([12, 13, 14]))
            plant(s, int(rng.integers(0, 300)), n - int(rng.integers(k, 350)), k, inv)
        elif kind == 6:                                 # a long run with mismatches 51 bases apart (runs of exactly 50 between)
            k = int(rng.integers(100, 300))
            a, b = int(rng.integers(0, 80)), n - int(rng.integers(k, 390))
            plant(s, a, b, k, inv)
            for q in (51, 102):
                if q < k - 1:
                    s[b + q] = acgt[(int(np.nonzero(acgt == s[b + q])[0][0]) + 1) % 4]
        elif kind == 7:                                 # records around twice the 4 000-base scan length, run at the scan's edge
            n = int(rng.choice([7999, 8000, 8001, 8050]))
            s = rand(n)
            k = int(rng.integers(51, 120))
            plant(s, 4000 - k, n - 4000, k, inv)
        else:                                           # lower case over half the run
            k = int(rng.integers(51, 150))
            a, b = int(rng.integers(0, 200)), n - int(rng.integers(k, 380))
            plant(s, a, b, k, inv)
            s[b:b + k // 2] |= 0x20
        seqs.append(bytes(s))
    bases, offsets = frag.concat_records(seqs)
    names = [f"r{i}" for i in range(len(seqs))]
    fa = frag.FastaBatch(names, bases, offsets)
    dev = HipDevice(0)
    quick = terminal_repeat_table(dev, fa, 500, report_min=REPORT_MIN_COLUMNS)
    L.check(dev.lib.jg_engine_set_option(dev.handle, L.JG_OPT_TERMINI_EXACT, 1))
    exact = terminal_repeat_table(dev, fa, 500)            # every alignment through the length / gap carrying kernel
    dev.close()
    assert (exact[:, 0] > 100).sum() + (exact[:, 5] > 100).sum() > 800
    for col in (0, 5):
        long_enough = exact[:, col + 1] >= REPORT_MIN_COLUMNS
        bad = np.nonzero((quick[long_enough, col:col + 5] != exact[long_enough, col:col + 5]).any(axis=1))[0]
        assert bad.size == 0, (col, bad[:5], quick[long_enough][bad[:5]], exact[long_enough][bad[:5]])
        assert (quick[~long_enough, col + 1] < REPORT_MIN_COLUMNS).all()
    a, b = RepeatColumns(exact, names, fa.lengths), RepeatColumns(quick, names, fa.lengths)
    assert (a.kind == b.kind).all()
    np.testing.assert_array_equal(a.length, b.length)
    np.testing.assert_array_equal(a.score, b.score)


def test_high_priority_stream_gives_the_same_table():
    """JG_OPT_STREAM_PRIORITY (run_core sets it on the repeat scan's side engine so that its kernels are not queued behind
    the network's): the engine's stream is re-created at the device's highest priority - same table; 0 switches back; any
    other value is refused."""
    from jaeger_amd import _lib as L
    from jaeger_amd import fragment as frag
    from jaeger_amd.engine import HipDevice
    from jaeger_amd.termini import terminal_repeat_table
    seqs = _one_run_line_records(seed=5, n_records=300)
    bases, offsets = frag.concat_records(seqs)
    fa = frag.FastaBatch([f"r{i}" for i in range(len(seqs))], bases, offsets)
    dev = HipDevice(0)
    plain = terminal_repeat_table(dev, fa, 500)
    dev.set_stream_priority(True)
    high = terminal_repeat_table(dev, fa, 500)
    dev.set_stream_priority(False)
    again = terminal_repeat_table(dev, fa, 500)
    assert dev.lib.jg_engine_set_option(dev.handle, L.JG_OPT_STREAM_PRIORITY, 7) != 0
    dev.close()
    np.testing.assert_array_equal(plain, high)
    np.testing.assert_array_equal(plain, again)
    assert (plain[:, 0] > 100).sum() > 100
