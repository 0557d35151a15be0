"""The two statements of the terminal-repeat alignment in oracle/termini.py agree (loops vs numpy rows)."""
import numpy as np

from oracle import termini as ot


def test_vectorised_oracle_matches_loops():
    rng = np.random.Generator(np.random.PCG64(3))
    for trial in range(40):
        n, m = int(rng.integers(20, 110)), int(rng.integers(20, 110))
        q = "".join(rng.choice(list("ACGT"), n))
        r = "".join(rng.choice(list("ACGT"), m))
        if trial % 2:                        # plant a common stretch, sometimes with a mismatch or a gap
            core = "".join(rng.choice(list("ACGTN" if trial % 8 == 1 else "ACGT"), int(rng.integers(14, 70))))
            var = core
            if trial % 3 == 0 and len(core) > 12:
                var = core[:len(core) // 2] + core[len(core) // 2 + 1:]
            elif trial % 5 == 0:
                mid = len(core) // 2
                var = core[:mid] + ("A" if core[mid] != "A" else "C") + core[mid + 1:]
            a, b = int(rng.integers(0, n // 2)), int(rng.integers(0, m // 2))
            q, r = q[:a] + core + q[a:], r[:b] + var + r[b:]
        assert ot.smith_waterman(q, r) == ot.smith_waterman_loops(q, r), (q, r)


def test_decision_rule():
    d = lambda s, n, g=0: {"score": s, "length": n, "fgaps": g}  # noqa: E731
    assert ot.classify(d(20, 10), d(24, 12)) == (None, None)
    assert ot.classify(d(40, 20), d(30, 15)) == ("DTR", 20)
    assert ot.classify(d(30, 15), d(40, 20)) == ("ITR", 20)
    assert ot.classify(d(40, 20), d(40, 20)) == ("DTR", 20)          # ties go to the direct repeat
    assert ot.classify(d(600, 300), d(0, 0)) == ("LTR_DTR", 300)
    assert ot.classify(d(398, 251, 2), d(0, 0)) == ("DTR", 251)      # 249 query bases: below the LTR cut-off
    assert ot.scan_length(44776) == 1791 and ot.scan_length(5000) == 400 and ot.scan_length(10 ** 6) == 4000
