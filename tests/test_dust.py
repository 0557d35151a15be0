"""Symmetric DUST soft-masking (jg_dust_mask) against the definitional evaluation in oracle/dust.py
and against the algorithm's defining properties.  (pydustmasker itself is not installable here.)"""
import numpy as np
import pytest

from jaeger_amd import fragment as frag
from oracle import dust as od


def _native(seqs, window=64, threshold=20, threads=1):
    bases, offsets = frag.concat_records(seqs)
    fa = frag.FastaBatch([f"r{i}" for i in range(len(seqs))], bases, offsets)
    n = frag.dust_mask(fa, window, threshold, threads)
    return [fa.sequence(i) for i in range(len(seqs))], n


def _random_cases(seed, n_cases):
    rng = np.random.Generator(np.random.PCG64(seed))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for _ in range(n_cases):
        n = int(rng.integers(3, 400))
        s = bytearray(acgt[rng.integers(0, 4, n)].tobytes())
        for _ in range(int(rng.integers(0, 4))):
            unit = acgt[rng.integers(0, 4, int(rng.integers(1, 5)))].tobytes()
            length, p = int(rng.integers(5, 90)), int(rng.integers(0, max(1, n - 5)))
            rep = (unit * (length // len(unit) + 1))[:min(length, n - p)]
            s[p:p + len(rep)] = rep
        if rng.random() < 0.4:
            p = int(rng.integers(0, n))
            s[p:p + 2] = b"NN"[:n - p]
        if rng.random() < 0.2:
            s = bytearray(bytes(s).lower())
        yield bytes(s[:n])


@pytest.mark.parametrize("window,threshold", [(64, 20), (16, 10), (32, 15)])
def test_matches_the_definition(window, threshold):
    seqs = list(_random_cases(window, 120))
    got, n = _native(seqs, window, threshold)
    masked = 0
    for s, g in zip(seqs, got):
        want = od.soft_mask(s, window, threshold)
        assert g == want, (s, g, want)
        masked += sum(1 for c in want if 97 <= c <= 122)
    assert n == masked and masked > 0


def test_known_shapes():
    rng = np.random.Generator(np.random.PCG64(2))
    rnd = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 300)].tobytes()
    (g,), _ = _native([rnd[:100] + b"A" * 40 + rnd[100:200] + b"AT" * 30 + rnd[200:]])
    assert g[100:140].islower() and g[240:300].islower()
    assert sum(1 for c in g if 97 <= c <= 122) < 130                    # little beyond the planted runs
    (g2,), n2 = _native([rnd])
    assert n2 <= 20                                                     # random sequence: (almost) nothing
    # an ambiguous base splits the scan: the two halves are masked as if they were separate records
    a, b = b"ACGTTGCA" + b"C" * 30, b"G" * 25 + b"TTGACA"
    (joined,), _ = _native([a + b"N" + b])
    (ga, gb), _ = _native([a, b])
    assert joined == ga + b"N" + gb


def test_idempotent_and_thread_invariant():
    seqs = list(_random_cases(11, 200))
    one, n1 = _native(seqs, threads=1)
    many, n8 = _native(seqs, threads=8)
    assert one == many and n1 == n8
    again, n_again = _native([s.upper() for s in one])
    assert again == one and n_again == n1
