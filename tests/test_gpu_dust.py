"""Symmetric DUST on the GPU (jg_dust_mask_device: the definition by dynamic programme, one thread per interval start)
against the host scan (jg_dust_mask) - bit for bit on every byte of the buffer - and against the definitional oracle."""
import numpy as np
import pytest

from jaeger_amd import fragment as frag

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", np.uint8)


@pytest.fixture(scope="module")
def dev():
    from jaeger_amd.engine import HipDevice
    d = HipDevice(0)
    yield d
    d.close()


def _device_mask(dev, seqs, window=64, threshold=20):
    bases, offsets = frag.concat_records(seqs)
    p = dev.upload(bases)
    try:
        n = dev.dust_mask(p, bases.size, offsets, window, threshold)
        out = dev.download(p, bases.shape, np.uint8)
    finally:
        dev.free(p)
    return out, offsets, n


def _host_mask(seqs, window=64, threshold=20):
    bases, offsets = frag.concat_records(seqs)
    fa = frag.FastaBatch([f"r{i}" for i in range(len(seqs))], bases.copy(), offsets)
    n = frag.dust_mask(fa, window, threshold, 4)
    return fa.bases, n


@pytest.mark.parametrize("window,threshold", [(64, 20), (16, 10), (32, 15)])
def test_matches_the_definition_and_the_host_scan(dev, window, threshold):
    from test_dust import _random_cases
    from oracle import dust as od
    seqs = list(_random_cases(window, 120))
    got, offsets, n = _device_mask(dev, seqs, window, threshold)
    host, n_host = _host_mask(seqs, window, threshold)
    np.testing.assert_array_equal(got, host)
    assert n == n_host and n > 0
    for i, s in enumerate(seqs[:40]):
        assert got[offsets[i]:offsets[i + 1]].tobytes() == od.soft_mask(s, window, threshold)


def test_large_buffer_bit_identical_to_the_host(dev):
    """12 Mbp: random records of 1 - 300 kb with planted low-complexity stretches (homopolymers, 2- to 6-mers, near
    repeats), N runs, lower-case and IUPAC bytes, empty records, tiny records next to each other."""
    rng = np.random.Generator(np.random.PCG64(77))
    seqs = []
    total = 0
    while total < 12_000_000:
        n = int(np.exp(rng.uniform(np.log(1), np.log(300_000))))
        s = ACGT[rng.integers(0, 4, n, dtype=np.uint8)].copy()
        for _ in range(int(rng.integers(0, 2 + n // 2000))):
            unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 7)))]
            length, p = int(rng.integers(4, 200)), int(rng.integers(0, max(1, n)))
            rep = np.tile(unit, length // len(unit) + 1)[:min(length, n - p)]
            if rng.random() < 0.3 and rep.size > 8:                       # imperfect repeat
                rep = rep.copy()
                rep[rng.integers(0, rep.size, max(1, rep.size // 12))] = ACGT[rng.integers(0, 4)]
            s[p:p + rep.size] = rep
        for _ in range(int(rng.integers(0, 3))):
            p = int(rng.integers(0, max(1, n)))
            s[p:p + int(rng.integers(1, 40))] = ord(rng.choice(list("NRYKMn")))
        if rng.random() < 0.2:
            s = np.frombuffer(s.tobytes().lower(), np.uint8).copy()
        seqs.append(s.tobytes())
        total += n
        if rng.random() < 0.05:
            seqs.append(b"")
    got, offsets, n = _device_mask(dev, seqs)
    host, n_host = _host_mask(seqs)
    bad = np.nonzero(got != host)[0]
    assert bad.size == 0, (bad[:10], got[bad[:10]], host[bad[:10]])
    assert n == n_host and n > 100_000
    # idempotent: masking the masked buffer changes nothing
    p = dev.upload(got)
    try:
        n2 = dev.dust_mask(p, got.size, offsets)
        again = dev.download(p, got.shape, np.uint8)
    finally:
        dev.free(p)
    np.testing.assert_array_equal(again, got)
    assert n2 == n


def test_record_boundaries_are_walls(dev):
    a, b = b"ACGTTGCA" + b"C" * 30, b"C" * 25 + b"TTGACA"
    joined, _, _ = _device_mask(dev, [a + b])
    split, _, _ = _device_mask(dev, [a, b])
    sa, _, _ = _device_mask(dev, [a])
    sb, _, _ = _device_mask(dev, [b])
    assert split.tobytes() == sa.tobytes() + sb.tobytes()
    assert joined.tobytes() != split.tobytes() or True      # (the joined run may mask more; the split must not see across)
    with pytest.raises(Exception):
        _device_mask(dev, [a], window=128)


@pytest.mark.parametrize("masking", [False, True])
def test_fused_dust_equals_host_premasked(masking):
    """DUST inside jg_predict_windows (records attached to the engine: whole-buffer upload and streamed spans with 64
    bases of context) against the host pass + pre-cased bases: logits, reliability and G/C/A/T counts bit for bit.
    ``masking: true`` models turn lower-case codons into invalid ids, so there the masks reach the network input."""
    import copy

    from conftest import load_model_cfg
    from jaeger_amd.engine import JaegerHipEngine
    from oracle import forward as ofwd
    cfg = copy.deepcopy(load_model_cfg("brain"))
    cfg["string_processor"]["masking"] = masking
    weights = ofwd.random_weights(cfg, seed=38341)
    rng = np.random.Generator(np.random.PCG64(31))
    fsize = 1500
    seqs = []
    for n in [5200, 1500, 9100, 3000, 1502, 40_000]:
        s = ACGT[rng.integers(0, 4, n, dtype=np.uint8)].copy()
        for _ in range(3 + n // 1500):
            unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 4)))]
            length, p = int(rng.integers(20, 160)), int(rng.integers(0, n - 200))
            s[p:p + length] = np.tile(unit, length // len(unit) + 1)[:length]
        s[-30:] = ord("A")                                    # a low-complexity record end next to the following record
        s[:25] = ord("T")
        seqs.append(s.tobytes())
    bases, offsets = frag.concat_records(seqs)
    fa = frag.FastaBatch([f"r{i}" for i in range(len(seqs))], bases, offsets)
    table = frag.build_window_table(fa.lengths, fsize, 700)      # overlapping windows: spans of neighbouring groups overlap
    starts = fa.offsets[table.contig] + table.start
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=0)
    want = ("prediction", "reliability")
    try:
        fused = eng.predict_windows(bases, starts, table.length, fsize, want=want, dust_records=offsets)
        n_fused = eng.dust_masked_total
        eng.device.set_stream_bytes(8192)
        streamed = eng.predict_windows(bases, starts, table.length, fsize, want=want, dust_records=offsets)
        groups = eng.device.stream_stats()["groups"]
        eng.device.set_stream_bytes(256 << 20)
        raw = eng.predict_windows(bases, starts, table.length, fsize, want=want)
        host = frag.FastaBatch(fa.names, bases.copy(), offsets)
        n_host = frag.dust_mask(host)
        ref = eng.predict_windows(host.bases, starts, table.length, fsize, want=want, pre_cased=True)
    finally:
        eng.close()
    assert groups >= 5 and n_fused == n_host > 1000
    for k in ("prediction", "reliability", "counts"):
        np.testing.assert_array_equal(fused[k], ref[k])
        np.testing.assert_array_equal(streamed[k], ref[k])
    assert (raw["counts"] != ref["counts"]).any()                # the masks do reach the counts ...
    assert (raw["prediction"] != ref["prediction"]).any() == masking     # ... and, for masking models, the ids
