"""GPU tests at the sizes of BASELINE.json configs[2] and configs[4], run as the per-rank shard one of the eight
GPUs would own, through ``predict_windows`` with host buffers:

* oracle parity (logits within 1e-4) on a seeded sample of >= 500 windows drawn across the whole run,
* full-size invariants: G/C/A/T counts exact, results bit-identical under a different stream budget / windows per
  pass (chunk invariance implies repeatability: the second run repeats every window in other launch groups),
* configs[4]: the host-DRAM -> HBM ingest really is streamed (several groups, the device never holds the whole buffer).

Also the N-rank launch of ``bench.py --gpus N`` (two ranks sharing the one GPU of the test box, gloo exchange).
"""
import json
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT, load_model_cfg

pytestmark = pytest.mark.gpu

TOL = 1e-4
ACGT = np.frombuffer(b"ACGT", np.uint8)


@pytest.fixture(scope="module")
def brain():
    from jaeger_amd.engine import JaegerHipEngine
    from oracle import forward as ofwd
    cfg = load_model_cfg("brain")
    weights = ofwd.random_weights(cfg, seed=38341)
    with pytest.warns(UserWarning):
        eng = JaegerHipEngine(model_cfg=cfg, weights=weights, device_id=0)
    assert eng.model.precision == "f16x3"
    yield cfg, weights, eng
    eng.close()


def _oracle_check(cfg, weights, bases, starts, lens, fsize, got, sample):
    """Sampled windows against the CPU oracle on its fused oneDNN form (``oracle.forward.FAST``: proven equal to the
    spelled-out form in tests/test_oracle_forward.py), 256 windows per batch - the suite's time on a slow host is this
    function's (the GPU suite has to fit the driver's step limit whatever host the box has)."""
    from jaeger_amd.engine import frame_length
    from oracle import encoder as oenc
    from oracle import forward as ofwd
    windows = [bases[starts[i]:starts[i] + lens[i]].tobytes() for i in sample]
    ids = oenc.encode_windows(windows, fsize, pad_to=frame_length(fsize))
    worst = {}
    t0 = time.time()
    ofwd.FAST = True
    try:
        for i in range(0, len(sample), 256):
            ref = ofwd.forward(cfg, weights, ids[i:i + 256])
            sl = sample[i:i + 256]
            for k in ("prediction", "reliability"):
                worst[k] = max(worst.get(k, 0.0), float(np.abs(got[k][sl] - ref[k]).max()))
    finally:
        ofwd.FAST = False
    print("oracle forward of", len(sample), "windows: %.1f s" % (time.time() - t0))
    ref_counts = np.array([oenc.window_counts(w) for w in windows], np.int32)
    np.testing.assert_array_equal(got["counts"][sample], ref_counts)
    print("oracle parity on", len(sample), "sampled windows:", worst)
    assert worst["prediction"] < TOL and worst["reliability"] < TOL, worst


def test_config3_shard_125k_fragments(brain):
    """configs[2] per-rank shard: 125 000 fragments of exactly 1 500 bp, one window each, PCG64(seed + 1)."""
    from jaeger_amd.fragment import build_window_table
    cfg, weights, eng = brain
    fsize, n_frag = 1500, 125_000
    rng = np.random.Generator(np.random.PCG64(20260923 + 1))
    bases = ACGT[rng.integers(0, 4, fsize * n_frag, dtype=np.uint8)]
    # SURVEY 8(d)'s variant of the workload: 1 % of the bases in N runs of 1 - 20 (round 6) - masks at work in every layer of
    # every sampled window that holds one, at the config's full size
    pos = rng.integers(0, bases.size - 20, int(bases.size * 0.01 / 10.5))
    for p_, l_ in zip(pos.tolist(), rng.integers(1, 21, pos.size).tolist()):
        bases[p_:p_ + l_] = ord("N")
    n_acgt = int((bases != ord("N")).sum())
    assert 0.985 * bases.size < n_acgt < 0.995 * bases.size
    lengths = np.full(n_frag, fsize, np.int64)
    table = build_window_table(lengths, fsize, fsize)
    assert len(table) == n_frag and table.is_last.all()
    starts = (np.arange(n_frag, dtype=np.int64) * fsize)[table.contig] + table.start
    want = ("prediction", "reliability")
    eng.device.set_stream_bytes(256 << 20)
    got = eng.predict_windows(bases, starts, table.length, fsize, want=want)
    assert got["prediction"].shape == (n_frag, 6) and np.isfinite(got["prediction"]).all()
    assert int(got["counts"].sum()) == n_acgt                         # every base but the Ns is an upper-case A/C/G/T
    # the gathered (n, 6) f32 logit matrix of the full config is 24 MB; this shard's part is 3 MB
    assert got["prediction"].nbytes == n_frag * 6 * 4
    # bit-identical in other launch groups: 1 000 windows per pass and a 16 MiB stream budget (12 groups)
    eng.device.set_stream_bytes(16 << 20)
    eng.chunk = 1000
    try:
        again = eng.predict_windows(bases, starts, table.length, fsize, want=want)
        stats = eng.device.stream_stats()
    finally:
        eng.chunk = 0
        eng.device.set_stream_bytes(256 << 20)
    assert stats["groups"] >= 11 and stats["peak_device_bases"] < 40 << 20
    for k in ("prediction", "reliability", "counts"):
        np.testing.assert_array_equal(got[k], again[k])
    # 1 024 sampled windows here, >= 2 000 in the configs[4] test below
    sample = np.sort(np.random.Generator(np.random.PCG64(3)).choice(n_frag, 1024, replace=False))
    _oracle_check(cfg, weights, bases, starts, table.length, fsize, got, sample)


def mixed_assembly(rng, total_bp: int):
    """configs[4] length mixture: by count 70 % log-uniform 1.5-20 kb, 25 % 20-200 kb,
    5 % 0.2-5 Mb, drawn until ``total_bp`` is reached; bases generated block-wise into one buffer."""
    lens = []
    acc = 0
    while acc < total_bp:
        u = rng.random(4096)
        lo = np.where(u < 0.70, 1500.0, np.where(u < 0.95, 20e3, 200e3))
        hi = np.where(u < 0.70, 20e3, np.where(u < 0.95, 200e3, 5e6))
        batch = np.exp(rng.uniform(np.log(lo), np.log(hi))).astype(np.int64)
        for v in batch:
            lens.append(int(v))
            acc += int(v)
            if acc >= total_bp:
                break
    lengths = np.asarray(lens, np.int64)
    bases = np.empty(int(lengths.sum()), np.uint8)
    step = 1 << 26
    for o in range(0, bases.size, step):
        n = min(step, bases.size - o)
        bases[o:o + n] = ACGT[rng.integers(0, 4, n, dtype=np.uint8)]
    return lengths, bases


def test_config5_shard_mixed_lengths_streamed(brain):
    """configs[4] per-rank shard: >= 1.25 Gbp of mixed-length contigs incl. 0.2-5 Mb ones, host buffers, streamed."""
    from jaeger_amd.fragment import build_window_table
    cfg, weights, eng = brain
    fsize = 1500
    rng = np.random.Generator(np.random.PCG64(20260923 + 4))
    lengths, bases = mixed_assembly(rng, 1_250_000_000)
    assert bases.size >= 1_250_000_000 and lengths.max() > 1_000_000 and (lengths < 20_000).mean() > 0.6
    offsets = np.zeros(lengths.size + 1, np.int64)
    np.cumsum(lengths, out=offsets[1:])
    table = build_window_table(lengths, fsize, fsize)
    starts = offsets[table.contig] + table.start
    n_win = len(table)
    assert n_win > 800_000
    want = ("prediction", "reliability")
    eng.device.set_stream_bytes(256 << 20)
    got = eng.predict_windows(bases, starts, table.length, fsize, want=want)
    stats = eng.device.stream_stats()
    print("streamed ingest:", stats, "windows:", n_win)
    # streamed: several groups, all bases went through the staging buffers once, and the device held two spans
    assert stats["groups"] >= 4
    assert stats["bytes"] <= bases.size and stats["bytes"] >= int(table.length.sum())
    assert stats["peak_device_bases"] <= 2 * (256 << 20) + 8192 < bases.size // 2
    assert np.isfinite(got["prediction"]).all()
    assert int(got["counts"].sum()) == n_win * fsize
    # other launch groups (64 MiB spans, 1 536 windows per pass): bit-identical
    eng.device.set_stream_bytes(64 << 20)
    eng.chunk = 1536
    try:
        again = eng.predict_windows(bases, starts, table.length, fsize, want=want)
        assert eng.device.stream_stats()["groups"] >= 16
    finally:
        eng.chunk = 0
        eng.device.set_stream_bytes(256 << 20)
    for k in ("prediction", "reliability", "counts"):
        np.testing.assert_array_equal(got[k], again[k])
    # sampled oracle parity across the run: windows of short, medium and megabase contigs, first / last windows
    srng = np.random.Generator(np.random.PCG64(5))
    sample = set(srng.choice(n_win, 2000, replace=False).tolist())
    big = int(np.argmax(lengths))
    w_big = np.nonzero(table.contig == big)[0]
    sample.update([0, n_win - 1, int(w_big[0]), int(w_big[-1]), int(w_big[len(w_big) // 2])])
    sample.update(np.nonzero(table.is_last == 1)[0][:16].tolist())
    sample = np.sort(np.fromiter(sample, np.int64))
    assert len(sample) >= 2000
    _oracle_check(cfg, weights, bases, starts, table.length, fsize, got, sample)


def test_bench_two_ranks_gather_equals_single_rank_runs(tmp_path):
    """BASELINE configs[2] in miniature: rank 0's gathered logits of a 2-rank ``--config frag1m`` launch are, bit for bit,
    rank 0's own logits followed by rank 1's - each reproduced by a single-rank run on that rank's contig set."""
    base = [sys.executable, str(ROOT / "bench.py"), "--config", "frag1m", "--contigs", "400", "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--no-exact-f32", "--no-e2e", "--no-also", "--no-box"]
    both = tmp_path / "both.npy"
    res = subprocess.run(base + ["--gpus", "2", "--oversubscribe", "--dump-gather", str(both)], capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    parts = []
    for r in (0, 1):
        one = tmp_path / f"r{r}.npy"
        res = subprocess.run(base + ["--rank-seed", str(r), "--dump-gather", str(one)], capture_output=True, text=True,
                             timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        parts.append(np.load(one))
    got = np.load(both)
    assert got.shape == (800, parts[0].shape[1]) and parts[0].shape[0] == 400
    np.testing.assert_array_equal(got, np.concatenate(parts, axis=0))
    assert np.abs(got).max() > 0


def test_bench_eight_ranks_unequal_shards_gather_equals_single_rank_runs(tmp_path):
    """The first real 8-GPU launch in miniature: eight ranks (here sharing the box's one GPU, gloo instead of RCCL) with
    UNEQUAL shards and one EMPTY shard; rank 0's padded gather is, bit for bit, the eight single-rank runs back to back,
    and the line carries the per-rank split (windows min / max, compute and gather time of the slowest rank)."""
    shards = [200, 0, 150, 40, 200, 7, 120, 64]
    base = [sys.executable, str(ROOT / "bench.py"), "--config", "frag1m", "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--no-exact-f32", "--no-e2e", "--no-also", "--no-box", "--rank-contigs",
            ",".join(map(str, shards))]
    both = tmp_path / "all.npy"
    res = subprocess.run(base + ["--gpus", "8", "--oversubscribe", "--dump-gather", str(both)], capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["per_rank"]["windows"] == shards
    assert line["config"]["windows_per_gpu_min"] == 0 and line["config"]["windows_per_gpu_max"] == 200
    pr = line["per_rank"]
    assert pr["ms_per_step_slowest"] >= pr["ms_per_step_fastest"] > 0 and pr["gather_ms_per_step"]["max"] > 0
    parts = []
    for r, n in enumerate(shards):
        if n == 0:
            continue
        one = tmp_path / f"r{r}.npy"
        res = subprocess.run(base + ["--rank-seed", str(r), "--dump-gather", str(one)], capture_output=True, text=True,
                             timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        parts.append(np.load(one))
        assert parts[-1].shape[0] == n
    got = np.load(both)
    assert got.shape[0] == sum(shards)
    np.testing.assert_array_equal(got, np.concatenate(parts, axis=0))
    assert np.abs(got).max() > 0


def test_bench_failing_rank_is_reported_and_times_out(tmp_path):
    """A rank that dies surfaces its stderr and a non-zero exit; ranks that hang are killed after --rank-timeout."""
    import os
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--oversubscribe", "--contigs", "50", "--steps", "1",
           "--warmup", "0", "--no-cpu-baseline", "--no-exact-f32", "--no-e2e", "--no-also", "--no-box", "--rank-timeout", "240"]
    t0 = time.time()
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, JAEGER_BENCH_FAIL_RANK="1"))
    assert res.returncode != 0
    assert "rank 1" in res.stderr and "JAEGER_BENCH_FAIL_RANK" in res.stderr, res.stderr[-2000:]
    assert time.time() - t0 < 200, "the surviving rank was not ended after the grace period"


def test_bench_gpus_flag_spawns_ranks():
    """``python bench.py --gpus 2`` with no torchrun environment launches two ranks by itself and reports n_gpus 2
    (here both ranks share the test box's one GPU and exchange over gloo: --oversubscribe)."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--oversubscribe", "--contigs", "200", "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline", "--no-exact-f32", "--no-e2e", "--no-also", "--no-box"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["windows_per_gpu"] > 0 and line["roofline"]["launches"] > 0
    # a --gpus that contradicts the torchrun environment is an error, not a silent single-GPU run
    import os
    bad = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert bad.returncode == 2 and "does not match WORLD_SIZE" in bad.stderr


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_streamed_ingest_random_window_tables(brain, seed):
    """Streamed vs whole-buffer ingest on random start-sorted window tables (overlapping strides, ragged lengths,
    windows touching both ends of the buffer) under tiny stream budgets: bit-identical outputs, bounded device bases."""
    cfg, weights, eng = brain
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    fsize = int(rng.choice([600, 900, 1500]))
    n_bases = int(rng.integers(40_000, 400_000))
    bases = ACGT[rng.integers(0, 4, n_bases, dtype=np.uint8)]
    bases[rng.integers(0, n_bases, 50)] = ord("N")
    n_win = int(rng.integers(5, 120))
    starts = np.sort(rng.integers(0, n_bases - fsize, n_win)).astype(np.int64)
    starts[0], starts[-1] = 0, n_bases - fsize
    lens = np.full(n_win, fsize, np.int32)
    lens[rng.random(n_win) < 0.2] = rng.integers(fsize // 2, fsize)
    want = ("prediction", "reliability")
    eng.device.set_stream_bytes(1 << 30)
    whole = eng.predict_windows(bases, starts, lens, fsize, want=want)
    assert eng.device.stream_stats()["groups"] == 0
    budget = int(rng.choice([4096, 8192, 32768]))          # < n_bases: always streamed
    eng.device.set_stream_bytes(budget)
    try:
        streamed = eng.predict_windows(bases, starts, lens, fsize, want=want)
        stats = eng.device.stream_stats()
    finally:
        eng.device.set_stream_bytes(256 << 20)
    assert stats["groups"] >= 2 and stats["bytes"] >= int(lens.sum()) // 2
    for k in ("prediction", "reliability", "counts"):
        np.testing.assert_array_equal(whole[k], streamed[k])


def test_bench_one_rank_over_rccl(tmp_path):
    """First contact for the collective code on the one GPU there is: ``--gpus 1 --collective nccl``
    initialises the RCCL process group on the device, gathers the logits with ``dist.gather`` on DEVICE tensors inside the
    timed region, all_gathers the per-rank statistics, takes the barriers and destroys the group - and rank 0's gathered
    matrix is the plain single-rank run's, bit for bit.  A failure exits non-zero with the RCCL / HIP error (no gloo)."""
    base = [sys.executable, str(ROOT / "bench.py"), "--config", "frag1m", "--contigs", "400", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline", "--no-exact-f32", "--no-e2e", "--no-also", "--no-box"]
    res = subprocess.run(base + ["--collective", "nccl", "--dump-gather", str(tmp_path / "rccl.npy")], capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    # exactly ONE line on stdout - RCCL's version banner (C stdio, flushed at exit) must not follow the JSON line
    assert len(res.stdout.strip().splitlines()) == 1, res.stdout[-600:]
    line = json.loads(res.stdout.strip())
    assert line["n_gpus"] == 1 and line["config"]["collective_backend"] == "nccl"
    assert line["config"]["parallelism"].endswith("final RCCL gather (executed)")
    assert line["per_rank"]["gather_ms_per_step"]["max"] > 0 and line["per_rank"]["windows"] == [400]
    res = subprocess.run(base + ["--dump-gather", str(tmp_path / "plain.npy")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    plain = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert plain["config"]["collective_backend"] is None and "no exchange" in plain["config"]["parallelism"]
    a, b = np.load(tmp_path / "rccl.npy"), np.load(tmp_path / "plain.npy")
    assert a.shape == (400, 6) and np.abs(a).max() > 0
    np.testing.assert_array_equal(a, b)
