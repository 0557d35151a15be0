"""Round 6: the HOST side of run_core on one million 500-bp records, without a GPU - ingest, window table, per-contig
aggregation in batches, repeat columns, table rows - each stage timed and the calling thread profiled.  The engine's outputs
are random logits; the repeat table says what the real scan says for these records (one direct repeat of 300 columns each).
usage: python scripts/r6_many_host.py [records] [profile]"""
import cProfile, io, pstats, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from jaeger_amd import fragment as frag  # noqa: E402
from jaeger_amd import predict as P  # noqa: E402
from jaeger_amd.termini import RepeatColumns  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
prof = len(sys.argv) > 2
tmp = Path("/dev/shm/r6"); tmp.mkdir(exist_ok=True)
fa_path = tmp / f"many_{n}.fasta"
if not fa_path.exists():
    rng = np.random.Generator(np.random.PCG64(20260925))
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * 500, dtype=np.uint8)]
    bench.write_fasta_records(fa_path, bases.reshape(n, 500))
class_map = {"num_classes": 3, "class": ["bacteria", "phage", "eukarya"], "index": [0, 1, 2]}


def run():
    T = {}
    t0 = time.perf_counter()
    fa = frag.load_fasta(str(fa_path)); T["ingest"] = time.perf_counter() - t0
    t1 = time.perf_counter()
    table = frag.build_window_table(fa.lengths, 500, 500, False, 10.0, 500, None)
    starts = fa.offsets[table.contig] + table.start; T["table"] = time.perf_counter() - t1
    rng = np.random.default_rng(1)
    out = {"prediction": rng.normal(size=(len(table), 3)).astype(np.float32), "reliability": rng.normal(size=(len(table), 1)).astype(np.float32),
           "counts": np.full((len(table), 4), 125, np.int32)}
    t2 = time.perf_counter()
    writer = P._LazyTableWriter(class_map, tmp / "o.tsv", tmp / "o_phages.tsv", 0.1, 3)
    agg = P._Aggregator(table, fa.names, out, dict(class_map=class_map, fsize=500, term_repeats=None, want_full=False), min_batch=len(table) // 16)
    T["agg_init"] = time.perf_counter() - t2
    t3 = time.perf_counter()
    res = np.full((n, 10), 0, np.int32); res[:, 0] = 600; res[:, 1] = 300; res[:, 3] = 399; res[:, 4] = 399
    rep = RepeatColumns(res, fa.names, fa.lengths); T["repeat_columns"] = time.perf_counter() - t3
    t4 = time.perf_counter(); agg.names_unique(); T["names_unique"] = time.perf_counter() - t4
    t5 = time.perf_counter()
    for done in np.linspace(0, len(table), 17)[1:]:
        agg.advance(int(done))
    agg.advance(len(table), final=True); T["aggregate"] = time.perf_counter() - t5
    t6 = time.perf_counter(); agg.flush(writer, rep); T["flush_rows"] = time.perf_counter() - t6
    t7 = time.perf_counter(); nw = writer.close(); T["close"] = time.perf_counter() - t7
    T["total"] = time.perf_counter() - t0
    return T, nw


for r in range(2):
    T, nw = run()
    print(f"run {r}: rows {nw}  " + "  ".join(f"{k} {v:.3f}" for k, v in T.items()), flush=True)
if prof:
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:7000])
print((tmp / "o.tsv").stat().st_size, "bytes of TSV")
