"""End-to-end run_core on MANY short records (one million 500-bp records, 500-bp model): per-record host work is the cost
here - ingest names, window table, aggregation, 138 MB of TSV.  Prints the stage split and a cProfile of the calling thread."""
import cProfile, io, pstats, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import make_model_dir  # noqa: E402
from jaeger_amd import predict as P  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tmp = Path("/dev/shm/jaeger_r4_many"); tmp.mkdir(exist_ok=True)
rng = np.random.Generator(np.random.PCG64(20260925))
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * 500, dtype=np.uint8)].reshape(n, 500)
fa = tmp / "many.fasta"
with open(fa, "wb") as fh:
    for i in range(0, n, 10000):
        fh.write(b"".join(b">r%d\n%s\n" % (j, bases[j].tobytes()) for j in range(i, min(n, i + 10000))))
mdir = make_model_dir(tmp / "m", name="baseline500", model_name="jaeger_500bp_baseline")
kw = dict(input=str(fa), output=str(tmp / "out"), model_path=str(mdir), fsize=500, stride=500, overwrite=True, dustmask=True,
          verbose=0, batch=96, rc=0.1, pc=3)
for r in range(2):
    t0 = time.perf_counter(); P.run_core(**kw); dt = time.perf_counter() - t0
    print(f"run {r}: {dt:.2f} s = {n * 500 / dt / 1e6:.1f} Mbp/s", P.LAST_RUN, flush=True)
pr = cProfile.Profile(); pr.enable(); P.run_core(**kw); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue()[:6000])
