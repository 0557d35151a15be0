"""Round 5: timelines (predict.LAST_RUN["timeline"]) of short-forward end-to-end runs with the 500-bp model:
``10k`` = the 10 000-contig FASTA of BASELINE configs[1], ``many`` = one million 500-bp records.
usage: python scripts/r5_e2e_timeline.py [10k|many] [repeats]"""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from conftest import make_model_dir  # noqa: E402
from jaeger_amd import predict as P  # noqa: E402
which = sys.argv[1] if len(sys.argv) > 1 else "10k"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tmp = Path("/dev/shm/jaeger_r5_e2e"); tmp.mkdir(exist_ok=True)
fa = tmp / f"{which}.fasta"
if which == "many":
    rng = np.random.Generator(np.random.PCG64(20260925))
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 1_000_000 * 500, dtype=np.uint8)]
    bench.write_fasta_records(fa, bases.reshape(-1, 500))
else:
    rng = np.random.Generator(np.random.PCG64(20260923))
    lengths, bases = bench.synth_contigs(rng, 10000)
    bench.write_fasta_contigs(fa, lengths, bases)
mdir = make_model_dir(tmp / "m", name="baseline500", model_name="jaeger_500bp_baseline")
kw = dict(input=str(fa), output=str(tmp / "out"), model_path=str(mdir), fsize=500, stride=500, overwrite=True, dustmask=True,
          verbose=0, batch=96, rc=0.1, pc=3)
import os
if os.environ.get("JAEGER_SCAN_FIRST"):
    kw["scan_first"] = True
for r in range(reps):
    e0 = time.time(); t0 = time.perf_counter(); P.run_core(**kw); dt = time.perf_counter() - t0; e1 = time.time()
    lr = dict(P.LAST_RUN); tl = lr.pop("timeline"); ts = lr.pop("t_start_epoch")
    print(f"   before run_core's clock {ts - e0:.4f} s, behind its last mark {e1 - ts - tl[-1][1]:.4f} s")
    print(f"== {which} run {r}: {dt:.3f} s = {bases.size / dt / 1e6:.1f} Mbp/s  {json.dumps(lr)}", flush=True)
    print("   ", "  ".join(f"{n}@{t:.3f}" for n, t in tl), flush=True)
    time.sleep(0.3)
