"""Round 4: where does an end-to-end run_core spend its wall time?  10 000-contig synthetic FASTA on tmpfs, brain (1 500 bp)
and baseline500 (500 bp); plain repeats, then a cProfile of the main thread for the 500-bp family.
usage: python scripts/r4_e2e_prof.py [brain|baseline500|both] [repeats]"""
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from bench import synth_contigs  # noqa: E402
from conftest import make_model_dir  # noqa: E402

import os
which = sys.argv[1] if len(sys.argv) > 1 else "both"
extra = {}
if os.environ.get("JAEGER_STREAM_BYTES"):
    extra["stream_bytes"] = int(os.environ["JAEGER_STREAM_BYTES"])
if os.environ.get("JAEGER_DUST_STREAM"):
    extra["dust_stream"] = int(os.environ["JAEGER_DUST_STREAM"])
if os.environ.get("JAEGER_NO_DUST"):
    extra["dustmask_off"] = True
if os.environ.get("JAEGER_NO_PIPELINE"):
    extra["no_pipeline"] = True
do_prof = not os.environ.get("JAEGER_NO_CPROFILE")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tmp = Path("/dev/shm/jaeger_r4_e2e")
tmp.mkdir(exist_ok=True)
fa = tmp / "synth10k.fasta"
rng = np.random.Generator(np.random.PCG64(20260923))
lengths, bases = synth_contigs(rng, 10000)
with open(fa, "wb") as fh:
    off = 0
    for i, n in enumerate(lengths.tolist()):
        fh.write(b">contig_%d len=%d\n" % (i, n))
        s = bases[off:off + n].tobytes()
        off += n
        fh.write(b"\n".join(s[j:j + 80] for j in range(0, n, 80)) + b"\n")
mbp = bases.size / 1e6
from jaeger_amd.predict import run_core  # noqa: E402

models = {"brain": (make_model_dir(tmp / "m_brain"), 1500),
          "baseline500": (make_model_dir(tmp / "m_b500", name="baseline500", model_name="jaeger_500bp_baseline"), 500),
          "dvf500": (make_model_dir(tmp / "m_dvf", name="dvf500", model_name="jaeger_500bp_dvf"), 500)}
for name, (mdir, fsize) in models.items():
    if which not in ("both", name):
        continue
    for r in range(reps):
        t0 = time.perf_counter()
        run_core(input=str(fa), output=str(tmp / f"out_{name}"), model_path=str(mdir), fsize=fsize, stride=fsize,
                 overwrite=True, dustmask=not extra.pop('dustmask_off', False) if False else ('dustmask_off' not in extra), verbose=0, batch=96, rc=0.1, pc=3,
                 **{k: v for k, v in extra.items() if k != 'dustmask_off'})
        dt = time.perf_counter() - t0
        print(f"== {name}: run {r}: {dt:.3f} s = {mbp / dt:.1f} Mbp/s", flush=True)
        log = sorted((tmp / f"out_{name}").rglob("*_jaeger.log"))[-1]
        for line in log.read_text().splitlines()[-6:]:
            print("   ", line.split("[jaeger]")[-1].strip())
if do_prof and which in ("both", "baseline500"):
    mdir, fsize = models["baseline500"]
    pr = cProfile.Profile()
    pr.enable()
    run_core(input=str(fa), output=str(tmp / "out_prof"), model_path=str(mdir), fsize=fsize, stride=fsize,
             overwrite=True, dustmask=True, verbose=0, batch=96, rc=0.1, pc=3)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue()[:9000])
