"""Round 5: dust_kernel alone - 400 Mbp of random contigs (the bench's synthetic file has no planted low-complexity
stretches: what masks there is chance homopolymers) and the same with 2 % of the tiles carrying a planted repeat.
usage: python scripts/r5_dust_time.py"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from jaeger_amd.engine import HipDevice  # noqa: E402
dev = HipDevice(0)
rng = np.random.Generator(np.random.PCG64(5))
n = 400_000_000
acgt = np.frombuffer(b"ACGT", np.uint8)
for label, planted in (("random", 0.0), ("2 % of 640-base tiles with a planted 40-base repeat", 0.02)):
    bases = acgt[rng.integers(0, 4, n, dtype=np.uint8)]
    if planted:
        for p in rng.integers(0, n - 64, int(n / 640 * planted)):
            bases[p:p + 40] = np.tile(acgt[rng.integers(0, 4, 2)], 20)
    offsets = np.arange(0, n + 1, 40_000, dtype=np.int64)
    ptr = dev.upload(bases)
    try:
        for rep in range(3):
            t0 = time.perf_counter()
            masked = dev.dust_mask(ptr, n, offsets)
            dt = time.perf_counter() - t0
        print(f"{label}: {dt * 1e3:.1f} ms = {n / dt / 1e9:.1f} Gbp/s, {masked} bases masked ({masked / n * 100:.2f} %)", flush=True)
    finally:
        dev.free(ptr)
dev.close()
