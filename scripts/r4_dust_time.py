"""dust_kernel alone: 400 Mbp of random bases (10 000 records) resident in HBM, jg_dust_mask_device, wall time of the
synchronous call (n_masked read back)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from jaeger_amd.engine import HipDevice  # noqa: E402
rng = np.random.Generator(np.random.PCG64(20260923))
lens = np.clip(np.exp(rng.uniform(np.log(1500), np.log(200000), 10000)).astype(np.int64), 1500, 200000)
off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(off[-1]), dtype=np.uint8)]
dev = HipDevice(0)
ptr = dev.upload(bases)
for rep in range(4):
    t0 = time.perf_counter(); n = dev.dust_mask(ptr, bases.size, off); dt = time.perf_counter() - t0
    print(f"run {rep}: {dt * 1e3:.1f} ms = {bases.size / dt / 1e9:.2f} Gbp/s, {n} bases masked")
import zlib
print("crc of the masked buffer", zlib.crc32(dev.download(ptr, bases.shape, np.uint8).tobytes()))
