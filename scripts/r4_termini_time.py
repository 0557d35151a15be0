"""Terminal-repeat scan alone: the 10 000-contig workload and one million 500-bp records (round 3: 0.11 s and 1.04 s),
packed score-only pass (default) and every alignment through the exact kernel (JG_OPT_TERMINI_EXACT)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from bench import synth_contigs
from jaeger_amd import _lib as L
from jaeger_amd import fragment as frag
from jaeger_amd.engine import HipDevice
from jaeger_amd.termini import terminal_repeat_table
d = HipDevice(0)
for label, n, exact_len, fsize in (("10 000 contigs 1.5 - 200 kb", 10000, None, 1500), ("1 000 000 records of 500 bp", 1_000_000, 500, 500)):
    rng = np.random.Generator(np.random.PCG64(20260923))
    lengths, bases = synth_contigs(rng, n, exact=exact_len)
    offs = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs[1:])
    fa = frag.FastaBatch([""] * len(lengths), bases, offs)
    for exact in (0, 1):
        L.check(d.lib.jg_engine_set_option(d.handle, L.JG_OPT_TERMINI_EXACT, exact))
        terminal_repeat_table(d, fa, fsize)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); tab = terminal_repeat_table(d, fa, fsize); ts.append(time.perf_counter() - t)
        print(f"{label}: {'exact kernel for every alignment' if exact else 'packed score-only pass + exact above 100'}: "
              f"{min(ts) * 1e3:.0f} ms (incl. H2D of the ends, host job tables), checksum {int(np.asarray(tab, np.int64).sum())}", flush=True)
d.close()
