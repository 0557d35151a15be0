"""Host-side cost of jg_predict_windows on a 407 Mbp host buffer (baseline500, 500-bp windows): stream budget sweep."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from conftest import load_model_cfg
from bench import synth_contigs
from jaeger_amd.engine import JaegerHipEngine
from jaeger_amd.fragment import build_window_table
from oracle import forward as ofwd
import warnings
warnings.simplefilter("ignore")
cfg = load_model_cfg("baseline500")
w = ofwd.random_weights(cfg, seed=1)
rng = np.random.Generator(np.random.PCG64(20260923))
lengths, bases = synth_contigs(rng, 10000)
offs = np.zeros(len(lengths) + 1, np.int64); np.cumsum(lengths, out=offs[1:])
tab = build_window_table(lengths, 500, 500)
starts = offs[tab.contig] + tab.start
for budget in (256 << 20, 1 << 30, 256 << 20, 1 << 30):
    eng = JaegerHipEngine(model_cfg=cfg, weights=w, device_id=0)
    eng.device.set_stream_bytes(budget)
    for rep in range(3):
        t = time.time()
        out = eng.predict_windows(bases, starts, tab.length, 500, want=("prediction", "reliability"),
                                  dust_records=offs if rep == 2 else None)
        dt = time.time() - t
        print(f"budget {budget >> 20} MiB call {rep}{' +dust' if rep == 2 else ''}: {dt * 1e3:.0f} ms, groups {eng.device.stream_stats()['groups']}", flush=True)
    eng.close()

eng = JaegerHipEngine(model_cfg=cfg, weights=w, device_id=0)
eng.device.set_stream_bytes(1 << 30)
eng.predict_windows(bases, starts, tab.length, 500, want=("prediction", "reliability"))
for rep in range(2):
    t = time.time(); p = eng.device.upload(bases); t1 = time.time() - t
    st = np.ascontiguousarray(starts, np.int64); ln = np.ascontiguousarray(tab.length, np.int32)
    t = time.time(); ps = eng.device.upload(st); pl = eng.device.upload(ln); t2 = time.time() - t
    n = len(st)
    import ctypes as C
    from jaeger_amd import _lib as L
    dp = eng.device.alloc(n * 6 * 4); dr = eng.device.alloc(n * 4); dc = eng.device.alloc(n * 16)
    t = time.time()
    eng.model.predict_windows_raw(p, bases.size, ps, pl, n, 500, eng.lut, eng.encode_flags, 165,
                                  {"prediction": dp, "reliability": dr}, counts_ptr=dc, chunk=0)
    eng.device.sync(); t3 = time.time() - t
    t = time.time(); a = eng.device.download(dp, (n, 6), np.float32); b = eng.device.download(dc, (n, 4), np.int32); t4 = time.time() - t
    print("precision", eng.model.precision, eng.model.placement(), eng.device.profile_read() if rep else "")
    eng.device.profile_enable(True)
    print(f"upload bases {t1*1e3:.0f} ms ({bases.size/t1/1e9:.1f} GB/s), window table {t2*1e3:.0f} ms, device-resident predict {t3*1e3:.0f} ms, download {t4*1e3:.0f} ms")
    for q in (p, ps, pl, dp, dr, dc): eng.device.free(q)
eng.close()
