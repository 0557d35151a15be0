"""Cold-start split of a CLI-like process: library load, jg_create (HIP runtime + device), model creation, first launches."""
import sys, time
t00 = time.perf_counter()
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
t0 = time.perf_counter(); from jaeger_amd import _lib; lib = _lib.load(); t_lib = time.perf_counter() - t0
t0 = time.perf_counter(); from jaeger_amd.engine import HipDevice, JaegerHipEngine; t_imp = time.perf_counter() - t0
t0 = time.perf_counter(); dev = HipDevice(0); t_dev = time.perf_counter() - t0
t0 = time.perf_counter(); p = dev.alloc(1 << 20); dev.free(p); dev.sync(); t_alloc = time.perf_counter() - t0
seq = np.frombuffer(b"ACGT" * 1000, np.uint8)
t0 = time.perf_counter(); dev.encode(seq, np.zeros(1, np.int64), np.full(1, 1500, np.int32), 1500, np.zeros(65, np.uint8)); t_k1 = time.perf_counter() - t0
t0 = time.perf_counter(); dev.encode(seq, np.zeros(1, np.int64), np.full(1, 1500, np.int32), 1500, np.zeros(65, np.uint8)); t_k2 = time.perf_counter() - t0
from conftest import load_model_cfg
from jaeger_amd.plan import build_plan
from jaeger_amd.weights import random_weights
cfg = load_model_cfg("brain"); w = random_weights(build_plan(cfg))
t0 = time.perf_counter(); eng = JaegerHipEngine(model_cfg=cfg, weights=w, device_id=0); t_eng = time.perf_counter() - t0
t0 = time.perf_counter(); eng.predict_windows(np.tile(seq, 4), np.arange(8, dtype=np.int64) * 1500, np.full(8, 1500, np.int32), 1500); t_f1 = time.perf_counter() - t0
t0 = time.perf_counter(); eng.predict_windows(np.tile(seq, 4), np.arange(8, dtype=np.int64) * 1500, np.full(8, 1500, np.int32), 1500); t_f2 = time.perf_counter() - t0
t0 = time.perf_counter(); import pandas; t_pd = time.perf_counter() - t0
print(f"numpy+path {t0 - t00 - 0:.2f}?  lib load {t_lib:.3f}  engine imports {t_imp:.3f}  jg_create {t_dev:.3f}  first alloc {t_alloc:.3f}  "
      f"first kernel {t_k1:.3f} (second {t_k2:.4f})  second engine + brain model {t_eng:.3f}  first forward {t_f1:.3f} (second {t_f2:.4f})  "
      f"import pandas {t_pd:.3f}")
