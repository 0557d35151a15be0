#!/usr/bin/env python3
"""Golden vectors for the prophage-segmentation inputs, produced by IMPORTING the reference's
postprocess/prophages.py:logits_to_df_v2 (its plotting / segmentation / alignment dependencies, which the
function does not touch, are replaced by empty stub modules).  Build container only; outputs are data.

    python tests/golden/make_golden_prophage.py   ->  prophage_inputs.npz
"""
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference/src")
for name in ("pyfastx", "parasail", "ruptures", "matplotlib", "matplotlib.pyplot", "matplotlib.patches",
             "matplotlib.lines", "kneed", "pycirclize"):
    m = types.ModuleType(name)
    for attr in ("Patch", "Line2D", "KneeLocator", "Circos"):
        setattr(m, attr, object)
    sys.modules.setdefault(name, m)
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
from jaeger.postprocess.prophages import logits_to_df_v2  # noqa: E402

rng = np.random.Generator(np.random.PCG64(99))
classes = ["bacteria", "phage", "eukarya", "archaea", "plasmid", "virus"]
cm = {"class": classes, "index": list(range(6))}
n_win = [3, 400, 7, 11, 350]                 # windows per contig; lengths decide who passes --lc
lengths = np.array([4000, 600_123, 9000, 510_000, 499_999])
preds = [rng.normal(0, 2, (k, 6)).astype(np.float32) for k in n_win]
preds[1][120:180, 1] += 6.0                   # a phage-like island
gc_skews = [np.round(rng.normal(0, 0.1, k), 2) for k in n_win]
gcs = [rng.uniform(0.3, 0.6, k) for k in n_win]
headers = np.array([f"ctg{i}" for i in range(5)])
kw = {"lc": 500_000, "stride": 1500, "fsize": 2000}
out = logits_to_df_v2(cm, kw, headers, [p.copy() for p in preds], lengths, [g.copy() for g in gc_skews],
                      [g.copy() for g in gcs])
save = {"n_win": np.array(n_win), "lengths": lengths, "headers": headers, "kept": np.array(list(out.keys()))}
for i in range(5):
    save[f"pred_{i}"], save[f"gc_skew_{i}"], save[f"gc_{i}"] = preds[i], gc_skews[i], gcs[i]
for key, (df, host, length) in out.items():
    save[f"df_{key}"] = df.to_numpy(dtype=np.float64)
    save[f"cols_{key}"] = np.array(list(df.columns))
    save[f"host_{key}"] = np.array(host)
np.savez(HERE / "prophage_inputs.npz", **save)
print("kept", list(out.keys()))
