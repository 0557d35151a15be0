import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, yaml, cProfile, pstats, io
t=time.time(); import torch; print("import torch", time.time()-t)
from conftest import make_model_dir
from pathlib import Path
from jaeger_amd.engine import JaegerHipEngine
from jaeger_amd.predict import AvailableModels
tmp=Path("/dev/shm/probe"); tmp.mkdir(exist_ok=True)
make_model_dir(tmp/"m")
info=AvailableModels(path=str(tmp/"m")).info
name=next(iter(info)); mi=info[name]
from jaeger_amd.weights import load_npz
for rep in range(3):
    t0=time.time(); w=load_npz(mi["weights_npz"]); t1=time.time()
    pr=cProfile.Profile(); pr.enable()
    e=JaegerHipEngine(mi, weights=w, device_id=0); 
    pr.disable(); t2=time.time()
    print(f"rep {rep}: load_npz {t1-t0:.3f} engine {t2-t1:.3f}")
    if rep==1:
        s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(12); print(s.getvalue()[:3000])
    if rep<2: e.close()
# keep e alive with a big workspace, create another
fs=1500
seq=np.frombuffer(b"ACGT",np.uint8)[np.random.default_rng(0).integers(0,4,fs*4096)]
st=(np.arange(4096)*fs).astype(np.int64); ln=np.full(4096,fs,np.int32)
e.predict_windows(seq, st, ln, fs, want=("prediction",))
for rep in range(2):
    t1=time.time(); e2=JaegerHipEngine(mi, weights=w, device_id=0); t2=time.time()
    print(f"with a live engine holding its workspace: engine {t2-t1:.3f}")
    e2.close()
