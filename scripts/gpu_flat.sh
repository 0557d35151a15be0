python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
# ablation switches live in the experiment build only (make -C jaeger_amd/csrc exp)
export JAEGER_HIP_LIB=${JAEGER_HIP_LIB:-${GRAFT_REPO_ROOT:-.}/jaeger_amd/libjaeger_hip_exp.so}
python - <<'PY'
import sys, time, numpy as np, warnings, os
warnings.simplefilter("ignore")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_model_cfg
from bench import synth_contigs
from jaeger_amd.engine import JaegerHipEngine
from jaeger_amd.fragment import build_window_table
from jaeger_amd.plan import build_plan
from jaeger_amd.weights import random_weights
rng = np.random.Generator(np.random.PCG64(1))
lengths, bases = synth_contigs(rng, 3000)
off = np.zeros(lengths.size + 1, np.int64); np.cumsum(lengths, out=off[1:])
cfg = load_model_cfg("brain")
eng = JaegerHipEngine(model_cfg=cfg, weights=random_weights(build_plan(cfg), 1))
for fsize, stride in ((2000, 2000), (1500, 1500), (2500, 2500), (1000, 1000)):
    tab = build_window_table(lengths, fsize, stride)
    starts = off[tab.contig] + tab.start
    eng.predict_windows(bases, starts[:3000], tab.length[:3000], fsize, want=("prediction",))
    t = time.time(); out = eng.predict_windows(bases, starts, tab.length, fsize, want=("prediction", "reliability")); dt = time.time() - t
    print(f"flat={'off' if os.environ.get('JG_NO_FLAT') else 'on '} fsize {fsize}: {len(tab)*fsize/dt/1e6:.1f} Mbp/s  checksum {float(np.abs(out['prediction']).sum()):.4f}")
eng.close()
PY
