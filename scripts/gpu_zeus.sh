cd $GRAFT_REPO_ROOT
for lib in libjaeger_hip.so libjaeger_hip_prev.so; do
JAEGER_HIP_LIB=$GRAFT_REPO_ROOT/jaeger_amd/$lib python - <<'PY'
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
lengths, bases = synth_contigs(rng, 1500)
off = np.zeros(lengths.size + 1, np.int64); np.cumsum(lengths, out=off[1:])
cfg = load_model_cfg("zeus")
for seed in (1, 38341):
    eng = JaegerHipEngine(model_cfg=cfg, weights=random_weights(build_plan(cfg), seed))
    tab = build_window_table(lengths, 1500, 1500)
    starts = off[tab.contig] + tab.start
    eng.predict_windows(bases, starts[:3000], tab.length[:3000], 1500, want=("prediction",))
    eng.device.profile_enable(True); eng.device.profile_read()
    t = time.time(); out = eng.predict_windows(bases, starts, tab.length, 1500, want=("prediction",)); dt = time.time() - t
    pr = eng.device.profile_read()
    print(os.environ["JAEGER_HIP_LIB"].split("/")[-1], "seed", seed, f"{len(tab)*1500/dt/1e6:.1f} Mbp/s", eng.model.precision,
          {k: (round(v["ms"]), v["launches"]) for k, v in pr.items() if isinstance(v, dict)}, "max|logit|", float(np.abs(out["prediction"]).max()))
    eng.close()
PY
done
