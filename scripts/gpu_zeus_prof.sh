cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/zeus
export TMPDIR=/tmp
cat > /tmp/zeus_run.py <<'PY'
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
lengths, bases = synth_contigs(rng, 600)
off = np.zeros(lengths.size + 1, np.int64); np.cumsum(lengths, out=off[1:])
cfg = load_model_cfg("zeus")
eng = JaegerHipEngine(model_cfg=cfg, weights=random_weights(build_plan(cfg), 38341))
tab = build_window_table(lengths, 1500, 1500)
starts = off[tab.contig] + tab.start
t = time.time(); out = eng.predict_windows(bases, starts, tab.length, 1500, want=("prediction",)); dt = time.time() - t
print(f"{len(tab)} windows {len(tab)*1500/dt/1e6:.1f} Mbp/s", eng.model.precision)
eng.close()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zeus/prof -o z -- python3 /tmp/zeus_run.py 2>&1 | grep "Mbp/s"
f=$(find gpurun_out/zeus/prof -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
rm -rf gpurun_out/zeus/prof
