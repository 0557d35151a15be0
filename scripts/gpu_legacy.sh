python -m pytest tests/test_gpu_legacy.py -x -q -s 2>&1 | grep -E "output|embedding|passed|failed|Error|error" | head -20
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from jaeger_amd import legacy
from bench import synth_contigs
from jaeger_amd.fragment import build_window_table
w = legacy.load_legacy_h5('tests/golden/legacy_data/models/default/WRes_1024.h5')
rng = np.random.Generator(np.random.PCG64(1))
lengths, bases = synth_contigs(rng, 1500)
off = np.zeros(lengths.size + 1, np.int64); np.cumsum(lengths, out=off[1:])
tab = build_window_table(lengths, 2000, 1500)
starts = off[tab.contig] + tab.start
for prec in ("f32", "f16x3"):
    eng = legacy.LegacyHipEngine(w, precision=prec)
    eng.predict_windows(bases, starts[:2000], tab.length[:2000], 2000)
    t = time.time(); out = eng.predict_windows(bases, starts, tab.length, 2000); dt = time.time() - t
    print(prec, len(tab), "windows", round(dt, 2), "s", round(len(tab) * 2000 / dt / 1e6, 1), "Mbp/s (2000-bp windows, host buffers)")
    eng.close()
PY
