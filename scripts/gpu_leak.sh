cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, warnings, numpy as np, torch
warnings.simplefilter("ignore")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_model_cfg
from jaeger_amd.engine import JaegerHipEngine
from oracle import forward as ofwd
A = np.frombuffer(b"ACGT", np.uint8)
rng = np.random.Generator(np.random.PCG64(1))
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
torch.zeros(1, device="cuda")
f0 = free()
for rep in range(3):
    for name, fsize in (("brain", 1500), ("baseline500", 500), ("pyramid", 2000), ("zeus", 1500)):
        cfg = load_model_cfg(name)
        eng = JaegerHipEngine(model_cfg=cfg, weights=ofwd.random_weights(cfg, seed=1), device_id=0)
        for it in range(40):
            n = int(rng.integers(1, 600))
            bases = A[rng.integers(0, 4, fsize * n, dtype=np.uint8)]
            starts = (np.arange(n) * fsize).astype(np.int64); lens = np.full(n, fsize, np.int32)
            offs = np.append(starts, fsize * n)
            eng.predict_windows(bases, starts, lens, fsize, want=("prediction",), dust_records=offs if it % 2 else None)
            if it % 10 == 0:
                eng.device.set_stream_bytes(4096 if it % 20 == 0 else 1 << 30)
        eng.close()
    print(f"round {rep}: free device memory {free():.0f} MiB (start {f0:.0f})")
PY
