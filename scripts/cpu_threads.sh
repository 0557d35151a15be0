# host-side: how the CPU baseline (oracle forward, batch 96) scales with torch threads on this box
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; nproc
python - <<'PY'
import sys, time, numpy as np, torch, yaml
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_model_cfg
from oracle import forward as ofwd
cfg = load_model_cfg("brain")
w = ofwd.random_weights(cfg, seed=38341)
ids = np.random.default_rng(0).integers(1, 65, (96, 6, 498)).astype(np.uint8)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    ofwd.forward(cfg, w, ids[:8])
    t = time.time(); ofwd.forward(cfg, w, ids); dt = time.time() - t
    print(f"threads {nt}: 96 windows in {dt:.2f} s = {96*1500/dt/1e6:.4f} Mbp/s")
PY
