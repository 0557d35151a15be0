cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, warnings, collections
import numpy as np
warnings.simplefilter("ignore")
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_fuzz import random_model
from jaeger_amd.engine import JaegerHipEngine
from oracle import forward as ofwd
stat = collections.Counter(); convs = f16 = 0; why = collections.Counter(); pats = collections.Counter()
for seed in range(400):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    cfg = random_model(rng)
    try:
        eng = JaegerHipEngine(model_cfg=cfg, weights=ofwd.random_weights(cfg, seed=seed))
    except Exception as e:
        stat["refused"] += 1; continue
    pl = eng.model.placement(); prec = eng.model.precision
    for line in eng.model.describe().splitlines():
        if "exact-f32" in line:
            why[line.split("exact-f32")[1].lstrip(": ") or "(model in f32 mode)"] += 1
            if "pattern" in line:
                pats[("narrow " if "other than 128" in line else "") + line[line.find("[") + 1:line.find("]")]] += 1
    eng.close()
    convs += pl["convs"]; f16 += pl["convs_f16x3"] if prec == "f16x3" else 0
    if pl["small_fused"]: stat["fused small"] += 1
    elif prec == "f32": stat["all f32"] += 1
    elif pl["convs_f16x3"] == pl["convs"]: stat["all split-f16"] += 1
    else: stat["mixed"] += 1
print(dict(stat), f"convs on split-f16: {f16}/{convs}")
for k, v in why.most_common(12): print(f"  {v:5d}  {k}")
for k, v in pats.most_common(25): print(f"  {v:5d}  [{k}]")
PY
