cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d; exec > gpurun_out/r5d/run.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "small or baseline500 or narrow_residual or pyramid" 2>&1 | tail -5
JAEGER_FUZZ_SEEDS=200 timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from conftest import load_model_cfg
from jaeger_amd.engine import JaegerHipEngine
from oracle import forward as ofwd
cfg = load_model_cfg("baseline500")
eng = JaegerHipEngine(model_cfg=cfg, weights=ofwd.random_weights(cfg, seed=1), precision="f16x3")
print(eng.model.describe())
PY
