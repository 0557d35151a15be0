"""Hand-built Keras-3 ``.weights.h5`` fixtures in BOTH layouts the reference's converter documents
(scripts/convert_legacy_classifier_checkpoint.py:29-32,76-175), written with the HDF5 library's own ``h5import`` tool
(build container only: /opt/conda/bin/h5import) from seeded weights of ``stacks2_project.yaml`` - a two-stack model small
enough to commit (second stack with a 1x1 bypass: conv3 / bn3):

* ``stacks2_keras3_flat.weights.h5``   - the current generation: custom ``ResidualBlockStack`` layers, Keras-3 container
  naming: ``layers/residual_block_stack[_N]/blocks/residual_block[_k]/{conv1,conv2,conv3,bn1,bn2,bn3}/vars/<i>`` (blocks
  numbered per container), stem and head layers under their global auto names.
* ``stacks2_keras3_nested.weights.h5`` - the older generation: Functional sub-models: ``layers/functional[_N]/layers/
  residual_block[_k]/{conv1,...}/vars/<i>`` with the blocks' auto names counting on GLOBALLY across the sub-models
  (``residual_block_2`` is the first block of ``functional_1``) and the head's Dense layers under ``layers/functional_8/layers/
  dense[_k]``.

    python tests/golden/make_golden_h5.py
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np
import yaml

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))

H5IMPORT = "/opt/conda/bin/h5import"


def groups_for(plan, layout: str) -> dict[str, str]:
    """canonical layer prefix -> HDF5 group of the layer, for ``flat`` or ``nested``."""
    from jaeger_amd.weights import _layer_order
    out = {}
    counters: dict[str, int] = {}

    def auto(name):                    # Keras' global per-class auto names: name, name_1, name_2 ...
        n = counters.get(name, 0)
        counters[name] = n + 1
        return name if n == 0 else f"{name}_{n}"
    stacks: list[str] = []
    blocks_seen: dict[str, list[str]] = {}
    global_block = 0
    for prefix, leaves in _layer_order(plan):
        parts = prefix.split("/")
        if len(parts) == 4 and parts[2].startswith("block"):       # rep/<i>/block<j>/<sub>
            stack, blk, sub = "/".join(parts[:2]), parts[2], parts[3]
            if stack not in stacks:
                stacks.append(stack)
                blocks_seen[stack] = []
            if blk not in blocks_seen[stack]:
                blocks_seen[stack].append(blk)
                if layout == "nested":
                    blocks_seen[stack + "#" + blk] = [f"residual_block{'' if global_block == 0 else '_' + str(global_block)}"]
                    global_block += 1
            s_idx, b_idx = stacks.index(stack), blocks_seen[stack].index(blk)
            if layout == "flat":
                sname = f"residual_block_stack{'' if s_idx == 0 else '_' + str(s_idx)}"
                out[prefix] = f"layers/{sname}/blocks/residual_block{'' if b_idx == 0 else '_' + str(b_idx)}/{sub}"
            else:
                fname = f"functional{'' if s_idx == 0 else '_' + str(s_idx)}"
                out[prefix] = f"layers/{fname}/layers/{blocks_seen[stack + '#' + blk][0]}/{sub}"
        elif prefix == "embedding":
            out[prefix] = "layers/" + auto("embedding")
        elif "kernel" in leaves and parts[0] == "rep":
            out[prefix] = "layers/" + auto("masked_conv1d")
        elif "moving_variance" in leaves:
            out[prefix] = "layers/" + auto("masked_batch_norm")
        elif parts[0] in ("classifier", "reliability"):
            name = auto("dense")
            out[prefix] = f"layers/functional_8/layers/{name}" if layout == "nested" else f"layers/{name}"
        else:
            raise ValueError(prefix)
    return out


def write_h5(path: Path, datasets: dict[str, np.ndarray]):
    path.unlink(missing_ok=True)
    with tempfile.TemporaryDirectory() as td:
        for key, arr in datasets.items():
            txt, cfg = Path(td) / "t.txt", Path(td) / "t.cfg"
            np.savetxt(txt, arr.reshape(-1, 1), fmt="%.9g")
            cfg.write_text(f"PATH {key}\nINPUT-CLASS TEXTFP\nRANK {arr.ndim}\n"
                           f"DIMENSION-SIZES {' '.join(map(str, arr.shape))}\nOUTPUT-CLASS FP\nOUTPUT-SIZE 32\n")
            subprocess.run([H5IMPORT, str(txt), "-c", str(cfg), "-o", str(path)], check=True, capture_output=True)


def main():
    from jaeger_amd.plan import build_plan
    from jaeger_amd.weights import _layer_order, random_weights
    plan = build_plan(yaml.safe_load((HERE / "stacks2_project.yaml").read_text())["model"])
    w = random_weights(plan, seed=2026)
    for layout in ("flat", "nested"):
        groups = groups_for(plan, layout)
        data = {}
        for prefix, leaves in _layer_order(plan):
            for i, leaf in enumerate(leaves):
                data[f"{groups[prefix]}/vars/{i}"] = w[f"{prefix}/{leaf}"]
        write_h5(HERE / f"stacks2_keras3_{layout}.weights.h5", data)
        print(layout, len(data), "datasets;", sorted({g for g in groups.values()})[:4], "...")


if __name__ == "__main__":
    main()
