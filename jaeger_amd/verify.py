"""``jaeger_amd verify-model``: does the SavedModel a reference user would run agree with the layer plan the MI355X
engine compiles from ``project.yaml`` (or, for the legacy ``default`` tower, from the fixed v1 architecture)?

``InferModel`` executes ``<name>_graph/`` (``nnlib/inference.py:307-325``); the engine never reads it.  A graph
exported by an older code version can differ from today's builder (mask propagation, GELU form,
batch-norm epsilon), so before parity is claimed for a supplied model this check compares, without TensorFlow:

* the float variables of the bundle with the plan's weight shapes (as multisets - Keras names are not canonical),
* the convolutions of the serving function (count per frame-folded conv, dilations via SpaceToBatchND blocks,
  SAME / VALID) with the plan's,
* the batch-norm epsilons, the GELU form (erf / tanh), the max-pool and mask ops.

It returns a list of findings; an empty list means "the graph computes what the plan says".  The health check of
the reference (``commands/health.py:216-250``) loads the same artefacts through TensorFlow.
"""

from __future__ import annotations

import ast
from collections import Counter
from pathlib import Path

from . import savedmodel_lite as S
from .plan import Conv, ModelPlan, Norm, ResBlock, weight_shapes


def _plan_expectations(plan: ModelPlan) -> dict:
    convs, eps, acts = [], Counter(), Counter()
    for layer in plan.rep:
        if isinstance(layer, Conv):
            convs.append(layer)
        elif isinstance(layer, ResBlock):
            convs += [c for c in (layer.conv1, layer.conv2, layer.conv3) if c is not None]
            for n in (layer.bn1, layer.bn2, layer.bn3):
                if n is not None and n.kind == "masked_batchnorm":
                    eps[round(float(n.epsilon), 9)] += 1
        elif isinstance(layer, Norm) and layer.kind == "masked_batchnorm":
            eps[round(float(layer.epsilon), 9)] += 1
    return {"convs": convs, "bn_eps": eps, "masking": plan.use_masking,
            "shapes": Counter(tuple(s) for s in weight_shapes(plan).values())}


def _legacy_expectations() -> dict:
    from . import legacy
    convs = [Conv(c, k, 0, legacy.WIDTH, 1, "same", d) for c, _, k, d, _, _ in legacy.tower_layers()]
    return {"convs": convs, "bn_eps": Counter({round(legacy.BN_EPS, 9): len(convs)}), "masking": False,
            "shapes": Counter(tuple(s) for s in legacy.weight_shapes().values()), "frames_unrolled": 6,
            "gelu": "erf", "maxpool": 2}


def verify_model(graph_dir, plan: ModelPlan | None = None, legacy: bool = False) -> list[str]:
    """Findings (strings) where the SavedModel's census disagrees with the plan; [] = consistent."""
    c = S.census(graph_dir)
    exp = _legacy_expectations() if legacy else _plan_expectations(plan)
    out: list[str] = []
    # 1. variables
    have = Counter(tuple(s) for _, s in c["variables"])
    if have != exp["shapes"]:
        missing = exp["shapes"] - have
        extra = have - exp["shapes"]
        out.append(f"variable shapes differ: plan expects {dict(missing)} that the bundle lacks; bundle holds "
                   f"{dict(extra)} the plan does not use")
    # 2. convolutions: the graph runs one Conv2D per conv (frames folded into the batch) or one per frame
    unroll = exp.get("frames_unrolled", 1)
    n_conv = sum(c["conv2d"].values())
    if n_conv not in (len(exp["convs"]) * unroll, len(exp["convs"])):
        out.append(f"{n_conv} Conv2D nodes in the serving function, plan has {len(exp['convs'])} convolutions"
                   + (f" x {unroll} frames" if unroll > 1 else ""))
    per = n_conv // max(len(exp["convs"]), 1) or 1
    want_dil = Counter(int(cv.dilation_rate) for cv in exp["convs"] if int(cv.dilation_rate) > 1)
    have_dil = Counter()
    for blk, n in c["space_to_batch_blocks"].items():
        have_dil[int(blk.strip("(),"))] += n // per
    # dilated convs appear either as SpaceToBatchND blocks or as Conv2D dilations attributes
    for key, n in c["conv2d"].items():
        dil = ast.literal_eval(key)[2]                       # a tuple literal produced by census()
        if dil and max(dil) > 1:
            have_dil[max(dil)] += n // per
    if have_dil != want_dil:
        out.append(f"dilations differ: graph {dict(have_dil)}, plan {dict(want_dil)}")
    # 3. batch norm epsilon
    have_eps = Counter({round(float(k), 9): v // per for k, v in c["batchnorm_eps"].items()})
    if have_eps != exp["bn_eps"]:
        out.append(f"batch-norm epsilons differ: graph {dict(have_eps)}, plan {dict(exp['bn_eps'])}")
    # 4. GELU form
    want_gelu = exp.get("gelu", "tanh")
    if c["gelu_form"] != want_gelu and c["n_gelu"]:
        out.append(f"GELU form: graph computes the {c['gelu_form']} form, the plan's kernels the {want_gelu} form")
    # 5. masking: a mask-propagating graph compares the conv of the mask (Greater / GreaterEqual / Equal)
    mask_cmp = sum(c["mask_ops"][k] for k in ("Greater", "GreaterEqual", "Equal"))
    if exp["masking"] and mask_cmp == 0:
        out.append("the plan propagates masks through the convolutions (use_masking: true) but the graph holds no mask "
                   "comparison ops: it was probably exported before mask propagation (run with use_masking: false)")
    if not exp["masking"] and mask_cmp > 0:
        out.append("the graph propagates masks but the plan runs mask-free")
    if "maxpool" in exp and not c["maxpool"]:
        out.append("plan has MaxPool layers, graph has none")
    return out


def report(graph_dir, plan: ModelPlan | None = None, legacy: bool = False) -> str:
    c = S.census(graph_dir)
    findings = verify_model(graph_dir, plan, legacy)
    lines = [f"SavedModel {Path(graph_dir)}", f"  serving function {c['function']}: {c['n_nodes']} nodes, inputs "
             f"{c['inputs']}, outputs {c['outputs']}", f"  {c['n_parameters']} parameters in {len(c['variables'])} "
             f"float variables, {sum(c['conv2d'].values())} Conv2D, dilations {c['space_to_batch_blocks']}, batch-norm "
             f"eps {c['batchnorm_eps']}, GELU {c['gelu_form']} x{c['n_gelu']}, MaxPool {c['maxpool']}"]
    lines += ["  OK: the graph agrees with the compiled plan"] if not findings else [f"  MISMATCH: {f}" for f in findings]
    return "\n".join(lines)
