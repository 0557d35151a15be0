"""Oracle: a TensorFlow-free interpreter for the serving function of a SavedModel (TEST INFRASTRUCTURE).

The reference's ``InferModel`` runs ``tf.saved_model.load(graph).signatures["serving_default"]``
(``nnlib/inference.py:307-325``).  TensorFlow cannot be installed here, but a SavedModel is data: a GraphDef function
plus a variable bundle.  This module executes that function op by op in numpy (convolution and max-pool through
torch-CPU), following the published TensorFlow op definitions, so that the numbers it produces are *the reference's
own graph and weights evaluated*, not a restatement of its Python layer sources.  It is what pins the floating-point
forward of the legacy tower (``tests/golden/make_golden_savedmodel.py`` -> ``legacy_savedmodel_logits.npz``) and, through
the shared op semantics (SAME-padding split, dilation via SpaceToBatchND, BiasAdd, batch norm as rsqrt / mul / sub,
erfc-GELU, MaxPool, GatherV2, MatMul), the conv arithmetic of ``oracle/forward.py``.

Ops implemented = the 29 kinds in the bundled ``jaeger_fragment_graph`` (``src/jaeger/data/models/test``); anything
else raises.  Readers (protobuf walk, tensor bundle) come from ``jaeger_amd/savedmodel_lite.py``.
"""

from __future__ import annotations

import sys

import numpy as np

from jaeger_amd.savedmodel_lite import Function, SavedModel, read_bundle

_TF_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 9: np.int64, 10: np.bool_}


def _strided_slice(x, begin, end, strides, node):
    bm, em = node.attr_i("begin_mask", 0), node.attr_i("end_mask", 0)
    elm, nam, sam = node.attr_i("ellipsis_mask", 0), node.attr_i("new_axis_mask", 0), node.attr_i("shrink_axis_mask", 0)
    if elm:
        raise NotImplementedError("StridedSlice with an ellipsis mask")
    idx = []
    for i in range(len(begin)):
        if nam & (1 << i):
            idx.append(np.newaxis)
        elif sam & (1 << i):
            idx.append(int(begin[i]))
        else:
            b = None if bm & (1 << i) else int(begin[i])
            e = None if em & (1 << i) else int(end[i])
            idx.append(slice(b, e, int(strides[i])))
    return x[tuple(idx)]


def _space_to_batch(x, block, paddings):
    m = len(block)
    pad = [(0, 0)] + [tuple(int(v) for v in p) for p in paddings] + [(0, 0)] * (x.ndim - 1 - m)
    x = np.pad(x, pad)
    batch, rest = x.shape[0], x.shape[1 + m:]
    shape = [batch]
    for i in range(m):
        shape += [x.shape[1 + i] // int(block[i]), int(block[i])]
    x = x.reshape(shape + list(rest))
    perm = [2 + 2 * i for i in range(m)] + [0] + [1 + 2 * i for i in range(m)] + list(range(1 + 2 * m, x.ndim))
    x = x.transpose(perm)
    out = [batch * int(np.prod(block))] + [shape[1 + 2 * i] for i in range(m)] + list(rest)
    return x.reshape(out)


def _batch_to_space(x, block, crops):
    m = len(block)
    prod = int(np.prod(block))
    batch = x.shape[0] // prod
    rest = x.shape[1 + m:]
    x = x.reshape([int(b) for b in block] + [batch] + list(x.shape[1:]))
    perm = [m]
    for i in range(m):
        perm += [m + 1 + i, i]
    perm += list(range(2 * m + 1, x.ndim))
    x = x.transpose(perm)
    x = x.reshape([batch] + [x.shape[1 + 2 * i] * x.shape[2 + 2 * i] for i in range(m)] + list(rest))
    idx = [slice(None)] + [slice(int(c[0]), x.shape[1 + i] - int(c[1])) for i, c in enumerate(crops)]
    return x[tuple(idx)]


def _conv2d(x, w, node):
    import torch
    import torch.nn.functional as F
    if node.attr_s("data_format", "NHWC") != "NHWC":
        raise NotImplementedError("Conv2D data_format")
    strides = node.attr_ints("strides") or [1, 1, 1, 1]
    dil = node.attr_ints("dilations") or [1, 1, 1, 1]
    kh, kw = w.shape[0], w.shape[1]
    if node.attr_s("padding") == "SAME":
        pads = []
        for size, k, s, d in ((x.shape[1], kh, strides[1], dil[1]), (x.shape[2], kw, strides[2], dil[2])):
            out = -(-size // s)
            total = max((out - 1) * s + (k - 1) * d + 1 - size, 0)
            pads.append((total // 2, total - total // 2))       # TF: the smaller half goes in front
        x = np.pad(x, [(0, 0), pads[0], pads[1], (0, 0)])
    elif node.attr_s("padding") != "VALID":
        raise NotImplementedError("Conv2D explicit padding")
    xt = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))
    wt = torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))
    y = F.conv2d(xt, wt, stride=(strides[1], strides[2]), dilation=(dil[1], dil[2]))
    return np.ascontiguousarray(y.numpy().transpose(0, 2, 3, 1))


def _maxpool(x, node):
    import torch
    import torch.nn.functional as F
    ks, st = node.attr_ints("ksize"), node.attr_ints("strides")
    if node.attr_s("padding") != "VALID":
        raise NotImplementedError("MaxPool padding")
    xt = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))
    y = F.max_pool2d(xt, kernel_size=(ks[1], ks[2]), stride=(st[1], st[2]))
    return np.ascontiguousarray(y.numpy().transpose(0, 2, 3, 1))


def run_function(fn: Function, feeds: dict[str, np.ndarray], captured: dict[str, np.ndarray],
                 float_dtype=np.float32) -> dict[str, np.ndarray]:
    """Evaluate ``fn``'s outputs for the given explicit inputs; float tensors are computed in ``float_dtype``."""
    from scipy.special import erfc
    fd = np.dtype(float_dtype)
    nodes = {n.name: n for n in fn.nodes}
    args = {}
    for name, dt in fn.inputs:
        if name in feeds:
            v = np.asarray(feeds[name])
            args[name] = v.astype(fd) if v.dtype.kind == "f" else v
        elif name in captured:
            v = captured[name]
            args[name] = v.astype(fd) if v.dtype.kind == "f" else v
    memo: dict[str, list] = {}
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 50000))

    def tensor(ref: str):
        parts = ref.split(":")
        name = parts[0]
        if name in args and len(parts) == 1:
            return args[name]
        idx = int(parts[-1]) if len(parts) > 1 else 0
        if name not in memo:
            memo[name] = evaluate(nodes[name])
        return memo[name][idx]

    def cast_float(a):
        return a.astype(fd) if a.dtype.kind == "f" and a.dtype != fd else a

    def evaluate(n):
        ins = [tensor(r) for r in n.inputs if not r.startswith("^")]
        op = n.op
        if op == "Const":
            return [cast_float(n.attr_tensor("value"))]
        if op in ("Identity", "ReadVariableOp", "StopGradient"):
            return [ins[0]]
        if op == "NoOp":
            return [None]
        if op == "Cast":
            dst = _TF_DTYPES[n.attr_type("DstT")]
            return [ins[0].astype(fd if np.dtype(dst).kind == "f" else dst)]
        if op == "Shape":
            return [np.asarray(ins[0].shape, _TF_DTYPES[n.attr_type("out_type", 3)])]
        if op == "StridedSlice":
            return [np.asarray(_strided_slice(ins[0], ins[1], ins[2], ins[3], n))]
        if op == "Pack":
            return [np.stack(ins, axis=n.attr_i("axis", 0))]
        if op == "ExpandDims":
            return [np.expand_dims(ins[0], int(ins[1]))]
        if op == "Squeeze":
            dims = n.attr_ints("squeeze_dims")
            return [np.squeeze(ins[0], axis=tuple(dims) if dims else None)]
        if op == "Reshape":
            return [ins[0].reshape([int(v) for v in ins[1]])]
        if op == "AddV2":
            return [ins[0] + ins[1]]
        if op == "Sub":
            return [ins[0] - ins[1]]
        if op == "Mul":
            return [ins[0] * ins[1]]
        if op == "Neg":
            return [-ins[0]]
        if op == "FloorMod":
            return [np.mod(ins[0], ins[1])]
        if op == "Rsqrt":
            return [(1.0 / np.sqrt(ins[0])).astype(ins[0].dtype)]
        if op == "Erfc":
            return [erfc(ins[0]).astype(ins[0].dtype)]
        if op == "Less":
            return [ins[0] < ins[1]]
        if op == "NotEqual":
            return [ins[0] != ins[1]]
        if op == "SelectV2":
            return [np.where(ins[0], ins[1], ins[2])]
        if op == "GatherV2":
            if n.attr_i("batch_dims", 0) != 0:
                raise NotImplementedError("GatherV2 batch_dims")
            return [np.take(ins[0], ins[1].astype(np.int64), axis=int(ins[2]))]
        if op == "BiasAdd":
            if n.attr_s("data_format", "NHWC") != "NHWC":
                raise NotImplementedError("BiasAdd data_format")
            return [ins[0] + ins[1]]
        if op == "MatMul":
            a = ins[0].T if n.attr_b("transpose_a") else ins[0]
            b = ins[1].T if n.attr_b("transpose_b") else ins[1]
            return [a @ b]
        if op == "Max":
            return [np.max(ins[0], axis=tuple(int(v) for v in np.ravel(ins[1])), keepdims=n.attr_b("keep_dims"))]
        if op == "SpaceToBatchND":
            return [_space_to_batch(ins[0], np.ravel(ins[1]), ins[2])]
        if op == "BatchToSpaceND":
            return [_batch_to_space(ins[0], np.ravel(ins[1]), ins[2])]
        if op == "Conv2D":
            return [_conv2d(ins[0], ins[1], n)]
        if op == "MaxPool":
            return [_maxpool(ins[0], n)]
        raise NotImplementedError(f"graphdef oracle: op {op!r} (node {n.name}) is not implemented")

    return {name: tensor(ref) for name, ref in fn.ret.items()}


def run_saved_model(graph_dir, feeds: dict[str, np.ndarray], float_dtype=np.float32) -> dict[str, np.ndarray]:
    """Outputs of the SavedModel's serving function for explicit ``feeds`` (argument name -> array)."""
    sm = SavedModel(graph_dir)
    fn = sm.serving_function()
    bundle = read_bundle(sm.dir / "variables")
    return run_function(fn, feeds, sm.captured_variables(fn, bundle), float_dtype)
