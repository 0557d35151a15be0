"""TensorFlow-free readers for a SavedModel directory (``<name>_graph/``).

``InferModel`` executes the *SavedModel graph* (``nnlib/inference.py:307-325``), not the ``project.yaml`` the MI355X
engine compiles its layer plan from.  A graph exported by an older code version can differ from what today's builder
would assemble from the same YAML (source-of-truth warning), so ``jaeger_amd verify-model`` reads
the graph and the variable bundle *without TensorFlow* and compares their census with the compiled plan:

* :func:`read_bundle` - ``variables/variables.index`` is a LevelDB-style SSTable (prefix-compressed blocks, 48-byte
  footer); its values are ``BundleEntryProto`` records pointing into ``variables.data-00000-of-00001``.
* :class:`SavedModel` - a schema-less protobuf walk over ``saved_model.pb``: the function library, the serving
  function's nodes / attributes, and the object graph that ties the function's captured resource arguments to
  checkpoint keys.
* :func:`census` - what the serving function computes, in the terms the plan is written in (convolutions with
  kernel size / dilation / padding, batch-norm epsilons, GELU form, pools, dense layers, mask ops).

Nothing here imports TensorFlow; nothing here is on the compute path.
"""

from __future__ import annotations

import struct
from collections import Counter
from pathlib import Path

import numpy as np


# ---- protobuf wire format ------------------------------------------------------------------------------------
def _varint(b: bytes, i: int) -> tuple[int, int]:
    r = s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return r, i


def pb_fields(b: bytes):
    """Yield ``(field_number, wire_type, value)``; value is an int (varint) or bytes (fixed / length-delimited)."""
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v, i = b[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v, i = b[i:i + ln], i + ln
        elif wt == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f"protobuf wire type {wt} is not supported")
        yield f, wt, v


def pb_all(b: bytes, field: int) -> list:
    return [v for f, _, v in pb_fields(b) if f == field]


def pb_one(b: bytes, field: int, default=None):
    for f, _, v in pb_fields(b):
        if f == field:
            return v
    return default


def _packed_varints(raw) -> list[int]:
    if isinstance(raw, int):
        return [raw]
    out, i = [], 0
    while i < len(raw):
        v, i = _varint(raw, i)
        out.append(v)
    return out


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


# ---- tensor bundle (variables.index + variables.data-*) ------------------------------------------------------------
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 6: np.int8, 9: np.int64, 10: np.bool_,
           19: np.float16}


def _sstable_block(data: bytes, offset: int, size: int) -> list[tuple[bytes, bytes]]:
    block = data[offset:offset + size]
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    out, i, key = [], 0, b""
    while i < end:
        shared, i = _varint(block, i)
        non_shared, i = _varint(block, i)
        vlen, i = _varint(block, i)
        key = key[:shared] + block[i:i + non_shared]
        i += non_shared
        out.append((key, block[i:i + vlen]))
        i += vlen
    return out


def read_bundle_index(index_path) -> dict[str, dict]:
    """Checkpoint key -> {dtype, shape, shard, offset, size} from a ``variables.index`` SSTable."""
    data = Path(index_path).read_bytes()
    footer = data[-48:]
    if footer[-8:] != bytes.fromhex("57fb808b247547db"):
        raise ValueError(f"{index_path}: not an SSTable (bad magic)")
    _, i = _varint(footer, 0)          # metaindex handle
    _, i = _varint(footer, i)
    idx_off, i = _varint(footer, i)
    idx_size, i = _varint(footer, i)
    entries = {}
    for _, handle in _sstable_block(data, idx_off, idx_size):
        off, j = _varint(handle, 0)
        size, j = _varint(handle, j)
        for key, val in _sstable_block(data, off, size):
            if not key:
                continue               # BundleHeaderProto
            shape = [_signed(pb_one(d, 1, 0)) for d in pb_all(pb_one(val, 2, b""), 2)]
            entries[key.decode()] = {"dtype": pb_one(val, 1, 0), "shape": shape, "shard": pb_one(val, 3, 0),
                                     "offset": pb_one(val, 4, 0), "size": pb_one(val, 5, 0)}
    return entries


def read_bundle(variables_dir) -> dict[str, np.ndarray]:
    """All tensors of ``<dir>/variables.index`` + ``variables.data-00000-of-00001`` by checkpoint key."""
    d = Path(variables_dir)
    index = read_bundle_index(d / "variables.index")
    shards = sorted(d.glob("variables.data-*"))
    blobs = {i: p.read_bytes() for i, p in enumerate(shards)}
    out = {}
    for key, e in index.items():
        if e["dtype"] not in _DTYPES:
            continue                   # strings (the object graph) and the like
        dt = np.dtype(_DTYPES[e["dtype"]])
        raw = blobs[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        out[key] = np.frombuffer(raw, dt).reshape(e["shape"]).copy()
    return out


# ---- writing a tensor bundle (tests, and exporting canonical weights in the reference's own container) ---------------
def _crc32c_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return tbl


_CRC_TABLE = _crc32c_table()


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli), the checksum of SSTable blocks and of BundleEntryProto.crc32c."""
    crc ^= 0xFFFFFFFF
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_field(field: int, wt: int, payload) -> bytes:
    key = _enc_varint((field << 3) | wt)
    if wt == 0:
        return key + _enc_varint(payload)
    if wt == 5:
        return key + struct.pack("<I", payload)
    return key + _enc_varint(len(payload)) + payload


def write_bundle(variables_dir, tensors: dict[str, np.ndarray]) -> None:
    """``variables.index`` (SSTable: one data block per <= 4 KB of entries, an index block, the 48-byte footer, every block
    followed by its type byte and masked CRC-32C) + ``variables.data-00000-of-00001`` holding ``tensors`` by checkpoint
    key - the container ``tf.train.Checkpoint`` / ``tf.saved_model.save`` write (tensor_bundle.proto, table_format)."""
    d = Path(variables_dir)
    d.mkdir(parents=True, exist_ok=True)
    rev = {np.dtype(v): k for k, v in _DTYPES.items()}
    data = bytearray()
    entries: list[tuple[bytes, bytes]] = [(b"", _pb_field(1, 0, 1) + _pb_field(3, 2, _pb_field(1, 0, 1)))]   # BundleHeaderProto{num_shards 1, version{producer 1}}
    for key in sorted(tensors):
        arr = np.ascontiguousarray(tensors[key])
        raw = arr.tobytes()
        shape = b"".join(_pb_field(2, 2, _pb_field(1, 0, int(n))) for n in arr.shape)
        entry = (_pb_field(1, 0, rev[arr.dtype]) + _pb_field(2, 2, shape) + _pb_field(4, 0, len(data))
                 + _pb_field(5, 0, len(raw)) + _pb_field(6, 5, _masked_crc(raw)))
        entries.append((key.encode(), entry))
        data += raw
    (d / "variables.data-00000-of-00001").write_bytes(bytes(data))

    def block(items: list[tuple[bytes, bytes]]) -> bytes:       # no prefix compression: every entry is a restart point
        body, restarts = bytearray(), []
        for k, v in items:
            restarts.append(len(body))
            body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
        for r in restarts:
            body += struct.pack("<I", r)
        body += struct.pack("<I", len(restarts))
        return bytes(body)

    out = bytearray()
    index_items = []

    def emit(raw_block: bytes) -> tuple[int, int]:
        off = len(out)
        out.extend(raw_block)
        out.extend(b"\x00" + struct.pack("<I", _masked_crc(raw_block + b"\x00")))
        return off, len(raw_block)

    chunk, size = [], 0
    for kv in entries:
        chunk.append(kv)
        size += len(kv[0]) + len(kv[1])
        if size >= 4096:
            off, ln = emit(block(chunk))
            index_items.append((chunk[-1][0], _enc_varint(off) + _enc_varint(ln)))
            chunk, size = [], 0
    if chunk:
        off, ln = emit(block(chunk))
        index_items.append((chunk[-1][0], _enc_varint(off) + _enc_varint(ln)))
    meta_off, meta_len = emit(block([]))
    idx_off, idx_len = emit(block(index_items))
    footer = _enc_varint(meta_off) + _enc_varint(meta_len) + _enc_varint(idx_off) + _enc_varint(idx_len)
    footer += b"\x00" * (40 - len(footer)) + bytes.fromhex("57fb808b247547db")
    out.extend(footer)
    (d / "variables.index").write_bytes(bytes(out))


def bundle_layer_groups(variables_dir) -> list[tuple[tuple[str, ...], dict[str, np.ndarray]]]:
    """The float variables of a bundle grouped by the object that owns them, in creation order: checkpoint keys are
    ``<object path>/<attribute>/.ATTRIBUTES/VARIABLE_VALUE`` where a Keras-3 model's object path is
    ``_operations/<n>[/<sub-layer attribute>...]`` (or ``layers/<n>/...``) with ``n`` = position in the model's operation
    list, i.e. graph order; the attribute (``_kernel``, ``bias``, ``gamma``, ``moving_mean``, ``_embeddings`` ...) names
    the variable - leading underscores dropped.  Optimizer slots and non-float entries are skipped."""
    import re
    bundle = read_bundle(variables_dir)
    groups: dict[tuple[str, ...], dict[str, np.ndarray]] = {}
    for key, arr in bundle.items():
        if not key.endswith("/.ATTRIBUTES/VARIABLE_VALUE") or arr.dtype != np.float32:
            continue
        parts = key[:-len("/.ATTRIBUTES/VARIABLE_VALUE")].split("/")
        if len(parts) < 2 or parts[0] in ("optimizer", "_optimizer") or ".OPTIMIZER_SLOT" in key:
            continue
        groups.setdefault(tuple(parts[:-1]), {})[parts[-1].lstrip("_")] = arr

    def nat(path):
        return [tuple((1, int(t), "") if t.isdigit() else (0, 0, t) for t in re.split(r"(\d+)", c) if t != "") for c in path]
    return [(k, groups[k]) for k in sorted(groups, key=nat)]


# ---- saved_model.pb ------------------------------------------------------------------------------------------------
class Node:
    """One NodeDef of a FunctionDef: ``name``, ``op``, ``inputs`` (strings) and raw ``attr`` (name -> AttrValue bytes)."""
    __slots__ = ("name", "op", "inputs", "attr")

    def __init__(self, raw: bytes):
        self.name = pb_one(raw, 1, b"").decode()
        self.op = pb_one(raw, 2, b"").decode()
        self.inputs = [v.decode() for v in pb_all(raw, 3)]
        self.attr = {pb_one(a, 1, b"").decode(): pb_one(a, 2, b"") for a in pb_all(raw, 5)}

    # AttrValue{1: list{2 s, 3 i (packed), 4 f, 5 b}, 2: s, 3: i, 4: f, 5: b, 6: type, 7: shape, 8: tensor}
    def attr_i(self, key, default=None):
        a = self.attr.get(key)
        v = None if a is None else pb_one(a, 3)
        return default if v is None else _signed(v)

    def attr_b(self, key, default=False):
        a = self.attr.get(key)
        v = None if a is None else pb_one(a, 5)
        return default if v is None else bool(v)

    def attr_s(self, key, default=None):
        a = self.attr.get(key)
        v = None if a is None else pb_one(a, 2)
        return default if v is None else v.decode()

    def attr_type(self, key, default=None):
        a = self.attr.get(key)
        v = None if a is None else pb_one(a, 6)
        return default if v is None else v

    def attr_ints(self, key) -> list[int]:
        a = self.attr.get(key)
        if a is None:
            return []
        lst = pb_one(a, 1)
        if lst is None:
            return []
        out = []
        for v in pb_all(lst, 3):
            out += [_signed(x) for x in _packed_varints(v)]
        return out

    def attr_tensor(self, key="value"):
        a = self.attr.get(key)
        return None if a is None else parse_tensor(pb_one(a, 8, b""))


def parse_tensor(t: bytes) -> np.ndarray:
    """TensorProto{1 dtype, 2 shape, 4 tensor_content, 5 float_val, 6 double_val, 7 int_val, 10 int64_val, 11 bool_val}."""
    dtype = pb_one(t, 1, 0)
    shape = [_signed(pb_one(d, 1, 0)) for d in pb_all(pb_one(t, 2, b""), 2)]
    if dtype not in _DTYPES:
        raise ValueError(f"TensorProto dtype {dtype} is not supported")
    dt = np.dtype(_DTYPES[dtype])
    n = int(np.prod(shape)) if shape else 1
    content = pb_one(t, 4)
    if content:
        return np.frombuffer(content, dt).reshape(shape).copy()
    vals: list = []
    if dt == np.float32:
        for v in pb_all(t, 5):
            vals += list(struct.unpack(f"<{len(v) // 4}f", v)) if isinstance(v, bytes) else [v]
    elif dt == np.float64:
        for v in pb_all(t, 6):
            vals += list(struct.unpack(f"<{len(v) // 8}d", v)) if isinstance(v, bytes) else [v]
    elif dt in (np.int32, np.uint8, np.int8):
        for v in pb_all(t, 7):
            vals += [_signed(x) for x in _packed_varints(v)]
    elif dt == np.int64:
        for v in pb_all(t, 10):
            vals += [_signed(x) for x in _packed_varints(v)]
    elif dt == np.bool_:
        for v in pb_all(t, 11):
            vals += [bool(x) for x in _packed_varints(v)]
    if not vals:
        vals = [0]
    arr = np.asarray(vals, dt)
    if arr.size == 1 and n != 1:
        arr = np.full(n, arr[0], dt)          # a splat constant
    return arr.reshape(shape)


class Function:
    """One FunctionDef: argument names / dtypes, nodes, and the ``ret`` map (output name -> tensor reference)."""

    def __init__(self, raw: bytes):
        sig = pb_one(raw, 1, b"")
        self.name = pb_one(sig, 1, b"").decode()
        self.inputs = [(pb_one(a, 1, b"").decode(), pb_one(a, 3, 0)) for a in pb_all(sig, 2)]
        self.outputs = [pb_one(a, 1, b"").decode() for a in pb_all(sig, 3)]
        self.nodes = [Node(n) for n in pb_all(raw, 3)]
        self.ret = {pb_one(r, 1, b"").decode(): pb_one(r, 2, b"").decode() for r in pb_all(raw, 4)}


class SavedModel:
    """``saved_model.pb`` of a SavedModel directory, decoded far enough to census and to execute its serving function."""

    def __init__(self, graph_dir):
        self.dir = Path(graph_dir)
        raw = (self.dir / "saved_model.pb").read_bytes()
        mg = pb_one(raw, 2)
        if mg is None:
            raise ValueError(f"{self.dir}: saved_model.pb holds no MetaGraphDef")
        graph_def = pb_one(mg, 2, b"")
        self.functions = {f.name: f for f in (Function(x) for x in pb_all(pb_one(graph_def, 2, b""), 1))}
        og = pb_one(mg, 7, b"")
        self._objects = pb_all(og, 1)
        self.bound_inputs = {}
        for cf in pb_all(og, 2):
            val = pb_one(cf, 2, b"")
            ids: list[int] = []
            for v in pb_all(val, 2):
                ids += _packed_varints(v)
            self.bound_inputs[pb_one(cf, 1, b"").decode()] = ids

    def serving_function(self) -> Function:
        cands = [f for n, f in self.functions.items() if n.startswith("__inference_serving_default")]
        if not cands:                       # fall back to the function with the most convolutions
            cands = sorted(self.functions.values(), key=lambda f: -sum(n.op == "Conv2D" for n in f.nodes))[:1]
        if not cands:
            raise ValueError(f"{self.dir}: no serving function found")
        return max(cands, key=lambda f: len(f.nodes))

    def object_paths(self) -> dict[int, str]:
        """Object-graph node id -> checkpoint path (breadth-first shortest path of child names, as TF names them)."""
        children = []
        for obj in self._objects:
            children.append([(pb_one(c, 1, 0), pb_one(c, 2, b"").decode()) for c in pb_all(obj, 1)])
        paths, queue = {0: ""}, [0]
        while queue:
            nxt = []
            for nid in queue:
                for cid, name in children[nid]:
                    if cid not in paths:
                        paths[cid] = f"{paths[nid]}/{name}" if paths[nid] else name
                        nxt.append(cid)
            queue = nxt
        return paths

    def captured_variables(self, fn: Function, bundle: dict[str, np.ndarray]) -> dict[str, np.ndarray]:
        """Argument name -> variable value for the resource arguments the function captures."""
        paths = self.object_paths()
        bound = self.bound_inputs.get(fn.name, [])
        n_explicit = len(fn.inputs) - len(bound)
        out = {}
        for (arg, _), nid in zip(fn.inputs[n_explicit:], bound):
            key = f"{paths.get(nid, '?')}/.ATTRIBUTES/VARIABLE_VALUE"
            if key not in bundle:
                raise KeyError(f"captured argument {arg!r} -> {key!r} is not in the variable bundle")
            out[arg] = bundle[key]
        return out


# ---- census ----------------------------------------------------------------------------------------------------------
def census(graph_dir) -> dict:
    """What the serving function computes, counted per op kind with the attributes the layer plan depends on."""
    sm = SavedModel(graph_dir)
    fn = sm.serving_function()
    by_name = {n.name: n for n in fn.nodes}
    index = read_bundle_index(Path(graph_dir) / "variables" / "variables.index")
    variables = sorted((k.replace("/.ATTRIBUTES/VARIABLE_VALUE", ""), tuple(e["shape"])) for k, e in index.items()
                       if k.endswith("VARIABLE_VALUE") and e["dtype"] == 1)
    ops = Counter(n.op for n in fn.nodes)

    def const_of(ref: str):
        node = by_name.get(ref.split(":")[0].lstrip("^"))
        while node is not None and node.op == "Identity":
            node = by_name.get(node.inputs[0].split(":")[0])
        return node.attr_tensor() if node is not None and node.op == "Const" else None

    convs = Counter()
    for n in fn.nodes:
        if n.op == "Conv2D":
            convs[(n.attr_s("padding"), tuple(n.attr_ints("strides")), tuple(n.attr_ints("dilations")))] += 1
    dilations = Counter()
    for n in fn.nodes:
        if n.op == "SpaceToBatchND":
            blk = const_of(n.inputs[1])
            dilations[tuple(int(x) for x in np.ravel(blk))] += 1
    eps = Counter()
    for n in fn.nodes:                  # batch norm: rsqrt(var + eps)
        if n.op == "Rsqrt":
            add = by_name.get(n.inputs[0].split(":")[0])
            if add is not None and add.op in ("AddV2", "Add"):
                for ref in add.inputs:
                    c = const_of(ref)
                    if c is not None and c.size == 1:
                        eps[float(np.float32(c.ravel()[0]))] += 1
    pools = Counter((tuple(n.attr_ints("ksize")), tuple(n.attr_ints("strides")), n.attr_s("padding"))
                    for n in fn.nodes if n.op == "MaxPool")
    gelu = "erf" if ops.get("Erfc", 0) + ops.get("Erf", 0) > 0 else ("tanh" if ops.get("Tanh", 0) > 0 else "none")
    return {
        "function": fn.name, "n_nodes": len(fn.nodes),
        "inputs": [a for a, dt in fn.inputs if dt != 20], "outputs": fn.outputs,
        "n_captured_variables": sum(1 for _, dt in fn.inputs if dt == 20),
        "ops": dict(ops), "conv2d": {str(k): v for k, v in convs.items()},
        "space_to_batch_blocks": {str(k): v for k, v in dilations.items()},
        "batchnorm_eps": dict(eps), "maxpool": {str(k): v for k, v in pools.items()},
        "gelu_form": gelu, "n_gelu": ops.get("Erfc", 0) + ops.get("Erf", 0) + ops.get("Tanh", 0),
        "mask_ops": {k: ops.get(k, 0) for k in ("NotEqual", "Greater", "GreaterEqual", "Equal")},
        "variables": variables,
        "n_parameters": int(sum(int(np.prod(s)) for _, s in variables)),
    }
