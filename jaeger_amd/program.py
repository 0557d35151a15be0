"""Layer plan -> op program + weight blob for ``libjaeger_hip.so``.

Fuses what the reference runs as separate Keras layers (``nnlib/builder.py:982-1193``
``_build_block``; ``nnlib/v2/layers.py:1882-1915`` ``ResidualBlock.call``) into
conv launches with epilogue stages:

* ``masked_conv1d -> [nmd] -> norm -> activation``            one conv op
* ``ResidualBlock``: ``conv1 -> bn1 -> act`` one conv op; ``conv2 -> bn2 -> (+ shortcut)
  -> act [-> nmd -> norm -> activation that follow the stack]`` one conv op;
  optional ``conv3 -> bn3`` bypass one conv op
* each conv with masking gets one tiny mask op (``layers.py:1226-1255``).

Pure host logic (numpy + ctypes structs): unit-tested on CPU.
"""

from __future__ import annotations

from dataclasses import dataclass, replace

import numpy as np

from . import _lib as L
from .plan import Act, Conv, Dense, ModelPlan, Nmd, Norm, ResBlock, UnsupportedLayer, weight_shapes

_ACT_CODE = {None: L.ACT_NONE, "linear": L.ACT_NONE, "gelu": L.ACT_GELU_TANH, "gelu_erf": L.ACT_GELU_ERF,
             "relu": L.ACT_RELU, "tanh": L.ACT_TANH, "sigmoid": L.ACT_SIGMOID}
_MASK_MODE = {"any": L.MASK_ANY, "majority": L.MASK_MAJORITY, "strict": L.MASK_STRICT}
_SIGNAL_CODE = {"max_prob": 1, "entropy": 2, "energy": 3, "margin": 4, "nmd_norm": 5}


def act_code(name) -> int:
    key = name.lower() if isinstance(name, str) else name
    if key not in _ACT_CODE:
        raise UnsupportedLayer(f"activation {name!r} is not supported by the MI355X engine")
    return _ACT_CODE[key]


class _Blob:
    """Flat f32 weight blob; every tensor starts on a 16-byte boundary."""

    def __init__(self):
        self.parts: list[np.ndarray] = []
        self.size = 0

    def add(self, arr) -> int:
        a = np.ascontiguousarray(arr, dtype=np.float32).ravel()
        off = self.size
        pad = (-a.size) % 4
        self.parts.append(a)
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
        self.size += a.size + pad
        return off

    def finish(self) -> np.ndarray:
        return np.concatenate(self.parts) if self.parts else np.zeros(4, np.float32)


class _Slots:
    def __init__(self, n: int, what: str):
        self.free = list(range(n))
        self.what = what

    def take(self) -> int:
        if not self.free:
            raise UnsupportedLayer(f"program needs more than {L.JG_MAX_BUFS} {self.what} buffers")
        return self.free.pop(0)

    def give(self, s: int) -> None:
        if s is not None and s >= 0 and s not in self.free:
            self.free.append(s)
            self.free.sort()


def pack_conv_kernel(kernel: np.ndarray) -> np.ndarray:
    """(k, cin, cout) -> (k, cin_pad, cout_pad) zero padded (cin to even, cout to x32)."""
    k, cin, cout = kernel.shape
    cin_pad, cout_pad = (cin + 1) & ~1, (cout + 31) // 32 * 32
    out = np.zeros((k, cin_pad, cout_pad), np.float32)
    out[:, :cin, :cout] = kernel
    return out


@dataclass
class Program:
    ops: list            # list[L.JgOp]
    blob: np.ndarray     # float32
    vocab: int
    n_classes: int
    has_reliability: bool
    nmd_dim: int
    embedding_dim: int
    strands: int = 1     # > 1: ids (W, strands, L), one frame per program row (OP_STRANDS closes the program)

    def op_array(self):
        arr = (L.JgOp * len(self.ops))()
        for i, op in enumerate(self.ops):
            arr[i] = op
        return arr

    def describe(self) -> list[str]:
        kinds = {v: k for k, v in vars(L).items() if k.startswith("OP_")}
        stk = {v: k[3:] for k, v in vars(L).items() if k.startswith("ST_")}
        rows = []
        for op in self.ops:
            st = "+".join(stk[op.stages[s].kind] for s in range(op.n_stages))
            rows.append(f"{kinds[op.kind][3:]:9s} in={op.in_buf} out={op.out_buf} m={op.in_mask}->{op.out_mask} "
                        f"k={op.k} c={op.cin}->{op.cout} s={op.stride} d={op.dilation} [{st}]")
        return rows


class _Compiler:
    def __init__(self, plan: ModelPlan, weights: dict[str, np.ndarray]):
        self.plan, self.w = plan, weights
        missing = [n for n in weight_shapes(plan) if n not in weights]
        if missing:
            raise KeyError(f"weights missing for: {missing[:6]}{'...' if len(missing) > 6 else ''}")
        for n, shp in weight_shapes(plan).items():
            if tuple(weights[n].shape) != tuple(shp):
                raise ValueError(f"weight {n}: shape {tuple(weights[n].shape)} != expected {shp}")
        self.blob = _Blob()
        self.ops: list = []
        self.bufs = _Slots(L.JG_MAX_BUFS, "activation")
        self.masks = _Slots(L.JG_MAX_BUFS, "mask")
        self.parts = _Slots(L.JG_MAX_BUFS, "nmd partial")
        self.nmd_off = 0
        # the taps write the model's nmd output itself - or, under an NMDMerge that projects (sum / mean / max / weighted),
        # a scratch vector the merge reads (the two last vector slots: the heads use the ones from VEC_SCRATCH0 up)
        self.nmd_vec = L.VEC_NMD if plan.nmd_merge_mode == "concat" else L.JG_MAX_VECS - 1

    # ---- helpers -----------------------------------------------------------
    def _op(self, kind, **kw):
        op = L.JgOp()
        op.kind = kind
        for f in ("in_buf", "out_buf", "in_mask", "out_mask", "in_vec", "out_vec"):
            setattr(op, f, -1)
        op.w_off = op.b_off = -1
        op.stride = op.dilation = 1
        for k, v in kw.items():
            setattr(op, k, v)
        return op

    @staticmethod
    def _stage(kind, arg=0, p0=-1, p1=-1, p2=-1, p3=-1, f0=0.0):
        st = L.JgStage()
        st.kind, st.arg, st.p0, st.p1, st.p2, st.p3, st.f0 = kind, arg, p0, p1, p2, p3, f0
        return st

    def _norm_stage(self, n: Norm, has_mask: bool):
        w = self.w
        if n.kind == "masked_batchnorm":
            var = w[f"{n.name}/moving_variance"].astype(np.float32)
            inv_std = (np.float32(1.0) / np.sqrt(var + np.float32(n.epsilon))).astype(np.float32)
            return self._stage(L.ST_BN, p0=self.blob.add(w[f"{n.name}/moving_mean"]), p1=self.blob.add(inv_std),
                               p2=self.blob.add(w[f"{n.name}/gamma"]), p3=self.blob.add(w[f"{n.name}/beta"]))
        if n.kind == "masked_dyt":
            return self._stage(L.ST_DYT, arg=1 if has_mask else 0, f0=float(w[f"{n.name}/alpha"].ravel()[0]),
                               p2=self.blob.add(w[f"{n.name}/gamma"]), p3=self.blob.add(w[f"{n.name}/beta"]))
        if n.kind == "masked_layernorm":
            # a reduction over the channel axis: cannot live in a conv epilogue; _emit_conv cuts the stage list here
            return self._stage(L.ST_LN, arg=1 if has_mask else 0, f0=float(n.epsilon),
                               p2=self.blob.add(w[f"{n.name}/gamma"]), p3=self.blob.add(w[f"{n.name}/beta"]))
        raise UnsupportedLayer(f"{n.name}: {n.kind} is not supported by the MI355X engine yet")

    def _emit_conv(self, c: Conv, in_buf: int, in_mask: int, stages: list, out_mask: int, out_buf: int):
        # MaskedLayerNormalization needs a whole channel row: the conv keeps the stages in front of it and an
        # element-wise op (LayerNorm kernel) runs the norm and everything behind it, in place on the conv's output
        tail = []
        for j, st in enumerate(stages):
            if st.kind == L.ST_LN:
                stages, tail = stages[:j], stages[j:]
                break
        self._emit_conv_op(c, in_buf, in_mask, stages, out_mask, out_buf)
        while tail:                                   # one op per LayerNorm (each leads its own op)
            nxt = next((j for j, st in enumerate(tail) if st.kind == L.ST_LN and j > 0), len(tail))
            if nxt > L.JG_MAX_STAGES:
                raise UnsupportedLayer(f"{c.name}: more than {L.JG_MAX_STAGES} stages behind a layer norm")
            op = self._op(L.OP_ELTWISE, in_buf=out_buf, out_buf=out_buf, out_mask=out_mask, cout=c.filters)
            op.n_stages = nxt
            for j, st in enumerate(tail[:nxt]):
                op.stages[j] = st
            self.ops.append(op)
            tail = tail[nxt:]

    def _emit_conv_op(self, c: Conv, in_buf: int, in_mask: int, stages: list, out_mask: int, out_buf: int):
        if len(stages) > L.JG_MAX_STAGES:
            raise UnsupportedLayer(f"{c.name}: more than {L.JG_MAX_STAGES} fused epilogue stages")
        op = self._op(L.OP_CONV, in_buf=in_buf, out_buf=out_buf,
                      in_mask=in_mask if c.use_masking else L.JG_BUF_NONE, out_mask=out_mask,
                      k=c.kernel_size, cin=c.cin, cout=c.filters, stride=c.strides, dilation=c.dilation_rate,
                      padding=L.PAD_SAME if c.padding == "same" else L.PAD_VALID,
                      mask_mode=_MASK_MODE[c.mask_mode],
                      w_off=self.blob.add(pack_conv_kernel(self.w[f"{c.name}/kernel"])))
        if in_buf == L.JG_BUF_IDS:
            op.b_off = self.emb_off
        op.n_stages = len(stages)
        for i, st in enumerate(stages):
            op.stages[i] = st
        self.ops.append(op)

    def _conv_mask(self, c: Conv, in_mask: int) -> int:
        """Emit the mask op of a conv; returns the output mask slot (or NONE)."""
        if not c.use_masking or in_mask == L.JG_BUF_NONE:
            return L.JG_BUF_NONE
        if c.padding not in ("same", "valid"):
            raise ValueError(f"{c.name}: Invalid padding type {c.padding!r}")
        out = self.masks.take()
        self.ops.append(self._op(L.OP_MASK, in_mask=in_mask, out_mask=out, k=c.kernel_size, stride=c.strides,
                                 dilation=c.dilation_rate, mask_mode=_MASK_MODE[c.mask_mode],
                                 padding=L.PAD_SAME if c.padding == "same" else L.PAD_VALID))
        return out

    def _base_stages(self, c: Conv) -> list:
        st = []
        if c.use_bias:
            st.append(self._stage(L.ST_BIAS, p0=self.blob.add(self.w[f"{c.name}/bias"])))
        if c.activation is not None:
            st.append(self._stage(L.ST_ACT, arg=act_code(c.activation)))
        return st

    def _fuse_tail(self, layers: list, i: int, stages: list, mask: int, channels: int, pending_nmd: list):
        """Greedily fuse following nmd / norm / activation layers as epilogue stages."""
        while i < len(layers) and len(stages) < L.JG_MAX_STAGES:
            nxt = layers[i]
            if isinstance(nxt, Nmd):
                slot = self.parts.take()
                stages.append(self._stage(L.ST_NMD, arg=slot))
                pending_nmd.append((nxt, slot, mask))
            elif isinstance(nxt, Norm) and nxt.kind in ("masked_batchnorm", "masked_dyt", "masked_layernorm"):
                stages.append(self._norm_stage(nxt, mask != L.JG_BUF_NONE))
                if nxt.kind == "masked_batchnorm" and not nxt.use_masking:
                    mask = L.JG_BUF_NONE          # supports_masking False drops the mask (layers.py:816)
            elif isinstance(nxt, Act):
                stages.append(self._stage(L.ST_ACT, arg=act_code(nxt.kind)))
            else:
                break
            i += 1
        return i, mask

    def _flush_nmd(self, pending: list, buf: int):
        for nmd, slot, mask in pending:
            self.ops.append(self._op(L.OP_NMD_FINAL, in_buf=buf, in_mask=mask, cout=nmd.channels, arg=slot,
                                     out_vec=self.nmd_vec, vec_off=self.nmd_off, f0=nmd.epsilon,
                                     b_off=self.blob.add(self.w[f"{nmd.name}/moving_mean"])))
            self.nmd_off += nmd.channels
            self.parts.give(slot)
        pending.clear()

    # ---- representation learner ----------------------------------------------
    def _rep(self):
        plan = self.plan
        layers = plan.rep
        if plan.embedding_kind == "embedding":
            table = self.w["embedding/embeddings"]
        else:       # one-hot input: row 0 (invalid codon = all-zero one-hot row) is zero, row id + 1 the Dense row / unit row
            rows = self.w["embedding/kernel"] if plan.embedding_kind == "onehot_dense" else \
                np.eye(plan.vocab - 1, dtype=np.float32)
            table = np.concatenate([np.zeros((1, rows.shape[1]), np.float32), np.asarray(rows, np.float32)])
        self.emb_off = self.blob.add(table)
        buf, mask = L.JG_BUF_IDS, L.JG_BUF_IDS      # Embedding(mask_zero=True), builder.py:858-867
        if plan.vocab > 256 or plan.positional_wavelength is not None:
            # codon pairs (codon: DICODON, 4 097 ids): 16-bit ids do not fit the convs' one-byte gather - the lookup runs as
            # an op of its own that writes the rows and the mask (id != 0); everything behind it reads those.  The same op
            # adds the rows of SinusoidalPositionEmbedding (use_positional_embeddings): a table of POSITION_ROWS positions
            # computed here once - the sum depends on the position, so it cannot live in the convs' id -> row gather
            buf, mask = self.bufs.take(), self.masks.take()
            pos = {}
            if plan.positional_wavelength is not None:
                if plan.embedding_dim % 4:
                    raise UnsupportedLayer("positional embeddings need an embedding width that is a multiple of 4")
                pe = sinusoidal_position_rows(POSITION_ROWS, plan.embedding_dim, plan.positional_wavelength)
                pos = dict(w_off=self.blob.add(pe), k=POSITION_ROWS)
            self.ops.append(self._op(L.OP_EMBED, out_buf=buf, out_mask=mask, cout=plan.embedding_dim, b_off=self.emb_off, **pos))
        i = 0
        if not layers or not isinstance(layers[0], Conv):
            # A norm / activation / nmd / residual block - or the pool itself - directly on the Embedding output (the
            # reference's own Embedding -> MaskedBatchNorm -> masked max pool case, tests/unit/test_masked_pooling.py:
            # 186-209): the table rows pass through a one-tap identity conv that does NOT multiply by the mask, so that
            # masked positions keep Embedding row 0 exactly as in the Keras graph, and the Embedding's mask (ids != 0)
            # becomes a mask slot of its own; the layers that follow fuse into that conv's epilogue.
            e = plan.embedding_dim
            ident = Conv("embedding/identity", 1, e, e, 1, "same", 1, False, None, False, "any")
            self.w = dict(self.w)
            self.w["embedding/identity/kernel"] = np.eye(e, dtype=np.float32)[None]
            om = self._conv_mask(replace(ident, use_masking=True), mask)
            stages, pending = [], []
            i, om2 = self._fuse_tail(layers, 0, stages, om, e, pending)
            out = self.bufs.take()
            self._emit_conv(ident, buf, L.JG_BUF_NONE, stages, om, out)
            self._flush_nmd(pending, out)
            buf, mask = out, om2
        while i < len(layers):
            layer = layers[i]
            pending: list = []
            if isinstance(layer, Conv):
                om = self._conv_mask(layer, mask)
                stages = self._base_stages(layer)
                i, om2 = self._fuse_tail(layers, i + 1, stages, om, layer.filters, pending)
                out = self.bufs.take()
                self._emit_conv(layer, buf, mask, stages, om, out)
                self._flush_nmd(pending, out)
                self.bufs.give(buf)
                if mask != om:
                    self.masks.give(mask)
                buf, mask = out, om2
            elif isinstance(layer, ResBlock):
                blk = layer
                in_mask = mask if blk.use_masking else L.JG_BUF_NONE
                m1 = self._conv_mask(blk.conv1, in_mask)
                b1 = self.bufs.take()
                st1 = self._base_stages(blk.conv1) + [self._norm_stage(blk.bn1, m1 != L.JG_BUF_NONE),
                                                      self._stage(L.ST_ACT, arg=act_code(blk.activation))]
                self._emit_conv(blk.conv1, buf, in_mask, st1, m1, b1)
                shortcut = buf
                b3 = None
                if blk.conv3 is not None:
                    m3 = self._conv_mask(blk.conv3, in_mask)
                    b3 = self.bufs.take()
                    st3 = self._base_stages(blk.conv3) + [self._norm_stage(blk.bn3, m3 != L.JG_BUF_NONE)]
                    self._emit_conv(blk.conv3, buf, in_mask, st3, m3, b3)
                    self.masks.give(m3)
                    shortcut = b3
                if shortcut == L.JG_BUF_IDS:
                    raise UnsupportedLayer(f"{blk.name}: a residual block cannot be the first layer")
                m2 = self._conv_mask(blk.conv2, m1)
                b2 = self.bufs.take()
                st2 = self._base_stages(blk.conv2)
                if blk.nmd is not None:             # bn2(return_nmd=True): tap on bn2's input, conv2's mask
                    slot = self.parts.take()
                    st2.append(self._stage(L.ST_NMD, arg=slot))
                    pending.append((blk.nmd, slot, m2))
                st2 += [self._norm_stage(blk.bn2, m2 != L.JG_BUF_NONE), self._stage(L.ST_ADD, arg=shortcut),
                        self._stage(L.ST_ACT, arg=act_code(blk.activation))]
                i, m2b = self._fuse_tail(layers, i + 1, st2, m2, blk.conv2.filters, pending)
                self._emit_conv(blk.conv2, b1, m1, st2, m2, b2)
                self._flush_nmd(pending, b2)
                for s in (buf, b1, b3):
                    self.bufs.give(s)
                for s in (mask, m1):
                    if s != m2:
                        self.masks.give(s)
                buf, mask = b2, m2b
            elif isinstance(layer, (Norm, Act)):
                if buf == L.JG_BUF_IDS:
                    raise UnsupportedLayer("a norm / activation directly on the embedding is not supported")
                stages: list = []
                i, mask2 = self._fuse_tail(layers, i, stages, mask, 0, pending)
                if pending:
                    raise UnsupportedLayer("an nmd layer must directly follow a conv or residual block")
                if not stages:
                    raise UnsupportedLayer(f"{getattr(layer, 'name', layer)}: unsupported standalone layer")
                channels = self._channels_before(layers, i)
                # a LayerNorm must lead its op (the LayerNorm kernel): cut the list in front of every LN stage
                cuts = [0] + [j for j, st in enumerate(stages) if st.kind == L.ST_LN and j > 0] + [len(stages)]
                for a_, b_ in zip(cuts[:-1], cuts[1:]):
                    op = self._op(L.OP_ELTWISE, in_buf=buf, out_buf=buf, out_mask=mask, cout=channels)
                    op.n_stages = b_ - a_
                    for j, st in enumerate(stages[a_:b_]):
                        op.stages[j] = st
                    self.ops.append(op)
                mask = mask2
            elif isinstance(layer, Nmd):
                raise UnsupportedLayer("an nmd layer must directly follow a conv or residual block")
            else:
                raise UnsupportedLayer(f"layer {layer!r} is not supported in the representation learner")
        if buf == L.JG_BUF_IDS:
            raise UnsupportedLayer("the representation learner has no conv layer")
        # max1d / average1d: the stock Keras Global*Pooling1D of a strand branch (builder.py:1706-1707) - no mask, no sentinel
        kind = L.POOL_MAX if plan.pooling in ("max", "max1d") else L.POOL_AVG      # (without a mask: plain max / mean)
        if plan.pooling in ("max1d", "average1d"):
            mask = L.JG_BUF_NONE
        self.ops.append(self._op(L.OP_POOL, in_buf=buf, in_mask=mask, out_vec=L.VEC_EMBEDDING, vec_off=0,
                                 cout=plan.rep_channels, arg=kind))

    def _channels_before(self, layers, i) -> int:
        c = self.plan.embedding_dim
        for layer in layers[:i]:
            if isinstance(layer, Conv):
                c = layer.filters
            elif isinstance(layer, ResBlock):
                c = layer.conv2.filters
        return c

    # ---- heads -----------------------------------------------------------------
    def _head(self, layers: list, in_vec: int, out_vec: int, scratch0: int) -> None:
        dense_idx = [j for j, l in enumerate(layers) if isinstance(l, Dense)]
        if not dense_idx:
            raise UnsupportedLayer("a head needs at least one dense layer")
        cur, scratch = in_vec, scratch0
        j = 0
        while j < len(layers):
            layer = layers[j]
            if not isinstance(layer, Dense):
                raise UnsupportedLayer(f"head layer {layer!r} must follow a dense layer")
            act = layer.activation
            if j + 1 < len(layers) and isinstance(layers[j + 1], Act):
                if act is not None:
                    raise UnsupportedLayer("dense activation followed by another activation")
                act = layers[j + 1].kind
                j += 1
            last = not any(isinstance(l, Dense) for l in layers[j + 1:])
            dst = out_vec if last else scratch
            if not last:
                scratch += 1
                if scratch >= L.JG_MAX_VECS:
                    raise UnsupportedLayer("head too deep for the vector slots")
            b_off = self.blob.add(self.w[f"{layer.name}/bias"]) if layer.use_bias else -1
            self.ops.append(self._op(L.OP_DENSE, in_vec=cur, out_vec=dst, vec_off=0, cin=layer.cin,
                                     cout=layer.units, arg=act_code(act),
                                     w_off=self.blob.add(self.w[f"{layer.name}/kernel"]), b_off=b_off))
            cur = dst
            j += 1

    def _nmd_merge(self) -> None:
        """NMDMerge(mode != "concat") (nnlib/v2/nmd.py:141-155) behind the representation learner.  Every tap's vector goes
        through its own bias-free projection and the results are added (sum), averaged (mean), weighted by
        softmax(layer_weights) (weighted) - all three ONE dense layer over the vectors side by side, its kernel the
        projections stacked and scaled - or maximised element by element (max: the projections as one block-diagonal
        dense layer, then JG_OP_VECMAX over the blocks)."""
        plan = self.plan
        n, t = len(plan.nmd_dims), plan.nmd_merge_dim
        proj = [np.asarray(self.w[f"rep/nmd_merge/proj_{i}/kernel"], np.float32) for i in range(n)]
        raw = plan.nmd_raw_dim
        act = getattr(plan, "nmd_merge_act", None)
        if plan.nmd_merge_mode == "max" or act is not None:
            # the projections as ONE block-diagonal dense layer (with the projections' activation, projection_kwargs: round 6)
            kernel = np.zeros((raw, n * t), np.float32)
            at = 0
            for i, (d, p) in enumerate(zip(plan.nmd_dims, proj)):
                kernel[at:at + d, i * t:(i + 1) * t] = p
                at += d
            blocks = L.JG_MAX_VECS - 2
            self.ops.append(self._op(L.OP_DENSE, in_vec=self.nmd_vec, out_vec=blocks, vec_off=0, cin=raw, cout=n * t,
                                     arg=act_code(act), w_off=self.blob.add(kernel), b_off=-1))
            if plan.nmd_merge_mode == "max":
                self.ops.append(self._op(L.OP_VECMAX, in_vec=blocks, out_vec=L.VEC_NMD, vec_off=0, k=n, cout=t))
                return
        if plan.nmd_merge_mode == "weighted":
            lw = np.asarray(self.w["rep/nmd_merge/layer_weights"], np.float32)
            e = np.exp(lw - lw.max())
            scale = (e / e.sum()).astype(np.float32)            # tf.nn.softmax(layer_weights), nmd.py:153-155
        else:
            scale = np.full(n, 1.0 if plan.nmd_merge_mode == "sum" else 1.0 / n, np.float32)
        if act is not None:
            # sum / mean / weighted of ACTIVATED projections: a second, linear dense layer over the blocks whose kernel is the
            # scaled identity matrices stacked (every product with a zero is exact: the sum is the weighted sum of the blocks)
            kernel = np.concatenate([np.eye(t, dtype=np.float32) * s for s in scale], axis=0).astype(np.float32)
            self.ops.append(self._op(L.OP_DENSE, in_vec=L.JG_MAX_VECS - 2, out_vec=L.VEC_NMD, vec_off=0, cin=n * t, cout=t,
                                     arg=act_code(None), w_off=self.blob.add(kernel), b_off=-1))
            return
        kernel = np.concatenate([p * s for p, s in zip(proj, scale)], axis=0).astype(np.float32)
        self.ops.append(self._op(L.OP_DENSE, in_vec=self.nmd_vec, out_vec=L.VEC_NMD, vec_off=0, cin=raw, cout=t,
                                 arg=act_code(None), w_off=self.blob.add(kernel), b_off=-1))

    def compile(self) -> Program:
        plan = self.plan
        self._rep()
        if plan.nmd_merge_mode != "concat":
            self._nmd_merge()
        self._head(plan.classifier, L.VEC_EMBEDDING, L.VEC_PREDICTION, L.VEC_SCRATCH0)
        if plan.reliability is not None:
            if plan.reliability_signals:
                order = 0
                for idx, s in enumerate(plan.reliability_signals):
                    if s not in _SIGNAL_CODE:
                        raise ValueError(f"Unsupported signal(s): {s!r}")       # layers.py:1626-1630
                    order |= _SIGNAL_CODE[s] << (4 * idx)
                self.ops.append(self._op(L.OP_OODSIG, in_vec=L.VEC_PREDICTION, k=L.VEC_NMD, cin=plan.n_classes,
                                         cout=len(plan.reliability_signals), out_vec=L.VEC_NMD,
                                         vec_off=plan.nmd_dim, arg=order, f0=1e-10, stride=plan.nmd_dim))
            self._head(plan.reliability, L.VEC_NMD, L.VEC_RELIABILITY, L.VEC_SCRATCH0 + 4)
        if plan.strands > 1:
            # the heads above ran per strand (shared weights); the window's outputs are the strands' merged
            # (tf.keras.layers.Average / Add / Maximum, builder.py:1251-1262; embedding: Average, :779-780)
            self.ops.append(self._op(L.OP_STRANDS, k=plan.strands,
                                     arg={"average": L.MERGE_AVERAGE, "sum": L.MERGE_SUM, "max": L.MERGE_MAX,
                                          "concat": L.MERGE_CONCAT}[plan.merge]))
        return Program(self.ops, self.blob.finish(), plan.vocab, plan.n_classes,
                       plan.reliability is not None, plan.nmd_dim, plan.rep_channels, plan.strands)


POSITION_ROWS = 8192        # rows of the position table a program carries: windows of up to 24 576 bases (a row = one frame's codons)


def sinusoidal_position_rows(n_pos: int, hidden: int, max_wavelength: float) -> np.ndarray:
    """SinusoidalPositionEmbedding.call (nnlib/v2/layers.py:2155-2195) for positions 0 .. n_pos - 1, in float32 like the
    layer's compute dtype: timescale_i = (1 / max_wavelength) ** (2 floor(i / 2) / hidden), row[p, i] = sin(p timescale_i) for
    even i, cos(p timescale_i) for odd i."""
    f = np.float32
    positions = np.arange(n_pos, dtype=f)
    dims = np.arange(hidden, dtype=f)
    even = np.floor(dims / f(2)) * f(2)
    timescales = np.power(f(1.0) / f(max_wavelength), even / f(hidden)).astype(f)
    angles = (positions[:, None] * timescales[None, :]).astype(f)
    sin_mask = (np.arange(hidden) % 2 == 0).astype(f)
    return (np.sin(angles) * sin_mask + np.cos(angles) * (f(1.0) - sin_mask)).astype(f)


def compile_plan(plan: ModelPlan, weights: dict[str, np.ndarray]) -> Program:
    return _Compiler(plan, weights).compile()
