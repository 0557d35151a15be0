"""``*_project.yaml`` -> explicit layer plan.

The reference turns the ``model:`` section of a project YAML into a Keras graph
in ``nnlib/builder.py`` (``build_fragment_classifier`` :442-838,
``_build_embedding`` :844-894, ``_build_block`` :982-1193, ``_get_pooler``
:1697-1714).  This module resolves the same section into a flat, fully
defaulted plan (every default below cites the constructor it comes from) that
``program.py`` compiles for the MI355X engine.  Layers outside the conv family
(attention, LSTM, Hyena, gated pooling ...) raise :class:`UnsupportedLayer`.

Canonical weight names (shared with the loaders in ``weights.py``):
``embedding/embeddings``; ``rep/<i>/{kernel,bias,gamma,beta,moving_mean,
moving_variance,alpha}`` for the i-th ``hidden_layers`` entry;
``rep/<i>/block<j>/{conv1,conv2,conv3,bn1,bn2,bn3}/<var>`` inside a
``residual_block``; ``classifier/<i>/...`` and ``reliability/<i>/...`` for the heads.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any

from . import maps

_ACT_ALIASES = ("relu", "gelu", "sigmoid", "softmax", "tanh")
_NORMS = ("masked_batchnorm", "masked_dyt", "masked_layernorm")
_DEFAULT_SIGNALS = ["max_prob", "entropy", "energy", "margin", "nmd_norm"]


class UnsupportedLayer(ValueError):
    """A layer / option of the project YAML that the MI355X engine does not implement."""


@dataclass
class Conv:
    name: str                      # weight prefix
    kernel_size: int
    cin: int
    filters: int
    strides: int = 1               # layers.py:1154
    padding: str = "valid"         # layers.py:1156 (MaskedConv1D default!)
    dilation_rate: int = 1         # layers.py:1157
    use_bias: bool = True          # layers.py:1159
    activation: str | None = None  # layers.py:1158
    use_masking: bool = True       # layers.py:1163
    mask_mode: str = "any"         # layers.py:1164


@dataclass
class Norm:
    name: str
    kind: str                      # masked_batchnorm | masked_dyt | masked_layernorm
    channels: int
    epsilon: float = 1e-5          # MaskedBatchNorm layers.py:804; LN 1e-3 layers.py:306
    use_masking: bool = True


@dataclass
class Act:
    kind: str


@dataclass
class Nmd:
    name: str
    channels: int
    epsilon: float = 1e-5          # nmd.py:19


@dataclass
class ResBlock:
    """One ResidualBlock (layers.py:1774-1915)."""
    name: str
    conv1: Conv
    bn1: Norm
    conv2: Conv
    bn2: Norm
    conv3: Conv | None
    bn3: Norm | None
    activation: str = "gelu"       # layers.py:1781
    use_masking: bool = True
    nmd: "Nmd | None" = None       # return_nmd (last block of a stack): bn2's NMD side output (layers.py:1896-1899)


@dataclass
class Dense:
    name: str
    cin: int
    units: int
    use_bias: bool = True
    activation: str | None = None


@dataclass
class ModelPlan:
    vocab: int
    embedding_dim: int
    rep: list[Any]
    pooling: str                   # "max" | "average"
    rep_channels: int
    classifier: list[Any]
    n_classes: int
    nmd_dims: list[int] = field(default_factory=list)
    reliability: list[Any] | None = None
    reliability_mode: str = "nmd"
    reliability_signals: list[str] = field(default_factory=list)
    use_masking: bool = True
    string_processor: dict = field(default_factory=dict)
    class_label_map: list[dict] = field(default_factory=list)
    # "embedding" = Embedding(vocab, E, mask_zero) on ids; "onehot_dense" = one-hot input -> Masking(0) -> bias-free
    # Dense(E); "onehot" = the one-hot rows themselves (embedding_size 0) (builder.py:844-880).  The one-hot forms
    # are the same gather with table row 0 = zeros (the all-zero one-hot row of an invalid codon, masked by Masking)
    embedding_kind: str = "embedding"
    # input_type "nucleotide" (builder.py:881-882): a (2, L, 4) one-hot input whose two strands run through ONE
    # shared-weight branch each - representation learner and classifier head - and are merged behind the head
    # (builder.py:488-494, :563-590, :1195-1266).  strands > 1 marks such a plan; merge = the classifier's merge method
    strands: int = 1
    merge: str = "average"
    # NMDMerge (nnlib/v2/nmd.py:93-170) over two or more NMD vectors: None / "concat" = the vectors side by side; "sum" /
    # "mean" / "max" / "weighted" = every vector through its own bias-free Dense(nmd_merge_dim) first
    # (rep/nmd_merge/proj_<i>/kernel; "weighted": + rep/nmd_merge/layer_weights, softmax-ed), then combined
    nmd_merge_mode: str = "concat"
    nmd_merge_dim: int = 0
    nmd_merge_act: str | None = None        # projection_kwargs.activation of the projections (round 6), None = linear
    # use_positional_embeddings (builder.py:886-892): SinusoidalPositionEmbedding(max_wavelength) rows (layers.py:2149-2195)
    # added to the embedded input; None = none
    positional_wavelength: float | None = None

    @property
    def nmd_raw_dim(self) -> int:
        """Width of the NMD vectors side by side (what the taps write)."""
        return sum(self.nmd_dims)

    @property
    def nmd_dim(self) -> int:
        """Width of the model's ``nmd`` output (what the reliability head reads)."""
        return self.nmd_merge_dim if self.nmd_merge_mode != "concat" else sum(self.nmd_dims)


def _norm(name: str, kind: str, channels: int, cfg: dict, use_masking: bool) -> Norm:
    if cfg.get("return_nmd") and kind != "masked_batchnorm":
        # MaskedDYT / MaskedLayerNormalization raise on return_nmd=True (layers.py:307-311, 398-402)
        raise UnsupportedLayer(f"{name}: return_nmd=True is only defined for masked_batchnorm")
    eps = {"masked_batchnorm": 1e-5, "masked_layernorm": 1e-3, "masked_dyt": 0.0}[kind]
    return Norm(name, kind, channels, float(cfg.get("epsilon", eps)), use_masking)


def _block(layers: list[dict], prefix: str, cin: int, use_masking_default: bool,
           nmd_dims: list[int] | None):
    """builder.py:982-1158: walk one ``hidden_layers`` list."""
    out: list[Any] = []
    for i, layer in enumerate(layers):
        name = str(layer.get("name", "")).lower()
        cfg = dict(layer.get("config", {}) or {})
        p = f"{prefix}/{i}"
        if name == "conv1d":
            # tf.keras.layers.Conv1D (builder.py:280): strides 1, padding "valid", dilation 1, bias, no activation, and no
            # mask handling of its own - an incoming Keras mask is dropped
            out.append(Conv(p, int(cfg["kernel_size"]), cin, int(cfg["filters"]), int(cfg.get("strides", 1)),
                            str(cfg.get("padding", "valid")).lower(), int(cfg.get("dilation_rate", 1)),
                            bool(cfg.get("use_bias", True)), cfg.get("activation"), False, "any"))
            if out[-1].padding not in ("valid", "same"):
                raise UnsupportedLayer(f"{p}: conv1d padding {out[-1].padding!r}")
            cin = int(cfg["filters"])
        elif name == "masked_conv1d":
            um = bool(cfg.get("use_masking", use_masking_default))   # builder.py:1019-1020
            mode = cfg.get("mask_mode", "any")
            if mode not in ("any", "majority", "strict"):
                raise ValueError(f"{p}: invalid mask_mode {mode!r}")
            out.append(Conv(p, int(cfg["kernel_size"]), cin, int(cfg["filters"]),
                            int(cfg.get("strides", 1)), str(cfg.get("padding", "valid")).lower(),
                            int(cfg.get("dilation_rate", 1)), bool(cfg.get("use_bias", True)),
                            cfg.get("activation"), um, mode))
            cin = int(cfg["filters"])
        elif name in _NORMS:
            um = bool(cfg.get("use_masking", use_masking_default)) if name == "masked_batchnorm" else True
            norm = _norm(p, name, cin, cfg, um)
            if cfg.get("return_nmd"):
                # MaskedBatchNorm(return_nmd=True) also returns the masked per-example channel mean of its INPUT
                # minus its own moving_mean (layers.py:943-954): an NMD tap in front of the norm that shares the
                # norm's moving_mean and epsilon
                if nmd_dims is None:
                    raise UnsupportedLayer(f"{p}: return_nmd norm outside the representation learner")
                out.append(Nmd(p, cin, norm.epsilon))
                nmd_dims.append(cin)
            out.append(norm)
        elif name == "nmd":
            if nmd_dims is None:
                raise UnsupportedLayer(f"{p}: nmd layer outside the representation learner")
            out.append(Nmd(p, cin, float(cfg.get("epsilon", 1e-5))))
            nmd_dims.append(cin)
        elif name == "activation" or name in _ACT_ALIASES:
            kind = name if name in _ACT_ALIASES else cfg.get("activation")   # builder.py:1022-1023
            out.append(Act(str(kind).lower()))
        elif name == "residual_block":
            um = bool(cfg.get("use_masking", use_masking_default))
            filters = int(cfg["filters"])
            k = int(cfg.get("kernel_size", 3))                 # layers.py:1788
            stride = int(cfg.get("strides", 1))
            pad = str(cfg.get("padding", "same")).lower()      # layers.py:1790
            dil = int(cfg.get("dilation_rate", 1))
            bias = bool(cfg.get("use_bias", True))
            nt = str(cfg.get("norm_type", "masked_batchnorm")).lower()
            if nt not in _NORMS:
                raise UnsupportedLayer(f"{p}: norm_type {nt!r}")
            act = cfg.get("activation", "gelu")
            for j in range(int(cfg.get("block_size", 1))):
                bp = f"{p}/block{j}"
                # use_1x1conv only reaches the first block (layers.py:2677-2679)
                bypass = (bool(cfg.get("use_1x1conv", False)) and j == 0) or stride > 1
                c1 = Conv(f"{bp}/conv1", k, cin, filters, stride, pad, dil, bias, None, um, "any")
                c2 = Conv(f"{bp}/conv2", k, filters, filters, 1, pad, dil, bias, None, um, "any")
                c3 = b3 = None
                if bypass:
                    c3 = Conv(f"{bp}/conv3", 1, cin, filters, stride, pad, dil, bias, None, um, "any")
                    b3 = _norm(f"{bp}/bn3", nt, filters, {}, um)
                blk = ResBlock(bp, c1, _norm(f"{bp}/bn1", nt, filters, {}, um), c2,
                               _norm(f"{bp}/bn2", nt, filters, {}, um), c3, b3, act, um)
                if cfg.get("return_nmd") and j == int(cfg.get("block_size", 1)) - 1:   # layers.py:2682-2686
                    if nt != "masked_batchnorm":
                        raise ValueError(f"{p}: return_nmd=True is only supported with norm_type='masked_batchnorm'")
                    if nmd_dims is None:
                        raise UnsupportedLayer(f"{p}: return_nmd block outside the representation learner")
                    blk.nmd = Nmd(f"{bp}/bn2", filters, blk.bn2.epsilon)
                    nmd_dims.append(filters)
                out.append(blk)
                cin = filters
        elif name == "dense":
            out.append(Dense(p, cin, int(cfg["units"]), bool(cfg.get("use_bias", True)),
                             cfg.get("activation")))
            cin = int(cfg["units"])
        elif name == "dropout":
            continue                                            # identity at inference
        else:
            raise UnsupportedLayer(
                f"{p}: layer {name!r} is outside the Conv1D -> norm -> pool -> dense family "
                "the MI355X engine implements")
    return out, cin


def resolve_string_processor(model_cfg: dict) -> dict:
    """``InferModel._load_string_processor_config`` (nnlib/inference.py:423-483)."""
    cfg = dict(model_cfg.get("embedding", {}) or {})
    cfg.update(model_cfg.get("string_processor", {}) or {})
    cfg["input_type"] = cfg.get("type", "translated")          # inference.py:443-444
    if cfg.get("codon") is not None and cfg.get("codon_id") is not None:
        codon_name, id_name = cfg["codon"], cfg["codon_id"]
        cfg["codon"] = maps.NAMED_MAPS.get(codon_name)
        cfg["codon_id"] = maps.NAMED_MAPS.get(id_name)
        if cfg["codon"] is None or cfg["codon_id"] is None:
            raise UnsupportedLayer(f"codon map {codon_name!r}/{id_name!r} is not supported")
        cfg["codon_depth"] = max(cfg["codon_id"]) + 1
        cfg["vocab_size"] = len(cfg["codon_id"]) + 1
        cfg["ngram_width"] = int(math.log(len(cfg["codon"]), 4))
        shape = (model_cfg.get("embedding", {}) or {}).get("input_shape")
        if cfg.get("seq_onehot") is None and shape is not None:
            cfg["seq_onehot"] = len(shape) == 3 and shape[-1] is not None and shape[-1] > 1
        cfg["seq_onehot"] = cfg.get("seq_onehot", False)
        if cfg["seq_onehot"] is False:
            cfg["codon_depth"] = 1
    if "crop_size" in cfg:
        size = cfg["crop_size"]
        units = cfg.setdefault("crop_units", "nucleotide" if cfg["input_type"] == "nucleotide" else "codon")
        if cfg["input_type"] == "nucleotide":
            cfg["crop_size_nt"] = int(size)
        elif units == "codon":                                  # seqops/crop.py:70-89
            cfg["crop_size_codons"], cfg["crop_size_nt"] = int(size), 3 * int(size) + 5
        else:
            cfg["crop_size_codons"], cfg["crop_size_nt"] = (int(size) - 5) // 3, int(size)
    return cfg


def build_plan(model_cfg: dict) -> ModelPlan:
    """Resolve the ``model:`` section of a project YAML into a :class:`ModelPlan`."""
    sp = resolve_string_processor(model_cfg)
    emb = model_cfg.get("embedding")
    if emb is None:
        raise ValueError("Missing 'embedding' section in config")   # builder.py:476
    graph_input = str(emb.get("input_type", "translated")).lower()      # what the builder builds (builder.py:846,854-884)
    branched = ["branch" in (model_cfg.get(section) or {}) for section in ("representation_learner", "classifier")]
    if graph_input == "nucleotide" or any(branched):
        return _build_strand_plan(model_cfg, sp, graph_input, branched)
    use_emb = bool(emb.get("use_embedding_layer", False))
    if use_emb == bool(sp.get("seq_onehot")):
        # Embedding needs ids, the Dense / pass-through branch needs one-hot rows (builder.py:856-880)
        raise UnsupportedLayer("use_embedding_layer and seq_onehot must be opposite (ids -> Embedding, one-hot -> Dense)")
    positional = None
    if emb.get("use_positional_embeddings", False):
        # builder.py:886-892: x = Add()([x, SinusoidalPositionEmbedding(max_wavelength=positional_embedding_length)(x)])
        positional = emb.get("positional_embedding_length")
        if not positional or float(positional) <= 0:
            raise ValueError("use_positional_embeddings needs embedding.positional_embedding_length (the max_wavelength)")
        positional = float(positional)
    if sp["input_type"] != "translated":
        raise UnsupportedLayer(f"input_type {sp['input_type']!r} on a graph built for translated input")
    if sp.get("ngram_width", 3) not in (3, 6):
        raise UnsupportedLayer(f"n-gram width {sp.get('ngram_width')} (codons: 3, dicodons: 6)")
    if sp.get("ngram_width", 3) == 6:
        # codon: DICODON (nnlib/inference.py:430-451 -> ngram_width 6, commands/predict.py:224, seqops/encode.py:272-284): ids
        # of 4 096 codon pairs, 16-bit on the device; an Embedding lookup runs as an op of its own in front of the first conv
        if sp["codon_id"] != maps.DICODON_ID or sp["codon"] != maps.DICODONS:
            raise UnsupportedLayer("6-gram encodings other than DICODON / DICODON_ID")
    if "projection" in model_cfg:
        pass  # training-only head, not part of the serving graph outputs
    use_masking = bool(model_cfg.get("use_masking", True))          # builder.py:259
    e = int(emb.get("embedding_size", 4))
    kind = "embedding"
    if not use_emb:
        depth = max(sp["codon_id"]) + 1                        # one_hot depth (seqops/encode.py:297-302)
        kind = "onehot_dense" if e > 0 else "onehot"
        e = e if e > 0 else depth
        sp["vocab_size"] = depth + 1                           # ids on the device: 0 = invalid / padding, id + 1 else
    nmd_dims: list[int] = []
    rep_cfg = model_cfg["representation_learner"]
    rep, rep_c = _block(rep_cfg.get("hidden_layers", []), "rep", e, use_masking, nmd_dims)
    pooling = str(rep_cfg.get("pooling", "")).lower()
    pooling = {"masked_max": "max", "masked_average": "average"}.get(pooling, pooling)
    if pooling not in ("max", "average"):
        raise UnsupportedLayer(f"pooling {pooling!r} is not supported (max / average only)")
    cls, n_cls = _block(model_cfg["classifier"].get("hidden_layers", []), "classifier", rep_c,
                        use_masking, None)
    plan = ModelPlan(vocab=sp["vocab_size"], embedding_dim=e, rep=rep, pooling=pooling,
                     rep_channels=rep_c, classifier=cls, n_classes=n_cls, nmd_dims=nmd_dims,
                     use_masking=use_masking, string_processor=sp,
                     class_label_map=list(model_cfg.get("class_label_map", []) or []), embedding_kind=kind,
                     positional_wavelength=positional)
    rel = model_cfg.get("reliability_model")
    if rel is not None and nmd_dims:
        merge = rel.get("merge") or {}
        if len(nmd_dims) > 1 and merge:                        # builder.py:1176-1180: NMDMerge(**merge) over the list
            mmode = merge.get("mode", "concat")
            if mmode not in ("concat", "sum", "mean", "max", "weighted"):
                raise ValueError(f"Unsupported NMD merge mode: {mmode}")          # nmd.py:110-111
            if merge.get("axis", -1) != -1:
                raise UnsupportedLayer("NMDMerge: only axis = -1 is supported")
            if mmode != "concat":
                target = merge.get("target_dim")
                if target is None:
                    if len(set(nmd_dims)) != 1:
                        raise ValueError(f"target_dim is required for merge mode '{mmode}' when NMD channel "
                                         f"dimensions differ.")                 # nmd.py:128-132
                    target = nmd_dims[0]
                pk = {k: v for k, v in (merge.get("projection_kwargs") or {}).items()
                      if not k.startswith("kernel_")}          # (initialisers / regularisers / constraints: training only)
                # Dense(target_dim, use_bias=False, name=..., **projection_kwargs) (nmd.py:133-141): `use_bias` or `units` here
                # would be a duplicate keyword in the reference; what is left that changes the graph is `activation`
                if set(pk) - {"activation"}:
                    raise UnsupportedLayer(f"NMDMerge projection_kwargs {sorted(set(pk) - {'activation'})} are not supported "
                                           f"(bias-free projections, optionally with an activation)")
                act = pk.get("activation")
                if act is not None and str(act).lower() not in (set(_ACT_ALIASES) | {"linear"}):
                    raise UnsupportedLayer(f"NMDMerge projection activation {act!r}")
                plan.nmd_merge_act = None if act is None or str(act).lower() == "linear" else str(act).lower()
                plan.nmd_merge_mode, plan.nmd_merge_dim = mmode, int(target)
        mode = rel.get("mode", "nmd")
        if mode not in ("nmd", "nmd_plus_signals"):
            raise ValueError(f"Unsupported reliability_model.mode: {mode!r}")   # builder.py:628-632
        sig = list(rel.get("signals", _DEFAULT_SIGNALS)) if mode == "nmd_plus_signals" else []
        rin = plan.nmd_dim + len(sig)
        expected = rel.get("input_shape")
        if expected is not None and expected != rin:
            raise ValueError(f"reliability_model.input_shape ({expected}) does not match computed "
                             f"reliability input dimension ({rin})")              # builder.py:662-667
        plan.reliability, _ = _block(rel.get("hidden_layers", []), "reliability", rin, use_masking, None)
        plan.reliability_mode = mode
        plan.reliability_signals = sig
    elif rel is not None:
        raise ValueError("reliability_model is configured but the representation learner "
                         "produced no NMD tensor")                                # builder.py:636-641
    return plan


def _build_strand_plan(model_cfg: dict, sp: dict, graph_input: str, branched: list[bool]) -> ModelPlan:
    """The branched nucleotide model (``train_config/nn_config_500bp_dvf.yaml``): ``embedding.input_type: nucleotide``
    makes the input the (2, L, 4) one-hot strands themselves (builder.py:881-882; ``Masking`` passes the values through
    and the first plain Keras layer drops its mask); ``representation_learner.branch`` is ONE ``_build_block`` model applied
    to each strand (:1195-1266, :488-494), ``classifier.branch`` one head applied to each strand's vector with a closing
    ``merge`` layer (:563-590); outputs ``prediction`` (merged) and ``embedding`` (Average of the strand vectors,
    :776-791); no reliability head.  Compiled as rows of ONE frame: a strand is a program row, ``strands`` rows make a
    window (program.py: ``OP_STRANDS``)."""
    emb = model_cfg["embedding"]
    if graph_input != "nucleotide" or not all(branched):
        raise UnsupportedLayer("a branched representation learner / classifier is supported for the two-strand "
                               "nucleotide model only (embedding.input_type: nucleotide, both sections branched)")
    if emb.get("use_embedding_layer", False) or emb.get("use_positional_embeddings", False):
        raise UnsupportedLayer("nucleotide input takes the one-hot strands as they are (no embedding layer)")
    strands = int((emb.get("input_shape") or [2])[0] or 2)
    if strands != 2:
        raise UnsupportedLayer(f"nucleotide input has two strands (input_shape[0] = {strands})")
    if model_cfg.get("reliability_model") is not None:
        raise UnsupportedLayer("reliability head on a branched model (the reference's combined model has none, builder.py:776-791)")
    if sp["input_type"] != "nucleotide":
        # nnlib/inference.py:443-444 reads embedding.type: without it the reference's engine hands the TRANSLATED tensor
        # to this nucleotide graph and exits; the graph decides here
        sp["input_type_note"] = ("embedding.type is not 'nucleotide': the reference's InferModel would feed "
                                 f"x[{sp['input_type']!r}] to this nucleotide graph (nnlib/inference.py:443-444)")
        sp["input_type"] = "nucleotide"
        if "crop_size" in sp:
            sp["crop_units"] = "nucleotide"
            sp["crop_size_nt"] = int(sp["crop_size"])
            sp.pop("crop_size_codons", None)
    sp["vocab_size"] = 5                                          # device ids: 0 = all-zero one-hot row, 1..4 = A, G, C, T
    rep_cfg = model_cfg["representation_learner"]["branch"]
    plain = {"conv1d", "dense", "dropout", "activation", "merge", *_ACT_ALIASES}   # stock Keras layers (builder.py:280-302)
    for section in ("representation_learner", "classifier"):
        for layer in model_cfg[section]["branch"].get("hidden_layers", []):
            if str(layer.get("name", "")).lower() not in plain:
                raise UnsupportedLayer(f"{section}.branch: layer {layer.get('name')!r} (a strand carries no mask and no "
                                       "frame axis: conv1d / activation / dense / dropout only)")
    rep, rep_c = _block(rep_cfg.get("hidden_layers", []), "rep", 4, False, None)
    if not rep or not isinstance(rep[0], Conv):
        raise UnsupportedLayer("the strand branch must start with a conv1d layer")
    pooling = {"max1d": "max1d", "average1d": "average1d"}.get(str(rep_cfg.get("pooling", "")).lower())
    if pooling is None:
        raise UnsupportedLayer(f"strand branch pooling {rep_cfg.get('pooling')!r} (max1d / average1d: builder.py:1706-1707)")
    hidden = list(model_cfg["classifier"]["branch"].get("hidden_layers", []))
    if not hidden or str(hidden[-1].get("name", "")).lower() != "merge":
        raise ValueError("Branched classifier must end with a 'merge' layer")           # builder.py:565-568
    merge = str((hidden[-1].get("config") or {}).get("method", "average")).lower()    # builder.py:569-570
    if merge not in ("average", "sum", "max", "concat"):
        raise ValueError(f"Unknown merge method: {merge}")                               # builder.py:1266
    cls, n_cls = _block(hidden[:-1], "classifier", rep_c, False, None)
    return ModelPlan(vocab=5, embedding_dim=4, rep=rep, pooling=pooling, rep_channels=rep_c, classifier=cls,
                     n_classes=n_cls, use_masking=False, string_processor=sp,
                     class_label_map=list(model_cfg.get("class_label_map", []) or []), embedding_kind="onehot",
                     strands=strands, merge=merge)


def weight_shapes(plan: ModelPlan) -> dict[str, tuple]:
    """Canonical variable names -> shapes."""
    out: dict[str, tuple] = {}
    if plan.embedding_kind == "embedding":
        out["embedding/embeddings"] = (plan.vocab, plan.embedding_dim)
    elif plan.embedding_kind == "onehot_dense":
        out["embedding/kernel"] = (plan.vocab - 1, plan.embedding_dim)      # Dense(E, use_bias=False) on one-hot rows

    def norm(n: Norm):
        vars_ = {"masked_batchnorm": ("gamma", "beta", "moving_mean", "moving_variance"),
                 "masked_dyt": ("alpha", "gamma", "beta"),
                 "masked_layernorm": ("gamma", "beta")}[n.kind]
        for v in vars_:
            out[f"{n.name}/{v}"] = (1,) if v == "alpha" else (n.channels,)

    def conv(c: Conv):
        out[f"{c.name}/kernel"] = (c.kernel_size, c.cin, c.filters)
        if c.use_bias:
            out[f"{c.name}/bias"] = (c.filters,)

    for seq in (plan.rep, plan.classifier, plan.reliability or []):
        for layer in seq:
            if isinstance(layer, Conv):
                conv(layer)
            elif isinstance(layer, Norm):
                norm(layer)
            elif isinstance(layer, Nmd):
                out[f"{layer.name}/moving_mean"] = (layer.channels,)
            elif isinstance(layer, ResBlock):
                for c in (layer.conv1, layer.conv2, layer.conv3):
                    if c is not None:
                        conv(c)
                for n in (layer.bn1, layer.bn2, layer.bn3):
                    if n is not None:
                        norm(n)
            elif isinstance(layer, Dense):
                out[f"{layer.name}/kernel"] = (layer.cin, layer.units)
                if layer.use_bias:
                    out[f"{layer.name}/bias"] = (layer.units,)
        if seq is plan.rep and plan.nmd_merge_mode != "concat":                 # NMDMerge is built behind the rep block's layers
            for i, d in enumerate(plan.nmd_dims):
                out[f"rep/nmd_merge/proj_{i}/kernel"] = (d, plan.nmd_merge_dim)
            if plan.nmd_merge_mode == "weighted":
                out["rep/nmd_merge/layer_weights"] = (len(plan.nmd_dims),)
    return out


def conv_flops_per_position(plan: ModelPlan) -> list[tuple[str, int, int, int, str, int]]:
    """(name, k, cin, cout, padding, stride) of every conv, in execution order."""
    rows = []
    for layer in plan.rep:
        if isinstance(layer, Conv):
            rows.append((layer.name, layer.kernel_size, layer.cin, layer.filters, layer.padding, layer.strides))
        elif isinstance(layer, ResBlock):
            for c in (layer.conv1, layer.conv2, layer.conv3):
                if c is not None:
                    rows.append((c.name, c.kernel_size, c.cin, c.filters, c.padding, c.strides))
    return rows
