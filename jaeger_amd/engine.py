"""Python face of the MI355X engine: the reference's inference-engine duck type.

``JaegerHipEngine`` provides what ``commands/predict.py:745-841`` expects of an
engine (``InferModel``, ``nnlib/inference.py:300-483``): ``class_map``,
``string_processor_config`` and ``predict(dataset) -> dict[str, np.ndarray]``
- plus the fused entry points that move the fragmenter's per-window work and
the encoder onto the GPU (``predict_windows``).  All compute goes through the
C-ABI in ``include/jaeger_hip.h``; there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
import weakref
from collections import defaultdict
from pathlib import Path
from typing import Any, Iterable

import numpy as np
import yaml

from . import _lib as L
from .plan import ModelPlan, build_plan
from .program import Program, compile_plan


def _yaml_load(text: str):
    """``yaml.safe_load`` on libyaml's loader when PyYAML was built with it (ten times faster: the project YAML is on
    the critical path of a run's first milliseconds); same safe constructor set."""
    loader = getattr(yaml, "CSafeLoader", None)
    return yaml.load(text, Loader=loader) if loader is not None else yaml.safe_load(text)


def frame_length(nucleotides: int) -> int:
    """Codons per frame of a ``nucleotides``-long crop (seqops/crop.py:44-61)."""
    nt = int(nucleotides)
    if nt < 3:
        return 0
    usable = nt - 5 + (-2, -1, 0)[nt % 3]
    return 0 if usable <= 0 else -(-usable // 3)


def dicodon_frame_length(nucleotides: int) -> int:
    """Entries per frame of a ``nucleotides``-long crop under ``codon: DICODON`` (6-grams, seqops/encode.py:272-284):
    ``ngrams`` leaves n - 5 of them, a frame keeps every sixth below the codon frames' own stop."""
    nt = int(nucleotides)
    usable = nt - 8 + (-2, -1, 0)[nt % 3]
    return 0 if usable <= 0 else -(-usable // 6)


def codon_lut(codon_id: list[int]) -> np.ndarray:
    """65-byte table for ``jg_encode``: 16*b0+4*b1+b2 over TCAG=0..3 -> codon_id+1."""
    from .maps import CODONS
    index = {c: i for i, c in enumerate(CODONS)}
    lut = np.zeros(65, np.uint8)
    alpha = "TCAG"
    for i0, a in enumerate(alpha):
        for i1, b in enumerate(alpha):
            for i2, c in enumerate(alpha):
                lut[16 * i0 + 4 * i1 + i2] = codon_id[index[a + b + c]] + 1
    return lut


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


class HipDevice:
    """One ``jg_engine`` (one GPU, one HIP stream)."""

    def __init__(self, device_id: int = 0):
        self.lib = L.load()
        self.handle = C.c_void_p()
        L.check(self.lib.jg_engine_create(int(device_id), C.byref(self.handle)), "jg_engine_create")
        self.device_id = device_id
        self._models = weakref.WeakSet()      # models living on this engine: destroyed before it, whatever the order
                                              # in which the garbage collector finalises a cycle that holds both

    def close(self):
        if getattr(self, "handle", None):
            for mdl in list(getattr(self, "_models", ())):
                mdl.close()
            self.lib.jg_engine_destroy(self.handle)
            self.handle = None

    __del__ = close

    def sync(self):
        L.check(self.lib.jg_engine_sync(self.handle), "jg_engine_sync")

    def set_stream_bytes(self, nbytes: int):
        """Host base buffers above ``nbytes`` are streamed through pinned staging buffers (JG_OPT_STREAM_BYTES)."""
        L.check(self.lib.jg_engine_set_option(self.handle, L.JG_OPT_STREAM_BYTES, int(nbytes)), "jg_engine_set_option")

    def set_stream_priority(self, high: bool):
        """Re-create this (idle) engine's stream at the device's highest / default stream priority (JG_OPT_STREAM_PRIORITY)."""
        L.check(self.lib.jg_engine_set_option(self.handle, L.JG_OPT_STREAM_PRIORITY, int(bool(high))), "jg_engine_set_option")

    def set_table_net_lds(self, on: bool):
        """A strand branch's conv + pool on the exact-f32 LDS-table kernel (True) instead of the matrix cores
        (JG_OPT_TABLE_NET_LDS; tests and A/B timing)."""
        L.check(self.lib.jg_engine_set_option(self.handle, L.JG_OPT_TABLE_NET_LDS, int(bool(on))), "jg_engine_set_option")

    def set_conv_pc(self, on: bool):
        """Experiment build only (libjaeger_hip_exp.so): 128 -> 128 five-tap convs on the producer / consumer kernel (1) or
        on the two-workgroup kernel with the pipelined main loop (2) instead of the two-workgroup kernel (0, the default
        and the only one in the shipped library); same results bit for bit (JG_OPT_CONV_PC)."""
        L.check(self.lib.jg_engine_set_option(self.handle, L.JG_OPT_CONV_PC, int(on)), "jg_engine_set_option")

    def windows_done(self) -> int:
        """Progress of the ``predict_windows`` call running on this engine in ANOTHER thread (JG_STAT_WINDOWS_DONE): the
        output rows and counts of windows [0, value) are final."""
        return int(self.lib.jg_engine_get_stat(self.handle, L.JG_STAT_WINDOWS_DONE))

    def set_fuse_resblock(self, on: bool):
        """Narrow residual blocks as one launch with the intermediate tensor in LDS (True, the default) or conv by conv
        (JG_OPT_FUSE_RESBLOCK; A/B timing and tests)."""
        L.check(self.lib.jg_engine_set_option(self.handle, L.JG_OPT_FUSE_RESBLOCK, int(bool(on))), "jg_engine_set_option")

    def reset_progress(self):
        """``windows_done()`` back to 0 (JG_OPT_RESET_PROGRESS): called before a ``predict_windows`` call is handed to another
        thread, so that a poller which starts first never reads the previous call's final count."""
        L.check(self.lib.jg_engine_set_option(self.handle, L.JG_OPT_RESET_PROGRESS, 0), "jg_engine_set_option")

    def stream_stats(self) -> dict:
        """Streaming statistics of the last ``jg_predict_windows`` call on this engine."""
        g = lambda k: int(self.lib.jg_engine_get_stat(self.handle, k))  # noqa: E731
        return {"groups": g(L.JG_STAT_STREAM_GROUPS), "bytes": g(L.JG_STAT_STREAM_BYTES),
                "peak_device_bases": g(L.JG_STAT_PEAK_DEVICE_BASES)}

    # raw device memory -------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        L.check(self.lib.jg_dev_alloc(self.handle, int(nbytes), C.byref(p)), "jg_dev_alloc")
        return p.value or 0

    def free(self, ptr: int):
        if ptr:
            L.check(self.lib.jg_dev_free(self.handle, C.c_void_p(ptr)), "jg_dev_free")

    def upload(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.alloc(max(arr.nbytes, 1))
        L.check(self.lib.jg_memcpy_h2d(self.handle, C.c_void_p(p), _ptr(arr), arr.nbytes), "jg_memcpy_h2d")
        return p

    def download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        L.check(self.lib.jg_memcpy_d2h(self.handle, _ptr(out), C.c_void_p(ptr), out.nbytes), "jg_memcpy_d2h")
        return out

    def attach_records(self, offsets, window: int = 64, threshold: int = 20):
        """DUST inside the fused path (``jg_engine_set_dust``): the following ``predict_windows`` / ``encode`` calls on
        host bases soft-mask their uploaded copy on the device before encoding it.  ``offsets=None`` detaches."""
        if offsets is None:
            L.check(self.lib.jg_engine_set_dust(self.handle, None, 0, 0, 0), "jg_engine_set_dust")
            return
        off = np.ascontiguousarray(offsets, np.int64)
        L.check(self.lib.jg_engine_set_dust(self.handle, _ptr(off), len(off) - 1, int(window), int(threshold)),
                "jg_engine_set_dust")

    def dust_masked(self) -> int:
        return int(self.lib.jg_engine_get_stat(self.handle, L.JG_STAT_DUST_MASKED))

    def dust_mask(self, bases_ptr: int, n_bases: int, offsets: np.ndarray, window: int = 64, threshold: int = 20,
                  count: bool = True) -> int:
        """Symmetric-DUST soft-masking of device-resident bases in place (``jg_dust_mask_device``); returns the number
        of masked bases (``count=False``: asynchronous on the engine stream, returns -1)."""
        off = np.ascontiguousarray(offsets, np.int64)
        n = C.c_int64(-1)
        L.check(self.lib.jg_dust_mask_device(self.handle, C.c_void_p(bases_ptr), int(n_bases), _ptr(off), L.JG_PTR_HOST,
                                             len(off) - 1, int(window), int(threshold), C.byref(n) if count else None,
                                             None), "jg_dust_mask_device")
        return int(n.value)

    # timing --------------------------------------------------------------------
    def timer_start(self):
        L.check(self.lib.jg_timer_start(self.handle, None))

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        L.check(self.lib.jg_timer_stop_ms(self.handle, None, C.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on: bool = True):
        L.check(self.lib.jg_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self) -> dict:
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        L.check(self.lib.jg_profile_read(self.handle, C.byref(ms), C.byref(n), C.byref(fl)))
        out = {"conv_ms": ms.value, "conv_launches": n.value, "conv_flops": fl.value}
        for cls, name in enumerate(("mfma_f16x3", "mfma_f32", "table", "fused_small")):
            L.check(self.lib.jg_profile_read_class(self.handle, cls, C.byref(ms), C.byref(n), C.byref(fl)))
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value}
        return out

    def box_calibrate(self, seconds: float = 0.5) -> dict:
        """``jg_box_calibrate``: a bare ``v_mfma_f32_32x32x16_f16`` loop on random register operands for about ``seconds``
        of back-to-back launches - the dense f16 matrix-core rate and the in-kernel shader clock THIS device holds under
        load (MI355X devices differ by up to 12 % there)."""
        tf, ghz = C.c_double(), C.c_double()
        info = (C.c_double * 5)()
        L.check(self.lib.jg_box_calibrate(self.handle, float(seconds), C.byref(tf), C.byref(ghz), info), "jg_box_calibrate")
        return {"mfma_loop_tflops": round(tf.value, 1), "clock_ghz": round(ghz.value, 4), "launches": int(info[0]),
                "last_launch_ms": round(info[1], 3), "mean_tflops": round(info[2], 1),
                "clock_ghz_min": round(info[3], 4), "clock_ghz_max": round(info[4], 4), "seconds": float(seconds),
                "what": "bare v_mfma_f32_32x32x16_f16 loop, random f16 operands in registers, two waves per SIMD on every "
                        "CU, no memory traffic; TFLOP/s of the last launch by HIP events, clock = d s_memtime / "
                        "d s_memrealtime x 100 MHz (median over workgroups)"}

    def encode(self, bases: np.ndarray, win_start: np.ndarray, win_len: np.ndarray, fsize: int,
               lut: np.ndarray, flags: int = 0, l_pad: int | None = None):
        """``jg_encode`` with host buffers -> (ids (W,6,l_pad) u8, counts (W,4) i32); with ``JG_ENC_NUCLEOTIDE`` in
        ``flags`` the nucleotide ids (W,2,l_pad) of ``input_type: nucleotide`` (``l_pad`` counts bases)."""
        bases = np.ascontiguousarray(bases, np.uint8)
        ws = np.ascontiguousarray(win_start, np.int64)
        wl = np.ascontiguousarray(win_len, np.int32)
        n = ws.size
        nt = bool(int(flags) & L.JG_ENC_NUCLEOTIDE)
        di = bool(int(flags) & L.JG_ENC_DICODON)           # codon pairs: 16-bit ids, six bases apart
        l_pad = (int(fsize) if nt else dicodon_frame_length(fsize) if di else frame_length(fsize)) if l_pad is None else int(l_pad)
        ids = np.zeros((n, 2 if nt else 6, l_pad), np.uint16 if di else np.uint8)
        counts = np.zeros((n, 4), np.int32)
        lut = np.ascontiguousarray(lut, np.uint8)
        L.check(self.lib.jg_encode(self.handle, _ptr(bases), bases.size, L.JG_PTR_HOST, _ptr(ws), _ptr(wl),
                                   L.JG_PTR_HOST, n, int(fsize), _ptr(lut), int(flags), l_pad,
                                   _ptr(ids), _ptr(counts), L.JG_PTR_HOST, None), "jg_encode")
        return ids, counts


class HipModel:
    """A compiled :class:`Program` resident on one :class:`HipDevice`."""

    def __init__(self, device: HipDevice, program: Program):
        self.dev, self.program, self.lib = device, program, device.lib
        self.handle = C.c_void_p()
        ops = program.op_array()
        blob = np.ascontiguousarray(program.blob, np.float32)
        L.check(self.lib.jg_model_create(device.handle, ops, len(program.ops), _ptr(blob), blob.size,
                                         program.vocab, C.byref(self.handle)), "jg_model_create")
        self.widths = {name: self.lib.jg_model_vec_width(self.handle, i)
                       for i, name in enumerate(("prediction", "reliability", "embedding", "nmd"))}
        self.strands = int(getattr(program, "strands", 1))      # > 1: nucleotide ids (W, strands, L), rows count bases
        self.wide_ids = int(program.vocab) > 256                 # dicodon model: 16-bit ids, rows count codon pairs
        device._models.add(self)

    def close(self):
        if getattr(self, "handle", None):
            if getattr(self.dev, "handle", None):     # (an engine that is already gone took its models with it)
                self.lib.jg_model_destroy(self.handle)
            self.handle = None

    __del__ = close

    @property
    def precision(self) -> str:
        return "f16x3" if self.lib.jg_model_get_precision(self.handle) == 1 else "f32"

    def set_precision(self, mode: str) -> None:
        """"f32" = exact-f32 MFMA kernels, "f16x3" = split-f16 fast path (f32-accurate)."""
        L.check(self.lib.jg_model_set_precision(self.handle, {"f32": 0, "f16x3": 1}[mode]),
                "jg_model_set_precision")

    def placement(self) -> dict:
        """How the program was placed on the kernels (``jg_model_get_stat``): number of convolutions, how many of them
        run on the split-f16 kernels in "f16x3" mode, queued F16S <-> f32 layout conversions, fused small-window kernel."""
        g = lambda k: int(self.lib.jg_model_get_stat(self.handle, k))  # noqa: E731
        return {"convs": g(L.JG_MSTAT_CONVS), "convs_f16x3": g(L.JG_MSTAT_CONVS_F16X3),
                "layout_conversions": g(L.JG_MSTAT_LAYOUT_CONVERSIONS), "small_fused": bool(g(L.JG_MSTAT_SMALL_FUSED))}

    def describe(self) -> str:
        """Conv by conv: geometry, the kernel it runs on and - for convs left on the exact-f32 kernel - why (``jg_model_describe``)."""
        buf = C.create_string_buffer(1 << 16)
        L.check(self.lib.jg_model_describe(self.handle, buf, len(buf)), "jg_model_describe")
        return buf.value.decode()

    def flops_per_window(self, l: int) -> float:
        return float(self.lib.jg_model_flops_per_window(self.handle, int(l)))

    def row_length(self, nucleotides: int) -> int:
        """Positions per id row of a window of ``nucleotides`` bases: codons per frame, or the bases themselves."""
        if self.strands > 1:
            return int(nucleotides)
        return dicodon_frame_length(nucleotides) if self.wide_ids else frame_length(nucleotides)

    def _host_outputs(self, n: int, want: Iterable[str]):
        outs = {}
        for name in ("prediction", "reliability", "embedding", "nmd"):
            w = self.widths[name]
            outs[name] = np.zeros((n, w), np.float32) if (w > 0 and name in want) else None
        return outs

    def forward(self, ids: np.ndarray, chunk: int = 0,
                want=("prediction", "reliability", "embedding", "nmd")) -> dict[str, np.ndarray]:
        """ids (W, 6, L) u8 on the host - (W, 2, L) nucleotide ids for a two-strand model - -> dict of host arrays
        (``jg_forward``)."""
        ids = np.ascontiguousarray(ids, np.uint16 if self.wide_ids else np.uint8)
        n, rows, l = ids.shape
        if rows != (self.strands if self.strands > 1 else 6):
            raise ValueError(f"id tensor has {rows} rows per window, the model takes {self.strands if self.strands > 1 else 6}")
        o = self._host_outputs(n, want)
        L.check(self.lib.jg_forward(self.handle, _ptr(ids), L.JG_PTR_HOST, n, l, _ptr(o["prediction"]),
                                    _ptr(o["reliability"]), _ptr(o["embedding"]), _ptr(o["nmd"]),
                                    L.JG_PTR_HOST, int(chunk), None), "jg_forward")
        return {k: v for k, v in o.items() if v is not None}

    def predict_windows(self, bases, n_bases: int, win_start, win_len, n_win: int, fsize: int, lut,
                        flags: int = 0, l_pad: int | None = None, chunk: int = 0, device_inputs=False,
                        want=("prediction", "reliability", "embedding", "nmd"), counts=True, out: dict | None = None):
        """``jg_predict_windows``.  With ``device_inputs`` the base / window buffers
        are raw device pointers (ints); outputs always land in host arrays - the caller's (``out``, from
        :meth:`host_outputs`: another thread may then read finished rows while the call runs) or fresh ones."""
        l_pad = self.row_length(fsize) if l_pad is None else int(l_pad)
        res = out if out is not None else self.host_outputs(n_win, want, counts)
        o = {k: res.get(k) for k in ("prediction", "reliability", "embedding", "nmd")}
        cnt = res.get("counts")
        loc = L.JG_PTR_DEVICE if device_inputs else L.JG_PTR_HOST
        lut = np.ascontiguousarray(lut, np.uint8)
        L.check(self.lib.jg_predict_windows(
            self.handle, _ptr(bases), int(n_bases), loc, _ptr(win_start), _ptr(win_len), loc, int(n_win),
            int(fsize), _ptr(lut), int(flags), l_pad, _ptr(o["prediction"]), _ptr(o["reliability"]),
            _ptr(o["embedding"]), _ptr(o["nmd"]), _ptr(cnt), L.JG_PTR_HOST, int(chunk), None),
            "jg_predict_windows")
        return res

    def host_outputs(self, n_win: int, want=("prediction", "reliability", "embedding", "nmd"), counts=True) -> dict:
        """Host arrays :meth:`predict_windows` fills: the model's outputs named in ``want`` (+ ``counts`` (n, 4) int32).
        The engine's progress mark is reset here: whoever polls ``device.windows_done()`` for the call these arrays are
        made for starts from 0, also on a reused engine."""
        self.dev.reset_progress()
        res = {k: v for k, v in self._host_outputs(n_win, want).items() if v is not None}
        if counts:
            res["counts"] = np.zeros((n_win, 4), np.int32)
        return res

    def predict_windows_raw(self, bases_ptr: int, n_bases: int, win_start_ptr: int, win_len_ptr: int,
                            n_win: int, fsize: int, lut, flags: int, l_pad: int, out_ptrs: dict,
                            counts_ptr: int = 0, chunk: int = 0) -> None:
        """``jg_predict_windows`` on raw DEVICE pointers for inputs and outputs (no host
        round trip); asynchronous on the engine stream - call ``HipDevice.sync()``."""
        lut = np.ascontiguousarray(lut, np.uint8)
        g = lambda k: C.c_void_p(out_ptrs[k]) if out_ptrs.get(k) else None  # noqa: E731
        L.check(self.lib.jg_predict_windows(
            self.handle, C.c_void_p(bases_ptr), int(n_bases), L.JG_PTR_DEVICE, C.c_void_p(win_start_ptr),
            C.c_void_p(win_len_ptr), L.JG_PTR_DEVICE, int(n_win), int(fsize), _ptr(lut), int(flags),
            int(l_pad), g("prediction"), g("reliability"), g("embedding"), g("nmd"),
            C.c_void_p(counts_ptr) if counts_ptr else None, L.JG_PTR_DEVICE, int(chunk), None),
            "jg_predict_windows")


class JaegerHipEngine:
    """Drop-in for ``InferModel`` (nnlib/inference.py:300-483) on one MI355X.

    ``path_dict`` keys as produced by ``AvailableModels`` (utils/misc.py:346-392):
    ``classes`` (yaml), ``project`` (yaml), ``graph`` (the SavedModel directory the reference executes,
    nnlib/inference.py:307-325), ``weights`` (Keras-3 ``.weights.h5``) or ``weights_npz`` (canonical names).  The layer
    plan comes from ``project.yaml``; the weights come from the graph's own variable bundle when there is one
    (``<name>_graph/variables``, mapped by object-graph order and variable names - ``weights.load_savedmodel_bundle``),
    else from the weights file (also when the bundle's keys cannot be mapped: loud warning).  ``trust_project=True``
    (``JAEGER_TRUST_PROJECT=1``, ``--trust-project``) skips the bundle and the census and takes the plan from the YAML and
    the weights from the file, as round 2 did.  When the graph also holds ``saved_model.pb`` its census (Conv2D count, dilations,
    batch-norm epsilons, GELU form, variable shapes) is compared with the plan first and a disagreement refuses the
    model: an engine that silently computed something else than the graph the reference runs is worse than none.
    Alternatively pass ``model_cfg`` + ``weights`` (canonical-name dict) directly.
    """

    def __init__(self, path_dict: dict | None = None, *, model_cfg: dict | None = None,
                 weights: dict[str, np.ndarray] | None = None, device_id: int = 0,
                 use_xla: bool = False, return_embedding: bool = False, chunk: int = 0,
                 precision: str | None = None, trust_project: bool | None = None):
        self.use_xla = use_xla                      # accepted for signature parity; no-op
        self.return_embedding = return_embedding
        self.chunk = chunk
        self.dust_masked_total = 0                  # bases soft-masked on the device by predict_windows(dust_records=...)
        self.class_map = None
        if path_dict is not None:
            self.class_map = self._load_class_map(path_dict.get("classes"))
            project = path_dict.get("project")
            if project is None:
                raise ValueError("JaegerHipEngine needs the model's *_project.yaml (layer plan source)")
            cfg = _yaml_load(Path(project).read_text()) or {}
            model_cfg = cfg.get("model")
            if weights is None:
                import os

                from .weights import load_weights
                if trust_project is None:
                    trust_project = os.environ.get("JAEGER_TRUST_PROJECT", "0") not in ("", "0")
                graph = path_dict.get("graph")
                have_file = path_dict.get("weights") is not None or path_dict.get("weights_npz") is not None
                if not trust_project and graph is not None and (Path(graph) / "saved_model.pb").exists() \
                        and (Path(graph) / "variables" / "variables.index").exists():
                    from .verify import verify_model
                    findings = verify_model(graph, build_plan(model_cfg))
                    if findings:
                        raise ValueError(
                            f"{graph}: the SavedModel the reference would execute disagrees with the layer plan compiled from "
                            f"{project}:\n  " + "\n  ".join(findings) +
                            ("\n(to run the project.yaml plan with the weights file anyway: trust_project=True / "
                             "JAEGER_TRUST_PROJECT=1 / --trust-project)" if have_file else ""))
                weights = load_weights(path_dict, build_plan(model_cfg), trust_project=bool(trust_project))
        if model_cfg is None or weights is None:
            raise ValueError("JaegerHipEngine: provide path_dict or model_cfg + weights")
        self.plan: ModelPlan = build_plan(model_cfg)
        sp = self.plan.string_processor
        if sp.get("shuffle") or sp.get("mutate"):
            # commands/predict.py:229-231 forwards these into the inference encoder,
            # which makes the reference's own output random; parity is undefined.
            import warnings
            warnings.warn("project.yaml sets string_processor.shuffle/mutate: the reference would "
                          "randomise the windows at inference; jaeger_amd runs deterministically "
                          "with both disabled", stacklevel=2)
        self.string_processor_config = sp
        if self.class_map is None and self.plan.class_label_map:
            cm = self.plan.class_label_map
            self.class_map = {"num_classes": len(cm), "class": [i["class"] for i in cm],
                              "index": [i["label"] for i in cm]}
        self.program = compile_plan(self.plan, weights)
        self.device = HipDevice(device_id)
        self.model = HipModel(self.device, self.program)
        if precision is not None:
            self.model.set_precision(precision)
        if return_embedding and self.model.widths["embedding"] == 0:
            raise ValueError("The selected model does not expose an 'embedding' output.")
        if sp.get("ngram_width", 3) == 6:                  # codon pairs: the plain codon table, combined on the device
            from .maps import CODON_ID
            self.lut = codon_lut(CODON_ID)
        else:
            self.lut = codon_lut(sp["codon_id"]) if sp.get("codon_id") else np.zeros(65, np.uint8)
        self.encode_flags = (2 if sp.get("masking") else 0) | (L.JG_ENC_DICODON if sp.get("ngram_width", 3) == 6 else 0)
        if self.plan.strands > 1:
            self.encode_flags |= L.JG_ENC_NUCLEOTIDE
            if sp.get("input_type_note"):
                import warnings
                warnings.warn(sp["input_type_note"] + "; jaeger_amd feeds the nucleotide strands the graph was built for",
                              stacklevel=2)

    @staticmethod
    def _load_class_map(path):
        """inference.py:411-421."""
        if path is None:
            return None
        with open(path) as f:
            cm = _yaml_load(f.read())["classes"]
        return {"num_classes": len(cm), "class": [i["class"] for i in cm], "index": [i["label"] for i in cm]}

    # -- InferModel.predict -----------------------------------------------------
    def predict(self, dataset, no_progress: bool = False) -> dict[str, np.ndarray]:
        """``dataset`` yields ``(inputs_dict, meta0..meta9)`` batches like the reference's
        tf.data pipeline (inference.py:341-373); ``inputs_dict["translated"]`` is the
        (B, 6, L) id tensor (float or int, 0 = invalid)."""
        acc: dict[str, list] = defaultdict(list)
        key = self.string_processor_config.get("input_type", "translated")
        for inputs, *meta in dataset:
            ids = np.asarray(inputs.get(key))
            if ids.ndim == 4:
                # seq_onehot=True batches (B, 6, L, D) (seqops/encode.py:297-302) and the nucleotide one-hot strands
                # (B, 2, L, 4) (:265-271): class c -> device id c + 1, an all-zero row (invalid codon / base, padding:
                # what Masking(0.0) masks, builder.py:850-852) -> 0
                if not self.string_processor_config.get("seq_onehot") and self.plan.strands == 1:
                    raise ValueError("one-hot batch given to a model that takes codon ids")
                ids = np.where(ids.any(axis=-1), ids.argmax(axis=-1) + 1, 0)
            elif ids.ndim != 3:
                raise ValueError("JaegerHipEngine.predict expects (B, 6, L) id or (B, 6, L, D) one-hot tensors")
            elif self.string_processor_config.get("seq_onehot"):
                raise ValueError("id batch given to a model that takes one-hot input")
            out = self.model.forward(ids.astype(np.uint8), chunk=self.chunk)
            for k, v in out.items():
                acc[k].append(v)
            for idx, m in enumerate(meta):
                acc[f"meta_{idx}"].append(np.asarray(m))
        return {k: np.concatenate(v, axis=0) for k, v in acc.items()}

    # -- fused path ---------------------------------------------------------------
    def predict_windows(self, bases: np.ndarray, win_start: np.ndarray, win_len: np.ndarray, fsize: int,
                        l_pad: int | None = None, pre_cased: bool = False,
                        want=("prediction", "reliability", "embedding", "nmd"),
                        dust_records: np.ndarray | None = None, out: dict | None = None) -> dict[str, np.ndarray]:
        """Encode + forward for windows given as (start, length) into ``bases``.  ``want`` limits the
        outputs copied back (the embedding and NMD vectors are 2.6 KB per window).  ``dust_records``: the record
        offsets of ``bases`` (n + 1 entries) - the uploaded copy is then DUST soft-masked on the GPU before it is
        encoded (what ``fragment_generator`` does per contig with pydustmasker, io.py:104-108); ``bases`` itself is
        left as it is.  ``out``: preallocated arrays from ``model.host_outputs`` - rows below ``device.windows_done()`` may
        be read by another thread while the call runs."""
        bases = np.ascontiguousarray(bases, np.uint8)
        ws = np.ascontiguousarray(win_start, np.int64)
        wl = np.ascontiguousarray(win_len, np.int32)
        flags = self.encode_flags | (1 if pre_cased else 0)
        if dust_records is None:
            return self.model.predict_windows(bases, bases.size, ws, wl, ws.size, fsize, self.lut, flags,
                                              l_pad, self.chunk, want=want, out=out)
        self.device.attach_records(dust_records)
        try:
            return self.model.predict_windows(bases, bases.size, ws, wl, ws.size, fsize, self.lut, flags,
                                              l_pad, self.chunk, want=want, out=out)
        finally:
            self.dust_masked_total += max(0, self.device.dust_masked())
            self.device.attach_records(None)

    def close(self):
        self.model.close()
        self.device.close()
