"""ctypes binding of ``libjaeger_hip.so`` (C-ABI in ``include/jaeger_hip.h``).

There is no CPU fallback: :func:`load` raises if the library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C jaeger_amd/csrc``) or cannot be loaded.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

JG_MAX_STAGES = 8
JG_MAX_BUFS = 6
JG_MAX_VECS = 12
JG_PTR_HOST, JG_PTR_DEVICE = 0, 1
JG_BUF_NONE, JG_BUF_IDS = -1, -2
JG_OPT_STREAM_BYTES = 1
JG_OPT_CONV_PC = 2
JG_OPT_TERMINI_EXACT = 3
JG_OPT_TERMINI_REPORT_MIN = 8
JG_OPT_DUST_ON_COPY_STREAM = 4
JG_OPT_TABLE_NET_LDS = 5
JG_OPT_RESET_PROGRESS = 6
JG_OPT_FUSE_RESBLOCK = 7
JG_OPT_STREAM_PRIORITY = 9
JG_COL_STRING, JG_COL_INT, JG_COL_FLOAT, JG_COL_BOOL, JG_COL_SPANS = 0, 1, 2, 3, 4     # jg_table_format column kinds
JG_STAT_STREAM_GROUPS, JG_STAT_STREAM_BYTES, JG_STAT_PEAK_DEVICE_BASES, JG_STAT_DUST_MASKED, JG_STAT_WINDOWS_DONE = 1, 2, 3, 4, 5
JG_MSTAT_CONVS, JG_MSTAT_CONVS_F16X3, JG_MSTAT_LAYOUT_CONVERSIONS, JG_MSTAT_SMALL_FUSED = 0, 1, 2, 3

# jg_op_kind
OP_CONV, OP_MASK, OP_POOL, OP_DENSE, OP_ELTWISE, OP_NMD_FINAL, OP_OODSIG, OP_MAXPOOL1D, OP_FRAMESUM, OP_STRANDS, OP_EMBED, OP_VECMAX = range(1, 13)
# jg_stage_kind
ST_NONE, ST_BIAS, ST_BN, ST_DYT, ST_ADD, ST_ACT, ST_NMD, ST_MASKMUL, ST_LN = range(9)
# jg_act
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_RELU, ACT_TANH, ACT_SIGMOID = range(6)
MASK_ANY, MASK_MAJORITY, MASK_STRICT = range(3)
PAD_VALID, PAD_SAME = 0, 1
POOL_MAX, POOL_AVG, POOL_MAX_NOMASK = 0, 1, 2
MERGE_AVERAGE, MERGE_SUM, MERGE_MAX, MERGE_CONCAT = 0, 1, 2, 3            # jg_merge_kind (OP_STRANDS)
JG_ENC_PRECASED, JG_ENC_CASE_SENSITIVE, JG_ENC_NUCLEOTIDE, JG_ENC_DICODON = 1, 2, 4, 8   # jg_encode / jg_predict_windows soft_mask bits
# vector slot convention (jg_api.hip: jg_model_vec_width)
VEC_EMBEDDING, VEC_NMD, VEC_PREDICTION, VEC_RELIABILITY, VEC_SCRATCH0 = 0, 1, 2, 3, 4


class JgStage(C.Structure):
    _fields_ = [("kind", C.c_int32), ("arg", C.c_int32),
                ("p0", C.c_int64), ("p1", C.c_int64), ("p2", C.c_int64), ("p3", C.c_int64),
                ("f0", C.c_float), ("pad_", C.c_int32)]


class JgOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_buf", C.c_int32), ("out_buf", C.c_int32),
                ("in_mask", C.c_int32), ("out_mask", C.c_int32),
                ("k", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("stride", C.c_int32), ("dilation", C.c_int32), ("padding", C.c_int32),
                ("mask_mode", C.c_int32), ("in_vec", C.c_int32), ("out_vec", C.c_int32),
                ("vec_off", C.c_int32), ("arg", C.c_int32),
                ("w_off", C.c_int64), ("b_off", C.c_int64),
                ("f0", C.c_float), ("n_stages", C.c_int32),
                ("stages", JgStage * JG_MAX_STAGES)]


class JaegerHipError(RuntimeError):
    """A libjaeger_hip call returned a negative jg_status."""


LIB_NAME = "libjaeger_hip.so"
_lib = None


def lib_path() -> Path:
    return Path(os.environ.get("JAEGER_HIP_LIB", Path(__file__).resolve().parent / LIB_NAME))


#: every symbol include/jaeger_hip.h declares -> (restype, argtypes)
_u8p, _i32p, _i64p, _f32p, _vp = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_float), C.c_void_p)
SYMBOLS = {
    "jg_abi_version": (C.c_int, []),
    "jg_sizeof": (C.c_int, [C.c_int]),
    "jg_last_error": (C.c_char_p, []),
    "jg_engine_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "jg_engine_destroy": (C.c_int, [_vp]),
    "jg_engine_sync": (C.c_int, [_vp]),
    "jg_engine_set_option": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "jg_engine_get_stat": (C.c_int64, [_vp, C.c_int]),
    "jg_model_create": (C.c_int, [_vp, C.POINTER(JgOp), C.c_int, _vp, C.c_int64, C.c_int32, C.POINTER(_vp)]),
    "jg_model_destroy": (C.c_int, [_vp]),
    "jg_model_set_precision": (C.c_int, [_vp, C.c_int]),
    "jg_model_get_precision": (C.c_int, [_vp]),
    "jg_model_get_stat": (C.c_int64, [_vp, C.c_int]),
    "jg_model_describe": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "jg_encode": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_int64, C.c_int32, _vp,
                            C.c_int32, C.c_int32, _vp, _vp, C.c_int, _vp]),
    "jg_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp, C.c_int,
                             C.c_int32, _vp]),
    "jg_predict_windows": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, C.c_int, C.c_int64, C.c_int32,
                                     _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                     C.c_int32, _vp]),
    "jg_model_vec_width": (C.c_int, [_vp, C.c_int]),
    "jg_model_flops_per_window": (C.c_double, [_vp, C.c_int32]),
    "jg_dev_alloc": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "jg_dev_free": (C.c_int, [_vp, _vp]),
    "jg_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "jg_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_int64]),
    "jg_timer_start": (C.c_int, [_vp, _vp]),
    "jg_timer_stop_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "jg_profile_enable": (C.c_int, [_vp, C.c_int]),
    "jg_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "jg_profile_read_class": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "jg_box_calibrate": (C.c_int, [_vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), _vp]),
    "jg_terminal_repeats": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, C.c_int64, C.c_int32, _vp]),
    "jg_viterbi_decode": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, C.c_int64, _vp, _vp]),
    "jg_dust_mask": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "jg_engine_set_dust": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.c_int32]),
    "jg_dust_mask_device": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int, C.c_int64, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_int64), _vp]),
    "jg_fasta_count": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "jg_fasta_index": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.POINTER(C.c_int64)]),
    "jg_fasta_parse": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.POINTER(C.c_int64),
                                 C.POINTER(C.c_int64)]),
    "jg_fasta_scan": (C.c_int, [_vp, C.c_int64, C.c_int32, C.POINTER(_vp), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                C.POINTER(C.c_int64)]),
    "jg_fasta_fill": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "jg_fasta_scan_free": (None, [_vp]),
    "jg_table_format": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp, C.c_int64, C.c_int32, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "jg_table_write": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "jg_names_unique": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]),
    "jg_table_free": (None, [_vp]),
    "jg_segment_mean_var": (C.c_int, [_vp, C.c_int64, C.c_int32, _vp, _vp, C.c_int64, _vp, _vp, C.c_int32]),
    "jg_segment_mean_1d": (C.c_int, [_vp, C.c_int32, C.c_int64, _vp, _vp, C.c_int64, _vp, C.c_int32]),
    "jg_run_summaries": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int64, _vp, C.c_int32, C.c_int32, C.POINTER(_vp),
                                   C.POINTER(C.c_int64)]),
}


def load():
    """Load the shared library (once) and attach prototypes.  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise JaegerHipError(
            f"{path} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C jaeger_amd/csrc`). "
            "jaeger_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; importing it first makes the loader
    # resolve our DT_NEEDED to that same copy, so torch tensors, RCCL and this library
    # share ONE HIP runtime in the process (torch is plumbing here, never compute).  A
    # single-GPU command-line run needs no torch at all: JAEGER_HIP_TORCH=0 (set by
    # `python -m jaeger_amd` outside torchrun) skips the ~2 s import and binds the ROCm copy.
    import sys
    want_torch = os.environ.get("JAEGER_HIP_TORCH", "1") != "0" or "torch" in sys.modules \
        or int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("JAEGER_SHARDED") == "1"
    if want_torch:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(str(path))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)            # AttributeError if the .so misses a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.jg_sizeof(0) != C.sizeof(JgOp) or lib.jg_sizeof(1) != C.sizeof(JgStage):
        raise JaegerHipError("jg_op / jg_stage layout mismatch between jaeger_hip.h and _lib.py")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().jg_last_error().decode(errors="replace")
        raise JaegerHipError(f"{what or 'libjaeger_hip'} failed ({rc}): {msg}")
