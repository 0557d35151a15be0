"""jaeger_amd - MI355X (gfx950) engine for the Jaeger ``predict`` hot path.

fragmenter -> 6-frame codon encoder -> Conv1D/BatchNorm/GlobalPool/Dense forward,
as hand-written HIP kernels behind the C-ABI in ``include/jaeger_hip.h``.
"""

__version__ = "0.1.0"
