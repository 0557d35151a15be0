"""The C-ABI library loads and exports every symbol include/jaeger_hip.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _declared_symbols():
    text = (ROOT / "include" / "jaeger_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from jaeger_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in jaeger_hip.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes prototype in jaeger_amd/_lib.py"
    assert lib.jg_abi_version() == 1
    assert lib.jg_sizeof(0) == ctypes.sizeof(_lib.JgOp) and lib.jg_sizeof(1) == ctypes.sizeof(_lib.JgStage)


def test_no_cpu_fallback(monkeypatch, tmp_path):
    """Without the built extension the package must fail loudly, not fall back."""
    import pytest
    from jaeger_amd import _lib
    monkeypatch.setenv("JAEGER_HIP_LIB", str(tmp_path / "missing.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.JaegerHipError, match="not built|not found"):
        _lib.load()


def test_product_never_imports_the_oracle():
    for py in (ROOT / "jaeger_amd").rglob("*.py"):
        src = py.read_text()
        assert "import oracle" not in src and "from oracle" not in src, py
