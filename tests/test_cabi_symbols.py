"""CPU: libeva_hip.so loads and exports every symbol include/eva_hip.h declares (no compute)."""
import ctypes
import os
import re

from eva_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "eva_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(evah_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = ctypes.CDLL(backend.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/eva_hip.h but not exported"


def test_binding_covers_header():
    assert sorted(backend.EXPORTED_SYMBOLS) == _declared()


def test_abi_version_and_error_string():
    lib = backend.load()
    assert lib.evah_abi_version() == 1
    assert isinstance(lib.evah_last_error(), bytes)


def test_no_cpu_fallback_without_device():
    """Without a GPU the product path must fail loudly, not compute on the CPU."""
    if backend.device_count() > 0:
        return
    import pytest
    with pytest.raises(backend.EvaHipError, match="no HIP device"):
        backend.Context(1024, [0xFFFFFFFFFFFC001 - 0, 0xFFFFFFFFFFE8001][:2])
