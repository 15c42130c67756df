"""Make sure a process holds exactly ONE HIP runtime.

PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7) under torch/lib,
while libeva_hip.so links the system one (/opt/rocm/lib, same SONAME).  If eva_amd is imported
before torch, the dynamic loader ends up with two runtimes and the second one sees no device.
Pre-loading torch's copy here (located without importing torch) makes both resolve to the same
library whatever the import order; without torch installed the system runtime is used.
"""
import ctypes
import importlib.util
import os
import sys


def _preload():
    if "torch" in sys.modules:
        return None  # torch already brought its runtime; ours will bind to it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            return ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            return None
    return None


_handle = _preload()
