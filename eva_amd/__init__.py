"""eva_amd — MI355X-native execution backend for EVA's CKKS execute() path.

`eva_amd.backend` is the ctypes view of the C-ABI (include/eva_hip.h); the HIP library has no
CPU fallback.  Importing this package does not import torch.
"""
__version__ = "0.1.0"
