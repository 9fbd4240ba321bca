"""MI355X-native hot path of faris-k/self-supervised-wafermaps (import as `ssl_wafermap_amd`).

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed over RCCL); all
device work of the path runs in hand-written HIP kernels behind the C ABI of include/wafer_hip.h
(libwafer_hip.so, built by build.py).  There is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
