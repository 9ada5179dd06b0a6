"""dns_slam_amd -- MI355X (gfx950) implementation of DNS-SLAM's volumetric-rendering hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd plumbing, torch.distributed);
the path itself is hand-written HIP behind the C ABI of ``include/dns_hip.h`` (``libdns_hip.so``).
There is no CPU fallback: every op raises if the HIP library is missing or a tensor is not on the GPU.

Mirrors of the reference interface (same names / argument meaning):
  tcnn_shim.Encoding / Network   <- tinycudann (reference models/pos_encoding.py, models/decoder.py)
  decoder.Decoder                <- reference models/decoder.py:7-27
  common.*                       <- reference utils/common.py (live render math)
  mapping.Mapper / tracking.Tracker optimise-step bodies <- reference slams/mapping.py, slams/tracking.py
"""
from . import _lib  # noqa: F401  (loads libdns_hip.so or raises)

__all__ = ["_lib"]
