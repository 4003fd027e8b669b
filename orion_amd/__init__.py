"""MI355X-native ray-casting / path-tracing hot path behind Orion's scene format.

The compute path is the HIP library ``liborion_hip.so`` (C ABI in
``include/orion_hip.h``); this package is the host-side plumbing around it:
scene loading (``host``), the ctypes binding and scene upload (``api``), and
sample-range sharding over GPUs (``dist``).
"""
from . import host  # noqa: F401

__all__ = ["host"]
