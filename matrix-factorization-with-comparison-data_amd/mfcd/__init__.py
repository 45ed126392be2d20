"""mfcd — host side of the MI355X triplet-comparison matrix-factorisation hot path.

`_lib`     ctypes binding of libmfcd_hip.so (C-ABI in include/mfcd.h); raises if it is not built.
`batching` DataLoader -> packed records + per-epoch order, RNG-stream compatible with the reference.
`engine`   fused training epochs / evaluation on the device, in place on the caller's model + Adam.
`metrics`  dense UV^T reconstruction / alignment metrics from the MFMA pass.
"""
from . import _lib, batching, engine, metrics  # noqa: F401

__all__ = ["_lib", "batching", "engine", "metrics"]
