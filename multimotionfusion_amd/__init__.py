"""multimotionfusion_amd -- MI355X (gfx950) implementation of MultiMotionFusion's dense
tracking hot path behind the reference's own interfaces.

The arithmetic lives in libmmf_hip.so (hand-written HIP, C ABI in include/mmf_hip.h); this
package is the host-side mirror used by tests and bench.py.  Importing the kernel modules
without the built library raises: there is no CPU fallback.
"""
from . import _capi  # noqa: F401
from ._capi import LIB_PATH, MmfError  # noqa: F401

__all__ = ["LIB_PATH", "MmfError"]
