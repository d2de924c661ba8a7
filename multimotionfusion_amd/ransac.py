"""RigidRANSAC (Core/Utils/RigidRANSAC.h) through the C ABI: keypoint-based pose initialisation, host code."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check


def _pts(a):
    a = np.ascontiguousarray(a, np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def fit(p0, p1, mask=None):
    """fit() of RigidRANSAC.cpp:73-120: least-squares T_01 (4x4) with p0 ~ T_01 p1."""
    p0, p1 = _pts(p0), _pts(p1)
    T = np.zeros((4, 4), np.float32)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    check(_capi.load().mmf_rigid_fit(_ptr(p0), _ptr(p1), p0.shape[0], None if m is None else _ptr(m), _ptr(T)))
    return T


def apply(T, p0, p1):
    """apply() of RigidRANSAC.cpp:122-126: || p0 - T p1 || per row."""
    p0, p1 = _pts(p0), _pts(p1)
    T = np.ascontiguousarray(T, np.float32)
    d = np.zeros(p0.shape[0], np.float32)
    check(_capi.load().mmf_rigid_apply(_ptr(T), _ptr(p0), _ptr(p1), p0.shape[0], _ptr(d)))
    return d


class RigidRANSAC:
    """RigidRANSAC(iterations, inlier_threshold, inlier_fraction).estimate(p0, p1, mask) -> (T, error, inlier)."""

    def __init__(self, iterations, inlier_threshold, inlier_fraction):
        self.lib = _capi.load()
        h = C.c_void_p()
        check(self.lib.mmf_ransac_create(int(iterations), float(inlier_threshold), float(inlier_fraction), C.byref(h)))
        self.handle = h

    def estimate(self, p0, p1, mask=None):
        p0, p1 = _pts(p0), _pts(p1)
        n = p0.shape[0]
        T = np.zeros((4, 4), np.float32)
        err = C.c_float()
        inl = np.zeros(n, np.uint8)
        has = C.c_int()
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        check(self.lib.mmf_ransac_estimate(self.handle, _ptr(p0), _ptr(p1), n, None if m is None else _ptr(m), _ptr(T),
                                           C.byref(err), _ptr(inl), C.byref(has)))
        return T, err.value, (inl.astype(bool) if has.value else None)

    def __del__(self):
        try:
            if self.handle:
                self.lib.mmf_ransac_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
