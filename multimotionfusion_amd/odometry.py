"""Host-side mirror of class RGBDOdometry (Core/Utils/RGBDOdometry.h:31-137) over the C ABI.

Same method names, argument order and meaning as the reference class; GPUTexture* become torch
CUDA tensors (RGBA32F [H, W, 4] float32 predictions, [H, W, C] uint8 colour), Eigen vectors /
matrices become numpy arrays.  All arithmetic runs in libmmf_hip.so.
"""
import ctypes as C
import math
import weakref

import numpy as np
import torch

from ._capi import check, fptr, mmf_odom_stats, MMF_NUM_PYRS
from .cudafuncs import Context, _p, _step


class RGBDOdometry:
    NUM_PYRS = MMF_NUM_PYRS  # RGBDOdometry.h:72

    def __init__(self, ctx: Context, width, height, cx, cy, fx, fy, maskID=0, distThresh=0.10,
                 angleThresh=math.sin(20.0 * 3.14159254 / 180.0)):  # RGBDOdometry.h:34-36
        self.ctx = ctx
        self.width, self.height = width, height
        self.maskID = maskID
        h = C.c_void_p()
        check(ctx.lib.mmf_odom_create(ctx.handle, width, height, cx, cy, fx, fy, distThresh, angleThresh,
                                      C.byref(h)))
        self.handle = h
        ctx._children.append(weakref.ref(self))
        self._refresh_stats()

    # -- data preparation ---------------------------------------------------------------------
    def buildDepthPyramid(self, depth):
        """Model::generateCUDATextures (Model.cpp:359-388): level-0 depth -> internal pyramid."""
        check(self.ctx.lib.mmf_odom_build_depth_pyramid(self.handle, _p(depth), _step(depth)))

    def initICP(self, depthPyramid=None, maskPyramid=None, depthCutoff=15.0, predictedVertices=None,
                predictedNormals=None):
        """Both overloads of RGBDOdometry::initICP (RGBDOdometry.h:41-44)."""
        if predictedVertices is not None:
            check(self.ctx.lib.mmf_odom_init_icp_from_prediction(self.handle, _p(predictedVertices),
                                                                 _p(predictedNormals), depthCutoff))
            return
        if depthPyramid is None:
            check(self.ctx.lib.mmf_odom_init_icp(self.handle, None, None, depthCutoff))
            return
        ptrs = (C.c_void_p * MMF_NUM_PYRS)(*[d.data_ptr() for d in depthPyramid])
        steps = (C.c_size_t * MMF_NUM_PYRS)(*[_step(d) for d in depthPyramid])
        check(self.ctx.lib.mmf_odom_init_icp(self.handle, ptrs, steps, depthCutoff))

    def initICPModel(self, predictedVertices, predictedNormals, depthCutoff, modelPose):
        pose = np.ascontiguousarray(np.asarray(modelPose, np.float32).reshape(16))
        check(self.ctx.lib.mmf_odom_init_icp_model(self.handle, _p(predictedVertices), _p(predictedNormals),
                                                   depthCutoff, fptr(pose)))

    def initRGB(self, rgb):
        check(self.ctx.lib.mmf_odom_init_rgb(self.handle, _p(rgb), _step(rgb), rgb.shape[2]))

    def initRGBModel(self, rgb):
        check(self.ctx.lib.mmf_odom_init_rgb_model(self.handle, _p(rgb), _step(rgb), rgb.shape[2]))

    def initFirstRGB(self, rgb):
        check(self.ctx.lib.mmf_odom_init_first_rgb(self.handle, _p(rgb), _step(rgb), rgb.shape[2]))

    # -- optimisation -------------------------------------------------------------------------
    def getIncrementalTransformation(self, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3,
                                     icpErrorSurface=None, rgbErrorSurface=None):
        """RGBDOdometry.h:56-58.  Returns the updated (trans[3], rot[3,3]) instead of mutating."""
        t = np.ascontiguousarray(np.asarray(trans, np.float32).reshape(3)).copy()
        r = np.ascontiguousarray(np.asarray(rot, np.float32).reshape(9)).copy()
        check(self.ctx.lib.mmf_odom_get_incremental_transformation(
            self.handle, fptr(t), fptr(r), int(bool(rgbOnly)), float(icpWeight), int(bool(pyramid)),
            int(bool(fastOdom)), int(bool(so3)), _p(icpErrorSurface), _p(rgbErrorSurface)))
        self._refresh_stats()
        return t, r.reshape(3, 3)

    def getCovariance(self):
        cov = np.zeros(36, np.float64)
        check(self.ctx.lib.mmf_odom_get_covariance(self.handle, cov.ctypes.data_as(C.POINTER(C.c_double))))
        return cov.reshape(6, 6)

    def _refresh_stats(self):
        s = mmf_odom_stats()
        check(self.ctx.lib.mmf_odom_get_stats(self.handle, C.byref(s)))
        self.lastICPError, self.lastICPCount = s.lastICPError, s.lastICPCount
        self.lastRGBError, self.lastRGBCount = s.lastRGBError, s.lastRGBCount
        self.lastSO3Error, self.lastSO3Count = s.lastSO3Error, s.lastSO3Count
        self.lastA = np.array(s.lastA, np.float64).reshape(6, 6)
        self.lastb = np.array(s.lastb, np.float64)
        self.iterations_run, self.so3_iterations_run = s.iterations_run, s.so3_iterations_run

    # -- test / bench hooks -------------------------------------------------------------------
    def sparseWalk(self):
        """(correspondences accepted outside the rectangle an object model's one-launch chain walks, was the last chain walked
        by the model's extents) -- the first is counted in checking mode only (mmf_debug_set_sparse_check)."""
        outside, by_extent, rect = C.c_uint(0), C.c_int(0), (C.c_int * 9)()
        check(self.ctx.lib.mmf_debug_odom_sparse_outside(self.handle, C.byref(outside), C.byref(by_extent), rect))
        self.sparseRect = list(rect)  # level-0 rectangle {x0, y0, width, rows}, most passes, derived, lanes needed at levels 0 / 1 / 2
        return int(outside.value), bool(by_extent.value)

    _DTYPES = {"vmaps_curr": (torch.float32, 3), "nmaps_curr": (torch.float32, 3),
               "vmaps_g_prev": (torch.float32, 3), "nmaps_g_prev": (torch.float32, 3),
               "last_depth": (torch.float32, 1), "next_depth": (torch.float32, 1), "depth_pyr": (torch.float32, 1),
               "last_image": (torch.uint8, 1), "next_image": (torch.uint8, 1), "last_next_image": (torch.uint8, 1),
               "dIdx": (torch.int16, 1), "dIdy": (torch.int16, 1)}

    def download(self, name, level):
        """Copy of an internal pyramid buffer as a numpy array (parity tests)."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        check(self.ctx.lib.mmf_odom_buffer(self.handle, name.encode(), level, C.byref(ptr), C.byref(nbytes)))
        host = np.empty(nbytes.value, np.uint8)
        check(self.ctx.lib.mmf_odom_download(self.handle, name.encode(), level, C.c_void_p(host.ctypes.data),
                                             nbytes.value))
        cols, rows = self.width >> level, self.height >> level
        if name in ("cloud", "cloud4", "prev_packed"):  # per-pixel records
            return host.view(np.float32).reshape(rows, cols, {"cloud": 3, "cloud4": 4, "prev_packed": 6}[name])
        if name == "corres":
            return host.reshape(rows, cols, 16)
        dt, planes = self._DTYPES[name]
        npdt = {torch.float32: np.float32, torch.uint8: np.uint8, torch.int16: np.int16}[dt]
        return host.view(npdt).reshape(planes * rows, cols)

    def enableTiming(self, mode=2):
        """Measurement mode: 0 off, 1 the duration of the whole Gauss-Newton chain, 2 also of each of its kernels
        (dispatch timestamps via HIP events; mode 2 perturbs the chain it sits in)."""
        check(self.ctx.lib.mmf_odom_enable_timing(self.handle, int(mode)))

    def getTiming(self):
        from ._capi import mmf_odom_timing
        t = mmf_odom_timing()
        check(self.ctx.lib.mmf_odom_get_timing(self.handle, C.byref(t)))
        out = {"chain_us": t.chain_us_sum / max(t.chains, 1), "chains": t.chains}
        for lvl in range(3):
            for name in ("producer", "rgb_step"):
                n = getattr(t, name + "_launches")[lvl]
                out[f"{name}_l{lvl}"] = {"launches": n, "mean_us": getattr(t, name + "_us_sum")[lvl] / max(n, 1),
                                         "min_us": getattr(t, name + "_us_min")[lvl]}
        return out

    def timeIcpKernel(self, level, reps, variant=0):
        us = C.c_float(0)
        check(self.ctx.lib.mmf_odom_time_icp_kernel(self.handle, level, reps, variant, C.byref(us)))
        return us.value

    def close(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.mmf_odom_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
