"""Host-side mirror of class Model + ModelProjection (Core/Model/Model.h:120-300,
Core/Model/ModelProjection.h:37-77) over the C ABI: same method names and argument meaning,
GPUTexture* become torch CUDA tensors.  All work runs in libmmf_hip.so.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from ._capi import check, fptr
from .cudafuncs import Context, _p

MAX_VERTICES = 1024 * 1024  # Model::MAX_VERTICES (Model.cpp:119-126)

_TEX = {  # name -> (torch dtype, trailing shape)
    "index": (torch.int32, ()), "vertConf": (torch.float32, (4,)), "colorTime": (torch.float32, (4,)),
    "normRad": (torch.float32, (4,)), "image": (torch.uint8, (4,)), "vertexConf": (torch.float32, (4,)),
    "normalRadius": (torch.float32, (4,)), "time": (torch.int16, ()), "fillVertex": (torch.float32, (4,)),
    "fillNormal": (torch.float32, (4,)), "fillImage": (torch.uint8, (4,)), "depth": (torch.float32, ()),
}


def filterDepth(ctx: Context, depth, maxD, out=None):
    """MultiMotionFusion::filterDepth (MultiMotionFusion.cpp:897-904)."""
    out = torch.empty_like(depth) if out is None else out
    rows, cols = depth.shape
    check(ctx.lib.mmf_filter_depth(ctx.handle, _p(depth), cols, rows, float(maxD), _p(out)))
    return out


class Model:
    def __init__(self, ctx: Context, width, height, cx, cy, fx, fy, id=0, confidenceThresh=10.0,
                 max_surfels=MAX_VERTICES):
        self.ctx, self.width, self.height, self.id = ctx, width, height, id
        h = C.c_void_p()
        check(ctx.lib.mmf_model_create(ctx.handle, width, height, cx, cy, fx, fy, id, confidenceThresh, max_surfels,
                                       C.byref(h)))
        self.handle = h
        ctx._children.append(weakref.ref(self))

    # -- pose ----------------------------------------------------------------------------------
    def overridePose(self, pose):
        p = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        check(self.ctx.lib.mmf_model_set_pose(self.handle, fptr(p)))

    def getPose(self):
        p = np.zeros(16, np.float32)
        check(self.ctx.lib.mmf_model_get_pose(self.handle, fptr(p)))
        return p.reshape(4, 4)

    def confidenceThreshold(self):
        return float(self.ctx.lib.mmf_model_confidence_threshold(self.handle))

    def setConfidenceThreshold(self, v):
        check(self.ctx.lib.mmf_model_set_confidence_threshold(self.handle, float(v)))

    def setMaxDepth(self, v):
        check(self.ctx.lib.mmf_model_set_max_depth(self.handle, float(v)))

    def lastCount(self):
        n = C.c_uint(0)
        check(self.ctx.lib.mmf_model_count(self.handle, C.byref(n)))
        return n.value

    # -- passes (reference names) ----------------------------------------------------------------
    def initialise(self, rgb, depthRaw, depthFiltered, time, maxDepth):
        check(self.ctx.lib.mmf_model_initialise(self.handle, _p(rgb), _p(depthRaw), _p(depthFiltered), int(time),
                                                float(maxDepth)))

    def predictIndices(self, time, depthCutoff, timeDelta):
        check(self.ctx.lib.mmf_model_predict_indices(self.handle, int(time), float(depthCutoff), int(timeDelta)))

    def combinedPredict(self, depthCutoff, time, maxTime, timeDelta):
        check(self.ctx.lib.mmf_model_combined_predict(self.handle, float(depthCutoff), int(time), int(maxTime),
                                                      int(timeDelta)))

    def synthesizeDepth(self, depthCutoff, confThreshold, time, maxTime, timeDelta):
        """ModelProjection::synthesizeDepth (ModelProjection.cpp:275-335); result in texture("depth")."""
        check(self.ctx.lib.mmf_model_synthesize_depth(self.handle, float(depthCutoff), float(confThreshold), int(time),
                                                      int(maxTime), int(timeDelta)))

    def fuse(self, time, rgb, mask, depthRaw, depthFiltered, depthCutoff, weighting):
        check(self.ctx.lib.mmf_model_fuse(self.handle, int(time), _p(rgb), _p(mask), _p(depthRaw), _p(depthFiltered),
                                          float(depthCutoff), float(weighting)))

    def clean(self, time, timeDelta, depthCutoff, depthFiltered, mask, outlierCoeff=3.0):
        check(self.ctx.lib.mmf_model_clean(self.handle, int(time), int(timeDelta), float(depthCutoff),
                                           _p(depthFiltered), _p(mask), float(outlierCoeff)))

    def performFillIn(self, rawRGB, rawDepth, frameToFrameRGB, lost):
        check(self.ctx.lib.mmf_model_perform_fill_in(self.handle, _p(rawRGB), _p(rawDepth), int(bool(frameToFrameRGB)),
                                                     int(bool(lost))))

    def requiresFillIn(self, ratio=0.75):
        r = C.c_int(0)
        check(self.ctx.lib.mmf_model_requires_fill_in(self.handle, float(ratio), C.byref(r)))
        return bool(r.value)

    def downloadMap(self):
        """[count, 12] float32: pos+conf | colour24, unused, initTime, timestamp | normal+radius."""
        n = self.lastCount()
        out = np.zeros((max(n, 1), 12), np.float32)
        got = C.c_uint(0)
        check(self.ctx.lib.mmf_model_download_map(self.handle, fptr(out), n, C.byref(got)))
        return out[: got.value]

    def uploadMap(self, surfels):
        s = np.ascontiguousarray(np.asarray(surfels, np.float32).reshape(-1, 12))
        check(self.ctx.lib.mmf_model_upload_map(self.handle, fptr(s), s.shape[0]))

    # -- projections (GPUTexture getters of the reference) --------------------------------------------
    def texture(self, name):
        """Zero-copy torch view of a projection image living inside the model's HBM slab."""
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        check(self.ctx.lib.mmf_model_texture(self.handle, name.encode(), C.byref(ptr), C.byref(nbytes)))
        dt, tail = _TEX[name]
        return _as_tensor(ptr.value, nbytes.value, dt, (self.height, self.width) + tail, self.ctx.device, self)

    def close(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.mmf_model_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevMem:
    """__cuda_array_interface__ holder so torch can wrap memory owned by the C library."""

    def __init__(self, ptr, nbytes, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}
        self._owner = owner


def _as_tensor(ptr, nbytes, dtype, shape, device, owner):
    t = torch.as_tensor(_DevMem(ptr, nbytes, owner), device=f"cuda:{device}")
    return t.view(dtype).reshape(shape)
