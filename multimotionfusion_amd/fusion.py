"""Host-side mirror of class MultiMotionFusion (Core/MultiMotionFusion.h:78-160) for the
static-scene configuration: processFrame / getCurrPose / getTick / getBackgroundModel.
The orchestration itself is native code inside libmmf_hip.so (mmf_fusion_*)."""
import ctypes as C
import weakref

import numpy as np

from ._capi import check, fptr, mmf_fusion_config
from .cudafuncs import Context, _p
from .model import Model
from .odometry import RGBDOdometry


class MultiMotionFusion:
    def __init__(self, ctx: Context, width, height, cx, cy, fx, fy, **overrides):
        self.ctx, self.width, self.height = ctx, width, height
        cfg = mmf_fusion_config()
        check(ctx.lib.mmf_fusion_default_config(C.byref(cfg)))
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown MultiMotionFusion option {k}")
            setattr(cfg, k, v)
        self.config = cfg
        h = C.c_void_p()
        check(ctx.lib.mmf_fusion_create(ctx.handle, width, height, cx, cy, fx, fy, C.byref(cfg), C.byref(h)))
        self.handle = h
        ctx._children.append(weakref.ref(self))
        # borrowed views of the objects the fusion owns
        self._model = Model.__new__(Model)
        self._model.ctx, self._model.width, self._model.height, self._model.id = ctx, width, height, 0
        self._model.handle = C.c_void_p(ctx.lib.mmf_fusion_model(h))
        self._model.close = lambda: None
        self._odom = RGBDOdometry.__new__(RGBDOdometry)
        self._odom.ctx, self._odom.width, self._odom.height = ctx, width, height
        self._odom.handle = C.c_void_p(ctx.lib.mmf_fusion_odometry(h))
        self._odom.close = lambda: None

    def processFrame(self, rgb, depth, timestamp=0, inPose=None, weightMultiplier=1.0, bootstrap=False, initTransform=None,
                     icpRefine=True):
        """MultiMotionFusion::processFrame: rgb [H,W,3] uint8, depth [H,W] float32 (CUDA tensors).
        initTransform: `-init kp` (MultiMotionFusion.cpp:312-384) -- the 4x4 of Model::getLastTrackTransform, applied
        to the pose before the dense tracker (which then refines it when icpRefine, `-icp_refine`)."""
        if initTransform is not None:
            assert inPose is None and not bootstrap
            T = np.ascontiguousarray(np.asarray(initTransform, np.float32).reshape(16))
            check(self.ctx.lib.mmf_fusion_process_frame_init(self.handle, _p(rgb), _p(depth), int(timestamp), fptr(T),
                                                             int(bool(icpRefine)), float(weightMultiplier)))
            return
        pose = None
        if inPose is not None:
            pose = np.ascontiguousarray(np.asarray(inPose, np.float32).reshape(16))
        check(self.ctx.lib.mmf_fusion_process_frame(self.handle, _p(rgb), _p(depth), int(timestamp),
                                                    fptr(pose) if pose is not None else None,
                                                    float(weightMultiplier), int(bool(bootstrap))))

    def prefetchFrame(self, rgb, depth):
        """Start the depth filter and the input-side preparation of the NEXT frame on a second stream (they overlap
        the fusion of the current one).  The next processFrame call must get the same tensors, unchanged."""
        check(self.ctx.lib.mmf_fusion_prefetch_frame(self.handle, _p(rgb), _p(depth)))

    def reset(self):
        check(self.ctx.lib.mmf_fusion_reset(self.handle))

    def getCurrPose(self):
        p = np.zeros(16, np.float32)
        check(self.ctx.lib.mmf_fusion_get_pose(self.handle, fptr(p)))
        return p.reshape(4, 4)

    def getTick(self):
        return self.ctx.lib.mmf_fusion_tick(self.handle)

    def getBackgroundModel(self):
        return self._model

    def getFrameOdometry(self):
        self._odom._refresh_stats()
        return self._odom

    def close(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.mmf_fusion_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
