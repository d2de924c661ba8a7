"""Host-side mirror of class MultiMotionFusion (Core/MultiMotionFusion.h:78-300): processFrame (one or many
rigid-body models), predict, getModels / getBackgroundModel / getTextures, the runtime setters, exportPoses.
The orchestration itself is native code inside libmmf_hip.so (mmf_fusion_*)."""
import ctypes as C
import weakref

import numpy as np

from ._capi import check, fptr, mmf_frame, mmf_fusion_config, mmf_segmentation, mmf_segmentation_model
from .cudafuncs import Context, _p
from .model import Model
from .odometry import RGBDOdometry


class HostFrame:
    """One frame in host memory (FrameData::rgb / depth as numpy arrays) with the arrays' addresses taken once: numpy's
    `.ctypes.data` costs ~2 us per access, 9 us per call for the four pointers of a frame and its successor."""
    __slots__ = ("rgb", "depth", "rgb_ptr", "depth_ptr")

    def __init__(self, rgb, depth):
        # raw addresses reach a C memcpy / DMA: the wrong dtype or stride would be an out-of-bounds host read (and an
        # `assert` disappears under python -O), so convertible inputs are converted and anything else is refused
        rgb, depth = np.ascontiguousarray(rgb), np.ascontiguousarray(depth)
        if rgb.dtype != np.uint8 or depth.dtype != np.float32:
            raise TypeError(f"HostFrame: rgb must be uint8 and depth float32 (metres), got {rgb.dtype} / {depth.dtype}")
        if rgb.ndim != 3 or rgb.shape[2] != 3 or depth.shape != rgb.shape[:2]:
            raise TypeError(f"HostFrame: rgb (H, W, 3) and depth (H, W) expected, got {rgb.shape} / {depth.shape}")
        self.rgb, self.depth = rgb, depth  # (kept alive with the addresses)
        self.rgb_ptr, self.depth_ptr = rgb.ctypes.data, depth.ctypes.data


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
        self._model = self._borrow_model(ctx.lib.mmf_fusion_model(h))
        self._odom = self._borrow_odom(ctx.lib.mmf_fusion_odometry(h))

    def _borrow_model(self, handle):
        m = Model.__new__(Model)
        m.ctx, m.width, m.height = self.ctx, self.width, self.height
        m.handle = C.c_void_p(handle)
        m.id = self.ctx.lib.mmf_model_id(m.handle)
        m.close = lambda: None
        return m

    def _borrow_odom(self, handle):
        o = RGBDOdometry.__new__(RGBDOdometry)
        o.ctx, o.width, o.height = self.ctx, self.width, self.height
        o.handle = C.c_void_p(handle)
        o.close = lambda: None
        return o

    def processFrame(self, rgb, depth, timestamp=0, inPose=None, weightMultiplier=1.0, bootstrap=False, initTransform=None,
                     icpRefine=True, mask=None, hasNewLabel=False, modelData=None, initTransforms=None, next=None):
        """MultiMotionFusion::processFrame: rgb [H,W,3] uint8, depth [H,W] float32 (CUDA tensors).
        initTransform: `-init kp` (MultiMotionFusion.cpp:312-384) -- the 4x4 of Model::getLastTrackTransform, applied
        to the pose before the dense tracker (which then refines it when icpRefine, `-icp_refine`).
        mask / hasNewLabel / modelData: the SegmentationResult of this frame when enable_multiple_models is set
        (fullSegmentation as a CUDA uint8 [H,W] tensor of model ids; modelData: dicts with id, super_pixel_count,
        avg_confidence, depth_mean, depth_std in list order).  initTransforms: one 4x4 per active model.
        next: (rgb, depth) the NEXT call will be given -- prefetchFrame folded into this call: the next frame's
        sensor-side preparation is enqueued while this one waits for its pose (mmf_frame::next_rgb / next_depth)."""
        if mask is not None or initTransforms is not None or next is not None:
            fr = mmf_frame()
            fr.rgb, fr.depth, fr.timestamp = _p(rgb), _p(depth), int(timestamp)
            if next is not None:
                fr.next_rgb, fr.next_depth = _p(next[0]), _p(next[1])
            if initTransform is not None:
                assert initTransforms is None
                initTransforms = [initTransform]
            fr.weight_multiplier, fr.bootstrap, fr.icp_refine = float(weightMultiplier), int(bool(bootstrap)), int(bool(icpRefine))
            keep = []
            if inPose is not None:
                pose = np.ascontiguousarray(np.asarray(inPose, np.float32).reshape(16))
                keep.append(pose)
                fr.in_pose = fptr(pose)
            if initTransforms is not None:
                T = np.ascontiguousarray(np.asarray(initTransforms, np.float32).reshape(-1, 16))
                keep.append(T)
                fr.init_transforms, fr.n_init_transforms = fptr(T), T.shape[0]
            if mask is not None:
                seg = mmf_segmentation()
                seg.mask, seg.has_new_label = _p(mask), int(bool(hasNewLabel))
                if modelData:
                    arr = (mmf_segmentation_model * len(modelData))()
                    for i, d in enumerate(modelData):
                        arr[i].id, arr[i].super_pixel_count = int(d["id"]), int(d["super_pixel_count"])
                        arr[i].avg_confidence = float(d["avg_confidence"])
                        arr[i].depth_mean, arr[i].depth_std = float(d["depth_mean"]), float(d["depth_std"])
                    seg.n_models, seg.model_data = len(modelData), arr
                    keep.append(arr)
                keep.append(seg)
                fr.segmentation = C.pointer(seg)
            check(self.ctx.lib.mmf_fusion_process_frame_ex(self.handle, C.byref(fr)))
            return
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

    def getModels(self):
        """std::list<std::shared_ptr<Model>>& getModels(): the active models in list order (index 0 = global)."""
        lib = self.ctx.lib
        return [self._borrow_model(lib.mmf_fusion_model_at(self.handle, i)) for i in range(lib.mmf_fusion_num_models(self.handle))]

    def getInactiveModels(self):
        lib = self.ctx.lib
        return [self._borrow_model(lib.mmf_fusion_inactive_model_at(self.handle, i))
                for i in range(lib.mmf_fusion_num_inactive_models(self.handle))]

    def getModelOdometry(self, index):
        o = self._borrow_odom(self.ctx.lib.mmf_fusion_odometry_at(self.handle, index))
        o._refresh_stats()
        return o

    def getNextModelID(self):
        return self.ctx.lib.mmf_fusion_next_model_id(self.handle)

    def scheduleDeactivation(self, model_id):
        check(self.ctx.lib.mmf_fusion_schedule_deactivation(self.handle, int(model_id)))

    def processFrameHost(self, rgb, depth=None, timestamp=0, mask=None, hasNewLabel=False, inPose=None, weightMultiplier=1.0,
                         bootstrap=False, next=None):
        """processFrame(const FrameData&) with HOST numpy arrays: staged through pinned buffers and uploaded inside.
        next = (rgb, depth) of the NEXT call (C-contiguous uint8 / float32 arrays, the very objects that call will pass):
        uploaded and prepared during this one (mmf_fusion_process_frame_host_next).  A frame may also be handed in as a
        HostFrame (rgb=HostFrame, next=HostFrame): the arrays' addresses are then taken once, not per call."""
        cur = rgb if isinstance(rgb, HostFrame) else HostFrame(rgb, depth)
        nxt = None if next is None else (next if isinstance(next, HostFrame) else HostFrame(*next))
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        pose = np.ascontiguousarray(np.asarray(inPose, np.float32).reshape(16)) if inPose is not None else None
        check(self.ctx.lib.mmf_fusion_process_frame_host_next(
            self.handle, cur.rgb_ptr, cur.depth_ptr, m.ctypes.data if m is not None else None, int(bool(hasNewLabel)),
            int(timestamp), fptr(pose) if pose is not None else None, float(weightMultiplier), int(bool(bootstrap)),
            nxt.rgb_ptr if nxt is not None else None, nxt.depth_ptr if nxt is not None else None))

    def predict(self):
        check(self.ctx.lib.mmf_fusion_predict(self.handle))

    def setShard(self, rank, world):
        """Per-rigid-body shard: this process runs the models whose list index k has k % world == rank."""
        check(self.ctx.lib.mmf_fusion_set_shard(self.handle, int(rank), int(world)))

    def ownsModel(self, index):
        return bool(self.ctx.lib.mmf_fusion_owns_model(self.handle, int(index)))

    def setModelPose(self, index, pose):
        p = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        check(self.ctx.lib.mmf_fusion_set_model_pose(self.handle, int(index), fptr(p)))

    def lastTimings(self):
        a, b = C.c_double(), C.c_double()
        check(self.ctx.lib.mmf_fusion_last_timings(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def setTick(self, val):
        check(self.ctx.lib.mmf_fusion_set_tick(self.handle, int(val)))

    def _set(self, name, val):
        check(getattr(self.ctx.lib, "mmf_fusion_set_" + name)(self.handle, val))

    def setRgbOnly(self, v): self._set("rgb_only", int(bool(v)))
    def setIcpWeight(self, v): self._set("icp_weight", float(v))
    def setOutlierCoefficient(self, v): self._set("outlier_coefficient", float(v))
    def setPyramid(self, v): self._set("pyramid", int(bool(v)))
    def setFastOdom(self, v): self._set("fast_odom", int(bool(v)))
    def setSo3(self, v): self._set("so3", int(bool(v)))
    def setFrameToFrameRGB(self, v): self._set("frame_to_frame_rgb", int(bool(v)))
    def setDepthCutoff(self, v): self._set("depth_cutoff", float(v))
    def setConfidenceThreshold(self, v): self._set("confidence_threshold", float(v))
    def setEnableMultipleModels(self, v): self._set("enable_multiple_models", int(bool(v)))

    def getConfig(self):
        cfg = mmf_fusion_config()
        check(self.ctx.lib.mmf_fusion_get_config(self.handle, C.byref(cfg)))
        return cfg

    def getTexture(self, name):
        """getTextures()[name]: "RGB", "DEPTH_METRIC", "DEPTH_METRIC_FILTERED", "MASK" as a CUDA tensor view."""
        import torch
        from .model import _as_tensor
        p, b = C.c_void_p(), C.c_size_t()
        check(self.ctx.lib.mmf_fusion_texture(self.handle, name.encode(), C.byref(p), C.byref(b)))
        dt = torch.uint8 if name in ("RGB", "MASK") else torch.float32
        shape = (self.height, self.width, 3) if name == "RGB" else (self.height, self.width)
        return _as_tensor(p.value, b.value, dt, shape, self.ctx.device, self)

    def getErrorTexture(self, index, which="icp"):
        """Model::getICPErrorTexture / getRGBErrorTexture of the index-th active model (CUDA float32 [H,W] view)."""
        import torch
        from .model import _as_tensor
        p = C.c_void_p()
        check(self.ctx.lib.mmf_fusion_error_texture(self.handle, int(index), 0 if which == "icp" else 1, C.byref(p)))
        return _as_tensor(p.value, self.width * self.height * 4, torch.float32, (self.height, self.width), self.ctx.device, self)

    def exportPoses(self, export_dir):
        check(self.ctx.lib.mmf_fusion_export_poses(self.handle, export_dir.encode()))

    def getPoseLog(self, index=0):
        n = C.c_int()
        check(self.ctx.lib.mmf_fusion_pose_log(self.handle, index, None, None, 0, C.byref(n)))
        ts = np.zeros(n.value, np.int64)
        p7 = np.zeros((n.value, 7), np.float32)
        if n.value:
            check(self.ctx.lib.mmf_fusion_pose_log(self.handle, index, ts.ctypes.data_as(C.POINTER(C.c_longlong)), fptr(p7),
                                                   n.value, C.byref(n)))
        return ts, p7

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
