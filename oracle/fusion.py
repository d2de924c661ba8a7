"""CPU restatement of MultiMotionFusion::processFrame / predict (Core/MultiMotionFusion.cpp:207-854, 863-875) and of
the Model methods it drives (Core/Model/Model.cpp:390-433, 876-891; Core/Model/Model.h:210-214, 301-305) on top of
the oracle's kernels (oracle.py).  TEST INFRASTRUCTURE ONLY: the checker of mmf_fusion_process_frame*.

PARITY UNPINNED (the reference holds no vectors for this path; see mmf_oracle.h).  Every step cites the lines
it follows.  Out of scope, as in the product: the segmentation itself (its RESULT -- the id image, hasNewLabel and
the per-model data -- is an input), relocalisation, ferns, deformation (closeLoops / reloc off), keypoint tracks
(the RigidRANSAC transformations of Model::getLastTrackTransform are inputs).
"""
import numpy as np

from . import oracle as orc

FLT_MAX = float(np.finfo(np.float32).max)


def _get_max_depth(d):
    """getMaxDepth lambda (:408): `data.depthMean + data.depthStd * 1.2` -- float operands, double arithmetic
    (1.2 is a double literal), returned as float."""
    return float(np.float32(float(np.float32(d["depth_mean"])) + float(np.float32(d["depth_std"])) * 1.2))


class OracleModel:
    """class Model (Core/Model/Model.h:120-360) as far as processFrame uses it."""

    def __init__(self, id, conf, w, h, K, fill_in):
        # Model::Model (Model.cpp:147-173): identity pose / lastPose, maxDepth = FLT_MAX (Model.h:129), frameToModel
        self.id = int(id)
        self.conf = float(np.float32(conf))
        self.fill_in = bool(fill_in)
        self.pose = np.eye(4, dtype=np.float32)
        self.last_pose = np.eye(4, dtype=np.float32)
        self.max_depth = FLT_MAX
        self.unseen = 0
        self.surfels = np.zeros((0, 12), np.float32)
        self.odom = orc.Odometry(w, h, K["cx"], K["cy"], K["fx"], K["fy"])
        self.pose_log = []

    def override_pose(self, p):  # Model.h:301-304
        self.pose = np.array(p, np.float32).reshape(4, 4).copy()
        self.last_pose = self.pose.copy()

    def compute_fusion_weight(self, multiplier):  # Model.cpp:876-891 (incl. rodrigues2's JacobiSVD, :1301-1342)
        return orc.compute_fusion_weight(self.pose, self.last_pose, multiplier)


class OracleFusion:
    """class MultiMotionFusion: models[0] is globalModel (id 0, fill-in on, MultiMotionFusion.cpp:69-71); object models
    are created with initConfThresObject and without fill-in (:944)."""

    def __init__(self, w, h, K, time_delta=200, conf=10.0, icp_weight=10.0, depth_cutoff=15.0, max_depth=20.0,
                 outlier_coeff=3.0, conf_object=0.01, enable_multiple_models=False, so3=True, pyramid=True,
                 fast_odom=False, rgb_only=False, pose_logging=False):
        self.w, self.h, self.K = w, h, K
        self.time_delta, self.icp_weight = time_delta, icp_weight
        self.depth_cutoff, self.max_depth, self.outlier_coeff = depth_cutoff, max_depth, outlier_coeff
        self.conf_object = conf_object
        self.enable_multiple_models = enable_multiple_models
        self.so3, self.pyramid, self.fast_odom, self.rgb_only = so3, pyramid, fast_odom, rgb_only
        self.pose_logging = pose_logging
        self.next_id = 0
        self.models = [OracleModel(self._next_model_id(True), conf, w, h, K, True)]
        self.inactive = []
        self.tick = 1  # MultiMotionFusion.cpp:36
        self.mask = np.zeros((h, w), np.uint8)  # textures[MASK]
        self.last_image_ring = None

    # -- compatibility with the single-model tests ----------------------------------------------------------
    @property
    def pose(self):
        return self.models[0].pose

    @property
    def surfels(self):
        return self.models[0].surfels

    @property
    def fill_in_taken(self):
        return self.models[0].fill_in_taken

    def _next_model_id(self, assign):  # getNextModelID (:983-999)
        nxt = self.next_id
        if assign:
            while True:
                self.next_id = (self.next_id + 1) & 255
                if not any(m.id == self.next_id for m in getattr(self, "models", [])):
                    break
        return nxt

    # -- Model::combinedPredict(ACTIVE) + performFillIn: the body of predict() (:863-875) ---------------------
    def _predict_model(self, m, rgb, fil):
        m.image, m.vertexConf, m.normalRadius, m.time_tex = orc.combined_predict(
            m.surfels, m.pose, self.K, self.w, self.h, self.max_depth, m.conf, self.tick, self.tick, self.time_delta)
        if m.fill_in:  # Model::performFillIn (Model.cpp:1607-1616), lost = false, frameToFrameRGB = false
            m.fillVertex, m.fillNormal, m.fillImage = orc.fill_in(m.vertexConf, m.normalRadius, m.image, fil, rgb,
                                                                  self.K, 0, 0)

    def predict(self, rgb, fil):
        for m in self.models:
            self._predict_model(m, rgb, fil)

    # -- predictIndices / fuse / predictIndices / clean of one model (:791-816) -------------------------------
    def _fuse_clean_model(self, m, rgb, depth, fil, weight, second_predict=True):
        index, vc, ct, nr = orc.predict_indices(m.surfels, m.pose, self.K, self.w, self.h, self.max_depth, self.tick,
                                                self.time_delta)
        # Model::fuse: maxDepth uniform = std::min(depthCutoff, maxDepth) (Model.cpp:928)
        s_upd, new = orc.fuse(m.surfels, rgb, depth, fil, self.mask, index, vc, nr, m.pose, self.K, self.tick, weight,
                              m.id, min(self.max_depth, m.max_depth))
        if second_predict:
            index, vc, ct, nr = orc.predict_indices(s_upd, m.pose, self.K, self.w, self.h, self.max_depth, self.tick,
                                                    self.time_delta)
        m.surfels = orc.clean(s_upd, new, m.pose, self.K, self.w, self.h, self.tick, self.time_delta, m.conf,
                              self.outlier_coeff, m.id, index, vc, ct, fil, self.mask)

    # -- Model::performTracking (Model.cpp:409-433) with Model::initICP (:390-407) ----------------------------
    def _perform_tracking(self, m, rgb, fil, do_fill_in):
        m.last_pose = m.pose.copy()  # :412
        if do_fill_in:
            m.odom.initICPModel(m.fillVertex, m.fillNormal, m.pose)
            m.odom.initRGBModel(m.fillImage)
        else:
            m.odom.initICPModel(m.vertexConf, m.normalRadius, m.pose)
            m.odom.initRGBModel(m.image)  # frameToFrameRGB = false
        m.odom.initICP(fil, self.max_depth)  # gpu.depth_tmp = pyramid of the filtered depth (Model.cpp:359-388, 402)
        m.odom.initRGB(rgb)
        t, R = m.odom.getIncrementalTransformation(m.pose[:3, 3], m.pose[:3, :3], self.rgb_only, self.icp_weight,
                                                   self.pyramid, self.fast_odom, self.so3)
        m.pose = np.eye(4, dtype=np.float32)
        m.pose[:3, :3], m.pose[:3, 3] = R, t

    def process_frame(self, rgb, depth, timestamp=0, in_pose=None, weight_multiplier=1.0, bootstrap=False,
                      init_transform=None, init_transforms=None, icp_refine=True, mask=None, has_new_label=False,
                      model_data=None):
        """mask / has_new_label / model_data: the SegmentationResult of this frame (fullSegmentation, hasNewLabel,
        modelData as dicts with id, super_pixel_count, avg_confidence, depth_mean, depth_std)."""
        if init_transform is not None:
            init_transforms = [init_transform]
        fil = orc.bilateral_filter(depth, self.depth_cutoff)  # filterDepth (:262)
        if not self.enable_multiple_models:
            self.mask = np.zeros((self.h, self.w), np.uint8)  # :268-275
        g = self.models[0]
        if self.tick == 1:  # :290-296
            g.surfels = orc.surfel_initialise(rgb, depth, fil, self.K, self.tick, self.max_depth)
            g.odom.initFirstRGB(rgb)
        else:
            if bootstrap or in_pose is None:  # :299
                for k, m in enumerate(self.models):  # :312-387
                    do_icp = True
                    if init_transforms is not None:  # odom_init == "kp" (:316-376)
                        do_icp = bool(icp_refine)
                        if k < len(init_transforms):
                            T = np.asarray(init_transforms[k], np.float32).reshape(4, 4)
                            tnew = orc.matmul4f(m.pose, T) if m.id == 0 else orc.matmul4f(T, m.pose)  # :331 / :334
                        else:
                            tnew = m.pose.copy()
                        m.override_pose(tnew)  # :350
                        self._predict_model(m, rgb, fil)  # :353-355
                        # fuse(..., weightMultiplier) -> computeFusionWeight(weightMultiplier) (:359-360, Model.cpp:918)
                        self._fuse_clean_model(m, rgb, depth, fil, m.compute_fusion_weight(weight_multiplier))
                    if do_icp:  # :377-381
                        do_fill = m.fill_in and bool(orc.requires_fill_in(m.image, 0.75))  # requiresFillIn (:877-895)
                        m.fill_in_taken = do_fill
                        self._perform_tracking(m, rgb, fil, do_fill)
                if bootstrap:  # :397-400
                    g.override_pose(orc.matmul4f(g.pose, np.asarray(in_pose, np.float32).reshape(4, 4)))
                if self.enable_multiple_models:  # :407-622
                    assert mask is not None
                    self.mask = np.ascontiguousarray(mask, np.uint8)  # :416
                    data = list(model_data) if model_data is not None else []
                    fresh = None
                    if has_new_label:  # :469-487
                        fresh = OracleModel(self._next_model_id(True), self.conf_object, self.w, self.h, self.K, False)
                        fresh.odom.initFirstRGB(rgb)  # spawnObjectModel (:946)
                        if data:
                            fresh.max_depth = _get_max_depth(data[-1])  # :486
                    for i in range(1, min(len(self.models), len(data))):  # :585-586
                        self.models[i].max_depth = _get_max_depth(data[i])
                    if fresh is not None:  # :588-601
                        self._fuse_clean_model(fresh, rgb, depth, fil, fresh.compute_fusion_weight(100.0),
                                               second_predict=False)  # the second predictIndices is commented out (:594)
                        self.models.append(fresh)  # moveNewModelToList (:600)
                    lost = []
                    for d in data:  # :606-613
                        m = next((x for x in self.models if x.id == int(d["id"])), None)
                        if m is None or m is fresh:
                            continue
                        if d["super_pixel_count"] <= 0:
                            m.unseen += 1
                            if m.unseen > 0 and m.id != 0:
                                lost.append(m)
                    for m in lost:  # inactivateModel (:962-981)
                        self.models.remove(m)
                        self.inactive.append(m)
                    for i in range(1, min(len(self.models), len(data))):  # :616-620
                        old = np.float32(self.models[i].conf)
                        self.models[i].conf = float(min(max(old, np.float32(data[i]["avg_confidence"])), np.float32(9.0)))
            else:
                g.override_pose(in_pose)  # :670
            self.predict(rgb, fil)  # :675
            if not self.rgb_only:  # :791-817 (trackingOk, !lost)
                for m in self.models:
                    self._fuse_clean_model(m, rgb, depth, fil, m.compute_fusion_weight(weight_multiplier))
        self.predict(rgb, fil)  # :821
        self.tick += 1  # :825
        if self.pose_logging:  # :829-846
            for k, m in enumerate(self.models):
                T = g.pose if k == 0 else orc.matmul4f(g.pose, orc.inverse4f(m.pose))
                m.pose_log.append((int(timestamp), T.copy()))
