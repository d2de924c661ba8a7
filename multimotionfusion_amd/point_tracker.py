"""Keypoint tracks for the `-init kp` front end: the host-side mirror of tracker::PointTracker
(Core/Utils/PointTracker.cpp:27-226) and Model::getLastTrackTransform (Core/Model/Model.cpp:739-775).

The track lists are host bookkeeping like in the reference; the N x N descriptor search
(cv::BFMatcher, PointTracker.cpp:100-102) runs on the device through mmf_match_descriptors and the rigid
fit through mmf_ransac_estimate -- no fallback.
"""
import numpy as np
import torch

from .cudafuncs import Context
from .matcher import matchDescriptors
from .ransac import RigidRANSAC


class Keypoint:
    __slots__ = ("timestamp", "xy", "coordinate", "descriptor")

    def __init__(self, timestamp, xy, coordinate, descriptor):
        self.timestamp, self.xy, self.coordinate, self.descriptor = timestamp, xy, coordinate, descriptor


def _cv_round(v):
    """cv::Point(cv::Vec2d) -> saturate_cast<int> = cvRound: round half to even"""
    return int(np.rint(v))


class PointTracker:
    def __init__(self, ctx: Context, intrinsics):
        """intrinsics: (fx, fy, cx, cy) of the pyramid level the keypoints come from (CameraModel::operator()(level))"""
        self.ctx = ctx
        self.fx, self.fy, self.cx, self.cy = (np.float32(v) for v in intrinsics)
        self.tracks = []  # each track: list of Keypoint or None, all of the same length

    def getTracks(self):
        return self.tracks

    def _construct_kp(self, coordinate, descriptor, timestamp, depth):
        # PointTracker.cpp:35-56: pixel = normalised coordinate * (cols, rows); back-projection in float
        rows, cols = depth.shape
        x, y = _cv_round(coordinate[0] * cols), _cv_round(coordinate[1] * rows)
        z = np.float32(depth[y, x])
        if z > 0:
            v = np.array([np.float32(np.float32(z * np.float32(np.float32(x) - self.cx)) / self.fx),
                          np.float32(np.float32(z * np.float32(np.float32(y) - self.cy)) / self.fy), z], np.float64)
        else:
            v = np.full(3, np.nan)
        return Keypoint(int(timestamp), (x, y), v, np.asarray(descriptor, np.float32))

    def addKeypoints(self, coordinates, descriptors, timestamp, depth, min_feature_distance=0.7, history=30):
        """coordinates [n,2] normalised to [0,1), descriptors [n,d], depth [rows,cols] float32 (host)"""
        coordinates, descriptors = np.asarray(coordinates, np.float64), np.asarray(descriptors)
        depth = np.asarray(depth, np.float32)
        n = coordinates.shape[0]
        assert descriptors.shape[0] == n
        if not self.tracks:  # :61-66 add without matching
            for ik in range(n):
                self.tracks.append([self._construct_kp(coordinates[ik], descriptors[ik], timestamp, depth)])
            return
        active = self.getLastActiveKeypoints(history)
        for track in self.tracks:  # :71-73 inactive by default
            track.append(None)
        if n == 0:
            return
        valid = [i for i, kp in enumerate(active) if kp is not None]  # map_valid_prev
        matched = {}
        if valid:
            dev = torch.device("cuda", self.ctx.device)
            previous = torch.from_numpy(np.stack([active[i].descriptor for i in valid])).to(dev)
            current = torch.from_numpy(np.ascontiguousarray(descriptors, np.float32)).to(dev)
            # cv::BFMatcher(cv::NORM_L2, true).match(current, previous) + the distance test of :108
            idx, _ = matchDescriptors(self.ctx, current, previous, min_feature_distance)
            for q, t in enumerate(idx.cpu().numpy()):
                if t >= 0:
                    matched[q] = valid[int(t)]
        for q, ti in matched.items():  # :107-112
            self.tracks[ti][-1] = self._construct_kp(coordinates[q], descriptors[q], timestamp, depth)
        curr_length = len(self.tracks[0])
        for q in range(n):  # :116-121 unmatched keypoints start new tracks
            if q not in matched:
                track = [None] * curr_length
                track[-1] = self._construct_kp(coordinates[q], descriptors[q], timestamp, depth)
                self.tracks.append(track)

    def prune(self, min_kps, min_time):
        """:168-203: drop tracks with fewer than min_kps keypoints whose last keypoint is older than min_time"""
        kept = []
        for track in self.tracks:
            nvalid = sum(kp is not None for kp in track)
            last_stamp = 0
            for kp in track:
                if kp is not None:
                    last_stamp = kp.timestamp
            if not (nvalid < min_kps and last_stamp < min_time):
                kept.append(track)
        self.tracks = kept

    def getLastActiveKeypoints(self, history=0):
        """:205-224: the last keypoint of every track within `history` steps from its end (0: anywhere)"""
        active = []
        for track in self.tracks:
            found = None
            for d, kp in enumerate(reversed(track)):
                if history and d >= history:
                    break
                if kp is not None:
                    found = kp
                    break
            active.append(found)
        return active


def getLastTrackTransform(tracks, config=(10, 0.03, 0.6)):
    """Model::getLastTrackTransform (Model.cpp:739-775): rigid transformation between the last two keypoints
    of the tracks.  -> (T 4x4 float32, inlier mask or None).  Fewer than 3 valid pairs: identity."""
    p0s, p1s = [], []
    for track in tracks:
        if len(track) < 2:
            continue
        kp0, kp1 = track[-2], track[-1]
        if kp0 is not None and kp1 is not None:
            if np.all(np.isfinite(kp0.coordinate)) and np.all(np.isfinite(kp1.coordinate)):
                p0s.append(kp0.coordinate.astype(np.float32))
                p1s.append(kp1.coordinate.astype(np.float32))
    if len(p0s) < 3:
        return np.eye(4, dtype=np.float32), None
    T, _, inlier = RigidRANSAC(*config).estimate(np.stack(p0s), np.stack(p1s))
    return T, inlier


class KeypointFrontEnd:
    """The keypoint half of processFrame for one pyramid level (MultiMotionFusion.cpp:223-248, 286-296, 322-337):
    SuperPoint features -> PointTracker -> getLastTrackTransform -> processFrame(initTransform=...)."""

    def __init__(self, ctx: Context, fusion, kp_predictor, intrinsics, icp_refine=True):
        self.ctx, self.fusion, self.kp, self.icp_refine = ctx, fusion, kp_predictor, icp_refine
        self.tracker = PointTracker(ctx, intrinsics)
        self.last_inlier = None

    def processFrame(self, rgb, depth, timestamp, weightMultiplier=1.0):
        coordinates, descriptors = self.kp.getFeatures(rgb)
        self.tracker.addKeypoints(coordinates, descriptors, timestamp, depth.cpu().numpy(), 0.7, 30)
        self.tracker.prune(30, max(int(timestamp) - int(1e9), 0))  # :246
        if self.fusion.getTick() == 1:
            self.fusion.processFrame(rgb, depth, timestamp=timestamp, weightMultiplier=weightMultiplier)
            return
        T, self.last_inlier = getLastTrackTransform(self.tracker.getTracks())
        self.fusion.processFrame(rgb, depth, timestamp=timestamp, weightMultiplier=weightMultiplier, initTransform=T,
                                 icpRefine=self.icp_refine)
