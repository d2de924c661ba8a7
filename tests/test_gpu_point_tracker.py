"""-m gpu: the keypoint-track front end (point_tracker.py: tracker::PointTracker, Model::getLastTrackTransform)
with the descriptor search on the device, and the whole `-init kp` chain SuperPoint -> tracks -> RANSAC ->
processFrame on a synthetic sequence."""
import numpy as np
import pytest
import torch

from multimotionfusion_amd import synth

pytestmark = pytest.mark.gpu

W, H = 320, 240
FX = FY = 264.0
CX, CY = 160.0, 120.0


def unit_rows(rng, n, dim=256):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def scene(rng, n):
    """n keypoints with integer pixels, a depth image holding their depth, their 3-D points"""
    px = np.stack([rng.choice(np.arange(10, W - 10), n, replace=False), rng.integers(10, H - 10, n)], 1)
    z = rng.uniform(1.0, 3.0, n).astype(np.float32)
    depth = np.zeros((H, W), np.float32)
    depth[px[:, 1], px[:, 0]] = z
    pts = np.stack([z * (px[:, 0] - CX) / FX, z * (px[:, 1] - CY) / FY, z], 1)
    return px, depth, pts


def norm_coords(px):
    return (px + 0.0) / np.array([W, H], np.float64)


def test_tracks_grow_match_and_prune(gpu_ctx):
    from multimotionfusion_amd.point_tracker import PointTracker
    rng = np.random.default_rng(0)
    tr = PointTracker(gpu_ctx, (FX, FY, CX, CY))
    px, depth, pts = scene(rng, 40)
    desc = unit_rows(rng, 40)
    tr.addKeypoints(norm_coords(px), desc, 1_000_000_000, depth)
    assert len(tr.getTracks()) == 40 and all(len(t) == 1 for t in tr.getTracks())
    kp = tr.getTracks()[7][0]
    assert kp.xy == (px[7, 0], px[7, 1]) and np.allclose(kp.coordinate, pts[7], atol=1e-5)

    # frame 2: keypoints 0..29 seen again (noisy descriptors, shuffled), 30..39 lost, 5 new ones
    order = rng.permutation(30)
    desc2 = np.concatenate([desc[order] + 0.01 * rng.standard_normal((30, 256)).astype(np.float32), unit_rows(rng, 5)])
    px2, depth2, _ = scene(rng, 35)
    tr.addKeypoints(norm_coords(px2), desc2, 1_033_000_000, depth2)
    tracks = tr.getTracks()
    assert len(tracks) == 45 and all(len(t) == 2 for t in tracks)
    for q, src in enumerate(order):  # the query keypoint q continues track src
        assert tracks[src][1] is not None and tracks[src][1].xy == (px2[q, 0], px2[q, 1])
    assert all(tracks[i][1] is None for i in range(30, 40))
    assert all(tracks[i][0] is None and tracks[i][1] is not None for i in range(40, 45))

    # frame 3: nothing detected: every track is extended by an inactive entry
    tr.addKeypoints(np.zeros((0, 2)), np.zeros((0, 256), np.float32), 1_066_000_000, depth2)
    assert all(len(t) == 3 and t[2] is None for t in tr.getTracks())
    active = tr.getLastActiveKeypoints(history=1)
    assert all(a is None for a in active)
    active = tr.getLastActiveKeypoints(history=2)
    assert sum(a is not None for a in active) == 35
    assert sum(a is not None for a in tr.getLastActiveKeypoints(0)) == 45

    # keypoints too far in descriptor space do not continue a track (min_feature_distance)
    tr.addKeypoints(norm_coords(px2[:3]), unit_rows(rng, 3), 1_100_000_000, depth2, min_feature_distance=0.7)
    assert len(tr.getTracks()) == 48

    # prune (:168-203): short tracks whose last keypoint is old go
    tr.prune(min_kps=2, min_time=1_050_000_000)
    kept = tr.getTracks()
    assert len(kept) == 30 + 3  # two-keypoint tracks stay, and the three born at 1.1 s are recent
    tr.prune(min_kps=30, min_time=0)
    assert len(tr.getTracks()) == 33  # nothing is older than time 0


def test_last_track_transform_recovers_the_motion(gpu_ctx):
    from multimotionfusion_amd.point_tracker import PointTracker, getLastTrackTransform
    rng = np.random.default_rng(1)
    tr = PointTracker(gpu_ctx, (FX, FY, CX, CY))
    px, depth, pts = scene(rng, 60)
    desc = unit_rows(rng, 60)
    tr.addKeypoints(norm_coords(px), desc, 0, depth)
    T_identity, inl = getLastTrackTransform(tr.getTracks())
    assert np.array_equal(T_identity, np.eye(4, dtype=np.float32)) and inl is None  # tracks of length 1
    # the camera moves by T (camera 1 in camera 0): p0 = T p1
    ang = 0.03
    T = np.eye(4)
    T[:3, :3] = [[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]
    T[:3, 3] = (0.04, -0.01, 0.02)
    p1 = (np.linalg.inv(T) @ np.c_[pts, np.ones(60)].T).T[:, :3]
    px1 = np.stack([np.rint(p1[:, 0] / p1[:, 2] * FX + CX), np.rint(p1[:, 1] / p1[:, 2] * FY + CY)], 1).astype(int)
    ok = (px1[:, 0] >= 0) & (px1[:, 0] < W) & (px1[:, 1] >= 0) & (px1[:, 1] < H)
    depth1 = np.zeros((H, W), np.float32)
    depth1[px1[ok, 1], px1[ok, 0]] = p1[ok, 2]
    depth1[px1[ok, 1][:4], px1[ok, 0][:4]] = 0.0  # four keypoints without depth: NaN coordinates, skipped
    tr.addKeypoints(norm_coords(px1[ok]), desc[ok], 33_000_000, depth1)
    got, inlier = getLastTrackTransform(tr.getTracks())
    assert inlier is not None and inlier.sum() >= 0.6 * (ok.sum() - 4)
    assert np.abs(got[:3, 3] - T[:3, 3]).max() < 0.01
    assert synth.rotation_angle(got[:3, :3].astype(np.float64), T[:3, :3]) < 0.01


class LandmarkPredictor:
    """stands in for SuperPoint::getFeatures with perfect keypoints: fixed world landmarks (taken from the first
    frame's depth) projected into every frame, one constant descriptor each"""

    def __init__(self, frames, poses, K, n=200, seed=0):
        rng = np.random.default_rng(seed)
        d0 = frames[0]["depth"]
        h, w = d0.shape
        ys, xs = np.nonzero(d0 > 0)
        pick = rng.choice(len(ys), n, replace=False)
        z = d0[ys[pick], xs[pick]].astype(np.float64)
        cam = np.stack([z * (xs[pick] - K["cx"]) / K["fx"], z * (ys[pick] - K["cy"]) / K["fy"], z, np.ones(n)], 0)
        self.world = poses[0] @ cam
        self.desc = unit_rows(rng, n)
        self.poses, self.K, self.w, self.h, self.frame = poses, K, w, h, 0

    def getFeatures(self, rgb):
        K = self.K
        cam = np.linalg.inv(self.poses[self.frame]) @ self.world
        self.frame += 1
        x = np.rint(cam[0] / cam[2] * K["fx"] + K["cx"])
        y = np.rint(cam[1] / cam[2] * K["fy"] + K["cy"])
        ok = (cam[2] > 0) & (x >= 0) & (x < self.w) & (y >= 0) & (y < self.h)
        return np.stack([x[ok] / self.w, y[ok] / self.h], 1), self.desc[ok].astype(np.float64)


@pytest.mark.parametrize("icp_refine", [True, False])
def test_keypoint_front_end_with_good_keypoints(gpu_ctx, icp_refine):
    """landmark keypoints -> PointTracker -> getLastTrackTransform -> processFrame(initTransform): the RANSAC
    transformation alone follows the camera to about a pixel's worth of depth noise; with -icp_refine the dense
    tracker brings it to the usual accuracy."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    from multimotionfusion_amd.point_tracker import KeypointFrontEnd
    w, h, n = 320, 240, 5
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n, seed=5)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    fusion = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    fe = KeypointFrontEnd(gpu_ctx, fusion, LandmarkPredictor(frames, poses, K), (K["fx"], K["fy"], K["cx"], K["cy"]),
                          icp_refine=icp_refine)
    for i, f in enumerate(frames):
        fe.processFrame(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda(), timestamp=1000 + 33_000_000 * i)
        if i > 0:
            assert fe.last_inlier is not None and fe.last_inlier.mean() > 0.6
        gt = np.linalg.inv(poses[0]) @ poses[i]
        err = np.linalg.norm(fusion.getCurrPose()[:3, 3] - gt[:3, 3])
        assert err < (0.01 if icp_refine else 0.03), (i, err)
    fusion.close()


def test_keypoint_front_end_runs_on_superpoint_features(gpu_ctx):
    """SuperPoint (random-init weights: meaningless but deterministic keypoints) through the same chain: the
    bookkeeping must hold whatever the matches are worth."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    from multimotionfusion_amd.point_tracker import KeypointFrontEnd
    from multimotionfusion_amd.superpoint import SuperPoint, random_weights
    w, h, n = 320, 240, 3
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n, seed=5)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    fusion = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    kp = SuperPoint(gpu_ctx, random_weights(3), max_width=w, max_height=h, max_keypoints=1024)
    fe = KeypointFrontEnd(gpu_ctx, fusion, kp, (K["fx"], K["fy"], K["cx"], K["cy"]), icp_refine=True)
    for i, f in enumerate(frames):
        fe.processFrame(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda(), timestamp=1000 + 33_000_000 * i)
        assert fusion.getTick() == i + 2
        tracks = fe.tracker.getTracks()
        assert tracks and all(len(t) == i + 1 for t in tracks)
        assert np.all(np.isfinite(fusion.getCurrPose()))
    kp.close()
    fusion.close()
