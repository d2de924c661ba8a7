"""CPU: the orchestration oracle (oracle/fusion.py) and the host pose algebra it rests on (oracle/mmf_oracle_pose.c):
JacobiSVD / rodrigues2 / computeFusionWeight known answers, the multi-model bookkeeping, and the measured
sensitivity of object tracking to one-ulp input noise that the GPU parity tolerances are derived from."""
import numpy as np

from multimotionfusion_amd import synth
from oracle import oracle as orc
from oracle.fusion import OracleFusion


def test_jacobi_svd_reconstructs_and_orders():
    rng = np.random.default_rng(0)
    for k in range(600):
        a = rng.normal(size=(3, 3)).astype(np.float32)
        if k % 3 == 0:  # nearly a rotation: what rodrigues2 is given
            a = (synth.rodrigues(rng.normal(size=3) * 10 ** rng.uniform(-6, 0)) + rng.normal(size=(3, 3)) * 1e-7).astype(np.float32)
        if k % 50 == 1:
            a[:, 2] = a[:, 0]  # rank deficient
        U, s, V = orc.jacobi_svd3f(a)
        rec = U.astype(np.float64) @ np.diag(s.astype(np.float64)) @ V.T.astype(np.float64)
        assert np.abs(rec - a).max() < 5e-6 * max(1.0, np.abs(a).max())
        assert np.abs(U.T @ U - np.eye(3)).max() < 2e-6 and np.abs(V.T @ V - np.eye(3)).max() < 2e-6
        assert s[0] >= s[1] >= s[2] >= 0
        assert np.abs(np.linalg.svd(a.astype(np.float64), compute_uv=False) - s).max() < 5e-6 * max(1.0, float(s[0]))
    U, s, V = orc.jacobi_svd3f(np.zeros((3, 3), np.float32))
    assert np.array_equal(U, np.eye(3)) and np.array_equal(V, np.eye(3)) and not s.any()


def test_rodrigues2_known_answers():
    axis = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    for ang in (0.3, 1.0, 2.5, 3.1):
        r = orc.rodrigues2(synth.rodrigues(axis * ang).astype(np.float32))
        assert np.abs(r - axis * ang).max() < 2e-5 * max(1.0, 1.0 / np.sin(ang))
    # below the float32 resolution of acos((trace - 1) / 2) the rotation vector is exactly zero (Model.cpp:1311-1318):
    # the quantisation the reference's fusion weight lives with
    assert not orc.rodrigues2(synth.rodrigues(axis * 5e-5).astype(np.float32)).any()
    assert abs(np.linalg.norm(orc.rodrigues2(synth.rodrigues(axis * 1e-3).astype(np.float32))) - 2.0 ** -10) < 1e-9
    # a scaled / sheared matrix is orthonormalised first (the JacobiSVD projection, :1302-1303)
    R = synth.rodrigues(axis * 0.7)
    assert np.abs(orc.rodrigues2((1.7 * R).astype(np.float32)) - axis * 0.7).max() < 1e-5
    # rotation by pi about x: the s < 1e-5, c < 0 branch
    assert np.abs(orc.rodrigues2(np.diag([1, -1, -1]).astype(np.float32)) - [np.pi, 0, 0]).max() < 1e-6


def test_compute_fusion_weight_known_answers():
    I = np.eye(4, dtype=np.float32)
    assert orc.compute_fusion_weight(I, I, 1.0) == 1.0 and orc.compute_fusion_weight(I, I, 100.0) == 100.0
    L = I.copy()
    L[:3, 3] = [0.003, 0.0, 0.004]  # |t| = 5 mm of the 10 mm that saturate the weight (Model.cpp:883-888)
    assert abs(orc.compute_fusion_weight(I, L, 1.0) - 0.5) < 1e-6
    L[:3, 3] = [0.03, 0.0, 0.0]
    assert orc.compute_fusion_weight(I, L, 2.0) == 1.0  # clamped to minWeight 0.5
    L = I.copy()
    L[:3, :3] = synth.rodrigues([0.0, 0.004, 0.0])
    w = orc.compute_fusion_weight(I, L, 1.0)
    assert 0.55 < w < 0.65  # the rotation term, quantised by acos near 1
    # getLastTransform() = pose^-1 * lastPose: invariant to a common left factor
    P = synth.make_pose([0.1, -0.2, 0.3], [0.5, 0.1, -0.4]).astype(np.float32)
    assert abs(orc.compute_fusion_weight(orc.matmul4f(P, I), orc.matmul4f(P, L), 1.0) - w) < 0.1


def _scene(w, h, n_frames, n_objects, seed=21):
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n_frames, seed=seed)
    objs = synth.make_objects(n_objects, seed=seed)
    traj = synth.object_trajectories(objs, n_frames, seed=seed)
    return K, [synth.render(p, w, h, seed=i, objects=objs, object_poses=[t[i] for t in traj]) for i, p in enumerate(poses)]


def _run(K, frames, w, h, eps=0.0, with_data=True):
    o = OracleFusion(w, h, K, enable_multiple_models=True, pose_logging=True)
    known, out = [0], []
    for i, f in enumerate(frames):
        spawn = 1 <= i <= 2
        if spawn:
            known.append(i)
        mask = np.where(np.isin(f["ids"], known), f["ids"], 0).astype(np.uint8)
        depth = f["depth"]
        if eps:
            rng = np.random.default_rng(i)
            depth = (depth.astype(np.float64) * (1 + eps * rng.standard_normal(depth.shape))).astype(np.float32)
        data = None
        if with_data and i > 0:
            data = [dict(id=k, super_pixel_count=int((mask == k).sum()) // 256, avg_confidence=0.4,
                         depth_mean=float(f["depth"][mask == k].mean()), depth_std=0.05) for k in known]
        o.process_frame(f["rgb"], depth, timestamp=i, mask=mask, has_new_label=spawn, model_data=data)
        out.append([m.pose.copy() for m in o.models])
    return o, out


def test_multi_model_bookkeeping():
    w, h = 160, 120
    K, frames = _scene(w, h, 4, 2)
    o, poses = _run(K, frames, w, h)
    assert [m.id for m in o.models] == [0, 1, 2] and o.next_id == 3 and o.tick == 5
    assert o.models[0].fill_in and not o.models[1].fill_in
    assert o.models[1].conf == np.float32(0.4) and o.models[0].conf == 10.0  # :616-620 raises object thresholds only
    assert 1.0 < o.models[1].max_depth < 3.0 and o.models[0].max_depth > 1e30  # getMaxDepth (:408, :586)
    assert [len(m.pose_log) for m in o.models] == [4, 3, 2]  # one entry per frame in the list (:829-846)
    assert all(m.surfels.shape[0] > 50 for m in o.models[1:])
    # a model that is not seen leaves the list (:606-613)
    mask = np.zeros((h, w), np.uint8)
    data = [dict(id=0, super_pixel_count=70, avg_confidence=0.4, depth_mean=2.0, depth_std=0.1),
            dict(id=1, super_pixel_count=0, avg_confidence=0.4, depth_mean=0.0, depth_std=0.0),
            dict(id=2, super_pixel_count=3, avg_confidence=0.4, depth_mean=1.5, depth_std=0.1)]
    o.process_frame(frames[-1]["rgb"], frames[-1]["depth"], timestamp=9, mask=mask, model_data=data)
    assert [m.id for m in o.models] == [0, 2] and [m.id for m in o.inactive] == [1]


def test_object_tracking_is_sensitive_to_one_ulp_noise():
    """The tolerance of the GPU parity tests for OBJECT models is a measured property of the restated algorithm:
    depth noise of 1e-7 relative (about one float32 ulp) moves the oracle's own object poses by orders of magnitude
    more than the global pose."""
    w, h = 320, 240
    K, frames = _scene(w, h, 5, 2)
    _, a = _run(K, frames, w, h, with_data=False)
    _, b = _run(K, frames, w, h, eps=1e-7, with_data=False)
    glob = max(float(np.abs(x[0] - y[0]).max()) for x, y in zip(a, b))
    obj = max(float(np.abs(p - q).max()) for x, y in zip(a, b) for p, q in zip(x[1:], y[1:]))
    assert glob < 1e-6, glob
    assert obj > 20 * glob, (glob, obj)  # measured: 1e-5 .. 1e-3 against < 1e-7
