"""-m gpu: multi-model MultiMotionFusion::processFrame (Core/MultiMotionFusion.cpp:312-387, 407-622, 791-816, 863-875) --
the global model plus object models spawned from a ground-truth id image (configs 4 / 5 of BASELINE.json), each on
its own stream -- against the oracle orchestration (oracle/fusion.py).

What is bit-exact: everything that does not pass through the tracker's float32 sums -- the first surfels of a model in
the frame it is spawned (identity pose), Model::computeFusionWeight, the pose log's quaternion.
Poses: the tracker differs from the oracle only in float32 summation order (~1e-7 in a pose).  For the global model
that stays below 1e-5 over a sequence.  An object model is a few thousand pixels of a map full of silhouette
artefacts: its 19 fixed Gauss-Newton iterations with hard-gated projective association are NOT stable against
one-ulp input noise -- the ORACLE ITSELF moves an object pose by 1e-4 .. 1e-3 when the depth image is perturbed by
1e-7 relative (tests/test_oracle_fusion.py::test_object_tracking_is_sensitive_to_one_ulp_noise), while the global
pose moves by < 1e-6.  Object poses are therefore compared (a) in a RE-SYNCHRONISED run, where every frame starts
from the oracle's maps and poses so that each frame's tracking is compared on identical inputs: north_star's tolerance
(1e-4 translation, 1e-3 rad) for every frame and a median <= 1e-5 over all (frame, object) pairs; (b) free running:
the object models stay on their objects (ground truth) and their surfel counts stay within 2 %."""
import numpy as np
import pytest
import torch

from helpers import OracleFusion
from multimotionfusion_amd import synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def scene(w, h, n_frames, n_objects, seed=21):
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n_frames, seed=seed)
    objs = synth.make_objects(n_objects, seed=seed)
    traj = synth.object_trajectories(objs, n_frames, seed=seed)
    frames = [synth.render(p, w, h, seed=i, objects=objs, object_poses=[t[i] for t in traj]) for i, p in enumerate(poses)]
    return K, poses, traj, frames, objs


def gt_mask(ids, known):
    """Segmentation.cpp:89-150 for a ground-truth id image whose ids already are model ids: ids of models that do not
    exist (yet) read as background."""
    return np.where(np.isin(ids, list(known)), ids, 0).astype(np.uint8)


def model_data(mask, depth, ids):
    """SegmentationResult::modelData of the pre-masked path (Segmentation.cpp:121-147): pixel count / 256 as the
    super-pixel count, avgConfidence 0.4, mean and mean absolute deviation of the depth per id."""
    out = []
    for i in ids:
        sel = mask == i
        n = int(sel.sum())
        mean = float(depth[sel].mean()) if n else 0.0
        std = float(np.abs(depth[sel] - mean).mean()) if n else 0.0
        out.append(dict(id=i, super_pixel_count=n // 256, avg_confidence=0.4, depth_mean=mean, depth_std=std))
    return out


@pytest.mark.parametrize("w,h,with_data,sync", [(320, 240, False, False), (320, 240, True, True)])
def test_three_objects_spawned_one_per_frame(gpu_ctx, orc, w, h, with_data, sync):
    objects_against_the_oracle(gpu_ctx, orc, w, h, with_data, sync, 6, 3)


def test_two_objects_at_640x480(gpu_ctx, orc):
    """the multi-model frame step at the size BASELINE.json names (configs[3]): camera + two objects spawned one per frame,
    every frame from the oracle orchestration's state (oracle/fusion.py)"""
    objects_against_the_oracle(gpu_ctx, orc, 640, 480, True, True, 4, 2)


def objects_against_the_oracle(gpu_ctx, orc, w, h, with_data, sync, n_frames, n_obj):
    from multimotionfusion_amd.fusion import MultiMotionFusion
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj)
    g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=1,
                          pose_logging=1)
    o = OracleFusion(orc, w, h, K, enable_multiple_models=True, pose_logging=True)
    known = [0]
    object_diffs, keep = [], []
    for i, f in enumerate(frames):
        spawn = 1 <= i <= n_obj  # one new label per frame (Segmentation.cpp:113-118: allowNew && !hasNewLabel)
        if spawn:
            assert g.getNextModelID() == i
            known.append(i)
        mask = gt_mask(f["ids"], known)
        assert (mask == known[-1]).sum() > 200, "the object must be visible"
        data = model_data(mask, f["depth"], known) if with_data and i > 0 else None
        keep.append((dev(f["rgb"]), dev(f["depth"]), dev(mask)))  # predict() re-reads the frame: keep it alive
        g.processFrame(*keep[-1][:2], timestamp=1000 + i, mask=keep[-1][2], hasNewLabel=spawn, modelData=data)
        o.process_frame(f["rgb"], f["depth"], timestamp=1000 + i, mask=mask, has_new_label=spawn, model_data=data)
        gm = g.getModels()
        assert [m.id for m in gm] == [m.id for m in o.models] == known[:len(gm)], (i, [m.id for m in gm])
        for k, (a, b) in enumerate(zip(gm, o.models)):
            pa = a.getPose()
            if k == 0:
                assert np.abs(pa - b.pose).max() <= 1e-5, (i, k, pa, b.pose)
            elif sync:
                assert np.abs(pa[:3, 3] - b.pose[:3, 3]).max() <= 1e-4, (i, k, pa, b.pose)
                assert synth.rotation_angle(pa[:3, :3].astype(np.float64), b.pose[:3, :3]) <= 1e-3
                if k < len(gm) - 1 or not spawn:  # tracked this frame
                    object_diffs.append(float(np.abs(pa - b.pose).max()))
            na, nb = a.lastCount(), b.surfels.shape[0]
            assert abs(na - nb) <= max(8, 0.002 * nb if k == 0 or sync else 0.02 * nb), (i, k, na, nb)
            if k > 0:
                assert abs(a.confidenceThreshold() - b.conf) < 1e-7 and a.id == k
        if spawn:  # the new model was created at the identity pose: nothing of it went through the tracker
            fresh_g, fresh_o = gm[-1].downloadMap(), o.models[-1].surfels
            assert fresh_o.shape[0] > 50
            assert np.array_equal(fresh_g.view(np.uint32), fresh_o.view(np.uint32)), (i, fresh_g.shape, fresh_o.shape)
        if sync:  # the next frame starts from the oracle's state: maps, poses, and the predictions rendered from them
            for a, b in zip(gm, o.models):
                a.uploadMap(b.surfels)
                a.overridePose(b.pose)
            g.predict()
    if sync:
        assert len(object_diffs) >= min(6, n_obj) and np.median(object_diffs) <= 1e-5, object_diffs
    # pose log (MultiMotionFusion.cpp:829-846): one entry per frame the model was in the list, object->world
    from multimotionfusion_amd.klg import pose_7d
    gm = g.getModels()
    for k in range(0, n_obj + 1):
        ts, p7 = g.getPoseLog(k)
        lo = o.models[k].pose_log
        assert len(ts) == len(lo) == n_frames - k and ts[-1] == 1000 + n_frames - 1
        if k == 0 or sync:
            assert np.abs(p7[-1] - pose_7d(lo[-1][1])).max() < (2e-5 if k == 0 else 2e-3)
    # the object models follow their objects.  Model frame = camera frame at the spawn frame s (the model is created at
    # the identity pose), so X_cam(t) = P(t)^-1 X_model with P_gt(t) = C_s^-1 T(s) T(t)^-1 C_t
    t = n_frames - 1
    for k in range(1, n_obj + 1 if not sync else 0):
        s = k
        p_gt = np.linalg.inv(poses[s]) @ traj[k - 1][s] @ np.linalg.inv(traj[k - 1][t]) @ poses[t]
        p_est = gm[k].getPose().astype(np.float64)
        # measured where the object is (a rotation error of the small object turns into centimetres at the camera origin)
        c = np.linalg.inv(poses[s]) @ np.append(traj[k - 1][s][:3, :3] @ objs[k - 1]["centre"] + traj[k - 1][s][:3, 3], 1.0)
        err = np.linalg.norm((np.linalg.inv(p_est) @ c - np.linalg.inv(p_gt) @ c)[:3])
        assert err < 0.01, (k, err)
        assert synth.rotation_angle(p_est[:3, :3], p_gt[:3, :3]) < 0.04
    g.close()


def test_batched_chain_equals_one_chain_per_model(gpu_ctx):
    """The Gauss-Newton chains of all models of a frame run as ONE chain of launches (gridDim.y = model, every model's
    buffers at its slab offset, the SO3 pre-alignment computed once and shared); batch_tracking = 0 runs one chain per
    model on the model's own stream.  Same kernels on the same data: poses and maps must agree bit for bit."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    w, h, n_frames, n_obj = 320, 240, 7, 4
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj, seed=31)
    rgb, depth = [dev(f["rgb"]) for f in frames], [dev(f["depth"]) for f in frames]

    def run(batch, hint=False):
        g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=n_obj,
                              batch_tracking=batch)
        known, out, keep = [0], [], []
        for i, f in enumerate(frames):
            spawn = 1 <= i <= n_obj
            if spawn:
                known.append(i)
            keep.append(dev(gt_mask(f["ids"], known)))
            nxt = (rgb[i + 1], depth[i + 1]) if hint and i + 1 < n_frames else None  # mmf_frame::next_*
            g.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn, next=nxt)
            out.append([m.getPose() for m in g.getModels()])
        maps = [m.downloadMap() for m in g.getModels()]
        stats = [(g.getModelOdometry(k).lastICPCount, g.getModelOdometry(k).lastRGBCount, g.getModelOdometry(k).iterations_run) for k in range(len(maps))]
        err = g.getErrorTexture(1, "icp").cpu().numpy().copy()
        g.close()
        return out, maps, stats, err

    # One chain per model means several chains in flight at once, which rules out the one-launch-per-iteration kernels
    # (a barrier inside every launch: csrc/gn_fused.hpp), so the bit-for-bit comparison runs on the producer + step chain;
    # the batched one-launch chain is compared bit for bit with single-model runs in
    # test_gpu_shard.py::test_a_model_leaving_the_list_moves_no_other_model and, below, within tolerance with this one.
    gpu_ctx.lib.mmf_debug_set_gn_fused(1)
    try:
        f1, f2 = run(1), run(1, hint=True)
    finally:
        gpu_ctx.lib.mmf_debug_set_gn_fused(0)
    try:
        a, b = run(1), run(0)
        c = run(1, hint=True)  # and with the next frame's sensor side (and SO3 pre-alignment) prepared on the side streams
        # Object models walk their images with a quarter of the workgroups and skip the blocks outside the model's own depth
        # (csrc/extent.hpp, ChainGeom), batched or alone -- a, b, c above.  Told not to, every model is tracked like the camera
        # model: again the same bits batched and alone, and the same poses up to the order of the float sums.
        gpu_ctx.lib.mmf_debug_set_track_cull(0)
        d, e = run(1), run(0)
    finally:
        gpu_ctx.lib.mmf_debug_set_gn_fused(-1)
        gpu_ctx.lib.mmf_debug_set_track_cull(-1)
    def apart(k, p, q):
        """how far two poses of object model k (spawned in frame k, at the identity) put the object's centre apart, in metres:
        measured where the object is -- a rotation error of a small object turns into centimetres at the camera origin"""
        c = np.linalg.inv(poses[k]) @ np.append(traj[k - 1][k][:3, :3] @ objs[k - 1]["centre"] + traj[k - 1][k][:3, 3], 1.0)
        return float(np.linalg.norm((np.linalg.inv(p.astype(np.float64)) @ c - np.linalg.inv(q.astype(np.float64)) @ c)[:3]))

    for i in range(n_frames):
        for k, (pd, pe) in enumerate(zip(d[0][i], e[0][i])):
            assert np.array_equal(pd, pe), (i, k)
        assert np.abs(a[0][i][0] - d[0][i][0]).max() <= 1e-5, i  # the camera
        for k, (pa, pd) in enumerate(zip(a[0][i][1:], d[0][i][1:])):  # free-running object models: see the bound below
            assert apart(k + 1, pa, pd) <= 1e-2, (i, k)
    for sd, se in zip(d[1], e[1]):
        assert np.array_equal(sd.view(np.uint32), se.view(np.uint32))
    assert d[2] == e[2]
    for i in range(n_frames):
        for pf, pg in zip(f1[0][i], f2[0][i]):
            assert np.array_equal(pf, pg), i
        assert np.abs(f1[0][i][0] - a[0][i][0]).max() <= 1e-5, i  # the camera; object poses amplify one-ulp noise
        # free-running object models over seven frames: the two chains add the same Jacobian rows up in different orders (exact
        # fixed-point totals against a float tree), and an object's few thousand pixels turn such one-ulp differences into
        # 1e-5 .. 1e-3 per frame in the ORACLE itself (test_oracle_fusion.py::test_object_tracking_is_sensitive_to_one_ulp_noise)
        # (the walks of the two chains are compared frame by frame from identical state in test_extent_walk_equals_dense_walk_frame_by_frame)
        for k, (pf, pa) in enumerate(zip(f1[0][i][1:], a[0][i][1:])):
            assert apart(k + 1, pf, pa) <= 1e-2, (i, k, apart(k + 1, pf, pa))
    assert all(s[2] == 19 for s in f1[2])
    for i in range(n_frames):
        for pa, pc in zip(a[0][i], c[0][i]):
            assert np.array_equal(pa, pc), i
    for sa, sc in zip(a[1], c[1]):
        assert np.array_equal(sa.view(np.uint32), sc.view(np.uint32))
    for i in range(n_frames):
        assert len(a[0][i]) == len(b[0][i]) == min(i, n_obj) + 1
        for k, (pa, pb) in enumerate(zip(a[0][i], b[0][i])):
            assert np.array_equal(pa, pb), (i, k, np.abs(pa - pb).max())
    for sa, sb in zip(a[1], b[1]):
        assert np.array_equal(sa.view(np.uint32), sb.view(np.uint32))
    assert a[2] == b[2] and all(s[2] == 19 for s in a[2])
    assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)) and (a[3] > 0).sum() > 100  # every model's own error image


def test_a_batched_chain_that_gives_up_is_tracked_again(gpu_ctx):
    """Two models in ONE one-launch chain (gridDim.y = 2): a launch that gives up (forced) voids the chain for both; the frame's
    tracking is enqueued again as the batched two-launch chain from both start poses.  Poses and maps equal, bit for bit, a run
    that switches to the two-launch chain at that frame by itself."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    lib = gpu_ctx.lib
    w, h, n_frames, n_obj, fault_at = 320, 240, 6, 1, 4
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj, seed=33)
    rgb, depth = [dev(f["rgb"]) for f in frames], [dev(f["depth"]) for f in frames]

    def run(forced):
        lib.mmf_debug_set_gn_fused(1)
        g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=n_obj)
        known, out, keep = [0], [], []
        try:
            for i, f in enumerate(frames):
                spawn = 1 <= i <= n_obj
                if spawn:
                    known.append(i)
                keep.append(dev(gt_mask(f["ids"], known)))
                if i == fault_at:
                    if forced:
                        lib.mmf_debug_force_gn_fault(1)
                    else:
                        lib.mmf_debug_set_gn_fused(0)
                g.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn)
                out.append([m.getPose() for m in g.getModels()])
            maps = [m.downloadMap() for m in g.getModels()]
        finally:
            g.close()
            lib.mmf_debug_force_gn_fault(0)
            lib.mmf_debug_set_gn_fused(-1)
        return out, maps

    pa, ma = run(True)
    pb, mb = run(False)
    assert len(pa[-1]) == 2
    for i in range(n_frames):
        for k, (a, b) in enumerate(zip(pa[i], pb[i])):
            assert np.array_equal(a, b), (i, k, np.abs(a - b).max())
    for a, b in zip(ma, mb):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_object_models_walked_by_their_extents_miss_nothing(gpu_ctx):
    """The one-launch chain walks an object model by its extents (csrc/gn_fused.hpp, gn_iter_mixed_kernel): the photometric term
    over the box of the model's own depth, the ICP term over the rectangle of sensor pixels the model's prediction can reach
    under the iteration's pose.  In checking mode such a model walks the WHOLE image instead and counts every correspondence
    icpStep (reduce.cu:231-397) accepts outside that rectangle, in every launch of every chain: there must be none -- which is
    the claim that the pixels the walk leaves out add exact zeros."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    lib = gpu_ctx.lib
    w, h, n_frames, n_obj = 640, 480, 7, 3
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj, seed=35)
    rgb, depth = [dev(f["rgb"]) for f in frames], [dev(f["depth"]) for f in frames]

    def run(check):
        lib.mmf_debug_set_gn_fused(1)
        lib.mmf_debug_set_sparse_check(check)
        g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=n_obj)
        known, keep, walked, out = [0], [], 0, []
        try:
            for i, f in enumerate(frames):
                spawn = 1 <= i <= n_obj
                if spawn:
                    known.append(i)
                keep.append(dev(gt_mask(f["ids"], known)))
                g.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn)
                n_models = len(g.getModels())
                for k in range(n_models):
                    od = g.getModelOdometry(k)
                    if od.iterations_run == 0:  # spawned this frame: not tracked yet
                        continue
                    outside, by_extent = od.sparseWalk()
                    assert od.iterations_run == 19
                    assert by_extent == (k > 0), (i, k)  # the camera model is walked densely, every object model by its extents
                    assert outside == 0, (i, k, outside)
                    walked += int(by_extent)
                    assert od.lastICPCount > (20000 if k == 0 else 300), (i, k, od.lastICPCount)
                out.append([m.getPose() for m in g.getModels()])
        finally:
            g.close()
            lib.mmf_debug_set_sparse_check(0)
            lib.mmf_debug_set_gn_fused(-1)
        return walked, out

    walked, pc = run(1)
    assert walked >= 12  # 3 + 3 + 3 + 2 + 1 object trackings at least
    import ctypes as C
    rec, in_use = C.c_int(0), C.c_int(0)
    lib.mmf_gn_chain_status(C.byref(rec), C.byref(in_use))
    assert in_use.value == 1
    # the same sequence walked by the rectangles (the default): the same Jacobian rows summed in another order
    _, pn = run(0)
    for i in range(n_frames):
        assert np.abs(pc[i][0] - pn[i][0]).max() <= 1e-5, i  # the camera
        for a, b in zip(pc[i][1:], pn[i][1:]):  # free-running objects (see test_batched_chain_equals_one_chain_per_model)
            assert np.abs(a - b).max() <= 1e-2, i


def test_an_extent_that_does_not_fit_is_tracked_again(gpu_ctx):
    """An object model whose extent does not fit the workgroups it is given (forced here: one workgroup) makes its chain report
    it (OdomState::gn_fault = 3): the frame is tracked again on the two-launch chain, the model's next chain is sized by what
    its box needed, and -- unlike a launch that could not become resident -- the process keeps the one-launch chain.  The frame's poses equal,
    bit for bit, those of a run that takes the two-launch chain at that frame by itself."""
    import ctypes as C
    from multimotionfusion_amd.fusion import MultiMotionFusion
    lib = gpu_ctx.lib
    w, h, n_frames, n_obj, fault_at = 320, 240, 5, 2, 4
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj, seed=33)
    rgb, depth = [dev(f["rgb"]) for f in frames], [dev(f["depth"]) for f in frames]

    def status():
        rec, in_use = C.c_int(0), C.c_int(0)
        lib.mmf_gn_chain_status(C.byref(rec), C.byref(in_use))
        return rec.value, in_use.value

    def run(forced):
        lib.mmf_debug_set_gn_fused(1)
        g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=n_obj)
        known, out, keep = [0], [], []
        try:
            for i, f in enumerate(frames):
                spawn = 1 <= i <= n_obj
                if spawn:
                    known.append(i)
                keep.append(dev(gt_mask(f["ids"], known)))
                if i == fault_at:
                    if forced:
                        lib.mmf_debug_set_sparse_groups(1)
                    else:
                        lib.mmf_debug_set_gn_fused(0)
                g.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn)
                out.append([m.getPose() for m in g.getModels()])
            walked = [g.getModelOdometry(k).sparseWalk()[1] for k in range(len(g.getModels()))]
        finally:
            g.close()
            lib.mmf_debug_set_sparse_groups(0)
            lib.mmf_debug_set_gn_fused(-1)
        return out, walked

    r0, _ = status()
    pa, wa = run(True)
    r1, in_use = status()
    assert r1 == r0 + 1 and in_use == 1  # tracked again once; the one-launch chain stays in use
    assert not any(wa)  # the frame that was tracked again went through the two-launch chain
    pb, _ = run(False)
    assert len(pa[-1]) == n_obj + 1
    for i in range(n_frames):
        for k, (a, b) in enumerate(zip(pa[i], pb[i])):
            assert np.array_equal(a, b), (i, k, np.abs(a - b).max())


@pytest.mark.parametrize("fast", [True, False])
def test_extent_walk_equals_dense_walk_frame_by_frame(gpu_ctx, fast):
    """The same frames through two fusion objects in lockstep, both on the one-launch chain: A walks its object models by their
    extents (gn_iter_mixed_kernel's sparse path), B walks every model like the camera model (mmf_debug_set_track_cull(0): the
    dense path for all).  After every frame B takes A's maps and poses over, so that each frame's tracking starts from identical
    state and the two walks add up the SAME Jacobian rows -- in different orders, hence not bit for bit, but: the Gauss-Newton
    system of the last iteration (lastA, lastb), the inlier and correspondence counts and the poses agree to what a changed
    summation order can do.  fast: three level-0 iterations only (fastOdom, no pyramid), where nothing has had time to amplify."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    lib = gpu_ctx.lib
    w, h, n_frames, n_obj = 320, 240, 6, 3
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj, seed=37)
    rgb, depth = [dev(f["rgb"]) for f in frames], [dev(f["depth"]) for f in frames]
    lib.mmf_debug_set_gn_fused(1)
    mk = lambda: MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=n_obj)  # noqa: E731
    A, B = mk(), mk()
    try:
        for g in (A, B):
            if fast:
                g.setFastOdom(True)
                g.setPyramid(False)
        known, keep, compared, diffs = [0], [], 0, []
        for i, f in enumerate(frames):
            spawn = 1 <= i <= n_obj
            if spawn:
                known.append(i)
            keep.append(dev(gt_mask(f["ids"], known)))
            lib.mmf_debug_set_track_cull(1)
            A.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn)
            lib.mmf_debug_set_track_cull(0)
            B.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn)
            ma, mb = A.getModels(), B.getModels()
            assert len(ma) == len(mb)
            for k in range(len(ma)):
                oa, ob = A.getModelOdometry(k), B.getModelOdometry(k)
                if oa.iterations_run == 0:
                    continue
                assert oa.sparseWalk()[1] == (k > 0) and not ob.sparseWalk()[1]
                assert oa.iterations_run == ob.iterations_run == (3 if fast else 19)
                na, nb = oa.lastICPCount, ob.lastICPCount
                assert abs(na - nb) <= max(2, (2e-4 if fast else 5e-3) * nb), (i, k, na, nb)
                assert abs(oa.lastRGBCount - ob.lastRGBCount) <= max(2, (2e-4 if fast else 5e-3) * ob.lastRGBCount), (i, k)
                sa, sb = np.abs(oa.lastA).max(), np.abs(oa.lastb).max()
                tolA = (1e-4 if fast else 5e-3)
                assert np.abs(oa.lastA - ob.lastA).max() <= tolA * sa, (i, k, np.abs(oa.lastA - ob.lastA).max() / sa)
                assert np.abs(oa.lastb - ob.lastb).max() <= tolA * max(sb, 1e-3 * sa), (i, k)
                pa, pb = ma[k].getPose(), mb[k].getPose()
                # (19 hard-gated iterations on a few thousand pixels amplify a changed summation order: 1e-5 .. 1e-3 per frame in
                # the ORACLE itself, test_oracle_fusion.py::test_object_tracking_is_sensitive_to_one_ulp_noise; the median stays small)
                assert np.abs(pa - pb).max() <= (2e-6 if fast or k == 0 else 1e-3), (i, k, np.abs(pa - pb).max())
                if k > 0:
                    diffs.append(float(np.abs(pa - pb).max()))
                    # the error images of the chain's last level-0 iteration (RGBDOdometry.cpp:367,408): A's launch walks the
                    # rectangle the SENSOR's depth range allows and zero-fills the rest, B's the whole image
                    for which, tol in (("icp", 2e-4), ("rgb", 0.0)):
                        ea, eb = A.getErrorTexture(k, which).cpu().numpy(), B.getErrorTexture(k, which).cpu().numpy()
                        assert (eb > 0).sum() > 100, (i, k, which)
                        off = np.abs(ea - eb) > tol
                        assert off.sum() <= (2 if fast else 0.03 * (eb > 0).sum()), (i, k, which, int(off.sum()), float(np.abs(ea - eb).max()))
                compared += int(k > 0)
            for a_, b_ in zip(ma, mb):  # B continues from A's state
                b_.uploadMap(a_.downloadMap())
                b_.overridePose(a_.getPose())
            B.predict()
        assert compared >= 6 and np.median(diffs) <= 2e-5, diffs
    finally:
        A.close()
        B.close()
        lib.mmf_debug_set_gn_fused(-1)
        lib.mmf_debug_set_track_cull(-1)


@pytest.mark.parametrize("n_obj", [4, 6])
def test_batched_passes_equal_passes_model_by_model(gpu_ctx, n_obj):
    """The object models' projection / fuse / clean / predict passes go out as ONE launch per pass for all of them (gridDim.y =
    model) instead of ~9 launches per model on the model's own stream (MultiMotionFusion.cpp:791-816, 863-875 loop over the
    models) -- restricted to where each model is (csrc/pass_rect.hpp: the boxes of its key-image writes, of its non-zero
    images and of its id in the id image) or covering the whole frame (csrc/surfel_kernels.hpp, *_batched_kernel).  Same
    per-texel code on the same data: every model's surfels (values AND order), poses, prediction images -- all of the
    image, also where the restricted passes never go -- and error images must agree bit for bit."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    lib = gpu_ctx.lib
    w, h, n_frames = 320, 240, n_obj + 4
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj, seed=39)
    rgb, depth = [dev(f["rgb"]) for f in frames], [dev(f["depth"]) for f in frames]

    def run(batch):
        lib.mmf_debug_set_pass_batch(batch)
        g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=n_obj)
        known, keep, out = [0], [], []
        try:
            for i, f in enumerate(frames):
                spawn = 1 <= i <= n_obj
                if spawn:
                    known.append(i)
                keep.append(dev(gt_mask(f["ids"], known)))
                nxt = (rgb[i + 1], depth[i + 1]) if i + 1 < n_frames else None
                g.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn, next=nxt)
                out.append([m.getPose() for m in g.getModels()])
            maps = [m.downloadMap() for m in g.getModels()]
            def tex_of(m, n):  # (the index-map getters copy on the MODEL's stream: wait for the device before reading)
                t = m.texture(n)
                torch.cuda.synchronize()
                return t.cpu().numpy().copy()
            tex = [[tex_of(m, n) for n in ("image", "vertexConf", "normalRadius", "index", "vertConf", "normRad")] for m in g.getModels()]
            err = [g.getErrorTexture(k, "icp").cpu().numpy().copy() for k in range(len(maps))]
        finally:
            g.close()
            lib.mmf_debug_set_pass_batch(-1)
        return out, maps, tex, err

    # restricted to where the models are (csrc/pass_rect.hpp) against model by model ... and the batched launches that cover the
    # whole frame.  Six objects: the DEFAULT (-1: model by model up to three object models on a GPU, restricted launches from the
    # fourth on -- the mode changes in the middle of the sequence, when the fourth object is spawned) against model by model
    a, b = run(2 if n_obj == 4 else -1), run(0)
    c = run(1) if n_obj == 4 else a
    for x, y in ((a, c),):
        for i in range(n_frames):
            for pa, pb in zip(x[0][i], y[0][i]):
                assert np.array_equal(pa, pb), i
        for ma, mb in zip(x[1], y[1]):
            assert ma.shape == mb.shape and np.array_equal(ma.view(np.uint32), mb.view(np.uint32))
    assert len(a[1]) == n_obj + 1 and all(m.shape[0] > 100 for m in a[1])
    for i in range(n_frames):
        for k, (pa, pb) in enumerate(zip(a[0][i], b[0][i])):
            assert np.array_equal(pa, pb), (i, k)
    for k, (ma, mb) in enumerate(zip(a[1], b[1])):
        assert ma.shape == mb.shape and np.array_equal(ma.view(np.uint32), mb.view(np.uint32)), k
    for k, (ta, tb) in enumerate(zip(a[2], b[2])):
        for x, y in zip(ta, tb):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), k
    for ea, eb in zip(a[3], b[3]):
        assert np.array_equal(ea.view(np.uint32), eb.view(np.uint32))


@pytest.mark.parametrize("fused", [1, 0])
def test_object_preparation_by_box_equals_whole_frame_preparation(gpu_ctx, fused):
    """An object model's model-side preparation (the pyramids, global-frame records and point clouds of initICPModel /
    initRGBModel, csrc/prep_batch.hpp) covers only the hull of the box its prediction is non-zero in now and at its previous
    preparation; told not to (mmf_debug_set_prep_rect(0)) it covers the frame.  The buffers must be the same everywhere:
    poses, maps and error images bit for bit under the one-launch chain (which reads them near the boxes) AND under the
    two-launch chain (which reads all of them), and the downloadable pyramid levels themselves."""
    from multimotionfusion_amd.fusion import MultiMotionFusion
    lib = gpu_ctx.lib
    w, h, n_frames, n_obj = 320, 240, 8, 3
    K, poses, traj, frames, objs = scene(w, h, n_frames, n_obj, seed=41)
    rgb, depth = [dev(f["rgb"]) for f in frames], [dev(f["depth"]) for f in frames]

    def run(rect):
        lib.mmf_debug_set_gn_fused(fused)
        lib.mmf_debug_set_prep_rect(rect)
        g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=n_obj)
        known, keep, out, bufs = [0], [], [], []
        try:
            for i, f in enumerate(frames):
                spawn = 1 <= i <= n_obj
                if spawn:
                    known.append(i)
                keep.append(dev(gt_mask(f["ids"], known)))
                nxt = (rgb[i + 1], depth[i + 1]) if i + 1 < n_frames else None
                g.processFrame(rgb[i], depth[i], timestamp=i, mask=keep[-1], hasNewLabel=spawn, next=nxt)
                out.append([m.getPose() for m in g.getModels()])
                torch.cuda.synchronize()
                bufs.append([[g.getModelOdometry(k).download(nm, lvl) for nm in ("last_depth", "last_image") for lvl in range(3)]
                             for k in range(len(g.getModels()))])
            maps = [m.downloadMap() for m in g.getModels()]
            err = [g.getErrorTexture(k, "icp").cpu().numpy().copy() for k in range(len(maps))]
        finally:
            g.close()
            lib.mmf_debug_set_prep_rect(-1)
            lib.mmf_debug_set_gn_fused(-1)
        return out, maps, err, bufs

    a, b = run(1), run(0)
    for i in range(n_frames):
        for k, (pa, pb) in enumerate(zip(a[0][i], b[0][i])):
            assert np.array_equal(pa, pb), (i, k, np.abs(pa - pb).max())
        for k, (ba, bb) in enumerate(zip(a[3][i], b[3][i])):
            for j, (x, y) in enumerate(zip(ba, bb)):
                assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (i, k, j)
    for ma, mb in zip(a[1], b[1]):
        assert ma.shape == mb.shape and np.array_equal(ma.view(np.uint32), mb.view(np.uint32))
    for ea, eb in zip(a[2], b[2]):
        assert np.array_equal(ea.view(np.uint32), eb.view(np.uint32))
