"""-m gpu: the pure-HIP surfel path (mmf_model_*, mmf_filter_depth) against the oracle's restatement
of the reference's GL passes.  Everything on this path is per-element work with integer decisions,
so every output must be BIT-EXACT: filtered depth, surfel arrays after initialise / fuse / clean
(including their order), the index map's pixel assignments and attributes, the splat prediction."""
import numpy as np
import pytest
import torch

from helpers import assert_bit_equal
from multimotionfusion_amd import synth

pytestmark = pytest.mark.gpu
MAXD = 20.0      # maxDepthProcessed
CUTOFF = 15.0    # depthCutoff of the bilateral filter (GUI default)
TIME_DELTA = 200
CONF = 10.0      # confGlobalInit


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_model(gpu_ctx, w, h, conf=CONF):
    from multimotionfusion_amd.model import Model
    K = synth.intrinsics(w, h)
    return K, Model(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], 0, conf)


@pytest.mark.parametrize("w,h", [(640, 480), (160, 120)])
def test_filter_depth_bit_exact(gpu_ctx, orc, w, h):
    from multimotionfusion_amd.model import filterDepth
    f = synth.render(np.eye(4), w, h, seed=3, depth_noise=1e-3)
    out = filterDepth(gpu_ctx, dev(f["depth"]), CUTOFF)
    assert_bit_equal(out.cpu().numpy(), orc.bilateral_filter(f["depth"], CUTOFF), "bilateral filter")


@pytest.fixture
def splat_bound(gpu_ctx, request):
    """combinedPredict's early depth test (splat_kernel<true>: the key image is read before a fragment is evaluated): -1 = by the surfel count, 1 = always"""
    gpu_ctx.lib.mmf_debug_set_splat_bound(request.param)
    yield request.param
    gpu_ctx.lib.mmf_debug_set_splat_bound(-1)


@pytest.mark.parametrize("splat_bound", [-1, 1], indirect=True)
@pytest.mark.parametrize("w,h", [(640, 480), (160, 120)])
def test_surfel_cycle_bit_exact(gpu_ctx, orc, w, h, splat_bound):
    """initialise -> (predictIndices, fuse, predictIndices, clean, combinedPredict, fill-in) x 3 frames."""
    from multimotionfusion_amd.model import filterDepth
    K, m = make_model(gpu_ctx, w, h)
    poses = synth.trajectory(4, seed=5)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    mask = np.zeros((h, w), np.uint8)
    d_mask = dev(mask)

    f0 = frames[0]
    fil0 = orc.bilateral_filter(f0["depth"], CUTOFF)
    d_fil0 = filterDepth(gpu_ctx, dev(f0["depth"]), CUTOFF)
    m.overridePose(poses[0])
    m.initialise(dev(f0["rgb"]), dev(f0["depth"]), d_fil0, 1, MAXD)
    s = orc.surfel_initialise(f0["rgb"], f0["depth"], fil0, K, 1, MAXD)
    assert m.lastCount() == s.shape[0] > 0.8 * w * h
    assert_bit_equal(m.downloadMap(), s, "initialise")

    for t in range(1, 4):
        tick = t + 1
        f = frames[t]
        pose = poses[t].astype(np.float32)  # ground-truth pose stands in for the tracker here
        fil = orc.bilateral_filter(f["depth"], CUTOFF)
        d_rgb, d_raw = dev(f["rgb"]), dev(f["depth"])
        d_fil = filterDepth(gpu_ctx, d_raw, CUTOFF)
        m.overridePose(pose)

        m.predictIndices(tick, MAXD, TIME_DELTA)
        index, vc, ct, nr = orc.predict_indices(s, pose, K, w, h, MAXD, tick, TIME_DELTA)
        assert_bit_equal(m.texture("index").cpu().numpy().view(np.uint32), index, f"index map t={t}")
        assert_bit_equal(m.texture("vertConf").cpu().numpy(), vc, f"vertConf t={t}")
        assert_bit_equal(m.texture("colorTime").cpu().numpy(), ct, f"colorTime t={t}")
        assert_bit_equal(m.texture("normRad").cpu().numpy(), nr, f"normRad t={t}")

        m.fuse(tick, d_rgb, d_mask, d_raw, d_fil, MAXD, 1.0)
        s_upd, new = orc.fuse(s, f["rgb"], f["depth"], fil, mask, index, vc, nr, pose, K, tick, 1.0, 0, MAXD)
        assert_bit_equal(m.downloadMap(), s_upd, f"fused surfels t={t}")

        m.predictIndices(tick, MAXD, TIME_DELTA)
        index, vc, ct, nr = orc.predict_indices(s_upd, pose, K, w, h, MAXD, tick, TIME_DELTA)
        assert_bit_equal(m.texture("index").cpu().numpy().view(np.uint32), index, f"index map after fuse t={t}")

        m.clean(tick, TIME_DELTA, MAXD, d_fil, d_mask, 3.0)
        s = orc.clean(s_upd, new, pose, K, w, h, tick, TIME_DELTA, CONF, 3.0, 0, index, vc, ct, fil, mask)
        assert m.lastCount() == s.shape[0]
        assert_bit_equal(m.downloadMap(), s, f"cleaned surfels t={t}")

        m.combinedPredict(MAXD, tick, tick, TIME_DELTA)
        image, vcp, nrp, tm = orc.combined_predict(s, pose, K, w, h, MAXD, CONF, tick, tick, TIME_DELTA)
        assert_bit_equal(m.texture("image").cpu().numpy(), image, f"splat image t={t}")
        assert_bit_equal(m.texture("vertexConf").cpu().numpy(), vcp, f"splat vertexConf t={t}")
        assert_bit_equal(m.texture("normalRadius").cpu().numpy(), nrp, f"splat normalRadius t={t}")
        assert_bit_equal(m.texture("time").cpu().numpy().view(np.uint16), tm, f"splat time t={t}")
        # ModelProjection::synthesizeDepth: the same sprites, depth only, explicit confidence threshold
        for conf in (CONF, 0.5):
            m.synthesizeDepth(MAXD, conf, tick, tick, TIME_DELTA)
            sd = orc.synthesize_depth(s, pose, K, w, h, MAXD, conf, tick, tick, TIME_DELTA)
            assert_bit_equal(m.texture("depth").cpu().numpy(), sd, f"synthesized depth t={t} conf={conf}")

        m.performFillIn(d_rgb, d_fil, False, False)
        vo, no, io = orc.fill_in(vcp, nrp, image, fil, f["rgb"], K, 0, 0)
        assert_bit_equal(m.texture("fillVertex").cpu().numpy(), vo, f"fill vertex t={t}")
        assert_bit_equal(m.texture("fillNormal").cpu().numpy(), no, f"fill normal t={t}")
        assert_bit_equal(m.texture("fillImage").cpu().numpy(), io, f"fill image t={t}")
        assert m.requiresFillIn(0.75) == orc.requires_fill_in(image, 0.75)
    m.close()


@pytest.mark.parametrize("w,h", [(320, 240), (640, 480)])
def test_bounded_splat_keeps_the_images_of_a_deep_store(gpu_ctx, orc, w, h):
    """A store with occluded layers (what a room seen from many sides leaves: here the map of one frame plus three copies
    pushed 2 / 4 / 6 cm behind the surfaces, all stable) is drawn with and without the early depth test (splat_kernel<true>): the four images
    must keep their bits -- and equal the oracle's -- while most fragments are never evaluated."""
    from multimotionfusion_amd.model import filterDepth
    K, m = make_model(gpu_ctx, w, h)
    poses = synth.trajectory(2, seed=9)
    f0 = synth.render(poses[0], w, h, seed=0)
    m.overridePose(poses[0])
    m.initialise(dev(f0["rgb"]), dev(f0["depth"]), filterDepth(gpu_ctx, dev(f0["depth"]), CUTOFF), 1, MAXD)
    base = m.downloadMap()
    layers = []
    cam = np.linalg.inv(poses[0]).astype(np.float64)
    for step in (0.0, 0.02, 0.04, 0.06):
        cp = base.copy()
        pc = cp[:, :3].astype(np.float64) @ cam[:3, :3].T + cam[:3, 3]  # into the first camera's frame, along its rays, back
        pc += step * pc / np.maximum(np.linalg.norm(pc, axis=1, keepdims=True), 1e-6)
        cp[:, :3] = ((pc - cam[:3, 3]) @ cam[:3, :3]).astype(np.float32)
        cp[:, 3] = 20.0
        layers.append(cp)
    rng = np.random.default_rng(4)
    deep = np.concatenate(layers)
    deep = deep[rng.permutation(deep.shape[0])][: 1024 * 1024 - 8]  # draw order unrelated to depth
    # surfels with a zero normal (a first frame has them): their fragments' z is NaN, which wins every depth test -- in both passes
    deep[rng.choice(deep.shape[0], 300, replace=False), 8:11] = 0.0
    m.uploadMap(deep)
    out = {}
    for pose_i, pose in enumerate(poses):
        m.overridePose(pose)
        for mode in (0, 1):
            gpu_ctx.lib.mmf_debug_set_splat_bound(mode)
            try:
                m.combinedPredict(MAXD, 2, 2, TIME_DELTA)
            finally:
                gpu_ctx.lib.mmf_debug_set_splat_bound(-1)
            out[mode] = [m.texture(n).cpu().numpy().copy() for n in ("image", "vertexConf", "normalRadius", "time")]
        for a, b, n in zip(out[0], out[1], ("image", "vertexConf", "normalRadius", "time")):
            assert_bit_equal(a, b, f"{n}, pose {pose_i}: bounded against plain depth test")
        if w <= 320:
            image, vcp, nrp, tm = orc.combined_predict(deep, pose.astype(np.float32), K, w, h, MAXD, CONF, 2, 2, TIME_DELTA)
            assert_bit_equal(out[1][0], image, "image against the oracle")
            assert_bit_equal(out[1][1], vcp, "vertexConf against the oracle")
        assert (out[1][1][..., 2] > 0).mean() > 0.2
    # two passes in a row: the bound is handed back full by the resolve pass
    gpu_ctx.lib.mmf_debug_set_splat_bound(1)
    try:
        m.combinedPredict(MAXD, 2, 2, TIME_DELTA)
        again = m.texture("vertexConf").cpu().numpy()
    finally:
        gpu_ctx.lib.mmf_debug_set_splat_bound(-1)
    assert_bit_equal(again, out[1][1], "second bounded pass")
    m.close()


def test_index_map_edge_cases(gpu_ctx, orc):
    w, h = 160, 120
    K, m = make_model(gpu_ctx, w, h)

    def surf(x, y, z, t=1.0, conf=20.0):
        return [x, y, z, conf, 255.0, 0, 1, t, 0, 0, -1, 0.02]
    s = np.array([surf(0, 0, 3.0), surf(0, 0, 2.0), surf(0, 0, 2.0), surf(0, 0, 30.0), surf(0.5, 0, 1.0, t=-500.0),
                  surf(100.0, 0, 1.0), surf(0, 0, -1.0), surf(0.1, 0.1, 0.0), surf(-0.3, 0.2, 1.5)], np.float32)
    m.uploadMap(s)
    m.overridePose(np.eye(4))
    m.predictIndices(10, MAXD, TIME_DELTA)
    index, vc, ct, nr = orc.predict_indices(s, np.eye(4), K, w, h, MAXD, 10, TIME_DELTA)
    assert_bit_equal(m.texture("index").cpu().numpy().view(np.uint32), index, "index edge cases")
    assert_bit_equal(m.texture("vertConf").cpu().numpy(), vc, "vertConf edge cases")
    m.combinedPredict(MAXD, 10, 10, TIME_DELTA)
    image, vcp, nrp, tm = orc.combined_predict(s, np.eye(4), K, w, h, MAXD, CONF, 10, 10, TIME_DELTA)
    assert_bit_equal(m.texture("vertexConf").cpu().numpy(), vcp, "splat edge cases")
    # empty store
    m.uploadMap(np.zeros((0, 12), np.float32))
    m.predictIndices(10, MAXD, TIME_DELTA)
    assert not m.texture("index").any()
    m.close()


def test_surfel_cycle_against_golden_fixture(gpu_ctx):
    """The HIP surfel path on the committed fixture (tests/golden/surfel_cycle_96x72.npz): no oracle call, the
    expected arrays travel with the repository."""
    import os
    from multimotionfusion_amd.model import Model, filterDepth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "surfel_cycle_96x72.npz"))
    h, w = g["depth0"].shape
    fx, fy, cx, cy = (float(v) for v in g["intr"])
    m = Model(gpu_ctx, w, h, cx, cy, fx, fy, 0, CONF)
    d_mask = dev(np.zeros((h, w), np.uint8))
    d0, d1 = dev(g["depth0"]), dev(g["depth1"])
    fil0, fil1 = filterDepth(gpu_ctx, d0, CUTOFF), filterDepth(gpu_ctx, d1, CUTOFF)
    assert_bit_equal(fil1.cpu().numpy(), g["filtered1"], "filtered depth")
    m.overridePose(g["pose0"])
    m.initialise(dev(g["rgb0"]), d0, fil0, 1, MAXD)
    assert_bit_equal(m.downloadMap(), g["surfels_init"], "initialise")
    m.overridePose(g["pose1"])
    m.predictIndices(2, MAXD, TIME_DELTA)
    m.fuse(2, dev(g["rgb1"]), d_mask, d1, fil1, MAXD, 1.0)
    m.predictIndices(2, MAXD, TIME_DELTA)
    assert_bit_equal(m.texture("index").cpu().numpy().view(np.uint32), g["index_after_fuse"], "index map")
    m.clean(2, TIME_DELTA, MAXD, fil1, d_mask, 3.0)
    s1 = m.downloadMap()
    assert_bit_equal(s1, g["surfels_final"], "surfels after clean")
    s1[:, 3] = 20.0
    m.uploadMap(s1)
    m.combinedPredict(MAXD, 2, 2, TIME_DELTA)
    assert_bit_equal(m.texture("vertexConf").cpu().numpy(), g["splat_vertexConf"], "splat vertexConf")
    assert_bit_equal(m.texture("image").cpu().numpy(), g["splat_image"], "splat image")
    m.close()


def test_surfel_capacity_is_a_hard_limit(gpu_ctx, orc):
    """Model::MAX_VERTICES bounds the vertex buffer (Model.cpp:119-126): a first frame with more valid pixels
    than that keeps the first `capacity` surfels of the draw order (GL transform feedback into a full buffer
    drops the rest) -- found the hard way at 1280x960, where a frame has more pixels than 1024^2 / 1 surfels.
    Later passes must respect the limit too."""
    from multimotionfusion_amd.model import Model, filterDepth
    w, h, cap = 160, 120, 5000
    K = synth.intrinsics(w, h)
    m = Model(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], 0, CONF, max_surfels=cap)
    poses = synth.trajectory(2, seed=5)
    f0, f1 = synth.render(poses[0], w, h, seed=0), synth.render(poses[1], w, h, seed=1)
    d0 = dev(f0["depth"])
    fil0 = filterDepth(gpu_ctx, d0, CUTOFF)
    m.overridePose(poses[0])
    m.initialise(dev(f0["rgb"]), d0, fil0, 1, MAXD)
    s = orc.surfel_initialise(f0["rgb"], f0["depth"], orc.bilateral_filter(f0["depth"], CUTOFF), K, 1, MAXD)
    assert s.shape[0] > cap and m.lastCount() == cap
    assert_bit_equal(m.downloadMap(), s[:cap], "first `capacity` surfels of the draw order")
    # a full fuse / clean cycle on the full buffer: new surfels have nowhere to go, nothing may fault
    d1 = dev(f1["depth"])
    fil1 = filterDepth(gpu_ctx, d1, CUTOFF)
    m.overridePose(poses[1])
    m.predictIndices(2, MAXD, TIME_DELTA)
    m.fuse(2, dev(f1["rgb"]), dev(np.zeros((h, w), np.uint8)), d1, fil1, MAXD, 1.0)
    m.predictIndices(2, MAXD, TIME_DELTA)
    m.clean(2, TIME_DELTA, MAXD, fil1, dev(np.zeros((h, w), np.uint8)), 3.0)
    assert 0 < m.lastCount() <= cap
    m.combinedPredict(MAXD, 2, 2, TIME_DELTA)
    assert m.downloadMap().shape[0] == m.lastCount()
    m.close()


def test_device_exponentials_against_float64(gpu_ctx):
    """The gfx950 build of mmf_expf AND the separately written packed exponential of the two-pixel bilateral filter
    (expf_nonpositive2, surfel_kernels.hpp) against numpy's float64 exp over dense argument sweeps: <= 2 ulp each, and
    bit-identical to each other where the packed one is defined (x <= 0).  An independent check: the oracle shares
    mmf_expf's source with the kernels, so agreement with it says nothing about the function itself."""
    import ctypes
    xs = np.concatenate([np.linspace(-104.0, 0.0, 2_000_001), -np.logspace(-8, 2, 200_001), -np.arange(0, 104, 0.693359375 / 2),
                         [0.0, -0.0, -87.3, -87.4, -88.0, -103.0, -103.5]]).astype(np.float32)
    pos = np.concatenate([np.linspace(0.0, 88.7, 500_001), [1e-8, 1.0]]).astype(np.float32)
    for args, packed_defined in ((xs, True), (pos, False)):
        x = torch.from_numpy(args).cuda()
        a, b = torch.empty_like(x), torch.empty_like(x)
        assert gpu_ctx.lib.mmf_debug_expf(gpu_ctx.handle, ctypes.c_void_p(x.data_ptr()), x.numel(), ctypes.c_void_p(a.data_ptr()),
                                          ctypes.c_void_p(b.data_ptr())) == 0
        torch.cuda.synchronize()
        a, b = a.cpu().numpy(), b.cpu().numpy()
        want = np.exp(args.astype(np.float64))
        normal = want > 1.2e-38  # below: subnormal results, compared absolutely
        ulp = np.abs(a[normal].astype(np.float64) - want[normal]) / np.spacing(want[normal].astype(np.float32)).astype(np.float64)
        assert ulp.max() <= 2.0, ulp.max()
        assert np.abs(a[~normal].astype(np.float64) - want[~normal]).max() <= 2 * 1.4e-45 * 2 ** 23 if (~normal).any() else True
        if packed_defined:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    nan = torch.tensor([float("nan"), -1.0, 89.0, -200.0], device="cuda")
    a, b = torch.empty_like(nan), torch.empty_like(nan)
    gpu_ctx.lib.mmf_debug_expf(gpu_ctx.handle, ctypes.c_void_p(nan.data_ptr()), 4, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()))
    torch.cuda.synchronize()
    assert torch.isnan(a[0]) and torch.isnan(b[0]) and a[1] == b[1] and torch.isinf(a[2]) and a[3] == 0 and b[3] == 0


@pytest.mark.parametrize("w,h,obj_id", [(320, 240, 2), (640, 480, 3)])
def test_object_model_cycle_bit_exact(gpu_ctx, orc, w, h, obj_id):
    """An OBJECT model as the segmentation spawns it (MultiMotionFusion.cpp:588-601, 791-816): id != 0, confidence
    threshold confObjectInit = 0.01, no fill-in, created empty at the identity pose, fed through a ground-truth id
    image that also holds other ids -- only pixels with mask == id may be fused (data.vert:118, copy_unstable.vert:120) --
    and with Model::setMaxDepth in force.  Every pass bit-exact against the oracle."""
    from multimotionfusion_amd.model import Model, filterDepth
    CONF_OBJ = 0.01
    K = synth.intrinsics(w, h)
    n = 4
    poses = synth.trajectory(n, seed=9)
    objs = synth.make_objects(4, seed=9)
    traj = synth.object_trajectories(objs, n, seed=9)
    frames = [synth.render(p, w, h, seed=i, objects=objs, object_poses=[t[i] for t in traj]) for i, p in enumerate(poses)]
    m = Model(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], obj_id, CONF_OBJ)
    assert m.lastCount() == 0
    s = np.zeros((0, 12), np.float32)
    for t in range(n):
        tick = t + 2
        f = frames[t]
        mask = f["ids"].astype(np.uint8)
        assert (mask == obj_id).sum() > 500 and len(np.unique(mask)) >= 4
        depth_of_object = f["depth"][mask == obj_id]
        max_depth = float(np.float32(depth_of_object.mean() + 1.2 * np.abs(depth_of_object - depth_of_object.mean()).mean()))
        m.setMaxDepth(max_depth)
        # model frame = camera frame of the first frame: P(t) = C_0^-1 T(0) T(t)^-1 C_t (see test_gpu_multimodel)
        pose = (np.linalg.inv(poses[0]) @ traj[obj_id - 1][0] @ np.linalg.inv(traj[obj_id - 1][t]) @ poses[t]).astype(np.float32)
        fil = orc.bilateral_filter(f["depth"], CUTOFF)
        d_rgb, d_raw, d_mask = dev(f["rgb"]), dev(f["depth"]), dev(mask)
        d_fil = filterDepth(gpu_ctx, d_raw, CUTOFF)
        m.overridePose(pose)
        weight = 100.0 if t == 0 else 0.75
        m.predictIndices(tick, MAXD, TIME_DELTA)
        index, vc, ct, nr = orc.predict_indices(s, pose, K, w, h, MAXD, tick, TIME_DELTA)
        assert_bit_equal(m.texture("index").cpu().numpy().view(np.uint32), index, f"index map t={t}")
        m.fuse(tick, d_rgb, d_mask, d_raw, d_fil, MAXD, weight)
        s_upd, new = orc.fuse(s, f["rgb"], f["depth"], fil, mask, index, vc, nr, pose, K, tick, weight, obj_id, min(MAXD, max_depth))
        assert_bit_equal(m.downloadMap(), s_upd, f"fused surfels t={t}")
        if t > 0:  # at the spawn the second predictIndices is commented out (MultiMotionFusion.cpp:594)
            m.predictIndices(tick, MAXD, TIME_DELTA)
            index, vc, ct, nr = orc.predict_indices(s_upd, pose, K, w, h, MAXD, tick, TIME_DELTA)
        m.clean(tick, TIME_DELTA, MAXD, d_fil, d_mask, 3.0)
        s = orc.clean(s_upd, new, pose, K, w, h, tick, TIME_DELTA, CONF_OBJ, 3.0, obj_id, index, vc, ct, fil, mask)
        assert m.lastCount() == s.shape[0]
        assert_bit_equal(m.downloadMap(), s, f"cleaned surfels t={t}")
        # every surfel came from a pixel of the object: within the object's depth range in the camera frame
        if t == 0:
            assert s.shape[0] > 100 and s[:, 2].max() <= max_depth + 1e-3
        m.combinedPredict(MAXD, tick, tick, TIME_DELTA)
        image, vcp, nrp, tm = orc.combined_predict(s, pose, K, w, h, MAXD, CONF_OBJ, tick, tick, TIME_DELTA)
        assert_bit_equal(m.texture("image").cpu().numpy(), image, f"splat image t={t}")
        assert_bit_equal(m.texture("vertexConf").cpu().numpy(), vcp, f"splat vertexConf t={t}")
        assert_bit_equal(m.texture("normalRadius").cpu().numpy(), nrp, f"splat normalRadius t={t}")
        assert (vcp[..., 2] > 0).sum() > 100
    m.close()


@pytest.mark.gpu
def test_depth_key_without_the_division_is_the_depth_key(gpu_ctx):
    """combo_splat.frag's depth is (z / (2 maxDepth)) + 0.5; the rasterising pass computes the quotient from a reciprocal of the
    launch's constant with one correction step (splat_depth24_fast).  Same 24-bit key for every float a depth can be -- random
    bit patterns (denormals, infinities, NaNs, negatives), the plausible range densely, values around key boundaries -- and for
    every cut-off, not only the ones a test scene uses."""
    import torch
    rng = np.random.default_rng(5)
    n = 1 << 20
    bits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    dense = rng.uniform(-1.0, 30.0, n).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38], np.float32)
    cutoffs = [0.7, 3.0, 4.5, 5.0, 12.3, 20.0, 100.0] + list(rng.uniform(0.05, 200.0, 24).astype(np.float32)) + \
              list(np.exp(rng.uniform(np.log(1e-3), np.log(1e6), 8)).astype(np.float32))
    fast = torch.empty(n, dtype=torch.int32, device="cuda")
    div = torch.empty(n, dtype=torch.int32, device="cuda")
    for md in cutoffs:
        md = float(np.float32(md))
        # depths whose quotient sits next to a key boundary: (k + 0.5) / 16777215 - 0.5, times 2 maxDepth, +- a few ulps
        k = rng.integers(0, 1 << 24, n // 4)
        edge = (((k + 0.5) / 16777215.0 - 0.5) * (2.0 * md)).astype(np.float32)
        edge = np.concatenate([edge, np.nextafter(edge, np.float32(np.inf)), np.nextafter(edge, np.float32(-np.inf)), edge[: n // 4 - special.size], special])
        for z in (bits, dense, edge):
            zd = torch.from_numpy(np.ascontiguousarray(z)).cuda()
            assert gpu_ctx.lib.mmf_debug_depth_keys(gpu_ctx.handle, zd.data_ptr(), int(zd.numel()), md, fast.data_ptr(), div.data_ptr()) == 0
            torch.cuda.synchronize()
            a, b = fast[: zd.numel()].cpu().numpy(), div[: zd.numel()].cpu().numpy()
            bad = np.nonzero(a != b)[0]
            assert bad.size == 0, (md, z[bad[:4]], a[bad[:4]], b[bad[:4]])
