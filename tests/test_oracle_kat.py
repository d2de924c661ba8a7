"""CPU tests of the oracle itself: known-answer properties of the restated algorithms
(the reference ships no vectors for this path -- SURVEY.md section 4 -- so these pin the
oracle's behaviour on analytic cases), and the committed golden fixtures."""
import os

import numpy as np
import pytest

from helpers import ANGLE_THRESH, DIST_THRESH, frame_pair
from multimotionfusion_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def plane_maps(orc, w, h, K, z=2.0):
    depth = np.full((h, w), z, np.float32)
    vm = orc.create_vmap(depth, K["fx"], K["fy"], K["cx"], K["cy"], 15.0)
    nm = orc.create_nmap(vm)
    return depth, vm, nm


def test_vmap_nmap_of_a_plane(orc):
    w, h = 64, 48
    K = synth.intrinsics(w, h)
    depth, vm, nm = plane_maps(orc, w, h, K)
    u, v = np.meshgrid(np.arange(w), np.arange(h))
    assert np.allclose(vm[:h], 2.0 * (u - K["cx"]) / K["fx"], atol=1e-6)
    assert np.allclose(vm[h:2 * h], 2.0 * (v - K["cy"]) / K["fy"], atol=1e-6)
    assert np.all(vm[2 * h:] == 2.0)
    # forward-difference normal of a fronto-parallel plane is +z; last row / column invalid
    assert np.isnan(nm[:h][:, -1]).all() and np.isnan(nm[:h][-1, :]).all()
    assert np.allclose(nm[2 * h:][:-1, :-1], 1.0, atol=1e-6)
    assert np.allclose(nm[:h][:-1, :-1], 0.0, atol=1e-6)


def test_vmap_invalid_depth_marks_only_x_plane(orc):
    w, h = 32, 32
    K = synth.intrinsics(w, h)
    depth = np.full((h, w), 1.0, np.float32)
    depth[4, 5] = 0.0
    depth[6, 7] = 20.0  # beyond the cutoff
    depth[8, 9] = np.nan
    vm = orc.create_vmap(depth, K["fx"], K["fy"], K["cx"], K["cy"], 15.0)
    for y, x in ((4, 5), (6, 7), (8, 9)):
        assert np.isnan(vm[y, x])
    assert np.isnan(vm[:h]).sum() == 3


def test_icp_identity_motion_has_zero_rhs(orc):
    """Same surface seen from the same pose: every residual vanishes => b = 0, A is SPD-ish."""
    w, h = 64, 48
    K = synth.intrinsics(w, h)
    f = synth.render(np.eye(4), w, h, noise=False, dropout=0)
    vm = orc.create_vmap(f["depth"], K["fx"], K["fy"], K["cx"], K["cy"], 15.0)
    nm = orc.create_nmap(vm)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    out, err = orc.icp_step(I, z, vm, nm, I, z, K["fx"], K["fy"], K["cx"], K["cy"], vm, nm, DIST_THRESH,
                            ANGLE_THRESH, want_err=True)
    A, b, res = orc.unpack_se3(out)
    assert res[1] == np.sum(~np.isnan(nm[:h]))  # every pixel with a normal is an inlier
    assert np.all(b == 0) and res[0] == 0 and np.all(err == 0)
    assert np.array_equal(A, A.T)
    assert np.all(np.linalg.eigvalsh(A.astype(np.float64)) > -1e-3)


def test_icp_plane_translation_known_answer(orc):
    """Fronto-parallel plane moved by dz along its normal: the point-to-plane system has the
    closed-form solution t_z = -dz... (sign per the reference's update rule) and no rotation."""
    w, h = 64, 48
    K = synth.intrinsics(w, h)
    _, vm_prev, nm_prev = plane_maps(orc, w, h, K, 2.0)
    _, vm_cur, nm_cur = plane_maps(orc, w, h, K, 2.01)
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    out, _ = orc.icp_step(I, z, vm_cur, nm_cur, I, z, K["fx"], K["fy"], K["cx"], K["cy"], vm_prev, nm_prev,
                          DIST_THRESH, ANGLE_THRESH)
    A, b, res = orc.unpack_se3(out)
    n = res[1]
    assert n > 0.9 * (w - 1) * (h - 1)
    # row = [n, s x n, n.(s-d)] with n = (0,0,1): A[2,2] = count, b[2] = sum r = n * 0.01
    assert abs(A[2, 2] - n) < 1e-3 * n
    assert abs(b[2] - 0.01 * n) < 1e-3 * n * 0.01 + 1e-4
    assert abs(res[0] - n * 1e-4) < 1e-2 * n * 1e-4


def test_so3_identity_has_zero_rhs(orc):
    w, h = 80, 60
    K = synth.intrinsics(w, h)
    img = orc.image_to_intensity(synth.render(np.eye(4), w, h)["rgb"])
    Km = np.array([[K["fx"], 0, K["cx"]], [0, K["fy"], K["cy"]], [0, 0, 1.0]])
    out = orc.so3_step(img, img, np.eye(3), np.linalg.inv(Km), Km)
    A, b, res = orc.unpack_so3(out)
    assert np.all(b == 0) and res[0] == 0 and res[1] == (w - 2) * (h - 2)
    assert np.array_equal(A, A.T)


def test_pyramid_sizes_and_nan_skipping(orc):
    src = np.arange(40 * 30, dtype=np.float32).reshape(30, 40)
    out = orc.pyrdown_gauss_f(src)
    assert out.shape == (15, 20)
    src2 = src.copy()
    src2[10, 10] = np.nan  # a NaN tap is skipped, weights renormalised: the result stays finite
    assert np.isfinite(orc.pyrdown_gauss_f(src2)).all()
    # interior taps of a constant image reproduce the constant
    c = np.full((30, 40), 3.5, np.float32)
    assert np.allclose(orc.pyrdown_gauss_f(c), 3.5)
    u8 = np.full((30, 40), 200, np.uint8)
    assert np.all(orc.pyrdown_uchar_gauss(u8) == 200)
    assert np.all(orc.pyrdown_uchar_gauss(np.zeros((30, 40), np.uint8)) == 0)  # empty window -> 0


def test_intensity_uses_uploaded_channel_order(orc):
    img = np.zeros((2, 2, 3), np.uint8)
    img[..., 0] = 100  # channel 0 weighted by 0.114 (cudafuncs.cu:634)
    assert np.all(orc.image_to_intensity(img) == int(np.float32(100) * np.float32(0.114)))
    img[...] = (10, 20, 30)
    assert np.all(orc.image_to_intensity(img) == int(10 * 0.114 + 20 * 0.299 + 30 * 0.587))


def test_derivative_of_a_ramp(orc):
    ramp = np.tile(np.arange(64, dtype=np.uint8) * 2, (16, 1))
    dx, dy = orc.derivative_images(ramp)
    # interior: (0.52201*2 + 0.79451) * (I[x+1]-I[x-1]) = 1.83853 * 4 -> truncated 7; kernel is flipped
    assert np.all(dx[1:-1, 1:-1] == 7)
    assert np.all(dy[1:-1, 1:-1] == 0)


def test_rgb_residual_identity(orc):
    w, h = 160, 120
    K = synth.intrinsics(w, h)
    f = synth.render(np.eye(4), w, h, noise=False, dropout=0)
    img = orc.image_to_intensity(f["rgb"])
    depth = orc.vertices_to_depth(f["vertex"], 6.0)
    dIdx, dIdy = orc.derivative_images(img)
    corres, sigma, count, _ = orc.rgb_residual(25.0, dIdx, dIdy, depth, depth, img, img, 0.07, np.zeros(3), np.eye(3))
    assert sigma == 0 and count > 0  # same image, identity warp: every accepted pixel has diff 0
    rec = corres.view(np.int16).reshape(h, w, 8)
    valid = corres[..., 12] == 1
    assert valid.sum() == count
    ys, xs = np.nonzero(valid)
    assert np.all(rec[ys, xs, 0] == xs) and np.all(rec[ys, xs, 1] == ys)  # zero == one == own pixel
    assert not valid[:, w - 5:].any() and not valid[h - 1, :].any()  # reduce.cu:773


def test_host_algebra(orc):
    import ctypes as C
    lib = orc.lib()
    rng = np.random.RandomState(0)
    M = rng.randn(6, 6)
    A = M @ M.T + 6 * np.eye(6)
    b = rng.randn(6)
    x = np.zeros(6)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    assert lib.orc_ldlt_solve(6, dp(np.ascontiguousarray(A)), dp(b), dp(x)) == 0
    assert np.allclose(A @ x, b, atol=1e-10)
    r = np.array([0.3, -0.2, 0.5])
    R = np.zeros(9)
    lib.orc_rodrigues(dp(r), dp(R))
    assert np.allclose(R.reshape(3, 3), synth.rodrigues(r), atol=1e-12)
    T = synth.make_pose((0.1, 0.2, -0.3), (1, 2, 3)).reshape(16).copy()
    Ti = np.zeros(16)
    lib.orc_inverse4d(dp(T), dp(Ti))
    assert np.allclose(Ti.reshape(4, 4) @ T.reshape(4, 4), np.eye(4), atol=1e-12)


@pytest.mark.parametrize("mode", [dict(rgbOnly=False, icpWeight=100.0, pyramid=True, fastOdom=False, so3=False),
                                  dict(rgbOnly=False, icpWeight=10.0, pyramid=True, fastOdom=False, so3=True)])
def test_oracle_odometry_recovers_known_motion(orc, mode):
    """Config 1 of BASELINE.json: one frame pair through the CPU path (plumbing, no GPU)."""
    w, h = 320, 240
    K, prev, cur, fp, fc = frame_pair(w, h)
    o = orc.Odometry(w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    o.initFirstRGB(fp["rgb"])
    o.initICPModel(fp["vertex"], fp["normal"], prev.astype(np.float32))
    o.initRGBModel(fp["rgb"])
    o.initICP(fc["depth"], 15.0)
    o.initRGB(fc["rgb"])
    t, R = o.getIncrementalTransformation(prev[:3, 3], prev[:3, :3], **mode)
    e0 = np.linalg.norm(prev[:3, 3] - cur[:3, 3])
    assert np.linalg.norm(t - cur[:3, 3]) < 0.35 * e0
    assert synth.rotation_angle(R.astype(np.float64), cur[:3, :3]) < 0.35 * synth.rotation_angle(prev[:3, :3],
                                                                                                 cur[:3, :3])
    s = o.stats()
    assert s.iterations_run == 19 and s.lastICPCount > 0.5 * w * h


def test_golden_fixtures(orc):
    """tests/golden/*.npz were produced by tests/golden/make_golden.py from THIS oracle; they guard
    the oracle (and therefore the parity bar) against silent drift."""
    path = os.path.join(GOLDEN, "icp_pair_64x48.npz")
    g = np.load(path)
    out, err = orc.icp_step(g["Rcurr"], g["tcurr"], g["vmap_curr"], g["nmap_curr"], g["Rprev_inv"], g["tprev"],
                            *g["intr"], g["vmap_g_prev"], g["nmap_g_prev"], DIST_THRESH, ANGLE_THRESH, want_err=True)
    assert np.array_equal(out, g["out29"])
    assert np.array_equal(err.view(np.uint32), g["err_map"].view(np.uint32))
    g = np.load(os.path.join(GOLDEN, "odometry_160x120.npz"))
    w, h = 160, 120
    K = synth.intrinsics(w, h)
    o = orc.Odometry(w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    o.initFirstRGB(g["rgb_prev"])
    o.initICPModel(g["vertex_prev"], g["normal_prev"], g["pose_prev"])
    o.initRGBModel(g["rgb_prev"])
    o.initICP(g["depth_cur"], 15.0)
    o.initRGB(g["rgb_cur"])
    t, R = o.getIncrementalTransformation(g["pose_prev"][:3, 3], g["pose_prev"][:3, :3], False, 10.0, True, False,
                                          True)
    assert np.allclose(t, g["trans"], atol=1e-7) and np.allclose(R, g["rot"], atol=1e-7)


def test_a_valid_depth_lies_inside_the_scaled_box_of_the_coarsest_level(orc):
    """csrc/extent.hpp (extent_of_level): an object model's chain skips the blocks outside a box that is noted at the COARSEST
    level of the model's depth pyramid and scaled up for the finer ones, plus the valid pixels of the finer levels' last column
    and row.  That is exact only if every valid depth of a finer level lies inside: a number makes its parent a number (the
    parent's 5x5 window holds it, cudafuncs.cu:333-364) -- except in the last column / row, which the window's quirk leaves out.
    Checked here on the oracle's pyramid (the kernels' pyramid is bit-identical to it: test_gpu_kernels.py) for sparse random
    masks, blobs at the borders and single pixels in the corners."""
    rng = np.random.default_rng(11)
    w, h = 64, 48

    def box(mask, shift):  # bounding box of a mask, scaled up by `shift` levels; None if empty
        ys, xs = np.nonzero(mask)
        if xs.size == 0:
            return None
        return (int(xs.min()) << shift, int(ys.min()) << shift, ((int(xs.max()) + 1) << shift) - 1, ((int(ys.max()) + 1) << shift) - 1)

    def hull(a, b):
        if a is None:
            return b
        if b is None:
            return a
        return (min(a[0], b[0]), min(a[1], b[1]), max(a[2], b[2]), max(a[3], b[3]))

    def border(mask):
        m = np.zeros_like(mask)
        m[:, -1] = mask[:, -1]
        m[-1, :] = mask[-1, :]
        return m

    cases = []
    for _ in range(40):
        d = np.full((h, w), np.nan, np.float32)
        for _ in range(int(rng.integers(1, 4))):  # a few blobs, some hanging over the right / bottom border
            cx, cy, r = int(rng.integers(0, w + 4)), int(rng.integers(0, h + 4)), int(rng.integers(1, 9))
            d[max(cy - r, 0):cy + r, max(cx - r, 0):cx + r] = rng.uniform(0.5, 3.0)
        cases.append(d)
    for x, y in ((w - 1, h - 1), (w - 1, 0), (0, h - 1), (w - 1, h // 2), (w - 2, h - 2), (w - 3, 5)):  # single pixels
        d = np.full((h, w), np.nan, np.float32)
        d[y, x] = 1.0
        cases.append(d)
    sparse = np.where(rng.random((h, w)) < 0.01, np.float32(2.0), np.float32(np.nan)).astype(np.float32)
    cases.append(sparse)
    for d0 in cases:
        d1 = orc.pyrdown_gauss_f(d0)
        d2 = orc.pyrdown_gauss_f(d1)
        v0, v1, v2 = ~np.isnan(d0), ~np.isnan(d1), ~np.isnan(d2)
        for level, valid in ((2, v2), (1, v1), (0, v0)):
            e = box(v2, 2 - level)
            if level <= 1:
                e = hull(e, box(border(v1), 1 - level))
            if level == 0:
                e = hull(e, box(border(v0), 0))
            ys, xs = np.nonzero(valid)
            if xs.size == 0:
                continue
            assert e is not None
            assert xs.min() >= e[0] and ys.min() >= e[1] and xs.max() <= e[2] and ys.max() <= e[3], (level, e, xs.min(), ys.min(), xs.max(), ys.max())
