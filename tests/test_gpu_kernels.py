"""-m gpu: every gfx950 kernel behind the C ABI against the CPU oracle on the same inputs.

Per-pixel outputs (maps, pyramids, correspondence records, error maps) and integer reductions
must be BIT-EXACT; float32 reductions are compared with the double-accumulating oracle within
2e-5 * sqrt(S_ii S_jj) (summation-order tolerance, helpers.se3_sum_tolerance).
"""
import numpy as np
import pytest
import torch

from helpers import ANGLE_THRESH, DIST_THRESH, assert_bit_equal, frame_pair, se3_sum_tolerance

pytestmark = pytest.mark.gpu

SIZES = [(640, 480), (160, 120), (100, 52), (106, 50)]  # (100, 52): not a multiple of the 64x4 tile; 106: no multiple of four (ragged pixel groups, byte-wise loads)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def maps_for(orc, K, frame, pose, cutoff=15.0):
    vm = orc.create_vmap(frame["depth"], K["fx"], K["fy"], K["cx"], K["cy"], cutoff)
    nm = orc.create_nmap(vm)
    vg, ng = orc.copy_maps(frame["vertex"], frame["normal"])
    vg, ng = orc.transform_maps(vg, ng, pose[:3, :3], pose[:3, 3])
    return vm, nm, vg, ng


@pytest.mark.parametrize("w,h", SIZES)
def test_map_kernels_bit_exact(gpu_ctx, orc, w, h):
    from multimotionfusion_amd import cudafuncs as cf
    K, prev, cur, fp, fc = frame_pair(w, h)
    intr = cf.CameraModel(K["fx"], K["fy"], K["cx"], K["cy"])
    depth = fc["depth"].copy()
    depth[3, 5] = np.nan  # NaN depth must behave like the reference's `z != 0 && z < cutoff`
    d = dev(depth)
    vmap = torch.zeros(3 * h, w, device="cuda")
    cf.createVMap(gpu_ctx, intr, d, vmap, 3.0)
    ref_v = orc.create_vmap(depth, K["fx"], K["fy"], K["cx"], K["cy"], 3.0)
    gv = vmap.cpu().numpy()
    valid = ~np.isnan(ref_v[:h])
    assert_bit_equal(np.isnan(gv[:h]), np.isnan(ref_v[:h]), "vmap validity")
    for p in range(3):  # y/z planes are only defined where the x plane is valid
        assert_bit_equal(gv[p * h:(p + 1) * h][valid], ref_v[p * h:(p + 1) * h][valid], f"vmap plane {p}")

    vfull = dev(ref_v)
    nmap = torch.zeros(3 * h, w, device="cuda")
    cf.createNMap(gpu_ctx, vfull, nmap)
    ref_n = orc.create_nmap(ref_v)
    gn = nmap.cpu().numpy()
    nvalid = ~np.isnan(ref_n[:h])
    assert_bit_equal(np.isnan(gn[:h]), np.isnan(ref_n[:h]), "nmap validity")
    for p in range(3):
        assert_bit_equal(gn[p * h:(p + 1) * h][nvalid], ref_n[p * h:(p + 1) * h][nvalid], f"nmap plane {p}")

    # copyMaps + tranformMaps (in place) + resize
    vr, nr = dev(fp["vertex"]), dev(fp["normal"])
    vd, nd = torch.zeros(3 * h, w, device="cuda"), torch.zeros(3 * h, w, device="cuda")
    cf.copyMaps(gpu_ctx, vr, nr, vd, nd)
    rv, rn = orc.copy_maps(fp["vertex"], fp["normal"])
    assert_bit_equal(vd.cpu().numpy(), rv, "copyMaps v")
    assert_bit_equal(nd.cpu().numpy(), rn, "copyMaps n")
    pose = cur.astype(np.float32)
    cf.tranformMaps(gpu_ctx, vd, nd, pose[:3, :3], pose[:3, 3], vd, nd)
    tv, tn = orc.transform_maps(rv, rn, pose[:3, :3], pose[:3, 3])
    assert_bit_equal(vd.cpu().numpy(), tv, "tranformMaps v")
    assert_bit_equal(nd.cpu().numpy(), tn, "tranformMaps n")
    if w % 2 == 0 and h % 2 == 0:
        ov = torch.zeros(3 * (h // 2), w // 2, device="cuda")
        on = torch.zeros(3 * (h // 2), w // 2, device="cuda")
        cf.resizeVMap(gpu_ctx, vd, ov)
        cf.resizeNMap(gpu_ctx, nd, on)
        assert_bit_equal(ov.cpu().numpy(), orc.resize_map(tv, False), "resizeVMap")
        assert_bit_equal(on.cpu().numpy(), orc.resize_map(tn, True), "resizeNMap")


@pytest.mark.parametrize("w,h", SIZES)
def test_pyramid_kernels_bit_exact(gpu_ctx, orc, w, h):
    from multimotionfusion_amd import cudafuncs as cf
    K, prev, cur, fp, fc = frame_pair(w, h)
    depth = fc["depth"].copy()
    depth[depth == 0] = np.nan  # pyrDownGaussF skips NaN
    out = torch.zeros(h // 2, w // 2, device="cuda")
    cf.pyrDownGaussF(gpu_ctx, dev(depth), out)
    assert_bit_equal(out.cpu().numpy(), orc.pyrdown_gauss_f(depth), "pyrDownGaussF")

    inten = torch.zeros(h, w, dtype=torch.uint8, device="cuda")
    cf.imageBGRToIntensity(gpu_ctx, dev(fc["rgb"]), inten)
    ref_i = orc.image_to_intensity(fc["rgb"])
    assert_bit_equal(inten.cpu().numpy(), ref_i, "imageBGRToIntensity (3 ch)")
    rgba = np.concatenate([fc["rgb"], np.full((h, w, 1), 255, np.uint8)], -1)
    cf.imageBGRToIntensity(gpu_ctx, dev(rgba), inten)
    assert_bit_equal(inten.cpu().numpy(), ref_i, "imageBGRToIntensity (4 ch)")

    ref_i2 = ref_i.copy()
    ref_i2[: h // 3, : w // 3] = 0  # zeros are skipped; an all-zero window must give 0
    o8 = torch.zeros(h // 2, w // 2, dtype=torch.uint8, device="cuda")
    cf.pyrDownUcharGauss(gpu_ctx, dev(ref_i2), o8)
    assert_bit_equal(o8.cpu().numpy(), orc.pyrdown_uchar_gauss(ref_i2), "pyrDownUcharGauss")

    dx = torch.zeros(h, w, dtype=torch.int16, device="cuda")
    dy = torch.zeros(h, w, dtype=torch.int16, device="cuda")
    cf.computeDerivativeImages(gpu_ctx, dev(ref_i), dx, dy)
    rdx, rdy = orc.derivative_images(ref_i)
    assert_bit_equal(dx.cpu().numpy(), rdx, "dIdx")
    assert_bit_equal(dy.cpu().numpy(), rdy, "dIdy")

    vd = torch.zeros(h, w, device="cuda")
    cf.verticesToDepth(gpu_ctx, dev(fp["vertex"]), vd, 3.0)
    ref_d = orc.vertices_to_depth(fp["vertex"], 3.0)
    assert_bit_equal(vd.cpu().numpy(), ref_d, "verticesToDepth")

    cloud = torch.zeros(h, w, 3, device="cuda")
    intr = cf.CameraModel(K["fx"], K["fy"], K["cx"], K["cy"])
    cf.projectToPointCloud(gpu_ctx, dev(ref_d), cloud, intr, 0)
    assert_bit_equal(cloud.cpu().numpy(), orc.project_to_cloud(ref_d, K["fx"], K["fy"], K["cx"], K["cy"]),
                     "projectToPointCloud")


@pytest.mark.parametrize("w,h", SIZES)
def test_icp_step_parity(gpu_ctx, orc, w, h):
    from multimotionfusion_amd import cudafuncs as cf
    K, prev, cur, fp, fc = frame_pair(w, h)
    vm, nm, vg, ng = maps_for(orc, K, fc, prev.astype(np.float32))
    intr = cf.CameraModel(K["fx"], K["fy"], K["cx"], K["cy"])
    Rprev = prev[:3, :3].astype(np.float32)
    Rprev_inv = np.linalg.inv(Rprev).astype(np.float32)
    tprev = prev[:3, 3].astype(np.float32)
    err = torch.full((h, w), -1.0, device="cuda")
    A, b, res = cf.icpStep(gpu_ctx, Rprev, tprev, dev(vm), dev(nm), Rprev_inv, tprev, intr, dev(vg), dev(ng),
                           DIST_THRESH, ANGLE_THRESH, err)
    out, ref_err = orc.icp_step(Rprev, tprev, vm, nm, Rprev_inv, tprev, K["fx"], K["fy"], K["cx"], K["cy"], vg, ng,
                                DIST_THRESH, ANGLE_THRESH, want_err=True)
    assert res[1] == out[28] and out[28] > 0.2 * w * h, (res, out[28])  # inlier count exact
    assert_bit_equal(err.cpu().numpy(), ref_err, "icp error map")
    # pack the GPU result like the oracle's out29 to compare sum by sum
    got = np.zeros(29)
    k = 0
    for i in range(6):
        for j in range(i, 7):
            got[k] = b[i] if j == 6 else A[i, j]
            k += 1
    got[27], got[28] = res
    tol = se3_sum_tolerance(out)
    assert np.all(np.abs(got - out) <= tol), np.abs(got - out) / np.maximum(tol, 1e-30)
    assert np.array_equal(A, A.T)
    # run-to-run determinism: same launch geometry -> identical bits
    A2, b2, res2 = cf.icpStep(gpu_ctx, Rprev, tprev, dev(vm), dev(nm), Rprev_inv, tprev, intr, dev(vg), dev(ng),
                              DIST_THRESH, ANGLE_THRESH)
    assert_bit_equal(A2, A, "icp A determinism")
    assert_bit_equal(b2, b, "icp b determinism")


def test_icp_step_image_larger_than_one_pass(gpu_ctx, orc):
    """More pixels than the ICP producer's largest grid takes in one pass (8192 workgroups x 256 lanes): the multi-pass
    instantiation walks the image with the grid's stride."""
    from multimotionfusion_amd import cudafuncs as cf
    w, h = 2048, 1040
    K, prev, cur, fp, fc = frame_pair(w, h)
    vm, nm, vg, ng = maps_for(orc, K, fc, prev.astype(np.float32))
    intr = cf.CameraModel(K["fx"], K["fy"], K["cx"], K["cy"])
    Rprev = prev[:3, :3].astype(np.float32)
    Rprev_inv = np.linalg.inv(Rprev).astype(np.float32)
    tprev = prev[:3, 3].astype(np.float32)
    err = torch.full((h, w), -1.0, device="cuda")
    A, b, res = cf.icpStep(gpu_ctx, Rprev, tprev, dev(vm), dev(nm), Rprev_inv, tprev, intr, dev(vg), dev(ng), DIST_THRESH, ANGLE_THRESH, err)
    out, ref_err = orc.icp_step(Rprev, tprev, vm, nm, Rprev_inv, tprev, K["fx"], K["fy"], K["cx"], K["cy"], vg, ng, DIST_THRESH,
                                ANGLE_THRESH, want_err=True)
    assert res[1] == out[28] and out[28] > 0.2 * w * h
    assert_bit_equal(err.cpu().numpy(), ref_err, "icp error map")
    got = np.zeros(29)
    k = 0
    for i in range(6):
        for j in range(i, 7):
            got[k] = b[i] if j == 6 else A[i, j]
            k += 1
    got[27], got[28] = res
    tol = se3_sum_tolerance(out)
    assert np.all(np.abs(got - out) <= tol), np.abs(got - out) / np.maximum(tol, 1e-30)


def test_icp_step_edge_cases(gpu_ctx, orc):
    from multimotionfusion_amd import cudafuncs as cf
    w, h = 160, 120
    K, prev, cur, fp, fc = frame_pair(w, h)
    vm, nm, vg, ng = maps_for(orc, K, fc, prev.astype(np.float32))
    intr = cf.CameraModel(K["fx"], K["fy"], K["cx"], K["cy"])
    I, z = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    # all-invalid current frame -> zero system, zero inliers
    nanmap = np.full_like(vm, np.nan)
    A, b, res = cf.icpStep(gpu_ctx, I, z, dev(nanmap), dev(nanmap), I, z, intr, dev(vg), dev(ng), DIST_THRESH,
                           ANGLE_THRESH)
    assert not A.any() and not b.any() and res[0] == 0 and res[1] == 0
    # a pose that throws every point behind / outside the camera -> no correspondences
    far = np.array([0, 0, -100.0], np.float32)
    A, b, res = cf.icpStep(gpu_ctx, I, far, dev(vm), dev(nm), I, z, intr, dev(vg), dev(ng), DIST_THRESH, ANGLE_THRESH)
    out, _ = orc.icp_step(I, far, vm, nm, I, z, K["fx"], K["fy"], K["cx"], K["cy"], vg, ng, DIST_THRESH,
                          ANGLE_THRESH)
    assert res[1] == out[28] == 0
    # pitched (non-dense) maps give the same bits as dense ones
    pad = 32
    def pitched(a):
        t = torch.zeros(a.shape[0], a.shape[1] + pad, device="cuda")
        t[:, : a.shape[1]] = dev(a)
        return t[:, : a.shape[1]]
    Rp = prev[:3, :3].astype(np.float32)
    tp = prev[:3, 3].astype(np.float32)
    A1, b1, r1 = cf.icpStep(gpu_ctx, Rp, tp, dev(vm), dev(nm), Rp.T.copy(), tp, intr, dev(vg), dev(ng), DIST_THRESH,
                            ANGLE_THRESH)
    A2, b2, r2 = cf.icpStep(gpu_ctx, Rp, tp, pitched(vm), pitched(nm), Rp.T.copy(), tp, intr, pitched(vg),
                            pitched(ng), DIST_THRESH, ANGLE_THRESH)
    assert_bit_equal(A1, A2, "pitched A")
    assert_bit_equal(b1, b2, "pitched b")


@pytest.mark.parametrize("w,h", SIZES)
def test_rgb_residual_and_step_parity(gpu_ctx, orc, w, h):
    from multimotionfusion_amd import cudafuncs as cf
    K, prev, cur, fp, fc = frame_pair(w, h)
    last_i, next_i = orc.image_to_intensity(fp["rgb"]), orc.image_to_intensity(fc["rgb"])
    last_d = orc.vertices_to_depth(fp["vertex"], 6.0)
    next_d = orc.vertices_to_depth(fc["vertex"], 6.0)
    next_d[5:9, 7:30] = np.nan
    last_i[20:24, 40:44] = 0
    dIdx, dIdy = orc.derivative_images(next_i)
    Km = np.array([[K["fx"], 0, K["cx"]], [0, K["fy"], K["cy"]], [0, 0, 1.0]])
    T = np.linalg.inv(np.linalg.inv(prev) @ cur)  # resultRt.inverse() stand-in
    krkinv = (Km @ T[:3, :3] @ np.linalg.inv(Km)).astype(np.float32)
    kt = (Km @ T[:3, 3]).astype(np.float32)
    corres = torch.zeros(h, w, 16, dtype=torch.uint8, device="cuda")
    err = torch.full((h, w), -1.0, device="cuda")
    sigma, count = cf.computeRgbResidual(gpu_ctx, 400.0, dev(dIdx), dev(dIdy), dev(last_d), dev(next_d),
                                         dev(last_i), dev(next_i), corres, 0.07, kt, krkinv, err)
    rc, rsigma, rcount, rerr = orc.rgb_residual(400.0, dIdx, dIdy, last_d, next_d, last_i, next_i, 0.07, kt, krkinv,
                                                want_err=True)
    assert (sigma, count) == (rsigma, rcount) and count > 0
    assert_bit_equal(corres.cpu().numpy(), rc, "DataTerm records")
    assert_bit_equal(err.cpu().numpy(), rerr, "rgb error map")

    cloud = orc.project_to_cloud(last_d, K["fx"], K["fy"], K["cx"], K["cy"])
    for sig in (float(count), 1.0, -1.0):
        A, b = cf.rgbStep(gpu_ctx, corres, sig, dev(cloud), K["fx"], K["fy"], dev(dIdx), dev(dIdy), 0.125)
        out = orc.rgb_step(rc, sig, cloud, K["fx"], K["fy"], dIdx, dIdy, 0.125)
        got = np.zeros(29)
        k = 0
        for i in range(6):
            for j in range(i, 7):
                got[k] = b[i] if j == 6 else A[i, j]
                k += 1
        tol = se3_sum_tolerance(out)
        assert np.all(np.abs(got[:27] - out[:27]) <= tol[:27]), (sig, np.abs(got - out)[:27] / tol[:27])


@pytest.mark.parametrize("w,h", [(160, 120), (100, 52)])
def test_so3_step_parity(gpu_ctx, orc, w, h):
    from multimotionfusion_amd import cudafuncs as cf
    from multimotionfusion_amd import synth
    K, prev, cur, fp, fc = frame_pair(w, h)
    last_i, next_i = orc.image_to_intensity(fp["rgb"]), orc.image_to_intensity(fc["rgb"])
    Km = np.array([[K["fx"], 0, K["cx"]], [0, K["fy"], K["cy"]], [0, 0, 1.0]])
    for rvec in ((0, 0, 0), (0.01, -0.02, 0.005)):
        R = synth.rodrigues(rvec)
        B = (Km @ R @ np.linalg.inv(Km)).astype(np.float32)
        kinv = np.linalg.inv(Km).astype(np.float32)
        krlr = (Km @ R).astype(np.float32)
        A, b, res = cf.so3Step(gpu_ctx, dev(last_i), dev(next_i), B, kinv, krlr)
        out = orc.so3_step(last_i, next_i, B, kinv, krlr)
        assert res[1] == out[10] and out[10] > 0
        got = np.array([A[0, 0], A[0, 1], A[0, 2], b[0], A[1, 1], A[1, 2], b[1], A[2, 2], b[2], res[0]])
        d = np.array([out[0], out[4], out[7], out[9]])  # diagonals aa bb cc rr
        ii = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (3, 3)]
        tol = np.array([2e-5 * np.sqrt(d[i] * d[j]) + 1e-9 for i, j in ii])
        assert np.all(np.abs(got - out[:10]) <= tol), np.abs(got - out[:10]) / tol
