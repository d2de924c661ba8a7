"""CPU: the geometric claim behind the object models' walk in the one-launch chain (csrc/gn_fused.hpp, gn_sparse_icp_box).

icpStep (reduce.cu:257-299) moves a sensor pixel's point v into the model's camera, q = Rprev^-1 (Rcurr v + tcurr - tprev),
projects it to a pixel (u, v) of the model's maps (rounded to the nearest) and accepts the correspondence only if the vertex
stored there is a number and lies within distThresh of the point.  With B the pixel box of the model's valid vertices and
[zlo, zhi] their depth range, an accepted correspondence therefore has (u, v) in B and q.z in [zlo - d, zhi + d].  The kernel
walks only the bounding rectangle of that slab's eight corners taken into the sensor camera (+ margins) -- here the same
construction in numpy (float32, the kernel's formulas) against brute force: every sensor pixel with ANY depth whose point
satisfies the two necessary conditions must lie inside the rectangle, for random boxes, depth ranges and relative poses up
to several degrees and centimetres, at all three pyramid levels; and the error-image launch's rectangle (the cone over B cut by
the sensor's depth range) must hold every pixel whose point merely PROJECTS into B.
"""
import numpy as np

from multimotionfusion_amd import synth

F = np.float32


def rectangle(Rprev, tprev, Rcurr, tcurr, intr, cols, rows, lo, hi, level, dist_thres, err=False, zs=(0.0, 0.0)):
    """gn_sparse_icp_box: lo / hi = pixel x, pixel y (level 0), camera-frame z of the model's valid vertices.  Returns
    (x0, y0, x1, y1) inclusive, or None for "no pass"; the whole image when the construction does not apply."""
    fx, fy, cx, cy = (F(v) for v in intr)
    d = F(dist_thres) * F(1.01) + F(1e-3)
    us, vs, oks = [], [], []
    for c in range(8):
        bu = F((int(hi[0]) >> level) + 1) if c & 1 else F((int(lo[0]) >> level) - 1)
        bv = F((int(hi[1]) >> level) + 1) if c & 2 else F((int(lo[1]) >> level) - 1)
        e = np.array([(bu - cx) / fx, (bv - cy) / fy, F(1)], F)
        h = (Rprev.astype(F) @ e).astype(F)
        w = (Rcurr.astype(F).T @ h).astype(F)
        o = (Rcurr.astype(F).T @ (tprev.astype(F) - tcurr.astype(F))).astype(F)
        if not err:
            za, zb = F(lo[2]) - d, F(hi[2]) + d
            s = zb if c & 4 else za
            ok = za > F(0.05)
        else:
            zp = F(zs[1]) if c & 4 else F(zs[0]) * F(0.999) - F(1e-3)
            s = (zp - o[2]) / w[2] if w[2] != 0 else F(np.inf)
            ok = zs[0] > F(0.06) and w[2] > F(1e-3) and s > 0
        v = o + s * w
        ok = bool(ok and v[2] > F(0.05) and v[2] < F(1e6) and abs(v[0]) < F(1e6) and abs(v[1]) < F(1e6))
        us.append(float(v[0] * fx / v[2] + cx) if ok else 0.0)
        vs.append(float(v[1] * fy / v[2] + cy) if ok else 0.0)
        oks.append(ok)
    if not all(oks):
        return (0, 0, cols - 1, rows - 1)
    x0, x1 = max(0, int(np.floor(min(us))) - 2), min(cols - 1, int(np.ceil(max(us))) + 2)
    y0, y1 = max(0, int(np.floor(min(vs))) - 2), min(rows - 1, int(np.ceil(max(vs))) + 2)
    return None if x1 < x0 or y1 < y0 else (x0, y0, x1, y1)


def brute_force(Rprev, tprev, Rcurr, tcurr, intr, cols, rows, B, zrange, depths):
    """sensor pixels (x, y) that have, at one of `depths`, a point whose projection (rounded) falls into B with q.z in zrange"""
    fx, fy, cx, cy = intr
    ys, xs = np.mgrid[0:rows, 0:cols]
    hit = np.zeros((rows, cols), bool)
    Rpi = np.linalg.inv(Rprev.astype(np.float64))
    for z in depths:
        v = np.stack([(xs - cx) * z / fx, (ys - cy) * z / fy, np.full(xs.shape, z)], -1).astype(np.float64)
        q = (v @ Rcurr.astype(np.float64).T + tcurr - tprev) @ Rpi.T
        with np.errstate(divide="ignore", invalid="ignore"):
            u = np.rint(q[..., 0] * fx / q[..., 2] + cx)
            w = np.rint(q[..., 1] * fy / q[..., 2] + cy)
        hit |= (q[..., 2] > 0) & (u >= B[0]) & (u <= B[2]) & (w >= B[1]) & (w <= B[3]) & (q[..., 2] >= zrange[0]) & (q[..., 2] <= zrange[1])
    return hit


def test_the_rectangle_holds_every_pixel_that_can_be_accepted():
    rng = np.random.default_rng(5)
    K0 = synth.intrinsics(640, 480)
    n_checked, areas = 0, []
    for trial in range(60):
        level = trial % 3
        cols, rows = 640 >> level, 480 >> level
        s = 1 << level
        intr = (K0["fx"] / s, K0["fy"] / s, K0["cx"] / s, K0["cy"] / s)
        bw, bh = rng.integers(20, 260), rng.integers(20, 200)
        bx, by = rng.integers(0, 640 - bw), rng.integers(0, 480 - bh)
        zlo = float(rng.uniform(0.5, 3.0))
        zhi = zlo + float(rng.uniform(0.0, 1.5))
        lo, hi = (bx, by, zlo), (bx + bw, by + bh, zhi)
        Pprev = synth.make_pose(rng.normal(size=3) * 0.2, rng.normal(size=3) * 0.3)
        big = trial % 5 == 0  # now and then a motion far beyond what tracking meets
        dP = synth.make_pose(rng.normal(size=3) * (0.12 if big else 0.02), rng.normal(size=3) * (0.15 if big else 0.02))
        Pcurr = Pprev @ dP
        Rprev, tprev, Rcurr, tcurr = Pprev[:3, :3], Pprev[:3, 3], Pcurr[:3, :3], Pcurr[:3, 3]
        rect = rectangle(Rprev, tprev, Rcurr, tcurr, intr, cols, rows, lo, hi, level, 0.10)
        B = (bx >> level, by >> level, (bx + bw) >> level, (by + bh) >> level)
        depths = np.concatenate([np.linspace(0.2, 6.0, 30), [zlo - 0.1, zlo, zhi, zhi + 0.1]])
        hit = brute_force(Rprev, tprev, Rcurr, tcurr, intr, cols, rows, B, (zlo - 0.10, zhi + 0.10), depths)
        if rect is None:
            assert not hit.any(), trial
            continue
        inside = np.zeros_like(hit)
        inside[rect[1]:rect[3] + 1, rect[0]:rect[2] + 1] = True
        assert not (hit & ~inside).any(), (trial, rect, B, np.argwhere(hit & ~inside)[:3])
        n_checked += int(hit.sum())
        if not big and hit.any():
            areas.append(inside.sum() / max(1, hit.sum()))
    assert n_checked > 100000
    assert np.median(areas) < 2.5, np.median(areas)  # and it is a rectangle around the object, not the image


def test_the_error_image_rectangle_holds_every_pixel_that_projects_into_the_box():
    rng = np.random.default_rng(6)
    K0 = synth.intrinsics(640, 480)
    cols, rows, level = 640, 480, 0
    intr = (K0["fx"], K0["fy"], K0["cx"], K0["cy"])
    for trial in range(25):
        bw, bh = rng.integers(20, 260), rng.integers(20, 200)
        bx, by = rng.integers(0, 640 - bw), rng.integers(0, 480 - bh)
        lo, hi = (bx, by, 1.0), (bx + bw, by + bh, 2.0)
        Pprev = synth.make_pose(rng.normal(size=3) * 0.2, rng.normal(size=3) * 0.3)
        Pcurr = Pprev @ synth.make_pose(rng.normal(size=3) * 0.03, rng.normal(size=3) * 0.03)
        Rprev, tprev, Rcurr, tcurr = Pprev[:3, :3], Pprev[:3, 3], Pcurr[:3, :3], Pcurr[:3, 3]
        zs_min, cutoff = float(rng.uniform(0.4, 1.5)), 15.0
        rect = rectangle(Rprev, tprev, Rcurr, tcurr, intr, cols, rows, lo, hi, level, 0.10, err=True, zs=(zs_min, cutoff))
        depths = np.concatenate([np.geomspace(zs_min, cutoff * 0.999, 40)])
        hit = brute_force(Rprev, tprev, Rcurr, tcurr, intr, cols, rows, (bx, by, bx + bw, by + bh), (-np.inf, np.inf), depths)
        assert rect is not None
        inside = np.zeros_like(hit)
        inside[rect[1]:rect[3] + 1, rect[0]:rect[2] + 1] = True
        assert not (hit & ~inside).any(), (trial, rect, np.argwhere(hit & ~inside)[:3])
