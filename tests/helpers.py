"""Shared helpers of the parity tests."""
import numpy as np

from multimotionfusion_amd import synth

ANGLE_THRESH = float(np.sin(20.0 * 3.14159254 / 180.0))  # RGBDOdometry.h:36
DIST_THRESH = 0.10  # RGBDOdometry.h:35


def frame_pair(width, height, seed=1, **kw):
    """(K, prev_pose, cur_pose, prev_frame, cur_frame) of the synthetic sequence."""
    K = synth.intrinsics(width, height)
    poses = synth.trajectory(3, seed=seed)
    prev, cur = poses[0 + (seed % 2)], poses[1 + (seed % 2)]
    return K, prev, cur, synth.render(prev, width, height, seed=seed, **kw), synth.render(cur, width, height,
                                                                                         seed=seed + 1, **kw)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view({8: np.uint64, 4: np.uint32, 2: np.uint16, 1: np.uint8}[a.dtype.itemsize])


def assert_bit_equal(a, b, what=""):
    """Bit-for-bit equality (NaN payloads included)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    ne = bits(a) != bits(b)
    assert not ne.any(), f"{what}: {int(ne.sum())} of {ne.size} elements differ, first at {np.argwhere(ne)[:4].tolist()}"


def se3_sum_tolerance(out29, rel=2e-5):
    """Bound for |float32 grid sum - double sum| of each of the 29 sums: rel * sqrt(S_ii S_jj)
    (Cauchy-Schwarz bound of the L1 norm of the summed products) + tiny absolute."""
    out29 = np.asarray(out29, np.float64)
    diag = np.zeros(7)
    k = 0
    idx = []
    for i in range(6):
        for j in range(i, 7):
            if i == j:
                diag[i] = out29[k]
            idx.append((i, j))
            k += 1
    diag[6] = out29[27]
    tol = np.zeros(29)
    for k, (i, j) in enumerate(idx):
        tol[k] = rel * np.sqrt(abs(diag[i] * diag[j])) + 1e-12
    tol[27] = rel * abs(out29[27]) + 1e-12
    tol[28] = 0.0  # inlier count: exact
    return tol


def OracleFusion(orc, w, h, K, **kw):
    """The processFrame checker: oracle/fusion.py (cited restatement of MultiMotionFusion.cpp:207-854)."""
    from oracle.fusion import OracleFusion as _OracleFusion
    return _OracleFusion(w, h, K, **kw)


def slic_like_labels(W, H, S, seed=0, empty_every=0):
    """A gSLICr-like segmentation mask: grid cells of size S with wobbling borders; `empty_every` > 0 hands
    every k-th super-pixel's pixels to its left/upper neighbour, so that label does not occur at all."""
    rng = np.random.default_rng(seed)
    spx, spy = W // S, H // S
    yy, xx = np.mgrid[0:H, 0:W]
    jx = (3.0 * np.sin(yy / 7.0 + seed) + rng.integers(-2, 3, (H, W))).astype(np.int64)
    jy = (3.0 * np.cos(xx / 5.0 + seed) + rng.integers(-2, 3, (H, W))).astype(np.int64)
    cx = np.clip((xx + jx) // S, 0, spx - 1)
    cy = np.clip((yy + jy) // S, 0, spy - 1)
    labels = (cy * spx + cx).astype(np.int32)
    if empty_every:
        for s in range(empty_every, spx * spy, empty_every):
            labels[labels == s] = s - 1 if s % spx else max(s - spx, 0)
    return labels
