"""CPU: known-answer tests of the descriptor-matcher oracle (oracle/mmf_oracle_match.c), the restatement of
cv::BFMatcher(cv::NORM_L2, crossCheck=True).match + the PointTracker distance gate
(Core/Utils/PointTracker.cpp:100-114)."""
import numpy as np


def unit_rows(rng, n, dim):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)


def brute_force(q, t, gate):
    """The published algorithm in float64 numpy: mutual nearest neighbours under L2, first minimum on ties."""
    d = np.sqrt(np.maximum(((q[:, None, :].astype(np.float64) - t[None, :, :].astype(np.float64)) ** 2).sum(-1), 0))
    idx = np.full(len(q), -1, np.int32)
    if d.size:
        fwd, bwd = d.argmin(1), d.argmin(0)
        for i, j in enumerate(fwd):
            if bwd[j] == i and (gate < np.finfo(np.float32).eps or d[i, j] <= gate):
                idx[i] = j
    return idx, d


def test_matches_agree_with_float64_brute_force(orc):
    rng = np.random.default_rng(3)
    t = unit_rows(rng, 120, 256)
    perm = rng.permutation(120)[:80]
    q = np.concatenate([t[perm] + 0.03 * rng.standard_normal((80, 256)).astype(np.float32), unit_rows(rng, 30, 256)])
    for gate in (0.7, 0.0, 0.2):
        idx, dist = orc.match_descriptors(q, t, gate)
        ref, d = brute_force(q, t, gate)
        assert np.array_equal(idx, ref), gate
        m = idx >= 0
        assert np.allclose(dist[m], d[np.nonzero(m)[0], idx[m]], atol=2e-6) and (dist[~m] == 0).all()
    idx, _ = orc.match_descriptors(q, t, 0.7)
    assert (idx[:80] == perm).all()  # the tracked keypoints are found again


def test_cross_check_and_first_minimum(orc):
    # query 0 and query 1 are both nearest to train 0; train 0 prefers query 1: only (1, 0) survives
    t = np.zeros((2, 8), np.float32)
    t[0, 0], t[1, 1] = 1.0, 1.0
    q = np.zeros((3, 8), np.float32)
    q[0, 0], q[0, 2] = 1.0, 0.5
    q[1, 0], q[1, 2] = 1.0, 0.1
    q[2, 1] = 1.0
    idx, dist = orc.match_descriptors(q, t, 0.0)
    assert idx.tolist() == [-1, 0, 1] and abs(dist[1] - 0.1) < 1e-6 and dist[2] == 0
    # exact duplicates: the first one wins on either axis
    idx, _ = orc.match_descriptors(np.stack([q[2], q[2]]), np.stack([t[1], t[1]]), 0.0)
    assert idx.tolist() == [0, -1]
    # the gate rejects distant mutual neighbours; empty sets match nothing
    far = np.zeros((1, 8), np.float32)
    far[0, 3] = 5.0
    assert orc.match_descriptors(far, t, 0.7)[0].tolist() == [-1]
    assert orc.match_descriptors(far, t, 0.0)[0].tolist() == [0]
    assert orc.match_descriptors(far, np.zeros((0, 8), np.float32), 0.0)[0].tolist() == [-1]
    assert orc.match_descriptors(np.zeros((0, 8), np.float32), t, 0.0)[0].shape == (0,)
