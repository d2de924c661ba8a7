"""-m gpu: the MFMA descriptor matcher (mmf_match_descriptors) against the oracle's restatement of
cv::BFMatcher(NORM_L2, crossCheck) + the PointTracker distance gate.  The f32 matrix cores accumulate an
fmaf chain in k order, which is how the oracle defines its dot products, so indices AND distances must be
bit-exact -- ties, ragged sizes and empty sets included."""
import numpy as np
import pytest
import torch

from helpers import assert_bit_equal

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def unit_rows(rng, n, dim):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)


def run_both(gpu_ctx, orc, q, t, gate):
    from multimotionfusion_amd.matcher import matchDescriptors
    gi, gd = matchDescriptors(gpu_ctx, dev(q), dev(t), gate)
    oi, od = orc.match_descriptors(q, t, gate)
    assert_bit_equal(gi.cpu().numpy(), oi, "train indices")
    assert_bit_equal(gd.cpu().numpy(), od, "distances")
    return oi, od


@pytest.mark.parametrize("nq,nt,dim", [(300, 280, 256), (33, 65, 256), (1, 1, 8), (64, 31, 64), (1000, 1024, 256)])
def test_matches_tracked_keypoints(gpu_ctx, orc, nq, nt, dim):
    """`train` = descriptors of the previous frame; most queries are noisy copies of distinct train rows, the
    rest are new keypoints."""
    rng = np.random.default_rng(nq * 1000 + nt)
    t = unit_rows(rng, nt, dim)
    ncopy = min(nq, nt) * 3 // 4
    perm = rng.permutation(nt)[:ncopy]
    q = np.concatenate([t[perm] + 0.02 * rng.standard_normal((ncopy, dim)).astype(np.float32),
                        unit_rows(rng, nq - ncopy, dim)])
    idx, dist = run_both(gpu_ctx, orc, q, t, 0.7)
    assert (idx[:ncopy] == perm).all() and (dist[:ncopy] < 0.7).all()
    idx0, _ = run_both(gpu_ctx, orc, q, t, 0.0)  # gate off: every mutual nearest neighbour is kept
    assert (idx0 >= 0).sum() >= (idx >= 0).sum()


def test_ties_and_duplicates(gpu_ctx, orc):
    """Exact duplicates give equal distances: the first minimum must win on both axes, and crossCheck must
    drop the later duplicates."""
    rng = np.random.default_rng(7)
    base = unit_rows(rng, 40, 256)
    t = np.concatenate([base, base[:10]])          # train rows 40..49 duplicate rows 0..9
    q = np.concatenate([base[:20], base[:5]])      # query rows 20..24 duplicate query rows 0..4
    idx, dist = run_both(gpu_ctx, orc, q, t, 0.0)
    assert (idx[:20] == np.arange(20)).all() and (idx[20:] == -1).all() and (dist[:20] == 0).all()


def test_empty_sets(gpu_ctx, orc):
    q = unit_rows(np.random.default_rng(1), 5, 256)
    idx, _ = run_both(gpu_ctx, orc, q, np.zeros((0, 256), np.float32), 0.7)
    assert (idx == -1).all()
    run_both(gpu_ctx, orc, np.zeros((0, 256), np.float32), q, 0.7)
