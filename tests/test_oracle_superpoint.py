"""The SuperPoint oracle (oracle/mmf_oracle_superpoint.c) against torch.nn.functional in fp32 on the CPU.

The reference's SuperPoint lives in an un-vendored dependency (parity unpinned, see the oracle's header); what
can be pinned is that the restatement IS the published network: same layers, layouts and post-processing as a
plain PyTorch statement of it, within float32 summation-order noise (1e-4).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as orc

TOL = 1e-4  # float32, different summation order (ours: one fmaf chain; torch: blocked / vectorised)


def torch_forward(inp, weights):
    x = torch.from_numpy(inp)[None, None]
    t = [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in weights]
    conv = lambda x, i, pad: F.conv2d(x, t[i][0], t[i][1], padding=pad)
    x = F.relu(conv(x, 0, 1)); x = F.relu(conv(x, 1, 1)); x = F.max_pool2d(x, 2)
    x = F.relu(conv(x, 2, 1)); x = F.relu(conv(x, 3, 1)); x = F.max_pool2d(x, 2)
    x = F.relu(conv(x, 4, 1)); x = F.relu(conv(x, 5, 1)); x = F.max_pool2d(x, 2)
    x = F.relu(conv(x, 6, 1)); x = F.relu(conv(x, 7, 1))
    semi = conv(F.relu(conv(x, 8, 1)), 9, 0)
    desc = conv(F.relu(conv(x, 10, 1)), 11, 0)
    desc = desc / torch.norm(desc, p=2, dim=1, keepdim=True)
    return semi[0].permute(1, 2, 0).numpy(), desc[0].permute(1, 2, 0).numpy()


def test_conv_matches_torch():
    rng = np.random.default_rng(3)
    for cin, cout, k in ((1, 64, 3), (64, 64, 3), (128, 65, 1), (96, 40, 3)):
        x = rng.normal(0, 1, (24, 40, cin)).astype(np.float32)
        w = rng.normal(0, 0.1, (cout, cin, k, k)).astype(np.float32)
        b = rng.normal(0, 0.1, cout).astype(np.float32)
        got = orc.sp_conv(x, w, b, relu=True)
        ref = F.relu(F.conv2d(torch.from_numpy(x).permute(2, 0, 1)[None], torch.from_numpy(w), torch.from_numpy(b),
                              padding=k // 2))[0].permute(1, 2, 0).numpy()
        assert np.abs(got - ref).max() <= TOL * max(1.0, np.abs(ref).max())


def test_forward_matches_torch():
    rng = np.random.default_rng(5)
    weights = orc.sp_random_weights(seed=1)
    inp = rng.random((48, 64), dtype=np.float32)
    semi, desc = orc.sp_forward(inp, weights)
    semi_t, desc_t = torch_forward(inp, weights)
    assert semi.shape == (6, 8, 65) and desc.shape == (6, 8, 256)
    assert np.abs(semi - semi_t).max() <= TOL * max(1.0, np.abs(semi_t).max())
    assert np.abs(desc - desc_t).max() <= TOL
    assert np.allclose(np.linalg.norm(desc, axis=2), 1.0, atol=1e-5)


def test_forward_rejects_sizes_not_multiple_of_8():
    with pytest.raises(ValueError):
        orc.sp_forward(np.zeros((50, 64), np.float32), orc.sp_random_weights())


def test_heatmap_is_softmax_depth_to_space():
    rng = np.random.default_rng(7)
    semi = rng.normal(0, 2, (5, 7, 65)).astype(np.float32)
    heat = orc.sp_heatmap(semi)
    e = np.exp(semi.astype(np.float64))
    dense = e / (e.sum(axis=2, keepdims=True) + 1e-5)
    ref = dense[:, :, :64].reshape(5, 7, 8, 8).transpose(0, 2, 1, 3).reshape(40, 56)
    assert np.abs(heat - ref).max() <= 1e-6


def nms_fast_reference(heat, conf_thresh, dist, border):
    """a literal restatement of the demo's nms_fast (padded grid, strongest first) + border removal"""
    H, W = heat.shape
    ys, xs = np.where(heat >= conf_thresh)
    conf = heat[ys, xs]
    order = np.lexsort((ys * W + xs, -conf))  # strongest first, ties in row-major order
    grid = np.zeros((H + 2 * dist, W + 2 * dist), np.int8)
    grid[ys + dist, xs + dist] = 1
    keep = []
    for k in order:
        y, x = ys[k] + dist, xs[k] + dist
        if grid[y, x] == 1:
            grid[y - dist:y + dist + 1, x - dist:x + dist + 1] = 0
            grid[y, x] = -1
            keep.append(k)
    keep = [k for k in keep if border <= xs[k] < W - border and border <= ys[k] < H - border]
    return np.stack([xs[keep], ys[keep]], 1).astype(np.int32).reshape(-1, 2), conf[keep]


def test_keypoints_are_greedy_nms():
    rng = np.random.default_rng(11)
    heat = (rng.random((40, 56)) ** 8).astype(np.float32) * 0.3
    heat[10, 10] = heat[10, 13] = 0.5  # a tie inside one window: the row-major first one wins
    xy, conf = orc.sp_keypoints(heat, 0.015, 4, 4)
    xy_r, conf_r = nms_fast_reference(heat, 0.015, 4, 4)
    assert np.array_equal(xy, xy_r) and np.array_equal(conf, conf_r)
    assert np.all(np.diff(conf) <= 0)
    assert [10, 10] in xy.tolist() and [13, 10] not in xy.tolist()
    # every pair of survivors is farther apart than the suppression radius
    d = np.abs(xy[:, None, :] - xy[None, :, :]).max(axis=2) + np.eye(len(xy), dtype=int) * 99
    assert d.min() > 4


def test_keypoints_empty_and_border():
    heat = np.zeros((16, 24), np.float32)
    xy, conf = orc.sp_keypoints(heat, 0.015, 4, 4)
    assert xy.shape == (0, 2) and conf.shape == (0,)
    heat[2, 2] = 0.9   # inside the border band: suppresses its neighbourhood, then is dropped
    heat[5, 5] = 0.5   # within 4 px of it: suppressed
    heat[8, 12] = 0.4
    xy, conf = orc.sp_keypoints(heat, 0.015, 4, 4)
    assert xy.tolist() == [[12, 8]]


def test_sampled_descriptors_match_grid_sample():
    rng = np.random.default_rng(13)
    desc = rng.normal(0, 1, (6, 8, 256)).astype(np.float32)
    desc /= np.linalg.norm(desc, axis=2, keepdims=True)
    H, W = 48, 64
    xy = np.stack([rng.integers(0, W, 50), rng.integers(0, H, 50)], 1).astype(np.int32)
    got = orc.sp_sample_descriptors(desc, xy, H, W)
    grid = torch.from_numpy(np.stack([xy[:, 0] / (W / 2.0) - 1.0, xy[:, 1] / (H / 2.0) - 1.0], 1).astype(np.float32))
    ref = F.grid_sample(torch.from_numpy(desc).permute(2, 0, 1)[None], grid[None, None], mode="bilinear",
                        align_corners=True)[0, :, 0].T.numpy()
    ref = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    assert np.abs(got - ref).max() <= 1e-5


def test_get_features_contract():
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    coords, descs = orc.sp_get_features(img, orc.sp_random_weights(seed=2))
    assert coords.dtype == np.float64 and descs.dtype == np.float64
    assert coords.shape[1] == 2 and descs.shape == (coords.shape[0], 256)
    assert coords.shape[0] > 0
    assert np.all((coords >= 0) & (coords < 1))
    assert np.allclose(np.linalg.norm(descs, axis=1), 1.0, atol=1e-5)


def test_golden_fixture():
    """tests/golden/superpoint_slic.npz (make_golden.py section 4) pins the oracle's SuperPoint and super-pixel
    outputs bit for bit; the weights are regenerated from their seed and checked by hash."""
    import hashlib
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "superpoint_slic.npz"))
    weights = orc.sp_random_weights(seed=int(g["weights_seed"]))
    assert hashlib.sha256(b"".join(a.tobytes() for w, b in weights for a in (w, b))).hexdigest() == str(g["weights_sha256"])
    semi, desc = orc.sp_forward(orc.sp_input(g["image"]), weights)
    assert np.array_equal(semi.view(np.uint32), g["semi"].view(np.uint32))
    assert np.array_equal(desc.view(np.uint32), g["desc"].view(np.uint32))
    heat = orc.sp_heatmap(semi)
    assert np.array_equal(heat.view(np.uint32), g["heat"].view(np.uint32))
    xy, conf = orc.sp_keypoints(heat)
    assert np.array_equal(xy, g["xy"]) and np.array_equal(conf, g["conf"])
    assert np.array_equal(orc.sp_sample_descriptors(desc, xy, 48, 64).view(np.uint32), g["kdesc"].view(np.uint32))
    assert np.array_equal(orc.slic_downsample(g["labels"], 16, g["smap"]).view(np.uint32), g["low"].view(np.uint32))
    assert np.array_equal(orc.slic_downsample(g["labels"], 16, g["sdepth"], threshold=0.02).view(np.uint32),
                          g["low_depth"].view(np.uint32))
    assert np.array_equal(orc.slic_downsample_rgb(g["labels"], 16, g["srgb"]), g["low_rgb"])
