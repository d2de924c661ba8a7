"""-m gpu: the SuperPoint network and keypoint post-processing on the device against the oracle
(oracle/mmf_oracle_superpoint.c).  The f32 matrix cores accumulate one fmaf chain per output in the order the
oracle defines (K blocks of 32 channels, taps, channels), exp is the shared mmf_expf and every other operation
is a correctly rounded IEEE one, so layers, logits, descriptors, heat map, keypoints and sampled descriptors
must all be BIT-EXACT -- partial tiles, pooled layers, every tile width, empty and saturated heat maps
included."""
import numpy as np
import pytest
import torch

from helpers import assert_bit_equal

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def oracle_layer(orc, x, w, b, relu, pool):
    y = orc.sp_conv(x, w, b, relu)
    if pool:
        H, W, C = y.shape
        y = y.reshape(H // 2, 2, W // 2, 2, C).max(axis=(1, 3))
    return y


@pytest.mark.parametrize("H,W,cin,cout,k,relu,pool,nt", [
    (8, 16, 32, 32, 3, True, False, 1),      # exactly one tile, one K block
    (24, 40, 64, 64, 3, True, False, 2),     # ragged right edge (40 = 2.5 tiles), two K blocks
    (24, 40, 64, 64, 3, True, True, 2),      # fused 2x2 max pool
    (20, 36, 128, 128, 3, True, True, 4),    # ragged both ways (20 = 2.5 tile rows), four K blocks, widest tile
    (20, 36, 128, 128, 3, True, False, 1),
    (6, 8, 128, 512, 3, True, False, 1),     # the fused detector | descriptor head
    (6, 8, 256, 65, 1, False, False, 1),     # 1x1, output channels not a multiple of 32, no ReLU
    (12, 20, 256, 256, 1, False, False, 4),
    (16, 16, 64, 96, 3, False, False, 0),    # automatic tile width, 96 channels (3 tiles of 32)
])
def test_conv_layer_bit_exact(gpu_ctx, orc, H, W, cin, cout, k, relu, pool, nt):
    from multimotionfusion_amd.superpoint import conv
    rng = np.random.default_rng(H * 1000 + W * 10 + cin + cout)
    x = rng.normal(0, 1, (H, W, cin)).astype(np.float32)
    w = rng.normal(0, np.sqrt(2.0 / (cin * k * k)), (cout, cin, k, k)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    got = conv(gpu_ctx, dev(x), w, b, relu=relu, pool=pool, nt=nt).cpu().numpy()
    assert_bit_equal(got, oracle_layer(orc, x, w, b, relu, pool), f"conv {cin}->{cout} k{k} pool={pool} nt={nt}")


def test_conv_rejects_bad_arguments(gpu_ctx):
    from multimotionfusion_amd import MmfError
    from multimotionfusion_amd.superpoint import conv
    x = torch.zeros((8, 16, 48), device="cuda")
    with pytest.raises(MmfError):
        conv(gpu_ctx, x, np.zeros((32, 48, 3, 3), np.float32), np.zeros(32, np.float32))  # cin % 32
    x = torch.zeros((7, 16, 32), device="cuda")
    with pytest.raises(MmfError):
        conv(gpu_ctx, x, np.zeros((32, 32, 3, 3), np.float32), np.zeros(32, np.float32), pool=True)  # odd height


@pytest.fixture(scope="module")
def net(gpu_ctx, orc):
    from multimotionfusion_amd.superpoint import SuperPoint
    weights = orc.sp_random_weights(seed=4)
    sp = SuperPoint(gpu_ctx, weights, max_width=160, max_height=120, max_keypoints=2048)
    yield sp, weights
    sp.close()


@pytest.mark.parametrize("H,W,ch", [(120, 160, 3), (48, 64, 1), (72, 104, 3), (8, 8, 1)])
def test_network_bit_exact(net, orc, H, W, ch):
    sp, weights = net
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W) if ch == 1 else (H, W, ch), dtype=np.uint8)
    semi, desc, heat = sp.forward(img)
    o_semi, o_desc = orc.sp_forward(orc.sp_input(img), weights)
    assert_bit_equal(semi, o_semi, "detector logits")
    assert_bit_equal(desc, o_desc, "coarse descriptors")
    assert_bit_equal(heat, orc.sp_heatmap(o_semi), "heat map")


def test_get_features_bit_exact(net, orc):
    sp, weights = net
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    xy, conf, desc = sp.keypoints(img)
    semi, cdesc = orc.sp_forward(orc.sp_input(img), weights)
    o_xy, o_conf = orc.sp_keypoints(orc.sp_heatmap(semi), sp.conf_thresh, sp.nms_dist, sp.border)
    assert len(o_xy) > 20
    assert_bit_equal(xy, o_xy, "keypoint pixels")
    assert_bit_equal(conf, o_conf, "keypoint confidences")
    assert_bit_equal(desc, orc.sp_sample_descriptors(cdesc, o_xy, 120, 160), "sampled descriptors")
    coords, descs = sp.getFeatures(img)
    o_coords, o_descs = orc.sp_get_features(img, weights)
    assert coords.dtype == np.float64 and descs.dtype == np.float64
    assert_bit_equal(coords, o_coords, "normalised coordinates")
    assert_bit_equal(descs, o_descs, "descriptors")


def test_keypoint_limit_keeps_the_strongest(gpu_ctx, orc):
    from multimotionfusion_amd.superpoint import SuperPoint
    weights = orc.sp_random_weights(seed=4)
    sp = SuperPoint(gpu_ctx, weights, max_width=160, max_height=120, max_keypoints=16)
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    xy, conf, _ = sp.keypoints(img)
    semi, _ = orc.sp_forward(orc.sp_input(img), weights)
    o_xy, o_conf = orc.sp_keypoints(orc.sp_heatmap(semi), sp.conf_thresh, sp.nms_dist, sp.border)
    assert len(xy) == 16
    assert_bit_equal(xy, o_xy[:16], "strongest 16")
    assert_bit_equal(conf, o_conf[:16], "their confidences")
    sp.close()


@pytest.mark.parametrize("case", ["flat", "ramp", "none", "radius0"])
def test_suppression_corner_cases(net, orc, case):
    """the fixed-point suppression against the sequential greedy one on adversarial heat maps, reached through
    the logits: a flat map (every pixel ties), a monotone ramp (longest dependency chain), nothing above the
    threshold, and radius 0 (every candidate survives)."""
    sp, weights = net
    H, W = 40, 56
    # the heat map cannot be written from outside, so it is shaped through the network: all-zero weights make
    # the logits of every cell equal the convPb bias, which is chosen per case
    from multimotionfusion_amd.superpoint import SuperPoint
    w = [(np.zeros_like(a), np.zeros_like(b)) for a, b in weights]
    bias = np.zeros(65, np.float32)
    if case == "ramp":
        bias[:64] = np.linspace(0.0, 3.0, 64, dtype=np.float32)
    elif case == "none":
        bias[64] = 20.0  # the dustbin takes all the mass
    w[9] = (w[9][0], bias)
    sp2 = SuperPoint(sp.ctx, w, max_width=W, max_height=H, max_keypoints=H * W)
    sp2.nms_dist = 0 if case == "radius0" else 4
    img = np.zeros((H, W), np.uint8)
    xy, conf, desc = sp2.keypoints(img)
    semi, cdesc = orc.sp_forward(orc.sp_input(img), w)
    o_xy, o_conf = orc.sp_keypoints(orc.sp_heatmap(semi), sp2.conf_thresh, sp2.nms_dist, sp2.border)
    assert_bit_equal(xy, o_xy, "keypoint pixels")
    assert_bit_equal(conf, o_conf, "keypoint confidences")
    if case == "none":
        assert len(xy) == 0
    if case == "radius0":
        assert len(xy) == (H - 8) * (W - 8)
    if case == "flat":
        assert len(xy) > 0 and np.all(conf == conf[0])
    sp2.close()


def test_forward_rejects_bad_sizes(net):
    from multimotionfusion_amd import MmfError
    sp, _ = net
    with pytest.raises(MmfError):
        sp.forward(np.zeros((50, 64), np.uint8))      # not a multiple of 8
    with pytest.raises(MmfError):
        sp.forward(np.zeros((240, 320), np.uint8))    # larger than the object was created for


def test_golden_fixture(gpu_ctx, orc):
    """the committed fixture (tests/golden/superpoint_slic.npz): the HIP path against stored outputs, without the
    oracle in the loop (only its seeded weight generator)"""
    import os
    from multimotionfusion_amd import slic
    from multimotionfusion_amd.superpoint import SuperPoint
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "superpoint_slic.npz"))
    sp = SuperPoint(gpu_ctx, orc.sp_random_weights(seed=int(g["weights_seed"])), max_width=64, max_height=48, max_keypoints=512)
    semi, desc, heat = sp.forward(g["image"])
    assert_bit_equal(semi, g["semi"], "logits")
    assert_bit_equal(desc, g["desc"], "coarse descriptors")
    assert_bit_equal(heat, g["heat"], "heat map")
    xy, conf, kdesc = sp.keypoints(g["image"])
    assert_bit_equal(xy, g["xy"], "keypoints")
    assert_bit_equal(conf, g["conf"], "confidences")
    assert_bit_equal(kdesc, g["kdesc"], "descriptors")
    sp.close()
    labels = dev(g["labels"])
    assert_bit_equal(slic.downsample(gpu_ctx, labels, 16, dev(g["smap"])).cpu().numpy(), g["low"], "super-pixel means")
    assert_bit_equal(slic.downsample(gpu_ctx, labels, 16, dev(g["sdepth"]), threshold=0.02).cpu().numpy(), g["low_depth"],
                     "thresholded means")
    assert_bit_equal(slic.downsample_rgb(gpu_ctx, labels, 16, dev(g["srgb"])).cpu().numpy(), g["low_rgb"], "RGB means")


def test_network_bit_exact_at_the_bench_size(gpu_ctx, orc):
    """640x480 (BASELINE's frame size): the whole forward pass against the oracle, plus properties that hold at
    any size -- every cell's 64 heat values and its dustbin share sum to 1 - 1e-5/sum, descriptors have unit norm"""
    from multimotionfusion_amd.superpoint import SuperPoint
    weights = orc.sp_random_weights(seed=9)
    sp = SuperPoint(gpu_ctx, weights, max_width=640, max_height=480, max_keypoints=4096)
    rng = np.random.default_rng(640)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    semi, desc, heat = sp.forward(img)
    o_semi, o_desc = orc.sp_forward(orc.sp_input(img), weights)
    assert_bit_equal(semi, o_semi, "detector logits 640x480")
    assert_bit_equal(desc, o_desc, "coarse descriptors 640x480")
    assert_bit_equal(heat, orc.sp_heatmap(o_semi), "heat map 640x480")
    cells = heat.reshape(60, 8, 80, 8).transpose(0, 2, 1, 3).reshape(60, 80, 64).astype(np.float64)
    e = np.exp(semi.astype(np.float64))
    dust = e[:, :, 64] / (e.sum(axis=2) + 1e-5)
    assert np.abs(cells.sum(axis=2) + dust - 1.0).max() < 1e-4
    assert np.abs(np.linalg.norm(desc.astype(np.float64), axis=2) - 1.0).max() < 1e-5
    xy, conf, kdesc = sp.keypoints(img)
    o_xy, o_conf = orc.sp_keypoints(orc.sp_heatmap(o_semi), sp.conf_thresh, sp.nms_dist, sp.border)
    assert_bit_equal(xy, o_xy[:4096], "keypoints 640x480")
    assert_bit_equal(conf, o_conf[:4096], "confidences 640x480")
    d = np.abs(xy[:, None, :].astype(np.int64) - xy[None, :, :]).max(axis=2) + np.eye(len(xy), dtype=np.int64) * 99
    assert d.min() > sp.nms_dist  # survivors are farther apart than the suppression radius
    sp.close()


@pytest.mark.parametrize("seed", range(10))
def test_conv_layer_random_shapes_bit_exact(gpu_ctx, orc, seed):
    """seeded random shapes: images smaller than a tile, single rows / columns, output channel counts that are
    not multiples of 32, 1x1 and 3x3, pooled where both sides are even, every tile width"""
    from multimotionfusion_amd.superpoint import conv
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(1, 41)), int(rng.integers(1, 51))
    cin = int(rng.choice([32, 64, 96, 128]))
    cout = int(rng.integers(1, 201))
    k = int(rng.choice([1, 3]))
    pool = bool(k == 3 and H % 2 == 0 and W % 2 == 0 and rng.integers(0, 2))
    nt = int(rng.choice([0, 1, 2, 4]))
    relu = bool(rng.integers(0, 2))
    x = rng.normal(0, 1, (H, W, cin)).astype(np.float32)
    w = rng.normal(0, np.sqrt(2.0 / (cin * k * k)), (cout, cin, k, k)).astype(np.float32)
    b = rng.normal(0, 0.1, cout).astype(np.float32)
    got = conv(gpu_ctx, dev(x), w, b, relu=relu, pool=pool, nt=nt).cpu().numpy()
    assert_bit_equal(got, oracle_layer(orc, x, w, b, relu, pool), f"{H}x{W} {cin}->{cout} k{k} pool={pool} nt={nt} relu={relu}")
