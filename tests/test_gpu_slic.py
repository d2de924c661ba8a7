"""-m gpu: super-pixel resampling on the device (mmf_slic_*) against the oracle.  Float sums run in pixel
order on both sides, counts and RGB sums are integers: everything must be bit-exact, empty super-pixels and
the reference's in-place division included."""
import numpy as np
import pytest
import torch

from helpers import assert_bit_equal, slic_like_labels

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("W,H,S,empty_every", [(640, 480, 16, 0), (640, 480, 16, 9), (160, 128, 16, 7), (330, 250, 20, 5),
                                               (64, 48, 11, 0)])
def test_downsample_float(gpu_ctx, orc, W, H, S, empty_every):
    from multimotionfusion_amd import slic
    labels = slic_like_labels(W, H, S, seed=W + S, empty_every=empty_every)
    n = (W // S) * (H // S)
    rng = np.random.default_rng(W)
    icp_err = (rng.random((H, W), dtype=np.float32) ** 4) * 0.1          # ICP error map: 1 channel
    vert_conf = rng.random((H, W, 4), dtype=np.float32) * 10.0           # vertex-confidence map: channel 3
    got, counts = slic.downsample(gpu_ctx, dev(labels), S, dev(icp_err), with_counts=True)
    assert_bit_equal(got.cpu().numpy(), orc.slic_downsample(labels, S, icp_err), "ICP error means")
    assert_bit_equal(counts.cpu().numpy().ravel(), orc.slic_counts(labels, n), "spixelCounts")
    got = slic.downsample(gpu_ctx, dev(labels), S, dev(vert_conf), channel=3)
    assert_bit_equal(got.cpu().numpy(), orc.slic_downsample(labels, S, vert_conf, channel=3), "confidence means")


def test_downsample_thresholded_depth(gpu_ctx, orc):
    from multimotionfusion_amd import slic
    W, H, S = 640, 480, 16
    labels = slic_like_labels(W, H, S, seed=8, empty_every=11)
    rng = np.random.default_rng(4)
    depth = rng.random((H, W), dtype=np.float32) * 4.0
    depth[rng.random((H, W)) < 0.25] = 0.0
    depth[np.isin(labels, (5, 6, 47, 300))] = 0.0  # super-pixels with pixels but no valid depth
    got = slic.downsample(gpu_ctx, dev(labels), S, dev(depth), threshold=0.02)
    assert_bit_equal(got.cpu().numpy(), orc.slic_downsample(labels, S, depth, threshold=0.02), "lowDepth")


def test_downsample_rgb_and_upsample(gpu_ctx, orc):
    from multimotionfusion_amd import slic
    W, H, S = 640, 480, 16
    labels = slic_like_labels(W, H, S, seed=9, empty_every=13)
    rng = np.random.default_rng(5)
    for ch in (3, 4):
        rgb = rng.integers(0, 256, (H, W, ch), dtype=np.uint8)
        got = slic.downsample_rgb(gpu_ctx, dev(labels), S, dev(rgb))
        assert_bit_equal(got.cpu().numpy(), orc.slic_downsample_rgb(labels, S, rgb), f"lowRGB ({ch} channels)")
    small = rng.integers(0, 256, (H // S) * (W // S), dtype=np.uint8)
    got = slic.upsample_u8(gpu_ctx, dev(labels), dev(small))
    assert_bit_equal(got.cpu().numpy(), orc.slic_upsample_u8(labels, small), "fullSegmentation")


def test_scattered_labels_are_still_summed_in_pixel_order(gpu_ctx, orc):
    """labels need not be compact: a random label image makes every bounding box the whole frame"""
    from multimotionfusion_amd import slic
    W, H, S = 96, 64, 16
    rng = np.random.default_rng(6)
    labels = rng.integers(0, (W // S) * (H // S), (H, W)).astype(np.int32)
    img = rng.random((H, W), dtype=np.float32)
    got = slic.downsample(gpu_ctx, dev(labels), S, dev(img))
    assert_bit_equal(got.cpu().numpy(), orc.slic_downsample(labels, S, img), "means over scattered labels")


def test_rejects_bad_superpixel_size(gpu_ctx):
    from multimotionfusion_amd import MmfError, slic
    labels = torch.zeros((48, 64), dtype=torch.int32, device="cuda")
    img = torch.zeros((48, 64), device="cuda")
    with pytest.raises(MmfError):
        slic.downsample(gpu_ctx, labels, 8, img)     # Slic.cpp:23: spixelSize > 10
    with pytest.raises(MmfError):
        slic.downsample(gpu_ctx, labels, 100, img)   # does not fit the image
