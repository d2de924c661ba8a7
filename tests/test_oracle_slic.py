"""The super-pixel resampling oracle (oracle/mmf_oracle_slic.c) against a plain numpy statement of
Slic::downsample / downsampleThresholded / upsample (Core/Segmentation/Slic.h:48-146, Slic.cpp:82-112)."""
import numpy as np

from helpers import slic_like_labels
from oracle import oracle as orc


def means_f64(labels, n, values, mask=None):
    m = np.ones(values.shape, bool) if mask is None else mask
    s = np.bincount(labels[m].ravel(), weights=values[m].astype(np.float64).ravel(), minlength=n)
    c = np.bincount(labels[m].ravel(), minlength=n)
    return s, c


def test_counts_and_upsample():
    labels = slic_like_labels(160, 120, 16, seed=1)
    n = 10 * 7
    counts = orc.slic_counts(labels, n)
    assert np.array_equal(counts, np.bincount(labels.ravel(), minlength=n))
    small = np.arange(n, dtype=np.uint8)
    assert np.array_equal(orc.slic_upsample_u8(labels, small), small[labels])


def test_downsample_is_the_superpixel_mean():
    labels = slic_like_labels(160, 120, 16, seed=2)
    rng = np.random.default_rng(0)
    img = rng.random((120, 160, 4), dtype=np.float32)
    got = orc.slic_downsample(labels, 16, img, channel=3)
    s, c = means_f64(labels, 70, img[:, :, 3])
    assert got.shape == (7, 10)
    assert np.all(c > 0)
    assert np.abs(got.ravel() - s / c).max() < 1e-5


def test_empty_superpixels_take_their_substitute_with_the_in_place_quirk():
    W, H, S = 160, 128, 16  # 10 x 8: index / spixelY differs from index / spixelX
    labels = slic_like_labels(W, H, S, seed=3, empty_every=7)
    n = 80
    rng = np.random.default_rng(1)
    img = rng.random((H, W), dtype=np.float32) + 0.5
    got = orc.slic_downsample(labels, S, img).ravel()
    s, c = means_f64(labels, n, img)
    assert (c == 0).sum() >= 5
    for idx in np.nonzero(c == 0)[0]:
        hx, hy = idx % 10, idx // 8  # sic
        cx, cy = min(int(hx * S + S * 0.5), W - 1), min(int(hy * S + S * 0.5), H - 1)
        r = labels[cy, cx]
        want = (s[r] / c[r]) / c[r] if r < idx else s[r] / c[r]  # a lower substitute is already a mean
        assert abs(got[idx] - want) < 1e-5 * max(1.0, abs(want))
    ok = c > 0
    assert np.abs(got[ok] - s[ok] / c[ok]).max() < 1e-5


def test_thresholded_downsample_ignores_small_values():
    labels = slic_like_labels(160, 120, 16, seed=4)
    rng = np.random.default_rng(2)
    depth = rng.random((120, 160), dtype=np.float32) * 3.0
    depth[rng.random((120, 160)) < 0.3] = 0.0  # invalid depth
    depth[labels == 11] = 0.0                  # a super-pixel without any valid depth
    got = orc.slic_downsample(labels, 16, depth, threshold=0.02).ravel()
    s, c = means_f64(labels, 70, depth, depth > 0.02)
    ok = c > 0
    assert np.abs(got[ok] - s[ok] / c[ok]).max() < 1e-5
    assert c[11] == 0 and got[11] == 0.0  # substitute = itself (it has pixels): raw sum 0 / its pixel count


def test_rgb_downsample_is_the_integer_mean_of_reversed_channels():
    labels = slic_like_labels(160, 120, 16, seed=5)
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    got = orc.slic_downsample_rgb(labels, 16, rgb).reshape(-1, 3)
    for k in range(3):
        s, c = means_f64(labels, 70, rgb[:, :, 2 - k].astype(np.float32))
        assert np.array_equal(got[:, k], (s.astype(np.int64) // c).astype(np.uint8))
