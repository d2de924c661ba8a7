"""CPU: known-answer / consistency tests of the surfel-path oracle (index map, splat, fuse, clean,
initialise, bilateral filter) and of the shared bit-exact expf."""
import ctypes as C
import os

import numpy as np

from multimotionfusion_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 160, 120
MAXD = 20.0  # maxDepthProcessed (MultiMotionFusion.cpp:53)


def scene(w=W, h=H, seed=0, pose=None):
    K = synth.intrinsics(w, h)
    pose = np.eye(4) if pose is None else pose
    f = synth.render(pose, w, h, seed=seed)
    return K, f


def test_mmf_expf_matches_libm(tmp_path):
    src = tmp_path / "e.c"
    src.write_text('#include "mmf_math.h"\nfloat e(float x){return mmf_expf(x);}\n')
    so = tmp_path / "e.so"
    assert os.system(f"gcc -O2 -ffp-contract=off -shared -fPIC -I{REPO}/include {src} -o {so} -lm") == 0
    lib = C.CDLL(str(so))
    lib.e.restype = C.c_float
    lib.e.argtypes = [C.c_float]
    xs = np.concatenate([np.linspace(-87, 88, 20001), np.linspace(-1, 1, 2001), [0.0, -0.0, 1e-8, -100.0, -104.0]]).astype(
        np.float32)
    got = np.array([lib.e(float(x)) for x in xs], np.float32)
    want = np.exp(xs.astype(np.float64))
    ok = want > 1e-37
    ulp = np.abs(got[ok].astype(np.float64) - want[ok]) / np.spacing(want[ok].astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 2.0, ulp.max()
    assert lib.e(0.0) == 1.0 and lib.e(float("inf")) == float("inf") and lib.e(-200.0) == 0.0


def test_bilateral_filter_properties(orc):
    K, f = scene()
    d = f["depth"]
    out = orc.bilateral_filter(d, 15.0)
    assert np.all(out[d < 0.3] == 0)
    const = np.full((40, 50), 2.0, np.float32)
    assert np.allclose(orc.bilateral_filter(const, 15.0), 2.0, atol=1e-6)
    assert np.all(orc.bilateral_filter(np.full((20, 20), 20.0, np.float32), 15.0) == 0)  # beyond maxD
    valid = (d > 0.3) & (out > 0)
    assert np.abs(out[valid] - d[valid]).max() < 0.05


def test_initialise_then_index_map_round_trip(orc):
    """Surfels created from a frame project back onto the pixels they came from."""
    K, f = scene()
    fil = orc.bilateral_filter(f["depth"], 15.0)
    s = orc.surfel_initialise(f["rgb"], f["depth"], fil, K, 1, MAXD)
    n_valid = int(((f["depth"] > 0) & (f["depth"] <= MAXD)).sum())
    assert s.shape[0] == n_valid
    assert np.all(s[:, 6] == 1) and np.all(s[:, 7] == 1) and np.all(s[:, 5] == 0)  # initTime, timestamp, unused
    # column-major emission: the first surfel comes from column 0
    index, vc, ct, nr = orc.predict_indices(s, np.eye(4), K, W, H, MAXD, 1, 200)
    hit = index > 0
    assert hit.mean() > 0.9 * (f["depth"] > 0).mean()
    ys, xs = np.nonzero(hit)
    # pixel = floor of the projection of the winning surfel
    p = s[index[ys, xs]]
    u = np.floor(K["fx"] * p[:, 0] / p[:, 2] + K["cx"]).astype(int)
    v = np.floor(K["fy"] * p[:, 1] / p[:, 2] + K["cy"]).astype(int)
    assert np.array_equal(u, xs) and np.array_equal(v, ys)
    assert np.allclose(vc[ys, xs, :3], p[:, :3], atol=1e-6) and np.array_equal(vc[ys, xs, 3], p[:, 3])
    # "empty" pixels are zero -- except the one won by surfel 0, whose id aliases "empty"
    # (index_map.vert:49, data.vert:142): its attributes are written, its index reads 0
    assert np.all(index[~hit] == 0) and int(vc[~hit].any(axis=-1).sum()) <= 1
    # colour survives the 24-bit float encoding
    c = s[0, 4].astype(np.int64)
    assert 0 <= c < 2 ** 24


def test_index_map_depth_test_and_culling(orc):
    K = synth.intrinsics(W, H)
    def surf(x, y, z, t=1.0):
        return [x, y, z, 5.0, 255.0, 0, 1, t, 0, 0, -1, 0.01]
    s = np.array([surf(0, 0, 3.0), surf(0, 0, 2.0), surf(0, 0, 2.0), surf(0, 0, 30.0), surf(0.5, 0, 1.0, t=-500.0)],
                 np.float32)
    index, vc, ct, nr = orc.predict_indices(s, np.eye(4), K, W, H, MAXD, 10, 200)
    cx, cy = int(np.floor(K["cx"])), int(np.floor(K["cy"]))
    assert index[cy, cx] == 1  # nearest wins; the tie between ids 1 and 2 goes to the lower id (GL_LESS)
    assert (index > 0).sum() == 1  # id 3 is beyond maxDepth, id 4 is older than timeDelta
    # vertexId 0 aliases "empty"
    index0, *_ = orc.predict_indices(s[:1], np.eye(4), K, W, H, MAXD, 10, 200)
    assert index0.max() == 0


def test_splat_prediction_reproduces_the_surface(orc):
    K, f = scene()
    fil = orc.bilateral_filter(f["depth"], 15.0)
    s = orc.surfel_initialise(f["rgb"], f["depth"], fil, K, 1, MAXD)
    s[:, 3] = 20.0  # confident
    image, vc, nr, tm = orc.combined_predict(s, np.eye(4), K, W, H, MAXD, 10.0, 1, 1, 200)
    cov = vc[..., 2] > 0
    assert cov.mean() > 0.9
    valid = cov & (f["depth"] > 0)
    assert np.median(np.abs(vc[..., 2][valid] - f["depth"][valid])) < 5e-3
    assert np.all(image[cov][:, 3] == 255) and np.all(tm[cov] == 1)
    # vertex x,y are the ray through the pixel centre scaled by z
    ys, xs = np.nonzero(cov)
    assert np.allclose(vc[ys, xs, 0], (xs + 0.5 - K["cx"]) * vc[ys, xs, 2] / K["fx"], atol=1e-5)
    # synthesizeDepth is the same splat keeping corrected_pos.z only
    sd = orc.synthesize_depth(s, np.eye(4), K, W, H, MAXD, 10.0, 1, 1, 200)
    assert np.array_equal(sd.view(np.uint32), np.ascontiguousarray(vc[..., 2]).view(np.uint32))
    # low-confidence surfels are not splatted
    s[:, 3] = 1.0
    image2, vc2, *_ = orc.combined_predict(s, np.eye(4), K, W, H, MAXD, 10.0, 1, 1, 200)
    assert not vc2.any() and not image2.any()


def test_fuse_merges_and_clean_keeps_order(orc):
    K, f = scene()
    fil = orc.bilateral_filter(f["depth"], 15.0)
    mask = np.zeros((H, W), np.uint8)
    s = orc.surfel_initialise(f["rgb"], f["depth"], fil, K, 1, MAXD)
    pose = np.eye(4, dtype=np.float32)
    index, vc, ct, nr = orc.predict_indices(s, pose, K, W, H, MAXD, 2, 200)
    s2, new = orc.fuse(s, f["rgb"], f["depth"], fil, mask, index, vc, nr, pose, K, 2, 1.0, 0, MAXD)
    merged = s2[:, 7] == 2  # timestamp updated
    # the same frame again: the quarter-rate pixels (x,y even at even time) merge with their own surfels
    assert merged.sum() > 0.15 * s.shape[0]
    assert np.all(s2[merged, 3] > s[merged, 3])  # confidence accumulates
    assert np.allclose(s2[merged, :3], s[merged, :3], atol=0.05)  # a merge may pair with a neighbouring pixel
    assert np.all(new[:, 7] == -2)
    index, vc, ct, nr = orc.predict_indices(s2, pose, K, W, H, MAXD, 2, 200)
    out = orc.clean(s2, new, pose, K, W, H, 2, 200, 10.0, 3.0, 0, index, vc, ct, fil, mask)
    assert out.shape[0] <= s2.shape[0] + new.shape[0]
    # order-preserving compaction: surviving old surfels keep their relative order
    kept_old = out[out[:, 6] == 1][:, :3]
    pos_bytes = {tuple(r) for r in s2[:, :3].round(6)}
    assert all(tuple(r) in pos_bytes for r in kept_old[:50].round(6))
    assert np.all(out[:, 7] >= 0)  # the -2 marker became the current time


def test_fill_in_and_thumbnail(orc):
    K, f = scene()
    fil = orc.bilateral_filter(f["depth"], 15.0)
    zero4 = np.zeros((H, W, 4), np.float32)
    img0 = np.zeros((H, W, 4), np.uint8)
    vo, no, io = orc.fill_in(zero4, zero4, img0, fil, f["rgb"], K, 0, 0)
    assert np.array_equal(vo[..., 2], fil) and np.array_equal(io[..., :3], f["rgb"])
    assert orc.requires_fill_in(img0) is True
    assert orc.requires_fill_in(io) is False


def test_surfel_cycle_golden_fixture(orc):
    """tests/golden/surfel_cycle_96x72.npz (written by make_golden.py from this oracle) pins the surfel-path
    restatement against silent changes: initialise -> predictIndices -> fuse -> predictIndices -> clean -> splat."""
    g = np.load(os.path.join(REPO, "tests", "golden", "surfel_cycle_96x72.npz"))
    h, w = g["depth0"].shape
    K = dict(zip(("fx", "fy", "cx", "cy"), (float(v) for v in g["intr"])))
    mask = np.zeros((h, w), np.uint8)
    fil0, fil1 = orc.bilateral_filter(g["depth0"], 15.0), orc.bilateral_filter(g["depth1"], 15.0)
    assert np.array_equal(fil1.view(np.uint32), g["filtered1"].view(np.uint32))
    s0 = orc.surfel_initialise(g["rgb0"], g["depth0"], fil0, K, 1, MAXD)
    assert np.array_equal(s0.view(np.uint32), g["surfels_init"].view(np.uint32))
    pose1 = g["pose1"]
    index, vc, ct, nr = orc.predict_indices(s0, pose1, K, w, h, MAXD, 2, 200)
    s_upd, new = orc.fuse(s0, g["rgb1"], g["depth1"], fil1, mask, index, vc, nr, pose1, K, 2, 1.0, 0, MAXD)
    index2, vc2, ct2, nr2 = orc.predict_indices(s_upd, pose1, K, w, h, MAXD, 2, 200)
    assert np.array_equal(index2, g["index_after_fuse"])
    s1 = orc.clean(s_upd, new, pose1, K, w, h, 2, 200, 10.0, 3.0, 0, index2, vc2, ct2, fil1, mask)
    assert np.array_equal(s1.view(np.uint32), g["surfels_final"].view(np.uint32))
    s1c = s1.copy()
    s1c[:, 3] = 20.0
    image, vcp, nrp, tm = orc.combined_predict(s1c, pose1, K, w, h, MAXD, 10.0, 2, 2, 200)
    assert np.array_equal(vcp.view(np.uint32), g["splat_vertexConf"].view(np.uint32)) and np.array_equal(image, g["splat_image"])


def test_libm_exp_variant(orc):
    """The checker shares mmf_expf (include/mmf_math.h) with the kernels, which by itself would make 'bit-exact' a
    comparison of a function with itself.  Here the SAME oracle is built against the C library's expf
    (-DORC_LIBM_EXP) and the outputs that pass through exp -- the bilateral filter and the confidence of new surfels --
    are compared: a handful of last-bit differences, nothing more, i.e. the bit-exact parity claims carry over to a
    libm-based statement of the shaders up to the ulp of exp itself."""
    path = orc.build(out="liboracle_libm.so", extra="-DORC_LIBM_EXP")
    alt = orc.lib(path)
    K, f = scene(320, 240, seed=3)
    d = np.ascontiguousarray(f["depth"], np.float32)
    rows, cols = d.shape
    pf = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    out_alt = np.zeros_like(d)
    alt.orc_bilateral_filter(pf(d), cols, rows, C.c_float(15.0), pf(out_alt))
    out = orc.bilateral_filter(d, 15.0)
    ne = out.view(np.uint32) != out_alt.view(np.uint32)
    ulp = np.abs(out.view(np.int32).astype(np.int64) - out_alt.view(np.int32).astype(np.int64))
    print(f"bilateral filter: {int(ne.sum())} of {ne.size} outputs differ between mmf_expf and libm expf, max {int(ulp.max())} ulp")
    # measured: 12.9 % of the outputs differ, by at most 5 ulp (a quotient of two 169-term sums of weights that are each within
    # 2 ulp of the other build's)
    assert ne.mean() < 0.35 and ulp.max() <= 8
    assert np.abs(out - out_alt).max() < 2e-6  # 1.2 micrometres at 3.3 m
    # surfel confidence (surfels.glsl:36-46) through Model::initialise
    rgb = np.ascontiguousarray(f["rgb"], np.uint8)
    s_alt = np.zeros((rows * cols, 12), np.float32)
    alt.orc_surfel_initialise.restype = C.c_int
    n = alt.orc_surfel_initialise(rgb.ctypes.data_as(C.POINTER(C.c_uint8)), pf(d), pf(out), cols, rows, C.c_float(K["cx"]),
                                  C.c_float(K["cy"]), C.c_float(K["fx"]), C.c_float(K["fy"]), 1, C.c_float(MAXD), pf(s_alt))
    s = orc.surfel_initialise(rgb, d, out, K, 1, MAXD)
    assert n == s.shape[0]
    s_alt = s_alt[:n]
    other = np.delete(np.arange(12), 3)
    assert np.array_equal(s[:, other].view(np.uint32), s_alt[:, other].view(np.uint32))  # only the confidence passes through exp
    du = np.abs(s[:, 3].view(np.int32).astype(np.int64) - s_alt[:, 3].view(np.int32).astype(np.int64))
    print(f"surfel confidence: {int((du > 0).sum())} of {n} differ, max {int(du.max())} ulp")
    assert du.max() <= 2
