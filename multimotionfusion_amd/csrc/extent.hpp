// extent.hpp -- the image extent of a model's prediction at one pyramid level.
//
// An object model predicts a few thousand surfels: its depth pyramid (Model.cpp:359-407, RGBDOdometry.cpp:108-179) is NaN
// outside a box of a hundred pixels, yet every pass of its Gauss-Newton chain walks the whole image
// (MultiMotionFusion.cpp:312-387 does too -- the reference's cost, not a design).  The preparation job that WRITES the
// COARSEST level of that pyramid notes the bounding box of what it writes that is valid (one ballot per 64-pixel row
// segment, four atomics per segment with a valid pixel: a few dozen segments at 160x120 -- noting it at every level, three
// thousand same-address atomics per model, doubled the preparation launches of eight models), the finer levels scale the box
// up, and the photometric passes of the two-launch chain skip the 256-pixel blocks outside it: a pixel
// takes part in computeRgbResidual / rgbStep only if ITS OWN model depth is a number (reduce.cu:600: `!isnan(d1)`; next_depth
// is the prediction's depth, RGBDOdometry.cpp:179), so a skipped block contributes exactly the zeros it would have added.
//
// Four 64-bit words per box (three boxes per model: extent_of_level), all raised by atomicMax: {gen << 32 | 0xFFFF - x0, gen << 32 | x1, gen << 32 | 0xFFFF - y0,
// gen << 32 | y1}.  `gen` is the frame's number: a newer frame's first note supersedes whatever an older frame left, nothing
// is ever reset, and a word whose upper half is not this frame's says "no valid pixel".
#pragma once
#include <hip/hip_runtime.h>

namespace mmf {

struct ExtentRef {
    unsigned long long* words;  // this level's four words (null: no extent is kept)
    unsigned gen;               // != 0
};

// called by all 64 lanes of a wave that holds pixels x_lane = x, one row y (prep_batch.hpp's tiles)
__device__ __forceinline__ void extent_note(unsigned long long* words, unsigned gen, int x, int y, bool valid) {
    if (words == nullptr) return;  // (uniform)
    const unsigned long long b = __ballot(valid);
    if (b == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u);
    if (lane != 0) return;
    const unsigned long long g = (unsigned long long)gen << 32;
    const unsigned x0 = (unsigned)(x + __builtin_ctzll(b)), x1 = (unsigned)(x + 63 - __builtin_clzll(b));
    atomicMax(&words[0], g | (0xFFFFu - x0));
    atomicMax(&words[1], g | x1);
    atomicMax(&words[2], g | (0xFFFFu - (unsigned)y));
    atomicMax(&words[3], g | (unsigned)y);
}

struct ExtentBox {
    int x0, y0, x1, y1;  // inclusive; empty: x1 < x0
};
// (uniform address: four scalar loads).  `shift`: the box was noted `shift` pyramid levels above the one that asks
__device__ __forceinline__ ExtentBox extent_load(const unsigned long long* words, unsigned gen, int shift = 0) {
    ExtentBox e{1, 1, 0, 0};
    const unsigned long long w0 = words[0], w1 = words[1], w2 = words[2], w3 = words[3];
    if ((unsigned)(w0 >> 32) != gen || (unsigned)(w1 >> 32) != gen || (unsigned)(w2 >> 32) != gen || (unsigned)(w3 >> 32) != gen) return e;
    e.x0 = 0xFFFF - (int)(unsigned)(w0 & 0xFFFFFFFFull), e.x1 = (int)(unsigned)(w1 & 0xFFFFFFFFull);
    e.y0 = 0xFFFF - (int)(unsigned)(w2 & 0xFFFFFFFFull), e.y1 = (int)(unsigned)(w3 & 0xFFFFFFFFull);
    e.x0 <<= shift, e.y0 <<= shift;
    e.x1 = ((e.x1 + 1) << shift) - 1, e.y1 = ((e.y1 + 1) << shift) - 1;
    return e;
}
__device__ __forceinline__ ExtentBox extent_hull(ExtentBox a, ExtentBox b) {
    if (a.x1 < a.x0) return b;
    if (b.x1 < b.x0) return a;
    return ExtentBox{min(a.x0, b.x0), min(a.y0, b.y0), max(a.x1, b.x1), max(a.y1, b.y1)};
}
// The box of a three-level depth pyramid's level `level` (0 = finest) from the model's 12 words: [8..11] = the valid pixels of
// level 2, [4..7] / [0..3] = the valid pixels in the LAST column or row of level 1 / level 0.  A depth that is a number makes
// its parent a number -- the parent's 5x5 window holds it (cudafuncs.cu:333-364) -- except in the last column and row of the
// finer image, which that window leaves out (its quirk: [max(0, 2x - 2), min(2x + 3, cols - 1))): hence the two extra boxes.
__device__ __forceinline__ ExtentBox extent_of_level(const unsigned long long* words, unsigned gen, int level) {
    ExtentBox e = extent_load(words + 8, gen, 2 - level);
    if (level <= 1) e = extent_hull(e, extent_load(words + 4, gen, 1 - level));
    if (level == 0) e = extent_hull(e, extent_load(words, gen, 0));
    return e;
}
// do the `count` pixels from linear index `first` on (row major, `cols` per row, cols_magic = floor(2^32 / cols) + 1) all
// lie outside the box?
__device__ __forceinline__ bool extent_misses(const ExtentBox& e, unsigned first, unsigned count, int cols, unsigned cols_magic) {
    const unsigned last = first + count - 1u;
    const int ya = (int)__umulhi(first, cols_magic), yb = (int)__umulhi(last, cols_magic);
    int xa = (int)first - ya * cols, xb = (int)last - yb * cols;
    if (yb != ya) xa = 0, xb = cols - 1;
    return e.x1 < e.x0 || xb < e.x0 || xa > e.x1 || yb < e.y0 || ya > e.y1;
}


// ---- the model's predicted VERTICES: their pixel box and depth range at level 0 ----
// Six more words behind the twelve above ([12..14] the minima, [15..17] the maxima of pixel x, pixel y and camera-frame z of the
// prediction's valid vertices), noted by the preparation job that writes the level-0 global-frame records (prep_batch.hpp:
// PREP_TEX_TP), one set of atomics per WORKGROUP with a valid vertex.  A vertex is valid where the prediction drew something
// (copy_maps_px: z != 0) -- NOT the same set as the valid depths above (verticesToDepth drops z > 6 m, cudafuncs.cu:602-613).
// The coarser levels' vertices are averages of four valid finer ones (cudafuncs.cu:366-417): their pixels are the level-0 box
// shifted right, their depths lie in the same range.  What it is for: gn_fused.hpp, gn_sparse_icp_box.
__device__ __forceinline__ unsigned aabb_key_fwd(float f) {
    const unsigned b = __builtin_bit_cast(unsigned, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float aabb_unkey_fwd(unsigned k) { return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }
constexpr int kExtentWords = 20;
// Words [18], [19]: the smallest valid depth of the SENSOR frame's level-0 vertex map (createVMap: z != 0 && z < cutoff), noted
// by the sensor-side preparation job that writes it (one atomic per workgroup), in slot gen & 1 -- the next frame's sensor
// side is prepared while this frame's chain still reads its own slot.  The coarser levels' depths are weighted means of
// finer ones (pyrDownGaussF): no smaller.  For gn_sparse_icp_box's error-image launch.
__device__ __forceinline__ void sensor_zmin_note(unsigned long long* words, unsigned gen, float zmin) {
    atomicMax(&words[18 + (gen & 1u)], ((unsigned long long)gen << 32) | (unsigned long long)(0xFFFFFFFFu - aabb_key_fwd(zmin)));
}
__device__ __forceinline__ bool sensor_zmin_load(const unsigned long long* words, unsigned gen, float& zmin) {
    const unsigned long long w = words[18 + (gen & 1u)];
    zmin = aabb_unkey_fwd(0xFFFFFFFFu - (unsigned)(w & 0xFFFFFFFFull));
    return gen != 0u && (unsigned)(w >> 32) == gen;
}
__device__ __forceinline__ unsigned aabb_key(float f) { return aabb_key_fwd(f); }  // monotonic in f (no NaN: only valid vertices are noted)
__device__ __forceinline__ float aabb_unkey(unsigned k) { return aabb_unkey_fwd(k); }
// one thread: lo / hi of {x, y, z} over the valid vertices its workgroup wrote (called only when there is one)
__device__ __forceinline__ void aabb_note(unsigned long long* words, unsigned gen, const float (&lo)[3], const float (&hi)[3]) {
    const unsigned long long g = (unsigned long long)gen << 32;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        atomicMax(&words[12 + k], g | (unsigned long long)(0xFFFFFFFFu - aabb_key(lo[k])));
        atomicMax(&words[15 + k], g | (unsigned long long)aabb_key(hi[k]));
    }
}
// (uniform address: six scalar loads)  false: this frame noted no valid vertex
__device__ __forceinline__ bool aabb_load(const unsigned long long* words, unsigned gen, float (&lo)[3], float (&hi)[3]) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const unsigned long long a = words[12 + k], b = words[15 + k];
        ok = ok && (unsigned)(a >> 32) == gen && (unsigned)(b >> 32) == gen;
        lo[k] = aabb_unkey(0xFFFFFFFFu - (unsigned)(a & 0xFFFFFFFFull));
        hi[k] = aabb_unkey((unsigned)(b & 0xFFFFFFFFull));
    }
    return ok;
}


// ---- boxes kept as plain ints {x0, y0, x1, y1} (one writer per launch, read by later launches) ----
__device__ __forceinline__ ExtentBox box_of_ints(const int* p) { return ExtentBox{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void box_to_ints(int* p, const ExtentBox& e) {
    const bool none = e.x1 < e.x0 || e.y1 < e.y0;
    p[0] = none ? 1 : e.x0, p[1] = none ? 1 : e.y0, p[2] = none ? 0 : e.x1, p[3] = none ? 0 : e.y1;
}
__device__ __forceinline__ ExtentBox box_clip(ExtentBox e, int cols, int rows) {
    e.x0 = max(e.x0, 0), e.y0 = max(e.y0, 0), e.x1 = min(e.x1, cols - 1), e.y1 = min(e.y1, rows - 1);
    return e;
}
// Per surfel model, device resident (pass_rect.hpp; prep_batch.hpp reads spl_nz): where the key image was last written and
// where the model's images are non-zero.
struct PassBoxes {
    unsigned long long key[2][4];  // [g & 1]: box of the key-image writes of projection launch g (the words above, generation g)
    int idx_nz[2][4];              // [g & 1]: where the index-map images are non-zero after resolve g: {x0, y0, x1, y1}, x1 < x0 = nowhere
    int spl_nz[2][4];              // likewise the prediction images
};

}  // namespace mmf
