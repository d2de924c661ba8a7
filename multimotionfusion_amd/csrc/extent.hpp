// extent.hpp -- the image extent of a model's prediction at one pyramid level.
//
// An object model predicts a few thousand surfels: its depth / vertex pyramids (Model.cpp:359-407, RGBDOdometry.cpp:108-179)
// are NaN outside a box of a hundred pixels, yet every pass of its Gauss-Newton chain walks the whole image
// (MultiMotionFusion.cpp:312-387 does too -- the reference's cost, not a design).  The preparation jobs that WRITE those
// pyramids note the bounding box of what they write that is valid (one ballot per 64-pixel row segment, four atomics per
// segment with a valid pixel), and the photometric passes of a batched chain skip the 256-pixel blocks outside it: a pixel
// takes part in computeRgbResidual / rgbStep only if ITS OWN model depth is a number (reduce.cu:600: `!isnan(d1)`; next_depth
// is the prediction's depth, RGBDOdometry.cpp:179), so a skipped block contributes exactly the zeros it would have added.
//
// Four 64-bit words per level, all raised by atomicMax: {gen << 32 | 0xFFFF - x0, gen << 32 | x1, gen << 32 | 0xFFFF - y0,
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
// (uniform address: four scalar loads)
__device__ __forceinline__ ExtentBox extent_load(const unsigned long long* words, unsigned gen) {
    ExtentBox e{1, 1, 0, 0};
    const unsigned long long w0 = words[0], w1 = words[1], w2 = words[2], w3 = words[3];
    if ((unsigned)(w0 >> 32) != gen || (unsigned)(w1 >> 32) != gen || (unsigned)(w2 >> 32) != gen || (unsigned)(w3 >> 32) != gen) return e;
    e.x0 = 0xFFFF - (int)(unsigned)(w0 & 0xFFFFFFFFull), e.x1 = (int)(unsigned)(w1 & 0xFFFFFFFFull);
    e.y0 = 0xFFFF - (int)(unsigned)(w2 & 0xFFFFFFFFull), e.y1 = (int)(unsigned)(w3 & 0xFFFFFFFFull);
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

}  // namespace mmf
