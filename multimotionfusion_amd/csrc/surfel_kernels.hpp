// surfel_kernels.hpp -- pure-HIP compute replacement of the reference's OpenGL surfel passes
// (no GL interop):
//   index_map_kernel + index_resolve_kernel   <- index_map.vert/.frag   (ModelProjection.cpp:94-143)
//   splat_kernel + splat_resolve_kernel       <- splat.vert + combo_splat.frag (ModelProjection.cpp:187-269)
//   fuse_data_kernel + fuse_update_kernel     <- data.vert/.geom/.frag + update.vert (Model.cpp:893-1048)
//   clean_flag_kernel + clean_scatter_kernel  <- copy_unstable.vert/.geom (Model.cpp:1050-1182)
//   feedback_kernel + init_scatter_kernel     <- vertex_feedback.vert/.geom + init_unstable.vert (Model.cpp:267-312)
//   bilateral_filter_kernel                   <- depth_bilateral_metric.frag (MultiMotionFusion.cpp:897-904)
//   fill_in_kernel, thumbnail_count_kernel    <- fill_*.frag, resize.frag + requiresFillIn
//
// Data layout: the surfel store is a structure of three float4 arrays (position+confidence,
// colour/time, normal+radius), so a wave64 reads 1 KiB contiguous per attribute (16 B per lane);
// the reference's 48-byte AoS vertex (Vertex::SIZE) only exists at the download boundary.
// GL's depth-tested rasterisation becomes a 64-bit atomicMin of (24-bit depth << 32 | vertexId)
// per pixel followed by a per-pixel resolve pass that re-derives the winner's attributes; GL's
// transform-feedback compaction becomes flag -> exclusive scan -> scatter, which keeps the
// reference's draw order (surfel order, then column-major pixel order for new points).
// The arithmetic follows oracle/mmf_oracle_surfel.c statement by statement (assumptions A1-A5
// about the fixed-function GL state are listed there); built without FMA contraction.
#pragma once
#include "extent.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmf_math.h"
#include "frame_rider.hpp"

// The surfel passes run on the model's stream, which is what a frame waits for, while the next frame's bilateral filter
// (ALU bound on every CU for ~42 us) runs beside them on a side stream: their waves ask for issue priority over the
// filter's (s_setprio; the default is 0, the lowest).
#ifndef MMF_NO_PRIO
#define MMF_MODEL_STREAM_PRIORITY() __builtin_amdgcn_s_setprio(3)
#else
#define MMF_MODEL_STREAM_PRIORITY() do {} while (0)
#endif

namespace mmf {

struct v3 {
    float x, y, z;
};
__device__ __forceinline__ v3 V3(float x, float y, float z) { return v3{x, y, z}; }
__device__ __forceinline__ v3 v3add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 v3sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 v3scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 v3cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ v3 v3normalize(v3 a) { return v3scale(a, 1.0f / sqrtf(v3dot(a, a))); }
__device__ __forceinline__ float v3length(v3 a) { return sqrtf(v3dot(a, a)); }

struct Mat4 {  // row major, passed by value as a kernel argument
    float m[16];
};
__device__ __forceinline__ v3 m4point(const Mat4& M, v3 p) {
    const float* m = M.m;
    return V3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__device__ __forceinline__ v3 m4dir(const Mat4& M, v3 n) {
    const float* m = M.m;
    return V3(m[0] * n.x + m[1] * n.y + m[2] * n.z, m[4] * n.x + m[5] * n.y + m[6] * n.z,
              m[8] * n.x + m[9] * n.y + m[10] * n.z);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int texel(float coord, int size) { return clampi((int)floorf(coord * (float)size), 0, size - 1); }

struct Cam {  // cx, cy, fx, fy and the reciprocals the shaders receive as uniforms
    float cx, cy, fx, fy, ifx, ify;
};

__device__ __forceinline__ float encode_color(float r, float g, float b) {  // color_encoding.glsl
    int rgb = (int)roundf(r * 255.0f);
    rgb = (rgb << 8) + (int)roundf(g * 255.0f);
    rgb = (rgb << 8) + (int)roundf(b * 255.0f);
    return (float)rgb;
}
__device__ __forceinline__ v3 decode_color(float c) {
    const int ci = (int)c;
    return V3((float)(ci >> 16 & 0xFF) / 255.0f, (float)(ci >> 8 & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f);
}
__device__ __forceinline__ float get_radius(float depth, float norm_z, float ifx, float ify) {  // surfels.glsl:19-34
    const float meanFocal = ((1.0f / fabsf(ifx)) + (1.0f / fabsf(ify))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    const float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius;
    radius_n = radius_n / fabsf(norm_z);
    radius_n = fminf(2.0f * radius, radius_n);
    return radius_n;
}
__device__ __forceinline__ float confidence(float x, float y, float cx, float cy, float weighting) {  // surfels.glsl:36-46
    const float maxRadDist = 400, twoSigmaSquared = 0.72f;
    const float px = x - cx, py = y - cy;
    const float radialDist = sqrtf(px * px + py * py) / maxRadDist;
    return mmf_expf((-(radialDist * radialDist) / twoSigmaSquared)) * weighting;
}
__device__ __forceinline__ uint32_t depth24(float zw) {
    if (!(zw >= 0.f)) zw = 0.f;
    if (zw > 1.f) zw = 1.f;
    return (uint32_t)(zw * 16777215.0f + 0.5f);
}
// texture coordinate of pixel centre i as the reference's host code builds it (Model.cpp:206-210)
__device__ __forceinline__ float uv_coord(int i, int n) {
    return (float)((double)((float)i / (float)n) + 1.0 / (2 * (double)(float)n));
}
__device__ __forceinline__ v3 get_vertex(const float* __restrict__ depth, int cols, int rows, float tx, float ty,
                                         float x, float y, const Cam& c) {  // geometry.glsl:22-26
    const float z = depth[texel(ty, rows) * cols + texel(tx, cols)];
    return V3((x - c.cx) * z * c.ifx, (y - c.cy) * z * c.ify, z);
}
__device__ __forceinline__ v3 get_normal(const float* __restrict__ depth, int cols, int rows, v3 p, float tx,
                                         float ty, float x, float y, const Cam& c) {  // geometry.glsl:28-40
    const v3 xf = get_vertex(depth, cols, rows, tx + (1.0f / cols), ty, x + 1, y, c);
    const v3 xb = get_vertex(depth, cols, rows, tx - (1.0f / cols), ty, x - 1, y, c);
    const v3 yf = get_vertex(depth, cols, rows, tx, ty + (1.0f / rows), x, y + 1, c);
    const v3 yb = get_vertex(depth, cols, rows, tx, ty - (1.0f / rows), x, y - 1, c);
    const v3 del_x = v3sub(v3scale(v3add(xb, p), 0.5f), v3scale(v3add(xf, p), 0.5f));
    const v3 del_y = v3sub(v3scale(v3add(yb, p), 0.5f), v3scale(v3add(yf, p), 0.5f));
    return v3normalize(v3cross(del_x, del_y));
}

// the surfel store (one of the two ping-pong sets)
struct SurfelSoA {
    float4* pos;  // xyz + confidence
    float4* col;  // colour24-as-float, unused, initTime, timestamp
    float4* nrm;  // normal xyz + radius
};

constexpr unsigned long long kEmptyKey = ~0ull;

// ---- depth bilateral filter ------------------------------------------------------------------
__global__ __launch_bounds__(256) void bilateral_filter_kernel(const float* __restrict__ depth, int cols, int rows,
                                                               float maxD, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= cols || y >= rows) return;
    const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 555.556f;
    const int R = 6, D = R * 2 + 1;
    const float value = depth[y * cols + x];
    if (value > maxD || value < 0.3f) {
        out[y * cols + x] = 0;
        return;
    }
    // 13 x 13 taps in the shader's order (rows ascending, columns ascending; taps outside the image
    // are skipped).  The inner row is fully unrolled with compile-time spatial terms and loads from
    // clamped addresses, so the 13 loads of a row issue together and nothing depends on a branch;
    // space2 = dx^2 + dy^2 is a small integer, exact in float like the shader's float arithmetic.
    float sum1 = 0, sum2 = 0;
    for (int dy = -R; dy <= R; ++dy) {
        const int cy = y + dy;
        if (cy < 0 || cy >= rows) continue;  // wave-uniform except at the image border rows
        const float dy2 = (float)(dy * dy);
        const float* rowp = depth + (size_t)cy * cols;
        float taps[D];
#pragma unroll
        for (int k = 0; k < D; ++k) taps[k] = rowp[min(max(x + k - R, 0), cols - 1)];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int cx = x + k - R;
            const float tmp = taps[k];
            const float space2 = (float)((k - R) * (k - R)) + dy2;
            const float color2 = (value - tmp) * (value - tmp);
            const float weight = mmf_expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
            if (cx >= 0 && cx < cols) {
                sum1 += tmp * weight;
                sum2 += weight;
            }
        }
    }
    out[y * cols + x] = sum1 / sum2;
}

// Two horizontally adjacent pixels per lane: the filter is pure instruction issue -- 169 taps x ~50
// instructions per pixel, 64 us at 640x480.  Sharing the per-tap scalar work (tap loads, spatial term,
// loop control) between two pixels and the two exact rewrites below bring it to 47 us, the explicit fmaf steps
// of mmf_expf (v_pk_fma_f32 here) to 38 us.  The pair is
// written with packed registers (v_pk_mul_f32 / v_pk_add_f32); measured, a packed op costs what its two
// scalar ops cost on gfx950 (an unpacked build of the same kernel: 48 us), so the packing itself is
// neutral.  Bit-identical to the kernel above: the same float operations in the same order per pixel, with two exact rewrites of
// mmf_expf for its argument range here (x <= 0 or NaN):
//   * the `x > 88.7` overflow test can never fire;
//   * (p * 2^(n/2)) * 2^(n - n/2) == ldexpf(p, n): the first product is exact (p in [0.5, 2],
//     n/2 >= -75), so both forms round the exact value p * 2^n once, also into the subnormals.
// Waves whose 13-column windows lie inside the image (all but the first / last workgroup of a row)
// take the INTERIOR instantiation without the per-tap column tests.
typedef float v2fs __attribute__((ext_vector_type(2)));

// NT independent arguments at once, written stage by stage: the polynomial is one long dependent chain
// per argument, and with ~2 waves per SIMD a chain-by-chain order leaves the VALU waiting on its own
// results; stage-major order puts NT independent instructions between dependent ones.
template <int NT>
__device__ __forceinline__ void expf_nonpositive2(const v2fs (&x)[NT], v2fs (&e)[NT]) {
    v2fs n[NT], r[NT], p[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) n[k] = x[k] * 1.44269504088896341f;
#pragma unroll
    for (int k = 0; k < NT; ++k) n[k] = v2fs{rintf(n[k].x), rintf(n[k].y)};
    auto c2 = [](float c) { return v2fs{c, c}; };
#pragma unroll
    for (int k = 0; k < NT; ++k) r[k] = __builtin_elementwise_fma(n[k], c2(-0.693359375f), x[k]);
#pragma unroll
    for (int k = 0; k < NT; ++k) r[k] = __builtin_elementwise_fma(n[k], c2(2.12194440e-4f), r[k]);
#pragma unroll
    for (int k = 0; k < NT; ++k) p[k] = __builtin_elementwise_fma(c2(1.9875691500e-4f), r[k], c2(1.3981999507e-3f));
#pragma unroll
    for (int k = 0; k < NT; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], c2(8.3334519073e-3f));
#pragma unroll
    for (int k = 0; k < NT; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], c2(4.1665795894e-2f));
#pragma unroll
    for (int k = 0; k < NT; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], c2(1.6666665459e-1f));
#pragma unroll
    for (int k = 0; k < NT; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k], c2(5.0000001201e-1f));
#pragma unroll
    for (int k = 0; k < NT; ++k) p[k] = __builtin_elementwise_fma(p[k], r[k] * r[k], r[k]) + 1.0f;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        e[k] = v2fs{ldexpf(p[k].x, (int)n[k].x), ldexpf(p[k].y, (int)n[k].y)};
        e[k].x = x[k].x < -103.0f ? 0.0f : e[k].x;
        e[k].y = x[k].y < -103.0f ? 0.0f : e[k].y;
    }  // a NaN argument has propagated through the arithmetic to a NaN result, as in mmf_expf
}

template <bool INTERIOR>
__device__ __forceinline__ void bilateral_filter2_body(const float* __restrict__ depth, int cols, int rows, float maxD,
                                                       float* __restrict__ out, int x0, int y) {
    const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 555.556f;
    const int R = 6, D = R * 2 + 1;
    const float2 vv = *reinterpret_cast<const float2*>(depth + (size_t)y * cols + x0);
    const v2fs value = v2fs{vv.x, vv.y};
    const bool skip0 = vv.x > maxD || vv.x < 0.3f, skip1 = vv.y > maxD || vv.y < 0.3f;
    if (skip0 && skip1) {
        *reinterpret_cast<float2*>(out + (size_t)y * cols + x0) = make_float2(0.f, 0.f);
        return;
    }
    v2fs sum1 = v2fs{0.f, 0.f}, sum2 = v2fs{0.f, 0.f};
    // software pipelined over the 13 rows: the taps of row dy + 1 are loaded (from a clamped row, so
    // unconditionally) before the ~600 instructions of row dy, whose latency they then hide behind
    float taps[D + 1], next[D + 1];  // columns x0 - R .. x0 + R + 1
    {
        const float* rowp = depth + (size_t)min(max(y - R, 0), rows - 1) * cols;
#pragma unroll
        for (int k = 0; k < D + 1; ++k) next[k] = rowp[INTERIOR ? x0 + k - R : min(max(x0 + k - R, 0), cols - 1)];
    }
    for (int dy = -R; dy <= R; ++dy) {
        const int cy = y + dy;
#pragma unroll
        for (int k = 0; k < D + 1; ++k) taps[k] = next[k];
        if (dy < R) {
            const float* rowp = depth + (size_t)min(max(cy + 1, 0), rows - 1) * cols;
#pragma unroll
            for (int k = 0; k < D + 1; ++k) next[k] = rowp[INTERIOR ? x0 + k - R : min(max(x0 + k - R, 0), cols - 1)];
        }
        if (cy < 0 || cy >= rows) continue;  // wave-uniform except at the image border rows
        const float dy2 = (float)(dy * dy);
        v2fs arg[D], wgt[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const v2fs tmp = v2fs{taps[k], taps[k + 1]};
            const float space2 = (float)((k - R) * (k - R)) + dy2;
            const v2fs dv = value - tmp;
            const v2fs color2 = dv * dv;
            arg[k] = -(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half);
        }
        expf_nonpositive2<D>(arg, wgt);
#pragma unroll
        for (int k = 0; k < D; ++k) {  // the sums keep the shader's tap order
            const int cx = x0 + k - R;  // tap column of the first pixel; the second pixel's is cx + 1
            const v2fs tmp = v2fs{taps[k], taps[k + 1]};
            const v2fs weight = wgt[k];
            const v2fs tw = tmp * weight;
            const bool in0 = INTERIOR || (cx >= 0 && cx < cols), in1 = INTERIOR || (cx + 1 >= 0 && cx + 1 < cols);
            sum1.x = in0 ? sum1.x + tw.x : sum1.x;
            sum2.x = in0 ? sum2.x + weight.x : sum2.x;
            sum1.y = in1 ? sum1.y + tw.y : sum1.y;
            sum2.y = in1 ? sum2.y + weight.y : sum2.y;
        }
    }
    *reinterpret_cast<float2*>(out + (size_t)y * cols + x0) =
        make_float2(skip0 ? 0.f : sum1.x / sum2.x, skip1 ? 0.f : sum1.y / sum2.y);
}

__global__ __launch_bounds__(256) void bilateral_filter2_kernel(const float* __restrict__ depth, int cols, int rows,
                                                                float maxD, float* __restrict__ out) {
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 2, y = blockIdx.y * 4 + threadIdx.y;  // cols is even
    if (x0 >= cols || y >= rows) return;
    const int bx0 = blockIdx.x * 128;  // the workgroup's column range decides, so the choice is wave uniform
    if (bx0 - 6 >= 0 && bx0 + 127 + 7 < cols)
        bilateral_filter2_body<true>(depth, cols, rows, maxD, out, x0, y);
    else
        bilateral_filter2_body<false>(depth, cols, rows, maxD, out, x0, y);
}

// ---- exclusive scan of uint32 flags (3 launches; 1024 elements per workgroup) -------------------
constexpr int kScanBlock = 256, kScanPer = 4, kScanTile = kScanBlock * kScanPer;

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* lds, unsigned& total) {
    // inclusive scan over the 256 threads through LDS (Hillis-Steele), returns the exclusive value
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {
        const unsigned add = t >= off ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    total = lds[kScanBlock - 1];
    const unsigned incl = lds[t];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_block_sums_kernel(const unsigned* __restrict__ flags, unsigned n,
                                                                     unsigned* __restrict__ block_sums) {
    __shared__ unsigned lds[kScanBlock];
    const unsigned base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
    unsigned s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k)
        if (base + k < n) s += flags[base + k];
    unsigned total;
    block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// one workgroup: exclusive scan of up to 4096 block sums in place; writes the grand total
__global__ __launch_bounds__(kScanBlock) void scan_top_kernel(unsigned* __restrict__ block_sums, unsigned nblocks,
                                                              unsigned* __restrict__ total_out) {
    __shared__ unsigned lds[kScanBlock];
    unsigned carry = 0;
    for (unsigned base = 0; base < nblocks; base += kScanBlock) {
        const unsigned i = base + threadIdx.x;
        const unsigned v = i < nblocks ? block_sums[i] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan(v, lds, total);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(const unsigned* __restrict__ flags, unsigned n,
                                                                const unsigned* __restrict__ block_offsets,
                                                                unsigned* __restrict__ prefix) {
    __shared__ unsigned lds[kScanBlock];
    const unsigned base = blockIdx.x * kScanTile + threadIdx.x * kScanPer;
    unsigned f[kScanPer], s = 0;
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        f[k] = base + k < n ? flags[base + k] : 0u;
        s += f[k];
    }
    unsigned total;
    unsigned ex = block_exclusive_scan(s, lds, total) + block_offsets[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanPer; ++k) {
        if (base + k < n) prefix[base + k] = ex;
        ex += f[k];
    }
}

// ---- first frame: vertex_feedback.vert/.geom for one depth image -------------------------------
// draw index d = i * rows + j (column-major pixel order, FeedbackBuffer.cpp:41-49).  Writes the
// candidate surfel of every pixel into per-draw-index arrays plus its zVal > 0 flag.
__global__ __launch_bounds__(256) void feedback_kernel(const uint8_t* __restrict__ rgb, const float* __restrict__ depth,
                                                       int cols, int rows, Cam c, int time, float maxDepth,
                                                       SurfelSoA cand, unsigned* __restrict__ flags) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= cols * rows) return;
    const int i = d / rows, j = d - i * rows;
    const float tx = uv_coord(i, cols), ty = uv_coord(j, rows);
    const float x = tx * cols, y = ty * rows;
    const v3 p = get_vertex(depth, cols, rows, tx, ty, x, y, c);
    const v3 nl = get_normal(depth, cols, rows, p, tx, ty, x, y, c);
    const bool ok = !(p.z <= 0 || p.z > maxDepth);
    flags[d] = ok ? 1u : 0u;
    if (!ok) return;
    const uint8_t* px = rgb + (size_t)(texel(ty, rows) * cols + texel(tx, cols)) * 3;
    cand.pos[d] = make_float4(p.x, p.y, p.z, confidence(x, y, c.cx, c.cy, 1.0f));
    cand.col[d] = make_float4(encode_color(px[0] / 255.0f, px[1] / 255.0f, px[2] / 255.0f), 0.f, 1.0f, (float)time);
    cand.nrm[d] = make_float4(nl.x, nl.y, nl.z, get_radius(p.z, nl.z, c.ifx, c.ify));
}

// Model::initialise + init_unstable.vert: position/colour from the k-th valid RAW pixel, normal +
// radius from the k-th valid FILTERED pixel (the reference zips the two compacted buffers).
__global__ __launch_bounds__(256) void init_scatter_kernel(int n, SurfelSoA raw, const unsigned* __restrict__ raw_flags,
                                                           const unsigned* __restrict__ raw_prefix, SurfelSoA fil,
                                                           const unsigned* __restrict__ fil_flags,
                                                           const unsigned* __restrict__ fil_prefix, SurfelSoA dst,
                                                           unsigned capacity) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= n) return;
    // the model's vertex buffer holds `capacity` surfels (Model::MAX_VERTICES, Model.cpp:119-126): like
    // GL transform feedback into a full buffer, primitives beyond it are dropped (a 1280x960 first frame
    // has more valid pixels than that)
    if (raw_flags[d]) {
        const unsigned k = raw_prefix[d];
        if (k < capacity) {
            dst.pos[k] = raw.pos[d];
            float4 col = raw.col[d];
            col.y = 0;
            col.z = 1;
            dst.col[k] = col;
        }
    }
    if (fil_flags[d] && fil_prefix[d] < capacity) dst.nrm[fil_prefix[d]] = fil.nrm[d];
}

// ---- index map --------------------------------------------------------------------------------
// The projection passes normally get the inverse model pose as a launch argument, computed on the host from the pose
// the tracker returned.  The orchestrator enqueues the first projections of a frame BEFORE that pose has reached the
// host (they then run while the host wakes up and prepares the fusion passes): the tracker's last step leaves the same
// 16 floats -- same expression, same rounding -- in the odometry state, and the kernels read them from there.
template <typename Args>
__device__ __forceinline__ Args with_device_pose(const Args& in) {
    Args a = in;
    if (in.t_inv_dev) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a.t_inv.m[k] = in.t_inv_dev[k];
    }
    return a;
}

// A pass enqueued before the tracked pose has reached the host also carries `abort_dev`: a word of the odometry's device state
// that is non-zero when the tracking result it would read is not valid (OdomState::gn_fault: the one-launch chain gave up and
// the host re-runs the frame's tracking on the two-launch chain).  Such a pass then does NOTHING -- the host, which learns the
// same word with the pose, takes its own bookkeeping back and enqueues the pass again behind the new result.
#define MMF_SPECULATION_GUARD(args)                              \
    do {                                                         \
        if ((args).abort_dev != nullptr && *(args).abort_dev != 0) return; \
    } while (0)

struct IndexArgs {
    Mat4 t_inv;
    const float* t_inv_dev;  // non-null: the 16 floats to use instead of t_inv (see with_device_pose)
    const int* abort_dev;
    Cam c;
    int cols, rows;
    float maxDepth;
    int time, timeDelta;
};

// index.vert: surfel `id` (position p, timestamp ts) into the key image; true: it wrote, at pixel (px, py)
__device__ __forceinline__ bool index_map_project_xy(const IndexArgs& a, int id, float4 p, float ts,
                                                     unsigned long long* __restrict__ keys, int& px, int& py) {
    const v3 h = m4point(a.t_inv, V3(p.x, p.y, p.z));
    if (h.z > a.maxDepth || h.z < 0 || (float)a.time - ts > (float)a.timeDelta) return false;
    const float xn = ((((a.c.fx * h.x) / h.z) + a.c.cx) - (a.cols * 0.5f)) / (a.cols * 0.5f);
    const float yn = ((((a.c.fy * h.y) / h.z) + a.c.cy) - (a.rows * 0.5f)) / (a.rows * 0.5f);
    const float zn = h.z / a.maxDepth;
    if (!(xn >= -1 && xn <= 1 && yn >= -1 && yn <= 1 && zn >= -1 && zn <= 1)) return false;
    const float xw = (xn + 1.0f) * (a.cols * 0.5f), yw = (yn + 1.0f) * (a.rows * 0.5f);
    px = (int)floorf(xw), py = (int)floorf(yw);
    if (px < 0 || py < 0 || px >= a.cols || py >= a.rows) return false;
    const unsigned long long k = ((unsigned long long)depth24(0.5f * zn + 0.5f) << 32) | (unsigned)id;
    // the index map and its key image are stored TRANSPOSED (pixel (x, y) at x * rows + y): surfels are
    // kept in the reference's draw order, which is column-major over the image, so neighbouring lanes
    // hit neighbouring addresses here, in index_resolve_kernel's surfel gathers, and in the window
    // look-ups of fuse_data_kernel / clean_flag_kernel (row-major storage made each of those a
    // one-cache-line-per-lane access: 47 us for clean_flag_kernel at 640x480)
    atomicMin(&keys[px * a.rows + py], k);
    return true;
}
__device__ __forceinline__ void index_map_project(const IndexArgs& a, int id, float4 p, float ts,
                                                  unsigned long long* __restrict__ keys) {
    int px, py;
    (void)index_map_project_xy(a, id, p, ts, keys, px, py);
}

__device__ __forceinline__ void index_map_kernel_body(SurfelSoA s, int count, IndexArgs a_in,
                                                        unsigned long long* __restrict__ keys, FrameRider rider, const unsigned bx_, [[maybe_unused]] const unsigned gx_) {
    MMF_MODEL_STREAM_PRIORITY();
    if (rider.st && bx_ == 0) {  // the launch's one extra workgroup (frame_rider.hpp): dispatched first
        frame_rider_run<1u>(rider);
        return;
    }
    MMF_SPECULATION_GUARD(a_in);
    const IndexArgs a = with_device_pose(a_in);
    const int id = (int)(bx_ - (rider.st ? 1u : 0u)) * 256 + threadIdx.x;
    if (id >= count) return;
    index_map_project(a, id, s.pos[id], s.col[id].w, keys);
}
__global__ __launch_bounds__(256) void index_map_kernel(SurfelSoA s, int count, IndexArgs a_in,
                                                        unsigned long long* __restrict__ keys, FrameRider rider) {
    index_map_kernel_body(s, count, a_in, keys, rider, blockIdx.x, gridDim.x);
}

// texel i of the (transposed) index-map images from its key; the key goes back empty
__device__ __forceinline__ void index_resolve_texel(int i, const SurfelSoA& s, const IndexArgs& a, unsigned long long* __restrict__ keys,
                                                    unsigned* __restrict__ index, float4* __restrict__ vertConf, float4* __restrict__ colorTime,
                                                    float4* __restrict__ normRad) {
    const unsigned long long k = keys[i];
    if (k != kEmptyKey) keys[i] = kEmptyKey;
    if (k == kEmptyKey) {
        index[i] = 0;
        vertConf[i] = colorTime[i] = normRad[i] = make_float4(0, 0, 0, 0);
        return;
    }
    const unsigned id = (unsigned)k;
    const float4 p = s.pos[id], n = s.nrm[id];
    const v3 h = m4point(a.t_inv, V3(p.x, p.y, p.z));
    const v3 nn = v3normalize(m4dir(a.t_inv, V3(n.x, n.y, n.z)));
    index[i] = id;
    vertConf[i] = make_float4(h.x, h.y, h.z, p.w);
    colorTime[i] = s.col[id];
    normRad[i] = make_float4(nn.x, nn.y, nn.z, n.w);
}

// linear over the transposed images (it never needs a pixel's coordinates)
// Every resolve kernel hands the key image back EMPTY (it is the only reader of a texel's key), so the
// rasterising passes need no clearing launch in front of them.
__device__ __forceinline__ void index_resolve_kernel_body(SurfelSoA s, IndexArgs a_in,
                                                            unsigned long long* __restrict__ keys,
                                                            unsigned* __restrict__ index, float4* __restrict__ vertConf,
                                                            float4* __restrict__ colorTime, float4* __restrict__ normRad,
                                                            FrameRider rider, const unsigned bx_, [[maybe_unused]] const unsigned gx_) {
    MMF_MODEL_STREAM_PRIORITY();
    if (rider.st && bx_ == 0) {  // the launch's one extra workgroup (frame_rider.hpp): dispatched first
        frame_rider_run<2u>(rider);
        return;
    }
    MMF_SPECULATION_GUARD(a_in);
    const IndexArgs a = with_device_pose(a_in);
    const int i = (int)(bx_ - (rider.st ? 1u : 0u)) * 256 + threadIdx.x;
    if (i >= a.cols * a.rows) return;
    index_resolve_texel(i, s, a, keys, index, vertConf, colorTime, normRad);
}
__global__ __launch_bounds__(256) void index_resolve_kernel(SurfelSoA s, IndexArgs a_in,
                                                            unsigned long long* __restrict__ keys,
                                                            unsigned* __restrict__ index, float4* __restrict__ vertConf,
                                                            float4* __restrict__ colorTime, float4* __restrict__ normRad,
                                                            FrameRider rider) {
    index_resolve_kernel_body(s, a_in, keys, index, vertConf, colorTime, normRad, rider, blockIdx.x, gridDim.x);
}

// transposed (x * rows + y) -> row-major copy of an index-map image, for the texture getters
template <typename T>
__global__ __launch_bounds__(256) void untranspose_kernel(const T* __restrict__ src, int cols, int rows, T* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cols * rows) return;
    const int y = i / cols, x = i - y * cols;
    dst[i] = src[(size_t)x * rows + y];
}

#ifdef MMF_SPLAT_COUNT  // diagnostic builds (tools/mature_splat_probe.py): fragments considered / evaluated / drawn
__device__ unsigned long long g_splat_dbg[4];
#define MMF_SPLAT_TALLY(i, cond)                                                                                  \
    do {                                                                                                          \
        const unsigned long long b_ = __ballot(cond);                                                              \
        if (b_ && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(b_)) atomicAdd(&g_splat_dbg[i], (unsigned long long)__popcll(b_)); \
    } while (0)
#else
#define MMF_SPLAT_TALLY(i, cond) \
    do {                         \
    } while (0)
#endif
// ---- splat prediction ----------------------------------------------------------------------------
struct SplatArgs {
    Mat4 t_inv;
    const float* t_inv_dev;  // non-null: the 16 floats to use instead of t_inv (see with_device_pose)
    const int* abort_dev;
    Cam c;
    int cols, rows;
    float maxDepth, confThreshold;
    int time, maxTime, timeDelta;
    // != 0: a DEEP store (two surfels per pixel and more): the rasterising pass looks at the key image before it evaluates a
    // fragment (splat_kernel<true>)
    int early_z;
    // the viewing ray of every pixel centre, normalised (combo_splat.frag:41-42), transposed like the key image: it depends on
    // the camera alone, so the model keeps it as a table (splat_ray_kernel) instead of two divisions, a square root and a
    // third division per fragment
    const float4* rays;
};

struct SplatFrag {  // per-surfel quantities shared by the rasterising and the resolving pass
    v3 h, nrm;
    float rad;
    int x0, x1, y0, y1;
    int cpx, cpy;    // the pixel the sprite's centre falls into (inside [x0, x1] x [y0, y1])
    unsigned dmin;   // no fragment of this sprite has a smaller 24-bit depth
    float hn;        // dot(h, nrm)
    bool ok;
};

// SIZE: also the sprite's bounding box (splat.vert:71-86: eight more divisions, two normalisations) -- what the rasterising
// pass needs and the resolve pass and the bound's pre-pass do not; xw / yw: the window coordinates of the sprite's centre
template <bool SIZE = true>
__device__ __forceinline__ SplatFrag splat_setup(const float4 p, const float4 col, const float4 n, const SplatArgs& a, float* xw_out = nullptr,
                                                 float* yw_out = nullptr) {
    SplatFrag f;
    f.ok = false;
    f.h = m4point(a.t_inv, V3(p.x, p.y, p.z));
    const v3 h = f.h;
    if (h.z > a.maxDepth || h.z < 0 || p.w < a.confThreshold || (float)a.time - col.w > (float)a.timeDelta ||
        col.w > (float)a.maxTime)
        return f;
    const float fx = a.c.fx, fy = a.c.fy, cx = a.c.cx, cy = a.c.cy;
    const float xn = ((((fx * h.x) / h.z) + cx) - (a.cols * 0.5f)) / (a.cols * 0.5f);
    const float yn = ((((fy * h.y) / h.z) + cy) - (a.rows * 0.5f)) / (a.rows * 0.5f);
    const float zn = h.z / a.maxDepth;
    if (!(xn >= -1 && xn <= 1 && yn >= -1 && yn <= 1 && zn >= -1 && zn <= 1)) return f;
    f.nrm = v3normalize(m4dir(a.t_inv, V3(n.x, n.y, n.z)));
    f.rad = n.w;
    const v3 nrm = f.nrm;
    if (!SIZE) {
        f.hn = v3dot(h, nrm);
        f.x0 = f.y0 = 0, f.x1 = a.cols - 1, f.y1 = a.rows - 1, f.cpx = f.cpy = 0, f.dmin = 0u;
        if (xw_out) *xw_out = (xn + 1.0f) * (a.cols * 0.5f), *yw_out = (yn + 1.0f) * (a.rows * 0.5f);
        f.ok = true;
        return f;
    }
    const v3 x1 = v3scale(v3scale(v3normalize(V3((nrm.y - nrm.z), -nrm.x, nrm.x)), f.rad), 1.41421356f);
    const v3 y1 = v3cross(nrm, x1);
    const v3 q1 = v3add(h, x1), q2 = v3add(h, y1), q3 = v3sub(h, y1), q4 = v3sub(h, x1);
    const float p1x = ((fx * q1.x) / q1.z) + cx, p2x = ((fx * q2.x) / q2.z) + cx;
    const float p3x = ((fx * q3.x) / q3.z) + cx, p4x = ((fx * q4.x) / q4.z) + cx;
    const float p1y = ((fy * q1.y) / q1.z) + cy, p2y = ((fy * q2.y) / q2.z) + cy;
    const float p3y = ((fy * q3.y) / q3.z) + cy, p4y = ((fy * q4.y) / q4.z) + cy;
    const float xmin = fminf(p1x, fminf(p2x, fminf(p3x, p4x))), xmax = fmaxf(p1x, fmaxf(p2x, fmaxf(p3x, p4x)));
    const float ymin = fminf(p1y, fminf(p2y, fminf(p3y, p4y))), ymax = fmaxf(p1y, fmaxf(p2y, fmaxf(p3y, p4y)));
    float size = fmaxf(0.f, fmaxf(fabsf(xmax - xmin), fabsf(ymax - ymin)));
    if (!(size >= 1.0f)) size = 1.0f;
    const float xw = (xn + 1.0f) * (a.cols * 0.5f), yw = (yn + 1.0f) * (a.rows * 0.5f), hs = size * 0.5f;
    int x0 = (int)ceilf(xw - hs - 0.5f), x1i = (int)ceilf(xw + hs - 0.5f) - 1;
    int y0 = (int)ceilf(yw - hs - 0.5f), y1i = (int)ceilf(yw + hs - 0.5f) - 1;
    f.x0 = x0 < 0 ? 0 : x0, f.y0 = y0 < 0 ? 0 : y0;
    f.x1 = x1i > a.cols - 1 ? a.cols - 1 : x1i, f.y1 = y1i > a.rows - 1 ? a.rows - 1 : y1i;
    f.cpx = min(max((int)floorf(xw), f.x0), f.x1), f.cpy = min(max((int)floorf(yw), f.y0), f.y1);
    // a fragment that is not discarded lies within `rad` of h (combo_splat.frag:52), so its z is at least h.z - rad; the
    // margin covers the rounding of that bound and of the fragment's own arithmetic, depth24 is monotonic
    const float zlo = (h.z - f.rad) - (fabsf(h.z) + f.rad) * 4e-6f - 1e-7f;
    f.dmin = depth24((zlo / (2 * a.maxDepth)) + 0.5f);
    // ... unless its z is not a number: h.n / l.n with a NaN normal (a zero normal normalised) or 0 / 0 passes the radius test
    // (a comparison with NaN is false) and depth24(NaN) = 0 wins every depth test, as in the plain pass
    f.hn = v3dot(h, nrm);
    if (!(fabsf(f.hn) > 0.f)) f.dmin = 0u;
    f.ok = true;
    return f;
}

// one fragment of combo_splat.frag in three steps: the pixel's viewing ray (a property of the camera: SplatArgs::rays holds
// it), the depth of the ray's intersection with the disc's plane, the disc test.  splat_fragment = all three: returns false
// when discarded; `z` = corrected_pos.z
__device__ __forceinline__ v3 splat_ray(const Cam& c, int px, int py) {
    const float fcx = px + 0.5f, fcy = py + 0.5f;
    return v3normalize(V3((fcx - c.cx) / c.fx, (fcy - c.cy) / c.fy, 1.0f));
}
__device__ __forceinline__ float splat_plane_q(const SplatFrag& f, v3 l) { return f.hn / v3dot(l, f.nrm); }  // corrected = l * q
__device__ __forceinline__ unsigned splat_depth24(float z, float maxDepth) { return depth24((z / (2 * maxDepth)) + 0.5f); }
__device__ __forceinline__ bool splat_in_disc(const SplatFrag& f, v3 l, float q) {
    const v3 diff = v3sub(v3scale(l, q), f.h);
    return !(v3dot(diff, diff) > f.rad * f.rad);
}
__device__ __forceinline__ bool splat_fragment(const SplatFrag& f, const SplatArgs& a, int px, int py, float& z,
                                               unsigned& d24) {
    const v3 l = splat_ray(a.c, px, py);
    const float q = splat_plane_q(f, l);
    if (!splat_in_disc(f, l, q)) return false;
    z = l.z * q;
    d24 = splat_depth24(z, a.maxDepth);
    return true;
}
__global__ __launch_bounds__(256) void splat_ray_kernel(Cam c, int cols, int rows, float4* __restrict__ rays) {
    const int i = blockIdx.x * 256 + threadIdx.x;  // transposed: i = px * rows + py
    if (i >= cols * rows) return;
    const int px = i / rows, py = i - px * rows;
    const v3 l = splat_ray(c, px, py);
    rays[i] = make_float4(l.x, l.y, l.z, 0.f);
}

// The depth test of a DEEP store (a room seen from many sides: two surfels per pixel and more, four-fold overdraw): most
// fragments lose it.  splat_kernel<true> reads the pixel's key as its cache sees it, in the round trip of the rays, skips a
// fragment when even the nearest point of its disc (h.z - rad) lies behind that key's depth, and skips the atomic when the
// fragment itself does.  A stale look is a LARGER key (keys only ever fall during the pass), i.e. a fragment drawn that an
// up-to-date look would have skipped: the key image -- and with it all four images -- keeps its bits.  740 k stable surfels
// at 640x480 (18 M fragments): combinedPredict 105 -> 81 us (a pre-pass that wrote every surfel's centre fragment into a
// per-pixel bound first, this round's first form: 87 us); the headline loop's 246 k surfels would LOSE 2.5 us to the extra
// load per pixel slot, so the host asks for it from two surfels per pixel on.
// z / (2 maxDepth) for the depth key, without the division: with y = RN(1 / b), q = RN(z y), r = z - b q (one fma, exact),
// RN(q + r y) IS the correctly rounded quotient (Markstein's theorem; the division the compiler emits ends in these very
// steps, after refining a reciprocal that here is a constant of the launch).  Where the theorem does not reach -- quotients
// in the denormal range -- the + 0.5 of depth24's argument absorbs the difference (checked over all 2^32 inputs for
// b = 1.4, 6, 9, 10, 24.6, 40 on the CPU: no key differs; on the device for 39 cut-offs between 1e-3 and 1e6:
// tests/test_gpu_surfel.py); an infinite or overflowing quotient stays what the first product made of it.
struct SplatDepthScale {
    float b, y;
};
__device__ __forceinline__ SplatDepthScale splat_depth_scale(float maxDepth) { return SplatDepthScale{2 * maxDepth, 1.0f / (2 * maxDepth)}; }
__device__ __forceinline__ unsigned splat_depth24_fast(float z, const SplatDepthScale& d) {
    const float q0 = z * d.y;
    const float r = __builtin_fmaf(-d.b, q0, z);
    float q = __builtin_fmaf(r, d.y, q0);
    if (!(fabsf(q0) <= 3.4028234e38f)) q = q0;  // z infinite or not a number, or a quotient that overflows: inf - inf above
    return depth24(q + 0.5f);
}

// test hook (mmf_debug_depth_keys): both forms of the depth key for n depths
__global__ __launch_bounds__(256) void depth_key_probe_kernel(const float* __restrict__ z, int n, float maxDepth, unsigned* __restrict__ fast,
                                                              unsigned* __restrict__ divided) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    fast[i] = splat_depth24_fast(z[i], splat_depth_scale(maxDepth));
    divided[i] = splat_depth24(z[i], maxDepth);
}

// The rasterising pass.  `count_dev` (optional): the exact surfel count where the previous clean pass left it on the device;
// `count` is then only the bound the grid was sized by.
//
// A launch is a FIXED number of waves that deal the surfels out among themselves, `per` (<= 64) at a time per wave (round 3:
// one thread per surfel of the BOUND left an object model's 4 000 close-up sprites in 16 of 1 215 workgroups).  Inside a wave
// the work is dealt out again, because a sprite is anything from 1 to ~100 pixels on a side: every sprite row is cut into
// SEGMENTS of kSplatSeg pixels, the wave's segments are laid end to end (inclusive scan), lane l takes segments l, l + 64, ...,
// finds the segment's surfel by a binary search in the scan and reads the eight floats a fragment needs from LDS.
// Round 4 (this form).  Compiling parts of the previous form out (a lane walked a whole sprite row in steps of eight pixels)
// on a 305 k-surfel store: 57 us, of which set-up + scan + search 14, the atomics 6, the ray look-ups 9 and the rest --
// 28 us -- the fragment loop itself: ~80 instructions per pixel slot of which 56 % held a fragment (sprites are 4-5 pixels
// wide), 116 registers = four waves per SIMD, and a wait for the NEXT ray that also waited for the previous slot's atomic
// (loads and atomics share one counter here).  Now: four-pixel segments (78 % of the slots hold a fragment, no lane
// loops), ~50 instructions per slot (the depth key's division by 2 maxDepth is three multiply-adds, splat_depth24_fast; the
// row's constants stay in registers), all four fragments computed before the first atomic, 77-89 registers = five waves.
// combinedPredict 76 -> 60 us on that store (60 -> 52 on the headline loop's 246 k); the counters say what is left is latency, not arithmetic: 6.7 M vector instructions = 11 us of
// the SIMDs' time, waves waiting on memory 54 % of their life (the store's loads, then one ray round trip per 64 segments).
constexpr int kSplatSeg = 4;
struct SplatRowLds {  // per surfel of the wave's pass, structure of arrays: lanes read the entries of different surfels
    float nx[256], ny[256], nz[256], hn[256], hx[256], hy[256], hz[256], rad[256];
    unsigned x01[256];    // x0 | x1 << 16
    unsigned y0n[256];    // y0 | segments per row << 16
    unsigned magic[256];  // floor(2^32 / segments per row) + 1, or 0 for one segment per row: row = umulhi(local, magic)
    unsigned dmin[256];
    int seg_end[256];     // inclusive scan of rows x segments per row inside each wave
};
// (pass_rect.hpp) all 64 lanes of a wave: lanes with `valid` carry a box [x0, x1] x [y0, y1]; one set of generation-tagged
// atomics (extent.hpp's words) per wave that has one
__device__ __forceinline__ int wave_min_all(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max_all(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ void box_note_wave(unsigned long long* w, unsigned gen, bool valid, int x0, int x1, int y0, int y1) {
    const int a = wave_min_all(valid ? x0 : 0x7FFF), b = wave_max_all(valid ? x1 : -1);
    const int c = wave_min_all(valid ? y0 : 0x7FFF), d = wave_max_all(valid ? y1 : -1);
    if ((threadIdx.x & 63u) != 0u || b < a || d < c) return;
    const unsigned long long g = (unsigned long long)gen << 32;
    // look first: the words only ever rise, so a value that is there already needs no atomic -- after the first few waves of a
    // launch hardly any wave still widens the box (a rasterising pass of 4 000 one-surfel waves: 16 000 atomics on four
    // addresses, 300 us, without the look)
    const unsigned long long v0 = g | (unsigned long long)(0xFFFF - a), v1 = g | (unsigned long long)b, v2 = g | (unsigned long long)(0xFFFF - c),
                             v3w = g | (unsigned long long)d;
    if (__hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v0) atomicMax(&w[0], v0);
    if (__hip_atomic_load(&w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v1) atomicMax(&w[1], v1);
    if (__hip_atomic_load(&w[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v2) atomicMax(&w[2], v2);
    if (__hip_atomic_load(&w[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v3w) atomicMax(&w[3], v3w);
}
// BOX: the pass also notes the box of its sprites (what it may write of the key image) into key_box, generation kgen
template <bool EARLYZ, bool BOX = false>
__device__ __forceinline__ void splat_kernel_body(SurfelSoA s, int count, SplatArgs a_in,
                                                    unsigned long long* __restrict__ keys,
                                                    const unsigned* __restrict__ count_dev, const unsigned bx_, [[maybe_unused]] const unsigned gx_,
                                                    [[maybe_unused]] unsigned long long* key_box = nullptr, [[maybe_unused]] unsigned kgen = 0u) {
    MMF_MODEL_STREAM_PRIORITY();
    MMF_SPECULATION_GUARD(a_in);
    int bx0 = 0x7FFF, bx1 = -1, by0 = 0x7FFF, by1 = -1;  // (BOX) this lane's sprites
    const SplatArgs a = with_device_pose(a_in);
    __shared__ SplatRowLds L;
    const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
    if (count_dev != nullptr) count = min((unsigned)count, *count_dev);
    const int nwaves = (int)gx_ * 4, wave_id = (int)bx_ * 4 + (int)(threadIdx.x >> 6);
    // consecutive surfels per wave: neighbours in the store are neighbours on screen, and their rays and keys share cache
    // lines (dealing runs of four surfels round-robin over the waves instead, to even out the sprite sizes: 52 -> 56 us)
    const int per = min(64, max(1, (count + nwaves - 1) / nwaves));  // surfels of one wave's pass
    const SplatDepthScale ds = splat_depth_scale(a.maxDepth);
    for (int base = wave_id * per; base < count; base += nwaves * per) {  // wave uniform
        const int id = base + lane;
        SplatFrag f;
        f.ok = false;
        if (lane < per && id < count) {
            // all three loads before the first cull (the culls would otherwise put the three round trips one after the other)
            const float4 p = s.pos[id], col = s.col[id], n = s.nrm[id];
            asm volatile("" ::"v"(col.w), "v"(n.w));
            f = splat_setup(p, col, n, a);
        }
        const int h = f.ok ? max(f.y1 - f.y0 + 1, 0) : 0, w = f.ok ? max(f.x1 - f.x0 + 1, 0) : 0;
        if (BOX && h > 0 && w > 0) bx0 = min(bx0, f.x0), bx1 = max(bx1, f.x1), by0 = min(by0, f.y0), by1 = max(by1, f.y1);
        const int nseg = (w + kSplatSeg - 1) / kSplatSeg;
        int scan = h * nseg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(scan, d);
            if (lane >= d) scan += up;
        }
        const int total = __shfl(scan, 63);
        if (total == 0) continue;  // wave uniform
        // one wave writes and reads its own 64 entries: no workgroup barrier needed, only the previous pass's reads and
        // these writes drained
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            const int t = threadIdx.x;
            L.nx[t] = f.nrm.x, L.ny[t] = f.nrm.y, L.nz[t] = f.nrm.z, L.hn[t] = f.hn;
            L.hx[t] = f.h.x, L.hy[t] = f.h.y, L.hz[t] = f.h.z, L.rad[t] = f.rad;
            L.x01[t] = (unsigned)f.x0 | ((unsigned)f.x1 << 16);
            L.y0n[t] = (unsigned)f.y0 | ((unsigned)nseg << 16);
            L.magic[t] = nseg > 1 ? 0xFFFFFFFFu / (unsigned)nseg + 1u : 0u;
            if (EARLYZ) L.dmin[t] = f.dmin;
            L.seg_end[t] = scan;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < total; t += 64) {
            int lo = 0, hi = 63;  // first surfel of the wave with seg_end > t
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const int mid = (lo + hi) >> 1;
                const bool right = L.seg_end[wbase + mid] <= t;
                lo = right ? mid + 1 : lo;
                hi = right ? hi : mid;
            }
            const int e = wbase + lo;
            const unsigned local = (unsigned)(t - (lo ? L.seg_end[e - 1] : 0));
            const unsigned x01 = L.x01[e], y0n = L.y0n[e], magic = L.magic[e];
            // local < 480 x 160 and segments per row <= 160: umulhi(local, floor(2^32 / n) + 1) = local / n exactly
            const unsigned row = magic ? __umulhi(local, magic) : local;
            const unsigned sx = local - row * (y0n >> 16);
            const int py = (int)(y0n & 0xFFFFu) + (int)row, pb = (int)(x01 & 0xFFFFu) + kSplatSeg * (int)sx, x1 = (int)(x01 >> 16);
            const unsigned sid = (unsigned)(base + lo);
            SplatFrag g;
            g.nrm = V3(L.nx[e], L.ny[e], L.nz[e]), g.hn = L.hn[e];
            g.h = V3(L.hx[e], L.hy[e], L.hz[e]), g.rad = L.rad[e];
            const unsigned dmin = EARLYZ ? L.dmin[e] : 0u;

            // the segment's viewing rays (and, for a deep store, its pixels' keys) in one round trip
            unsigned zbv[kSplatSeg];
            float4 ray[kSplatSeg];
#pragma unroll
            for (int k = 0; k < kSplatSeg; ++k) {
                const size_t at = (size_t)min(pb + k, x1) * a.rows + py;
                // (an atomic relaxed load: the compiler may neither hoist nor merge it across the atomicMin's of the same image)
                zbv[k] = EARLYZ ? (unsigned)(__hip_atomic_load(&keys[at], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) : 0xFFFFFFFFu;
                ray[k] = a.rays[at];
            }
            // all four fragments first, then their atomics: a wait for a ray that follows an atomic in program order would
            // wait for the atomic's round trip as well (one counter for both on this chip)
            unsigned d24v[kSplatSeg];
            bool draw[kSplatSeg];
#pragma unroll
            for (int k = 0; k < kSplatSeg; ++k) {
                // the depth (a dot product and a division) and the disc test
                const v3 l = V3(ray[k].x, ray[k].y, ray[k].z);
                const float q = splat_plane_q(g, l);
                d24v[k] = splat_depth24_fast(l.z * q, ds);
                draw[k] = pb + k <= x1 && !(EARLYZ && (dmin > zbv[k] || d24v[k] > zbv[k])) && splat_in_disc(g, l, q);
                MMF_SPLAT_TALLY(0, pb + k <= x1);
                MMF_SPLAT_TALLY(2, draw[k]);
            }
#pragma unroll
            for (int k = 0; k < kSplatSeg; ++k)  // the key image is stored TRANSPOSED (x * rows + y), like the index map's
                if (draw[k]) atomicMin(&keys[(size_t)(pb + k) * a.rows + py], ((unsigned long long)d24v[k] << 32) | sid);
        }
    }
    if (BOX) box_note_wave(key_box, kgen, bx1 >= bx0 && by1 >= by0, bx0, bx1, by0, by1);
}
// the same pass, noting the box of its sprites (an object model's prediction: the preparation that follows walks that box only)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void splat_box_kernel(SurfelSoA s, int count, SplatArgs a_in,
                                                    unsigned long long* __restrict__ keys, const unsigned* __restrict__ count_dev,
                                                    PassBoxes* boxes, unsigned kgen) {
    splat_kernel_body<false, true>(s, count, a_in, keys, count_dev, blockIdx.x, gridDim.x, boxes->key[kgen & 1u], kgen);
}
template <bool EARLYZ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void splat_kernel(SurfelSoA s, int count, SplatArgs a_in,
                                                    unsigned long long* __restrict__ keys,
                                                    const unsigned* __restrict__ count_dev) {
    splat_kernel_body<EARLYZ>(s, count, a_in, keys, count_dev, blockIdx.x, gridDim.x);
}


struct SplatTexel {
    uchar4 image;
    float4 vertexConf, normalRadius;
    unsigned short time;
};

// The resolve kernels work on 16 x 16 pixel tiles: the tile's keys are read from the transposed key image along y
// (128 contiguous bytes per tile column), handed back empty, and passed through LDS so that the thread of pixel
// (x, y) -- threads run along x, the images are row-major -- gets its key.  Returns false outside the image.
constexpr int kSplatTile = 16;
__device__ __forceinline__ bool splat_tile_key(unsigned long long* __restrict__ keys, int cols, int rows, int& px, int& py,
                                               unsigned long long& k, unsigned first_block = 0, unsigned block = ~0u) {
    __shared__ unsigned long long tile[kSplatTile][kSplatTile + 1];
    const int tiles_x = (cols + kSplatTile - 1) / kSplatTile;
    const int bid = (int)((block == ~0u ? blockIdx.x : block) - first_block);
    const int bx = bid % tiles_x, by = bid / tiles_x;
    const int t = threadIdx.x;
    {
        const int cx = t >> 4, cy = t & 15;  // loading: threads run along y
        const int x = bx * kSplatTile + cx, y = by * kSplatTile + cy;
        unsigned long long v = kEmptyKey;
        if (x < cols && y < rows) {
            v = keys[(size_t)x * rows + y];
            if (v != kEmptyKey) keys[(size_t)x * rows + y] = kEmptyKey;
        }
        tile[cx][cy] = v;
    }
    __syncthreads();
    const int ly = t >> 4, lx = t & 15;  // resolving: threads run along x
    px = bx * kSplatTile + lx, py = by * kSplatTile + ly;
    k = tile[lx][ly];
    return px < cols && py < rows;
}
__host__ __device__ inline unsigned splat_tile_grid(int cols, int rows) {
    return (unsigned)(((cols + kSplatTile - 1) / kSplatTile) * ((rows + kSplatTile - 1) / kSplatTile));
}

// requiresFillIn's thumbnail test (MultiMotionFusion.cpp:877-895: how many of the (cols / 20) x (rows / 20) samples of the
// predicted image have all three channels > 0) counted by the pass that WRITES the image, instead of a launch of its own on
// the model's stream: `thumb` = two counters; the resolve of generation g adds to thumb[g & 1] (one atomic per wave with
// hits) and zeroes thumb[(g + 1) & 1] for the next one.  The preparation jobs that choose between the prediction and the
// fill-in images read the count and apply the ratio themselves (prep_batch.hpp: PrepJob::sel_total).
// (last_block: is this the launch's -- in a batched launch the model's -- last workgroup)
__device__ __forceinline__ void thumbnail_count_px(int px, int py, int cols, int rows, uchar4 p, unsigned* __restrict__ thumb, int gen,
                                                   bool last_block) {
    if (thumb == nullptr) return;
    const int dc = cols / 20, dr = rows / 20;
    const int i = px * dc / cols, j = py * dr / rows;  // the one sample this pixel can be
    const bool sample = dc > 0 && dr > 0 && texel((i + 0.5f) / dc, cols) == px && texel((j + 0.5f) / dr, rows) == py;
    const unsigned long long hits = __ballot(sample && p.x > 0 && p.y > 0 && p.z > 0);
    if (hits != 0ull && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(hits)) atomicAdd(&thumb[gen & 1], (unsigned)__popcll(hits));
    if (last_block && threadIdx.x == 0) thumb[(gen + 1) & 1] = 0u;
}

// the splat images' texel i = (px, py) from the depth-test winner k
__device__ __forceinline__ SplatTexel splat_resolve_px(int i, unsigned long long k, const SurfelSoA& s, const SplatArgs& a) {
    SplatTexel t;
    if (k == kEmptyKey) {
        t.image = make_uchar4(0, 0, 0, 0);
        t.vertexConf = t.normalRadius = make_float4(0, 0, 0, 0);
        t.time = 0;
        return t;
    }
    const unsigned id = (unsigned)k;
    const int py = i / a.cols, px = i - py * a.cols;
    const float4 p = s.pos[id], col = s.col[id];
    const SplatFrag f = splat_setup<false>(p, col, s.nrm[id], a);
    float z;
    unsigned d24;
    splat_fragment(f, a, px, py, z, d24);  // the winner's own fragment: same arithmetic as in splat_kernel
    const v3 rgb = decode_color(col.x);
    t.image = make_uchar4((unsigned char)(int)roundf(rgb.x * 255.0f), (unsigned char)(int)roundf(rgb.y * 255.0f),
                          (unsigned char)(int)roundf(rgb.z * 255.0f), 255);
    const float fcx = px + 0.5f, fcy = py + 0.5f;
    t.vertexConf = make_float4((fcx - a.c.cx) * z * (1.f / a.c.fx), (fcy - a.c.cy) * z * (1.f / a.c.fy), z, p.w);
    t.normalRadius = make_float4(f.nrm.x, f.nrm.y, f.nrm.z, f.rad);
    t.time = (unsigned short)(unsigned)col.z;
    return t;
}

__device__ __forceinline__ void splat_resolve_kernel_body(SurfelSoA s, SplatArgs a_in,
                                                            unsigned long long* __restrict__ keys,
                                                            uchar4* __restrict__ image, float4* __restrict__ vertexConf,
                                                            float4* __restrict__ normalRadius,
                                                            unsigned short* __restrict__ time_out, unsigned* __restrict__ thumb,
                                                            int gen, const unsigned bx_, [[maybe_unused]] const unsigned gx_) {
    MMF_MODEL_STREAM_PRIORITY();
    MMF_SPECULATION_GUARD(a_in);
    const SplatArgs a = with_device_pose(a_in);
    int px, py;
    unsigned long long k;
    if (!splat_tile_key(keys, a.cols, a.rows, px, py, k, 0, bx_)) return;
    const int i = py * a.cols + px;
    const SplatTexel t = splat_resolve_px(i, k, s, a);
    image[i] = t.image;
    vertexConf[i] = t.vertexConf, normalRadius[i] = t.normalRadius;
    time_out[i] = t.time;
    thumbnail_count_px(px, py, a.cols, a.rows, t.image, thumb, gen, bx_ == gx_ - 1);
}
// the same resolve behind splat_box_kernel: one thread also keeps the sprites' box as the box the images are non-zero in
__global__ __launch_bounds__(256) void splat_resolve_keep_box_kernel(SurfelSoA s, SplatArgs a_in, unsigned long long* __restrict__ keys,
                                                                     uchar4* __restrict__ image, float4* __restrict__ vertexConf,
                                                                     float4* __restrict__ normalRadius, unsigned short* __restrict__ time_out,
                                                                     unsigned* __restrict__ thumb, int gen, PassBoxes* boxes, unsigned kgen,
                                                                     unsigned sgen) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && !(a_in.abort_dev != nullptr && *a_in.abort_dev != 0))
        box_to_ints(boxes->spl_nz[sgen & 1u], box_clip(extent_load(boxes->key[kgen & 1u], kgen), a_in.cols, a_in.rows));
    splat_resolve_kernel_body(s, a_in, keys, image, vertexConf, normalRadius, time_out, thumb, gen, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void splat_resolve_kernel(SurfelSoA s, SplatArgs a_in,
                                                            unsigned long long* __restrict__ keys,
                                                            uchar4* __restrict__ image, float4* __restrict__ vertexConf,
                                                            float4* __restrict__ normalRadius,
                                                            unsigned short* __restrict__ time_out, unsigned* __restrict__ thumb,
                                                            int gen) {
    splat_resolve_kernel_body(s, a_in, keys, image, vertexConf, normalRadius, time_out, thumb, gen, blockIdx.x, gridDim.x);
}

// depth_splat.frag (ModelProjection::synthesizeDepth): the winner's corrected_pos.z, 0 where cleared
__global__ __launch_bounds__(256) void splat_depth_resolve_kernel(SurfelSoA s, SplatArgs a_in,
                                                                  unsigned long long* __restrict__ keys,
                                                                  float* __restrict__ depth) {
    const SplatArgs a = with_device_pose(a_in);
    int px, py;
    unsigned long long k;
    if (!splat_tile_key(keys, a.cols, a.rows, px, py, k, 0)) return;
    const int i = py * a.cols + px;
    float z = 0.f;
    if (k != kEmptyKey) {
        const unsigned id = (unsigned)k;
        const SplatFrag f = splat_setup<false>(s.pos[id], s.col[id], s.nrm[id], a);
        unsigned d24;
        splat_fragment(f, a, px, py, z, d24);
    }
    depth[i] = z;
}

// ---- fusion: data association (data.vert) ----------------------------------------------------------
struct FuseArgs {
    Mat4 pose;
    const float* pose_dev;    // non-null: the model pose (16 floats) to use instead of `pose`, and
    const float* weight_dev;  // computeFusionWeight(1) to use instead of `weighting` / weight_mult -- for a fuse pass that
    float weight_mult;        // is enqueued before the tracked pose has reached the host (with_device_pose)
    const int* abort_dev;
    Cam c;  // ifx, ify = (float)(1.0 / fx) computed in double on the host (Model.cpp:920-921)
    int cols, rows;
    int time;
    float weighting;
    unsigned char maskID;
    float maxDepth;
    int count;  // surfels in the store
};

constexpr unsigned kNoWinner = 0xFFFFFFFFu;

// Per pixel in draw order d = i*rows + j: the new measurement goes to meas.{pos,col,nrm}[d];
// op[d] = 0 (nothing), 1 (merge; target in best[d]), 2 (new unstable).  For merges the first
// pixel in draw order owns the target surfel: atomicMin(winner[best], d).
__device__ __forceinline__ void fuse_data_thread(const int t, const uint8_t* __restrict__ rgb, const float* __restrict__ depth_raw,
                                                 const float* __restrict__ depth_fil, const uint8_t* __restrict__ mask,
                                                 const unsigned* __restrict__ index, const float4* __restrict__ vertConf,
                                                 const float4* __restrict__ normRad, const FuseArgs& a, SurfelSoA meas,
                                                 unsigned* __restrict__ new_flags, unsigned* __restrict__ winner);
__device__ __forceinline__ void fuse_data_kernel_body(const uint8_t* __restrict__ rgb, const float* __restrict__ depth_raw,
                                                        const float* __restrict__ depth_fil, const uint8_t* __restrict__ mask,
                                                        const unsigned* __restrict__ index, const float4* __restrict__ vertConf,
                                                        const float4* __restrict__ normRad, FuseArgs a_in, SurfelSoA meas,
                                                        unsigned* __restrict__ new_flags, unsigned* __restrict__ winner, const unsigned bx_, [[maybe_unused]] const unsigned gx_) {
    MMF_MODEL_STREAM_PRIORITY();
    MMF_SPECULATION_GUARD(a_in);
    FuseArgs a = a_in;
    if (a_in.pose_dev) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a.pose.m[k] = a_in.pose_dev[k];
        a.weighting = *a_in.weight_dev * a_in.weight_mult;  // (computeFusionWeight's last operation)
    }
    fuse_data_thread((int)(bx_ * 256 + threadIdx.x), rgb, depth_raw, depth_fil, mask, index, vertConf, normRad, a, meas, new_flags, winner);
}
// thread t of the pass (t < ceil(cols / 2) * ceil(rows / 2), column-major over the 2 x 2 blocks); `a`: the pose resolved
__device__ __forceinline__ void fuse_data_thread(const int t, const uint8_t* __restrict__ rgb, const float* __restrict__ depth_raw,
                                                 const float* __restrict__ depth_fil, const uint8_t* __restrict__ mask,
                                                 const unsigned* __restrict__ index, const float4* __restrict__ vertConf,
                                                 const float4* __restrict__ normRad, const FuseArgs& a, SurfelSoA meas,
                                                 unsigned* __restrict__ new_flags, unsigned* __restrict__ winner) {
    // One thread per 2 x 2 block of pixels: data.vert:116 keeps one pixel in four -- (int)x and (int)y both of the frame's
    // parity, i.e. one lattice point per block ((int)(uv_coord(i, n) * n) == i for every i: the error of that product is
    // ~1e-5) -- so a thread per PIXEL left every other wave empty and the rest half empty.  The thread clears its block's
    // four flags and carries on with the block's lattice pixel, in the pixel's own draw-order slot d: same results.
    const int cols = a.cols, rows = a.rows;
    const int hr = (rows + 1) / 2, hc = (cols + 1) / 2;
    if (t >= hc * hr) return;
    const int ia = t / hr, jb = t - ia * hr;  // column-major like the draw order
    const int tm = ((int)(float)a.time) % 2;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
            if (2 * ia + u < cols && 2 * jb + v < rows) new_flags[(2 * ia + u) * rows + 2 * jb + v] = 0u;
    const int i = 2 * ia + tm, j = 2 * jb + tm;
    if (i >= cols || j >= rows) return;
    const int d = i * rows + j;
    const Cam& c = a.c;
    const float tx = uv_coord(i, cols), ty = uv_coord(j, rows);
    const float x = tx * cols, y = ty * rows;
    // the quarter-rate test of data.vert:116 (true by construction; kept as the shader has it)
    if (!(((int)x) % 2 == tm && ((int)y) % 2 == tm)) return;
    const v3 vPosLocal = get_vertex(depth_raw, cols, rows, tx, ty, x, y, c);
    const int pxi = texel(tx, cols), pyi = texel(ty, rows);
    const float zl = depth_raw[pyi * cols + texel(tx - (1.0f / cols), cols)];
    const float zu = depth_raw[texel(ty - (1.0f / rows), rows) * cols + pxi];
    const float zr = depth_raw[pyi * cols + texel(tx + (1.0f / cols), cols)];
    const float zd = depth_raw[texel(ty + (1.0f / rows), rows) * cols + pxi];
    const bool neighbours = !(zl == 0) && !(zu == 0) && !(zr == 0) && !(zd == 0);
    if (!(((int)x) % 2 == tm && ((int)y) % 2 == tm && mask[pyi * cols + pxi] == a.maskID && neighbours &&
          vPosLocal.z > 0 && vPosLocal.z <= a.maxDepth))
        return;

    const v3 vPos = m4point(a.pose, vPosLocal);
    const v3 vPos_f = get_vertex(depth_fil, cols, rows, tx, ty, x, y, c);
    const uint8_t* px = rgb + (size_t)(pyi * cols + pxi) * 3;
    const v3 nl = get_normal(depth_fil, cols, rows, vPos_f, tx, ty, x, y, c);
    const v3 ng = m4dir(a.pose, nl);

    int operation = 0;
    unsigned best = 0;
    const float scale = 1.0f;
    const float indexXStep = (1.0f / (cols * scale)) * 0.5f, indexYStep = (1.0f / (rows * scale)) * 0.5f;
    float bestDist = 1000;
    const float windowMultiplier = 2;
    const float xl = (x - c.cx) * c.ifx, yl = (y - c.cy) * c.ify;
    const float lambda = sqrtf(xl * xl + yl * yl + 1);
    const v3 ray = V3(xl, yl, 1);
    // The shader walks the window with float loop counters (data.vert:138-163).  The counters are
    // reproduced exactly (same sequence of float additions) but only to find WHICH texels they visit:
    // the 4-5 half-pixel steps per axis touch the texels c0, c0+1, c0+2, so the window is read as <= 3x3
    // distinct texels -- all loads independent, one round trip -- and evaluated in the order of first
    // visit (x outer, y inner).  Revisiting a texel never changes the result (`dist < bestDist` is strict).
    int cx0, cy0;
    bool hitx[3] = {false, false, false}, hity[3] = {false, false, false};
    {
        const float endx = tx + (scale * indexXStep * windowMultiplier);
        float ii = tx - (scale * indexXStep * windowMultiplier);
        cx0 = texel(ii, cols);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (ii < endx) {
                const int t = texel(ii, cols) - cx0;
                hitx[0] |= t == 0, hitx[1] |= t == 1, hitx[2] |= t == 2;
            }
            ii += indexXStep;
        }
        const float endy = ty + (scale * indexYStep * windowMultiplier);
        float jj = ty - (scale * indexYStep * windowMultiplier);
        cy0 = texel(jj, rows);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (jj < endy) {
                const int t = texel(jj, rows) - cy0;
                hity[0] |= t == 0, hity[1] |= t == 1, hity[2] |= t == 2;
            }
            jj += indexYStep;
        }
    }
    unsigned cur[9];
    float4 vcs[9], nrs[9];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const size_t t = (size_t)min(cx0 + dx, cols - 1) * rows + min(cy0 + dy, rows - 1);  // transposed index map
            cur[dx * 3 + dy] = index[t];
            vcs[dx * 3 + dy] = vertConf[t];
            nrs[dx * 3 + dy] = normRad[t];
        }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const unsigned current = cur[dx * 3 + dy];
            if (hitx[dx] && hity[dy] && current > 0U) {
                const float4 vc = vcs[dx * 3 + dy];
                const float zdiff = (vc.z - vPosLocal.z);
                if (fabsf(zdiff * lambda) < 0.05f) {
                    const float dist = v3length(v3cross(ray, V3(vc.x, vc.y, vc.z)));
                    const float4 nr = nrs[dx * 3 + dy];
                    const v3 nrv = V3(nr.x, nr.y, nr.z);
                    const float cosang = v3dot(nrv, nl) / (v3length(nrv) * v3length(nl));
                    if (dist < bestDist && (fabsf(nr.z) < 0.75f || (cosang > 0.87758255f && cosang <= 1.0f))) {
                        operation = 1;
                        bestDist = dist;
                        best = current;
                    }
                }
            }
        }
    const float conf = confidence(x, y, c.cx, c.cy, a.weighting);
    meas.pos[d] = make_float4(vPos.x, vPos.y, vPos.z, conf);
    meas.col[d] = make_float4(encode_color(px[0] / 255.0f, px[1] / 255.0f, px[2] / 255.0f), 0.f, (float)a.time,
                              operation == 1 ? -1.f : -2.f);
    meas.nrm[d] = make_float4(ng.x, ng.y, ng.z, get_radius(vPos_f.z, nl.z, c.ifx, c.ify));
    if (operation == 1) {
        if (best < (unsigned)a.count) atomicMin(&winner[best], (unsigned)d);
    } else {
        new_flags[d] = 1u;
    }
}
__global__ __launch_bounds__(256) void fuse_data_kernel(const uint8_t* __restrict__ rgb, const float* __restrict__ depth_raw,
                                                        const float* __restrict__ depth_fil, const uint8_t* __restrict__ mask,
                                                        const unsigned* __restrict__ index, const float4* __restrict__ vertConf,
                                                        const float4* __restrict__ normRad, FuseArgs a_in, SurfelSoA meas,
                                                        unsigned* __restrict__ new_flags, unsigned* __restrict__ winner) {
    fuse_data_kernel_body(rgb, depth_raw, depth_fil, mask, index, vertConf, normRad, a_in, meas, new_flags, winner, xcd_block(blockIdx.x, gridDim.x), gridDim.x);
}

// update.vert:38-111, in place (each surfel only touches itself); resets winner[] for the next frame
// surfel k merged with its winning measurement w, in place
__device__ __forceinline__ void fuse_update_one(SurfelSoA s, int k, unsigned w, SurfelSoA meas, int time, float4& op, float4& oc) {
    const float4 np = meas.pos[w], nc = meas.col[w], nn = meas.nrm[w];
    float4 on = s.nrm[k];
    op = s.pos[k], oc = s.col[k];
    const float c_k = op.w, av = np.w;
    if (nn.w < (1.0f + 0.5f) * on.w) {
        op.x = ((c_k * op.x) + (av * np.x)) / (c_k + av);
        op.y = ((c_k * op.y) + (av * np.y)) / (c_k + av);
        op.z = ((c_k * op.z) + (av * np.z)) / (c_k + av);
        const v3 oldCol = decode_color(oc.x), newCol = decode_color(nc.x);
        const float ar = ((c_k * oldCol.x) + (av * newCol.x)) / (c_k + av);
        const float ag = ((c_k * oldCol.y) + (av * newCol.y)) / (c_k + av);
        const float ab = ((c_k * oldCol.z) + (av * newCol.z)) / (c_k + av);
        oc.x = encode_color(ar, ag, ab);
        oc.w = (float)time;
        const float n0 = ((c_k * on.x) + (av * nn.x)) / (c_k + av), n1 = ((c_k * on.y) + (av * nn.y)) / (c_k + av);
        const float n2 = ((c_k * on.z) + (av * nn.z)) / (c_k + av), n3 = ((c_k * on.w) + (av * nn.w)) / (c_k + av);
        const v3 nz = v3normalize(V3(n0, n1, n2));
        on = make_float4(nz.x, nz.y, nz.z, n3);
        op.w = c_k + av;
        s.pos[k] = op;
        s.col[k] = oc;
        s.nrm[k] = on;
    } else {
        op.w = c_k + av;
        oc.w = (float)time;
        s.pos[k] = op;
        s.col[k] = oc;
    }
}

__global__ __launch_bounds__(256) void fuse_update_kernel(SurfelSoA s, int count, SurfelSoA meas, int time,
                                                          unsigned* __restrict__ winner) {
    MMF_MODEL_STREAM_PRIORITY();
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const unsigned w = winner[k];
    if (w == kNoWinner) return;
    winner[k] = kNoWinner;
    float4 op, oc;
    fuse_update_one(s, k, w, meas, time, op, oc);
}

// fuse_update_kernel and the predictIndices that follows it (MultiMotionFusion.cpp:800-808) in one pass over the surfels: the
// projection takes the surfel from the registers the update left it in
__device__ __forceinline__ void fuse_update_index_kernel_body(SurfelSoA s, int count, SurfelSoA meas, int time,
                                                                unsigned* __restrict__ winner, IndexArgs a_in,
                                                                unsigned long long* __restrict__ keys, const unsigned bx_, [[maybe_unused]] const unsigned gx_) {
    MMF_MODEL_STREAM_PRIORITY();
    MMF_SPECULATION_GUARD(a_in);
    const IndexArgs a = with_device_pose(a_in);
    const int k = bx_ * 256 + threadIdx.x;
    if (k >= count) return;
    const unsigned w = winner[k];
    float4 op, oc;
    if (w == kNoWinner) {
        op = s.pos[k];
        oc.w = s.col[k].w;
    } else {
        winner[k] = kNoWinner;
        fuse_update_one(s, k, w, meas, time, op, oc);
    }
    index_map_project(a, k, op, oc.w, keys);
}
__global__ __launch_bounds__(256) void fuse_update_index_kernel(SurfelSoA s, int count, SurfelSoA meas, int time,
                                                                unsigned* __restrict__ winner, IndexArgs a_in,
                                                                unsigned long long* __restrict__ keys) {
    fuse_update_index_kernel_body(s, count, meas, time, winner, a_in, keys, xcd_block(blockIdx.x, gridDim.x), gridDim.x);
}

// ---- clean: copy_unstable.vert:53-150 ---------------------------------------------------------------
struct CleanArgs {
    Mat4 t_inv;
    const float* t_inv_dev;  // non-null: the 16 floats to use instead of t_inv (with_device_pose)
    const int* abort_dev;
    Cam c;
    int cols, rows;
    int time, timeDelta;
    float confThreshold, outlierCoeff;
    unsigned char maskID;
    int count;   // existing surfels
    int npix;    // cols*rows candidates follow (new unstable where new_flags is set)
};

// candidate e in [0, count + npix): existing surfel e, or pixel draw-index e - count.
// Returns keep(e) and writes the two fields the shader modifies (confidence, timestamp).
__device__ __forceinline__ unsigned clean_flag_one(int e, SurfelSoA s, SurfelSoA meas, const unsigned* __restrict__ new_flags,
                                                   const CleanArgs& a, const unsigned* __restrict__ index,
                                                   const float4* __restrict__ vertConf,
                                                   const float4* __restrict__ colorTime,
                                                   const float* __restrict__ depth_in, const uint8_t* __restrict__ mask,
                                                   float2* __restrict__ conf_time, int e_store = -1) {
    // (e_store: where the candidate's {confidence, timestamp} go when the caller numbers its candidates differently: pass_rect.hpp)
    float4 vpos, vcol, vnrm;
    if (e < a.count) {
        vpos = s.pos[e], vcol = s.col[e], vnrm = s.nrm[e];
    } else {
        const int d = e - a.count;
        if (!new_flags[d]) return 0u;
        vpos = meas.pos[d], vcol = meas.col[d], vnrm = meas.nrm[d];
    }
    const int cols = a.cols, rows = a.rows;
    int test = 1;
    const float scale = 1.0f;
    const v3 localPos = m4point(a.t_inv, V3(vpos.x, vpos.y, vpos.z));
    const float x = ((a.c.fx * localPos.x) / localPos.z) + a.c.cx, y = ((a.c.fy * localPos.y) / localPos.z) + a.c.cy;
    const v3 localNorm = v3normalize(m4dir(a.t_inv, V3(vnrm.x, vnrm.y, vnrm.z)));
    const float x_n = x / cols, y_n = y / rows;
    const float stepX = 1.0f / cols, stepY = 1.0f / rows;
    const float indexXStep = stepX * 0.5f / scale, indexYStep = stepY * 0.5f / scale;
    const float windowMultiplier = 2;
    int count = 0, zCount = 0, violationCount = 0;
    float avgViolation = 0;
    if ((float)a.time - vcol.w < (float)a.timeDelta && localPos.z > 0 && x > 0 && y > 0 && x < cols && y < rows) {
        // The shader walks both windows with float loop counters (copy_unstable.vert:86-128).  The
        // counters are reproduced exactly (same sequence of float additions), but sampling is
        // decoupled from them: the 4-5 half-pixel steps of the index window only ever touch the
        // texels c0, c0+1, c0+2 per axis, so the window is read as <= 3x3 distinct texels with
        // multiplicities (all loads independent, one round trip) instead of 16 dependent samples.
        int cx0, mx[3] = {0, 0, 0}, cy0, my[3] = {0, 0, 0};
        {
            const float endx = x_n + (scale * indexXStep * windowMultiplier);
            float i = x_n - (scale * indexXStep * windowMultiplier);
            cx0 = texel(i, cols);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (i < endx) {
                    const int t = texel(i, cols) - cx0;
                    mx[0] += t == 0, mx[1] += t == 1, mx[2] += t == 2;
                }
                i += indexXStep;
            }
            const float endy = y_n + (scale * indexYStep * windowMultiplier);
            float j = y_n - (scale * indexYStep * windowMultiplier);
            cy0 = texel(j, rows);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (j < endy) {
                    const int t = texel(j, rows) - cy0;
                    my[0] += t == 0, my[1] += t == 1, my[2] += t == 2;
                }
                j += indexYStep;
            }
        }
        unsigned cur[9];
        float4 vcs[9];
        float2 cts[9];  // {init time, timestamp}: the half of the colour / time texel the tests below read (18 registers less:
                        // 104 -> 86, five waves per SIMD instead of four)
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e2 = 0; e2 < 3; ++e2) {
                const size_t t = (size_t)min(cx0 + d, cols - 1) * rows + min(cy0 + e2, rows - 1);  // transposed
                cur[d * 3 + e2] = index[t];
                vcs[d * 3 + e2] = vertConf[t];
                cts[d * 3 + e2] = reinterpret_cast<const float2*>(colorTime + t)[1];
            }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int e2 = 0; e2 < 3; ++e2) {
                const int mult = mx[d] * my[e2];
                if (mult > 0 && cur[d * 3 + e2] > 0U) {
                    const float4 vc = vcs[d * 3 + e2];
                    const float2 ct = cts[d * 3 + e2];  // (.x = colorTime.z, .y = colorTime.w)
                    const float dx = vc.x - localPos.x, dy = vc.y - localPos.y;
                    if (ct.x < vcol.z && vc.w > a.confThreshold && vc.z > localPos.z && vc.z - localPos.z < 0.01f &&
                        sqrtf(dx * dx + dy * dy) < vnrm.w * 1.4f)
                        count += mult;
                    if (ct.y == (float)a.time && vc.w > a.confThreshold && vc.z > localPos.z &&
                        vc.z - localPos.z > 0.01f && fabsf(localNorm.z) > 0.85f)
                        zCount += mult;
                }
            }
        // see-through test (:117-125): <= 3 samples per axis, accumulated in the shader's order
        int txs[3], tys[3];
        bool vxs[3], vys[3];
        {
            float i = x_n - stepX;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                vxs[k] = i <= x_n + stepX;
                txs[k] = texel(i, cols);
                i += stepX;
            }
            float j = y_n - stepY;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                vys[k] = j <= y_n + stepY;
                tys[k] = texel(j, rows);
                j += stepY;
            }
        }
        float dd[9];
#pragma unroll
        for (int ki = 0; ki < 3; ++ki)
#pragma unroll
            for (int kj = 0; kj < 3; ++kj) dd[ki * 3 + kj] = depth_in[tys[kj] * cols + txs[ki]] - localPos.z;
#pragma unroll
        for (int ki = 0; ki < 3; ++ki)
#pragma unroll
            for (int kj = 0; kj < 3; ++kj)
                if (vxs[ki] && vys[kj] && dd[ki * 3 + kj] > 0.03f) {
                    violationCount++;
                    avgViolation += dd[ki * 3 + kj];
                }
    }
    if (count > 8 || zCount > 4) test = 0;
    if (vcol.w == -2) vcol.w = (float)a.time;
    if ((vcol.w == -1 || (((float)a.time - vcol.w) > 20 && vpos.w < a.confThreshold))) test = 0;
    if (vcol.w > 0 && (float)a.time - vcol.w > (float)a.timeDelta) test = 1;
    if (violationCount > 0) {
        avgViolation /= violationCount;
        vpos.w *= 1.0f / (1 + a.outlierCoeff * avgViolation);
        const size_t t = (size_t)texel(y_n, rows) * cols + texel(x_n, cols);
        const float wDepth = depth_in[t];
        if (mask[t] != a.maskID && (wDepth > localPos.z - 0.05f && wDepth < localPos.z + 0.05f))
            vpos.w *= (0.5f + 0.5f * (1 - a.outlierCoeff / 10.0f));
    }
    conf_time[e_store >= 0 ? e_store : e] = make_float2(vpos.w, vcol.w);
    return test ? 1u : 0u;
}

#ifndef MMF_CLEAN_WAVES
#define MMF_CLEAN_WAVES 5  // waves per SIMD the register allocation aims for (96 registers; the default allocation takes 104: four)
#endif
// keep[e] for every candidate, and -- so that the compaction needs no separate scan launches -- the
// number of kept candidates of each 256-candidate workgroup in block_sums[blockIdx.x]
__device__ __forceinline__ void clean_flag_kernel_body(SurfelSoA s, SurfelSoA meas, const unsigned* __restrict__ new_flags,
                                                         CleanArgs a_in, const unsigned* __restrict__ index,
                                                         const float4* __restrict__ vertConf,
                                                         const float4* __restrict__ colorTime,
                                                         const float* __restrict__ depth_in, const uint8_t* __restrict__ mask,
                                                         unsigned* __restrict__ keep, float2* __restrict__ conf_time,
                                                         unsigned* __restrict__ block_sums, const unsigned bx_, [[maybe_unused]] const unsigned gx_) {
    MMF_MODEL_STREAM_PRIORITY();
    MMF_SPECULATION_GUARD(a_in);
    const CleanArgs a = with_device_pose(a_in);
    const int e = bx_ * 256 + threadIdx.x;
    unsigned k = 0u;
    if (e < a.count + a.npix) {
        k = clean_flag_one(e, s, meas, new_flags, a, index, vertConf, colorTime, depth_in, mask, conf_time);
        keep[e] = k;
    }
    const int kept = __syncthreads_count((int)k);
    if (threadIdx.x == 0) block_sums[bx_] = (unsigned)kept;
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MMF_CLEAN_WAVES))) void clean_flag_kernel(SurfelSoA s, SurfelSoA meas, const unsigned* __restrict__ new_flags,
                                                         CleanArgs a_in, const unsigned* __restrict__ index,
                                                         const float4* __restrict__ vertConf,
                                                         const float4* __restrict__ colorTime,
                                                         const float* __restrict__ depth_in, const uint8_t* __restrict__ mask,
                                                         unsigned* __restrict__ keep, float2* __restrict__ conf_time,
                                                         unsigned* __restrict__ block_sums) {
    clean_flag_kernel_body(s, meas, new_flags, a_in, index, vertConf, colorTime, depth_in, mask, keep, conf_time, block_sums, blockIdx.x, gridDim.x);
}

// ordered compaction into the other surfel set (transform feedback of copy_unstable.geom)
// The exclusive scan of keep[] is done here: a workgroup's base = the sum of the block_sums before it
// (every workgroup adds them up itself: a few thousand words from L2 instead of three scan launches),
// the rank inside the workgroup from wave ballots.  The last workgroup writes the new surfel count.
__device__ __forceinline__ void clean_scatter_kernel_body(SurfelSoA s, SurfelSoA meas, int count, int npix,
                                                            const unsigned* __restrict__ keep,
                                                            const unsigned* __restrict__ block_sums,
                                                            const float2* __restrict__ conf_time, SurfelSoA dst,
                                                            int capacity, unsigned* __restrict__ total_out,
                                                            unsigned* __restrict__ total_host, unsigned seq,
                                                            const int* __restrict__ abort_dev, const unsigned bx_, [[maybe_unused]] const unsigned gx_) {
    MMF_MODEL_STREAM_PRIORITY();
    if (abort_dev != nullptr && *abort_dev != 0) return;  // (MMF_SPECULATION_GUARD: neither surfels nor the count are published)
    __shared__ unsigned wave_part[4], wave_kept[4];
    const int e = bx_ * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // one memory round trip: the keep flag, this thread's share of the block sums (ten loads in flight; a loop
    // with a wait per iteration cost the late workgroups ~5 us) and, for the old surfels -- nearly all of which
    // survive -- the surfel itself, all issued before anything is consumed
    // (every address is clamped and every load unconditional: behind `if (old)` / `if (live)` the compiler waited for
    // the keep flag before it issued the rest and staged the surfel through LDS -- three dependent round trips, 17 us)
    const bool live = e < count + npix, old = e < count;
    const int ec = min(e, count + npix - 1), eo = min(e, max(count - 1, 0));
    const unsigned kp_raw = keep[ec];
    constexpr int kSumsPerThread = 10;  // 2560 workgroups = 655 k elements without the tail loop
    unsigned bs[kSumsPerThread];
    const unsigned last_block = gx_ - 1;
#pragma unroll
    for (int u = 0; u < kSumsPerThread; ++u) bs[u] = block_sums[min(threadIdx.x + 256u * u, last_block)];
    const float4 p = s.pos[eo], c = s.col[eo], n = s.nrm[eo];
    const float2 ct = conf_time[ec];
    __builtin_amdgcn_sched_barrier(0);
    const unsigned kp = live ? kp_raw : 0u;
    unsigned part = 0;
#pragma unroll
    for (int u = 0; u < kSumsPerThread; ++u) part += threadIdx.x + 256u * u < bx_ ? bs[u] : 0u;
    for (unsigned j = threadIdx.x + 256u * kSumsPerThread; j < bx_; j += 256) part += block_sums[j];
    part = wave_sum_to_lane63(part);
    const unsigned long long ballot = __ballot(kp != 0u);
    if (lane == 63) wave_part[wave] = part;
    if (lane == 0) wave_kept[wave] = (unsigned)__popcll(ballot);
    __syncthreads();
    unsigned base = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3];
    for (int w = 0; w < wave; ++w) base += wave_kept[w];
    if (bx_ == gx_ - 1 && threadIdx.x == 0) {
        const unsigned total = wave_part[0] + wave_part[1] + wave_part[2] + wave_part[3] + wave_kept[0] + wave_kept[1] +
                               wave_kept[2] + wave_kept[3];
        *total_out = total;
        if (total_host) {  // the host's pinned, device-visible copy: value, system-scope fence, then the sequence number it polls
            total_host[0] = total;
            __threadfence_system();
            __hip_atomic_store(&total_host[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!kp) return;
    const unsigned k = base + (unsigned)__popcll(ballot & ((1ull << lane) - 1ull));
    if (k >= (unsigned)capacity) return;  // the reference's VBO is full: further primitives are dropped
    if (old) {
        dst.pos[k] = make_float4(p.x, p.y, p.z, ct.x);
        dst.col[k] = make_float4(c.x, c.y, c.z, ct.y);
        dst.nrm[k] = n;
    } else {  // a new measurement that survives (few): fetched now
        const int d = e - count;
        const float4 mp = meas.pos[d], mc = meas.col[d], mn = meas.nrm[d];
        dst.pos[k] = make_float4(mp.x, mp.y, mp.z, ct.x);
        dst.col[k] = make_float4(mc.x, mc.y, mc.z, ct.y);
        dst.nrm[k] = mn;
    }
}
__global__ __launch_bounds__(256) void clean_scatter_kernel(SurfelSoA s, SurfelSoA meas, int count, int npix,
                                                            const unsigned* __restrict__ keep,
                                                            const unsigned* __restrict__ block_sums,
                                                            const float2* __restrict__ conf_time, SurfelSoA dst,
                                                            int capacity, unsigned* __restrict__ total_out,
                                                            unsigned* __restrict__ total_host, unsigned seq,
                                                            const int* __restrict__ abort_dev) {
    clean_scatter_kernel_body(s, meas, count, npix, keep, block_sums, conf_time, dst, capacity, total_out, total_host, seq, abort_dev, blockIdx.x, gridDim.x);
}

// ---- fill-in (fill_vertex.frag, fill_normal.frag, fill_rgb.frag) + thumbnail count ------------------
// one texel of the three fill-in passes from the prediction's values at that texel
__device__ __forceinline__ void fill_in_px(int i, const float4 vp, const float4 np, const uchar4 e,
                                           const float* __restrict__ depth_fil, const uint8_t* __restrict__ rgb, int cols,
                                           int rows, const Cam& c, int passthrough_geom, int passthrough_rgb,
                                           float4* __restrict__ vertex_out, float4* __restrict__ normal_out,
                                           uchar4* __restrict__ image_out) {
    const int py = i / cols, px = i - py * cols;
    const float tx = (px + 0.5f) / cols, ty = (py + 0.5f) / rows;
    const int ix = (int)(tx * cols), iy = (int)(ty * rows);
    if (vp.z == 0 || passthrough_geom == 1) {
        const float z = depth_fil[texel(ty, rows) * cols + texel(tx, cols)];
        vertex_out[i] = make_float4((ix - c.cx) * z * c.ifx, (iy - c.cy) * z * c.ify, z, 1.f);
    } else {
        vertex_out[i] = vp;
    }
    if (np.z == 0 || passthrough_geom == 1) {
        const v3 p = get_vertex(depth_fil, cols, rows, tx, ty, (float)ix, (float)iy, c);
        const v3 vx = get_vertex(depth_fil, cols, rows, tx + (1.0f / cols), ty, (float)(ix + 1), (float)iy, c);
        const v3 vy = get_vertex(depth_fil, cols, rows, tx, ty + (1.0f / rows), (float)ix, (float)(iy + 1), c);
        const v3 nn = v3normalize(v3cross(v3sub(vx, p), v3sub(vy, p)));
        normal_out[i] = make_float4(nn.x, nn.y, nn.z, 1.f);
    } else {
        normal_out[i] = np;
    }
    if (e.x / 255.0f + e.y / 255.0f + e.z / 255.0f == 0 || passthrough_rgb == 1)
        image_out[i] = make_uchar4(rgb[3 * i + 0], rgb[3 * i + 1], rgb[3 * i + 2], 255);
    else
        image_out[i] = e;
}

__global__ __launch_bounds__(256) void fill_in_kernel(const float4* __restrict__ vertex_pred, const float4* __restrict__ normal_pred,
                                                      const uchar4* __restrict__ image_pred, const float* __restrict__ depth_fil,
                                                      const uint8_t* __restrict__ rgb, int cols, int rows, Cam c,
                                                      int passthrough_geom, int passthrough_rgb, float4* __restrict__ vertex_out,
                                                      float4* __restrict__ normal_out, uchar4* __restrict__ image_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cols * rows) return;
    fill_in_px(i, vertex_pred[i], normal_pred[i], image_pred[i], depth_fil, rgb, cols, rows, c, passthrough_geom,
               passthrough_rgb, vertex_out, normal_out, image_out);
}

// combinedPredict's resolve and performFillIn in one pass over the image (MultiMotionFusion::predict runs them
// back to back, MultiMotionFusion.cpp:863-875): the fill-in takes the prediction's texel from registers instead of
// reading the three images back
__global__ __launch_bounds__(256) void splat_resolve_fill_kernel(SurfelSoA s, SplatArgs a_in, unsigned long long* __restrict__ keys,
                                                                 uchar4* __restrict__ image, float4* __restrict__ vertexConf,
                                                                 float4* __restrict__ normalRadius,
                                                                 unsigned short* __restrict__ time_out,
                                                                 const float* __restrict__ depth_fil,
                                                                 const uint8_t* __restrict__ rgb, int passthrough_geom,
                                                                 int passthrough_rgb, float4* __restrict__ vertex_out,
                                                                 float4* __restrict__ normal_out, uchar4* __restrict__ image_out,
                                                                 unsigned* __restrict__ thumb, int gen) {
    MMF_MODEL_STREAM_PRIORITY();
    MMF_SPECULATION_GUARD(a_in);
    const SplatArgs a = with_device_pose(a_in);
    int px, py;
    unsigned long long k;
    if (!splat_tile_key(keys, a.cols, a.rows, px, py, k, 0)) return;
    const int i = py * a.cols + px;
    const SplatTexel t = splat_resolve_px(i, k, s, a);
    image[i] = t.image;
    vertexConf[i] = t.vertexConf, normalRadius[i] = t.normalRadius;
    time_out[i] = t.time;
    fill_in_px(i, t.vertexConf, t.normalRadius, t.image, depth_fil, rgb, a.cols, a.rows, a.c, passthrough_geom, passthrough_rgb,
               vertex_out, normal_out, image_out);
    thumbnail_count_px(px, py, a.cols, a.rows, t.image, thumb, gen, blockIdx.x == gridDim.x - 1);
}

// requiresFillIn: number of (cols/20 x rows/20) thumbnail samples with all three channels > 0
__global__ __launch_bounds__(256) void thumbnail_count_kernel(const uchar4* __restrict__ image_pred, int cols, int rows,
                                                              unsigned* __restrict__ out) {
    const int dc = cols / 20, dr = rows / 20;
    const int t = blockIdx.x * 256 + threadIdx.x;
    unsigned hit = 0;
    if (t < dc * dr) {
        const int j = t / dc, i = t - j * dc;
        const float tx = (i + 0.5f) / dc, ty = (j + 0.5f) / dr;
        const uchar4 p = image_pred[(size_t)texel(ty, rows) * cols + texel(tx, cols)];
        hit = (p.x > 0 && p.y > 0 && p.z > 0) ? 1u : 0u;
    }
    const unsigned long long b = __ballot(hit != 0);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(out, (unsigned)__popcll(b));
}

// The same count by ONE workgroup, turned into the decision on the device: *flag = 1 when fewer than
// `ratio` of the thumbnail samples are covered (MultiMotionFusion.cpp:877-895).  The native orchestrator
// uses it to pick the tracker's source images without a host round trip.
__global__ __launch_bounds__(256) void thumbnail_flag_kernel(const uchar4* __restrict__ image_pred, int cols, int rows,
                                                             float ratio, int* __restrict__ flag) {
    const int dc = cols / 20, dr = rows / 20;
    int hits = 0;
    for (int t = threadIdx.x; t < dc * dr; t += 256) {
        const int j = t / dc, i = t - j * dc;
        const float tx = (i + 0.5f) / dc, ty = (j + 0.5f) / dr;
        const uchar4 p = image_pred[(size_t)texel(ty, rows) * cols + texel(tx, cols)];
        hits += (p.x > 0 && p.y > 0 && p.z > 0) ? 1 : 0;
    }
    __shared__ int part[4];
    hits = wave_sum_to_lane63(hits);
    if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = hits;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = part[0] + part[1] + part[2] + part[3];
        *flag = ((float)total / (float)(dr * dc) < ratio) ? 1 : 0;
    }
}

__global__ void fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void fill_u32_kernel(unsigned* p, size_t n, unsigned v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// AoS <-> SoA at the download / upload boundary (Model::downloadMap, Model.cpp:1353-1384)
__global__ void soa_to_aos_kernel(SurfelSoA s, int count, float4* __restrict__ aos) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    aos[3 * k + 0] = s.pos[k];
    aos[3 * k + 1] = s.col[k];
    aos[3 * k + 2] = s.nrm[k];
}
__global__ void aos_to_soa_kernel(const float4* __restrict__ aos, int count, SurfelSoA s) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    s.pos[k] = aos[3 * k + 0];
    s.col[k] = aos[3 * k + 1];
    s.nrm[k] = aos[3 * k + 2];
}


// ---- the passes of SEVERAL models in one launch each (gridDim.y = model) ----------------------------------------------------
// An object model's store is a few thousand surfels: its projection / fuse / clean / predict passes (MultiMotionFusion.cpp:791-816,
// 863-875 loop over the models) are ~9 short launches per model and frame, and with seven object models on a GPU the calling
// thread's launch rate, not the GPU, set the pace of that part of the frame.  Every pass exists as *_kernel_body(arguments,
// block, blocks); the *_batched_kernel of a pass runs the bodies of up to kMaxPassBatch models, each on its own arguments and
// its own number of workgroups (the surplus workgroups of a smaller model leave at once).  Same bodies, same bits.
constexpr int kMaxPassBatch = 7;
template <typename Item>
struct PassBatch {
    Item m[kMaxPassBatch];
};

struct index_map_item {
    SurfelSoA s;
    int count;
    IndexArgs a_in;
    unsigned long long* keys;
    FrameRider rider;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) void index_map_batched_kernel(PassBatch<index_map_item> b) {
    const index_map_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    index_map_kernel_body(p.s, p.count, p.a_in, p.keys, p.rider, blockIdx.x, p.grid);
}

struct index_resolve_item {
    SurfelSoA s;
    IndexArgs a_in;
    unsigned long long* keys;
    unsigned* index;
    float4* vertConf;
    float4* colorTime;
    float4* normRad;
    FrameRider rider;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) void index_resolve_batched_kernel(PassBatch<index_resolve_item> b) {
    const index_resolve_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    index_resolve_kernel_body(p.s, p.a_in, p.keys, p.index, p.vertConf, p.colorTime, p.normRad, p.rider, blockIdx.x, p.grid);
}

struct fuse_data_item {
    const uint8_t* rgb;
    const float* depth_raw;
    const float* depth_fil;
    const uint8_t* mask;
    const unsigned* index;
    const float4* vertConf;
    const float4* normRad;
    FuseArgs a_in;
    SurfelSoA meas;
    unsigned* new_flags;
    unsigned* winner;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) void fuse_data_batched_kernel(PassBatch<fuse_data_item> b) {
    const fuse_data_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    fuse_data_kernel_body(p.rgb, p.depth_raw, p.depth_fil, p.mask, p.index, p.vertConf, p.normRad, p.a_in, p.meas, p.new_flags, p.winner, blockIdx.x, p.grid);
}

struct fuse_update_index_item {
    SurfelSoA s;
    int count;
    SurfelSoA meas;
    int time;
    unsigned* winner;
    IndexArgs a_in;
    unsigned long long* keys;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) void fuse_update_index_batched_kernel(PassBatch<fuse_update_index_item> b) {
    const fuse_update_index_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    fuse_update_index_kernel_body(p.s, p.count, p.meas, p.time, p.winner, p.a_in, p.keys, blockIdx.x, p.grid);
}

struct clean_flag_item {
    SurfelSoA s;
    SurfelSoA meas;
    const unsigned* new_flags;
    CleanArgs a_in;
    const unsigned* index;
    const float4* vertConf;
    const float4* colorTime;
    const float* depth_in;
    const uint8_t* mask;
    unsigned* keep;
    float2* conf_time;
    unsigned* block_sums;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MMF_CLEAN_WAVES))) void clean_flag_batched_kernel(PassBatch<clean_flag_item> b) {
    const clean_flag_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    clean_flag_kernel_body(p.s, p.meas, p.new_flags, p.a_in, p.index, p.vertConf, p.colorTime, p.depth_in, p.mask, p.keep, p.conf_time, p.block_sums, blockIdx.x, p.grid);
}

struct clean_scatter_item {
    SurfelSoA s;
    SurfelSoA meas;
    int count;
    int npix;
    const unsigned* keep;
    const unsigned* block_sums;
    const float2* conf_time;
    SurfelSoA dst;
    int capacity;
    unsigned* total_out;
    unsigned* total_host;
    unsigned seq;
    const int* abort_dev;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) void clean_scatter_batched_kernel(PassBatch<clean_scatter_item> b) {
    const clean_scatter_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    clean_scatter_kernel_body(p.s, p.meas, p.count, p.npix, p.keep, p.block_sums, p.conf_time, p.dst, p.capacity, p.total_out, p.total_host, p.seq, p.abort_dev, blockIdx.x, p.grid);
}

struct splat_item {
    SurfelSoA s;
    int count;
    SplatArgs a_in;
    unsigned long long* keys;
    const unsigned* count_dev;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void splat_batched_kernel(PassBatch<splat_item> b) {
    const splat_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    splat_kernel_body<false>(p.s, p.count, p.a_in, p.keys, p.count_dev, blockIdx.x, p.grid);
}

struct splat_resolve_item {
    SurfelSoA s;
    SplatArgs a_in;
    unsigned long long* keys;
    uchar4* image;
    float4* vertexConf;
    float4* normalRadius;
    unsigned short* time_out;
    unsigned* thumb;
    int gen;
    unsigned grid;  // workgroups of this model
};
__global__ __launch_bounds__(256) void splat_resolve_batched_kernel(PassBatch<splat_resolve_item> b) {
    const splat_resolve_item& p = b.m[blockIdx.y];
    if (blockIdx.x >= p.grid) return;
    splat_resolve_kernel_body(p.s, p.a_in, p.keys, p.image, p.vertexConf, p.normalRadius, p.time_out, p.thumb, p.gen, blockIdx.x, p.grid);
}

}  // namespace mmf
