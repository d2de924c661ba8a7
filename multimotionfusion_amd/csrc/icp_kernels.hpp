// icp_kernels.hpp -- the point-to-plane ICP reduction for gfx950 (icpKernel + reduceSum,
// Core/Cuda/reduce.cu:231-473), second generation.
//
// The ISA of the first version showed what its 8 us at 640x480 were made of: ~280 VALU
// instructions per pixel (2.2 us of pure issue time over 1024 SIMDs), four dependent memory round
// trips (hipcc sank the normal loads and split the gathers behind the `found` branch), a 32-bit
// integer division per pass and a 174-instruction DPP reduction per wave.  This version:
//   * computes TWO pixels per lane (in packed registers: v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32),
//     sharing the per-lane scalar work; a packed op costs what its two scalar ops cost on gfx950
//     (the 1-pixel instantiation times the same), so this halves the instruction stream, not the time;
//   * is branch free per pixel, with scheduling barriers around the two load groups, so there
//     are exactly two memory round trips: 6 coalesced loads, then all gathers;
//   * optionally gathers the model vertex + normal as two 12-byte loads from a pixel-interleaved
//     copy (built once per frame by the transform kernel) instead of six 4-byte loads;
//   * compares squared distances against exact squared thresholds (no sqrt per pixel unless the
//     error map is being written);
//   * replaces i / cols by a multiply-high with a host-computed magic number;
//   * reduces the 29 sums over the wave with a transposed (halving) butterfly: 32 -> 16 -> 8 ...
//     values per lane (v_permlane32_swap / v_permlane16_swap across rows, bank-masked DPP adds
//     inside a row): ~65 instructions instead of 174.
// Per-pixel arithmetic is still evaluated in the oracle's operation order without contraction,
// so every Jacobian row is bit-identical to oracle/mmf_oracle.c; only the summation order
// differs (fixed by the launch geometry, hence run-to-run deterministic).
#pragma once
#include "device_math.hpp"
#include "grid_reduce.hpp"
#include "odom_state.hpp"

namespace mmf {

using v2f = float __attribute__((ext_vector_type(2)));

// ---- lane-vector helpers: T = float (1 pixel) or v2f (2 pixels) ---------------------------
template <typename T>
struct f3t {
    T x, y, z;
};
template <typename T>
__device__ __forceinline__ f3t<T> operator-(f3t<T> a, f3t<T> b) {
    return f3t<T>{a.x - b.x, a.y - b.y, a.z - b.z};
}
template <typename T>
__device__ __forceinline__ f3t<T> cross(f3t<T> a, f3t<T> b) {
    return f3t<T>{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T>
__device__ __forceinline__ T dot(f3t<T> a, f3t<T> b) {
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
// row-major 3x3 times vector, operation order of device_math.hpp's operator*
template <typename T>
__device__ __forceinline__ f3t<T> mul(const float (&M)[9], f3t<T> a) {
    return f3t<T>{M[0] * a.x + M[1] * a.y + M[2] * a.z, M[3] * a.x + M[4] * a.y + M[5] * a.z,
                  M[6] * a.x + M[7] * a.y + M[8] * a.z};
}

template <typename T>
struct lanevec;
template <>
struct lanevec<float> {
    static constexpr int W = 1;
    static __device__ __forceinline__ float get(float v, int) { return v; }
    static __device__ __forceinline__ void set(float& v, int, float s) { v = s; }
    static __device__ __forceinline__ float splat(float s) { return s; }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __device__ __forceinline__ float hsum(float v) { return v; }
};
template <>
struct lanevec<v2f> {
    static constexpr int W = 2;
    static __device__ __forceinline__ float get(v2f v, int i) { return i ? v.y : v.x; }
    static __device__ __forceinline__ void set(v2f& v, int i, float s) {
        if (i) v.y = s;
        else v.x = s;
    }
    static __device__ __forceinline__ v2f splat(float s) { return v2f{s, s}; }
    static __device__ __forceinline__ v2f fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ float hsum(v2f v) { return v.x + v.y; }
};

struct MapView {  // planar 3-plane map: element (plane k, row y, col x) at base[(y + k*rows)*stride + x]
    const float* base;
    int stride;  // in floats
};

struct IcpArgs {
    MapView vmap_curr, nmap_curr, vmap_g_prev, nmap_g_prev;
    const float* prev_packed;  // optional: [pixel] {vertex xyz, normal xyz} (24 bytes, two 12-byte loads), dense rows
    LevelIntr intr;
    float dist_thres, angle_thres;
    float dist_sq_max;   // largest x with sqrtf(x) <= dist_thres   (host: exact_sq_thresholds)
    float sine_sq_min;   // smallest x with sqrtf(x) >= angle_thres
    unsigned cols_magic;  // floor(2^32 / cols) + 1: i / cols == umulhi(i, magic) for i * cols < 2^32
    int cols, rows;
    float* err_map;  // optional
    int err_stride;
};

// PX consecutive floats of one plane row as ONE load instruction (4, 8 or 16 bytes per lane)
template <int PX>
__device__ __forceinline__ void load_px(const float* __restrict__ p, float (&out)[PX]) {
    if constexpr (PX == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        out[0] = t.x, out[1] = t.y, out[2] = t.z, out[3] = t.w;
    } else if constexpr (PX == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        out[0] = t.x, out[1] = t.y;
    } else {
        out[0] = *p;
    }
}
template <int PX>
__device__ __forceinline__ void store_px(float* __restrict__ p, const float (&v)[PX]) {
    if constexpr (PX == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (PX == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    } else {
        *p = v[0];
    }
}

// ---- the per-pixel work ---------------------------------------------------------------------
struct IcpPose {  // wave-uniform copy of the pose being optimised (scalar registers)
    float Rcurr[9], tcurr[3], Rprev_inv[9], tprev[3];
};
__device__ __forceinline__ IcpPose load_icp_pose(const OdomState* __restrict__ st) {
    IcpPose p;
#pragma unroll
    for (int k = 0; k < 9; ++k) p.Rcurr[k] = st->Rcurr[k], p.Rprev_inv[k] = st->Rprev_inv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) p.tcurr[k] = st->tcurr[k], p.tprev[k] = st->tprev[k];
    return p;
}

// ICPReduction::search + getProducts for the W pixels of one lane vector (reduce.cu:257-368).
// Phase A (project): needs only the current vertex; yields the gather coordinates.
template <typename T>
struct IcpProj {
    f3t<T> vcurr_g, vcurr_cp;
    int ux[lanevec<T>::W], uy[lanevec<T>::W];
    bool inside[lanevec<T>::W];
};

template <typename T>
__device__ __forceinline__ IcpProj<T> icp_project_v(const IcpPose& P, const IcpArgs& a, f3t<T> vcurr) {
    using L = lanevec<T>;
    IcpProj<T> p;
    const f3t<T> rv = mul(P.Rcurr, vcurr);
    p.vcurr_g = f3t<T>{rv.x + P.tcurr[0], rv.y + P.tcurr[1], rv.z + P.tcurr[2]};
    const f3t<T> d = f3t<T>{p.vcurr_g.x - P.tprev[0], p.vcurr_g.y - P.tprev[1], p.vcurr_g.z - P.tprev[2]};
    p.vcurr_cp = mul(P.Rprev_inv, d);
    const T px = p.vcurr_cp.x * a.intr.fx / p.vcurr_cp.z + a.intr.cx;
    const T py = p.vcurr_cp.y * a.intr.fy / p.vcurr_cp.z + a.intr.cy;
#pragma unroll
    for (int e = 0; e < L::W; ++e) {
        const int ux = float2int_rn(L::get(px, e)), uy = float2int_rn(L::get(py, e));
        const bool in = !(ux < 0 || uy < 0 || ux >= a.cols || uy >= a.rows || L::get(p.vcurr_cp.z, e) < 0);
        p.inside[e] = in;
        p.ux[e] = in ? ux : 0;  // outside: read element 0 (a valid address), masked afterwards
        p.uy[e] = in ? uy : 0;
    }
    return p;
}

// Phase B: Jacobian rows from the gathered model vertex / normal, accumulated into sum[29].
// FIRST: sum[] is initialised with the products instead of accumulated into.
template <bool ERR, bool FIRST, typename T>
__device__ __forceinline__ void icp_rows_v(const IcpPose& P, const IcpArgs& a, const IcpProj<T>& p, bool live,
                                           f3t<T> ncurr, f3t<T> vprev_g, f3t<T> nprev_g, T (&sum)[29],
                                           float (&err)[lanevec<T>::W]) {
    using L = lanevec<T>;
    const f3t<T> ncurr_g = mul(P.Rcurr, ncurr);
    const f3t<T> dv = vprev_g - p.vcurr_g;
    const T dist_sq = dot(dv, dv);
    const f3t<T> cr = cross(ncurr_g, nprev_g);
    const T sine_sq = dot(cr, cr);
    // rows (reduce.cu:320-329); s_cp is the projection's vcurr_cp (same expression)
    const f3t<T> dp = f3t<T>{vprev_g.x - P.tprev[0], vprev_g.y - P.tprev[1], vprev_g.z - P.tprev[2]};
    const f3t<T> d_cp = mul(P.Rprev_inv, dp);
    const f3t<T> n_cp = mul(P.Rprev_inv, nprev_g);
    const f3t<T> c = cross(p.vcurr_cp, n_cp);
    const T r6 = dot(n_cp, p.vcurr_cp - d_cp);
    T row[7] = {n_cp.x, n_cp.y, n_cp.z, c.x, c.y, c.z, r6};
    T found_f;
#pragma unroll
    for (int e = 0; e < L::W; ++e) {
        // sqrtf is monotonic and correctly rounded, so sqrtf(x) <= T  <=>  x <= dist_sq_max and
        // sqrtf(x) < S  <=>  x < sine_sq_min for the exact host-computed squared thresholds;
        // a NaN fails both forms alike.
        const float d2 = L::get(dist_sq, e), s2 = L::get(sine_sq, e);
        const float ncx = L::get(ncurr.x, e), npx = L::get(nprev_g.x, e);
        const bool found = live && p.inside[e] && (s2 < a.sine_sq_min && d2 <= a.dist_sq_max && !(ncx != ncx) && !(npx != npx));
#pragma unroll
        for (int k = 0; k < 7; ++k) L::set(row[k], e, found ? L::get(row[k], e) : 0.f);
        L::set(found_f, e, found ? 1.0f : 0.0f);
        if (ERR) {  // reduce.cu:275,299
            const float dist = sqrtf(d2);
            err[e] = p.inside[e] ? (isfinite(dist) ? dist : 0.0f) : 0.0f;
        }
    }
    // 27 upper-triangular products + residual^2 + inliers, member order of JtJJtrSE3
    // (types.cuh:101-112, reduce.cu:331-365); the running sums use fused multiply-adds
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 7; ++j) {
            sum[k] = FIRST ? row[i] * row[j] : L::fma(row[i], row[j], sum[k]);
            ++k;
        }
    sum[27] = FIRST ? row[6] * row[6] : L::fma(row[6], row[6], sum[27]);
    sum[28] = FIRST ? found_f : sum[28] + found_f;
}

// T = lane vector (float: 1 px, v2f: 2 px); NV lane vectors per lane per pass, loaded as ONE
// 4*W*NV-byte load per plane.  PACKED: gather from a.prev_packed.
// CHECK_BREAK: return when st->level_break is set -- tested only after the current-frame loads have been
// issued: the state was written by the previous kernel's finishing lane and reading it is a cold
// ~1 us round trip that should overlap those loads, not precede them.
// MULTI: the grid is smaller than the image: every lane walks it with the grid's stride (sums start from zero and
// every pass accumulates); else ONE pass and the first vector's products initialise the sums.
template <typename T, int NV, int BLOCK, bool PACKED, bool ERR, bool CHECK_BREAK = false, bool MULTI = false>
__device__ __forceinline__ void icp_block2(const OdomState* __restrict__ st, const IcpArgs& a,
                                           float* __restrict__ partials, GridReduceLds<float, BLOCK>& lds,
                                           unsigned bid, unsigned nblocks) {
    using L = lanevec<T>;
    constexpr int W = L::W, PX = W * NV;
    const IcpPose P = load_icp_pose(st);
    const int level_break = CHECK_BREAK ? st->level_break : 0;

    T sum[29];
    if (MULTI) {
#pragma unroll
        for (int k = 0; k < 29; ++k) sum[k] = L::splat(0.f);
    }
    const unsigned N = (unsigned)(a.cols * a.rows);
    const int rows = a.rows;
    const unsigned stride = nblocks * BLOCK * PX;
    // ONE pass unless MULTI: the host sizes the grid to cover the image (N % PX == 0).  Lanes past the end
    // stay active for the wave reduction: they recompute pixel 0 and contribute zeros.
    for (unsigned first = (bid * BLOCK + threadIdx.x) * PX;; first += stride) {
        unsigned i0 = first;
        const bool live = i0 < N;
        i0 = live ? i0 : 0u;
        // cols_magic == 0: the image is too large for the multiply-high (i0 * cols >= 2^32)
        const unsigned y = a.cols_magic ? __umulhi(i0, a.cols_magic) : i0 / (unsigned)a.cols;
        const unsigned x = i0 - y * (unsigned)a.cols;

        // ---- round trip 1: the lane's current vertices and normals, six loads in flight ----
        float cur[6][PX];
        {
            const float* pv = a.vmap_curr.base + (size_t)y * a.vmap_curr.stride + x;
            const float* pn = a.nmap_curr.base + (size_t)y * a.nmap_curr.stride + x;
            const size_t sv = (size_t)rows * a.vmap_curr.stride, sn = (size_t)rows * a.nmap_curr.stride;
            load_px<PX>(pv, cur[0]);
            load_px<PX>(pv + sv, cur[1]);
            load_px<PX>(pv + 2 * sv, cur[2]);
            load_px<PX>(pn, cur[3]);
            load_px<PX>(pn + sn, cur[4]);
            load_px<PX>(pn + 2 * sn, cur[5]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (level_break) return;  // wave-uniform

        IcpProj<T> pr[NV];
#pragma unroll
        for (int g = 0; g < NV; ++g) {
            f3t<T> v;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                L::set(v.x, e, cur[0][g * W + e]);
                L::set(v.y, e, cur[1][g * W + e]);
                L::set(v.z, e, cur[2][g * W + e]);
            }
            pr[g] = icp_project_v<T>(P, a, v);
        }

        // ---- round trip 2: every gather of the lane issued before any is consumed ----------
        f3t<T> vp[NV], np[NV];
        __builtin_amdgcn_sched_barrier(0);
        if (PACKED) {
            struct f3pk {
                float x, y, z;
            };
            f3pk gv[PX], gn[PX];
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const f3pk* src = reinterpret_cast<const f3pk*>(a.prev_packed) +
                                  2 * ((size_t)pr[q / W].uy[q % W] * a.cols + pr[q / W].ux[q % W]);
                gv[q] = src[0];
                gn[q] = src[1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                L::set(vp[q / W].x, q % W, gv[q].x);
                L::set(vp[q / W].y, q % W, gv[q].y);
                L::set(vp[q / W].z, q % W, gv[q].z);
                L::set(np[q / W].x, q % W, gn[q].x);
                L::set(np[q / W].y, q % W, gn[q].y);
                L::set(np[q / W].z, q % W, gn[q].z);
            }
        } else {
            float g6[PX][6];
            const size_t pv = (size_t)rows * a.vmap_g_prev.stride, pn = (size_t)rows * a.nmap_g_prev.stride;
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const size_t ov = (size_t)pr[q / W].uy[q % W] * a.vmap_g_prev.stride + pr[q / W].ux[q % W];
                const size_t on = (size_t)pr[q / W].uy[q % W] * a.nmap_g_prev.stride + pr[q / W].ux[q % W];
                g6[q][0] = a.vmap_g_prev.base[ov];
                g6[q][1] = a.vmap_g_prev.base[ov + pv];
                g6[q][2] = a.vmap_g_prev.base[ov + 2 * pv];
                g6[q][3] = a.nmap_g_prev.base[on];
                g6[q][4] = a.nmap_g_prev.base[on + pn];
                g6[q][5] = a.nmap_g_prev.base[on + 2 * pn];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                L::set(vp[q / W].x, q % W, g6[q][0]);
                L::set(vp[q / W].y, q % W, g6[q][1]);
                L::set(vp[q / W].z, q % W, g6[q][2]);
                L::set(np[q / W].x, q % W, g6[q][3]);
                L::set(np[q / W].y, q % W, g6[q][4]);
                L::set(np[q / W].z, q % W, g6[q][5]);
            }
        }

        float errs[PX];
#pragma unroll
        for (int g = 0; g < NV; ++g) {
            f3t<T> n;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                L::set(n.x, e, cur[3][g * W + e]);
                L::set(n.y, e, cur[4][g * W + e]);
                L::set(n.z, e, cur[5][g * W + e]);
            }
            float er[W];
            if (g == 0 && !MULTI)
                icp_rows_v<ERR, true, T>(P, a, pr[g], live, n, vp[g], np[g], sum, er);
            else
                icp_rows_v<ERR, false, T>(P, a, pr[g], live, n, vp[g], np[g], sum, er);
            if (ERR) {
#pragma unroll
                for (int e = 0; e < W; ++e) errs[g * W + e] = er[e];
            }
        }
        if (ERR && live) store_px<PX>(a.err_map + (size_t)y * a.err_stride + x, errs);
        if (!MULTI || first + stride >= N || first + stride < first) break;  // (the lanes of a wave leave together or at the tail)
    }

    float s32[32];
#pragma unroll
    for (int k = 0; k < 29; ++k) s32[k] = L::hsum(sum[k]);
    s32[29] = s32[30] = s32[31] = 0.f;
    block_reduce_store<32, BLOCK, false>(s32, partials, lds, bid, nblocks);
}

// dense planar model maps -> the pixel-interleaved copy the packed gather reads
__global__ __launch_bounds__(256) void pack_prev_kernel(const float* __restrict__ vmap, const float* __restrict__ nmap,
                                                        int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2* o = reinterpret_cast<float2*>(out + 6 * (size_t)i);  // 24-byte records: 8-byte aligned
    o[0] = make_float2(vmap[i], vmap[i + n]);
    o[1] = make_float2(vmap[i + 2 * n], nmap[i]);
    o[2] = make_float2(nmap[i + n], nmap[i + 2 * n]);
}

}  // namespace mmf
