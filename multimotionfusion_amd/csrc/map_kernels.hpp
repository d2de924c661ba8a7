// map_kernels.hpp -- vertex/normal map and image-pyramid kernels for gfx950
// (replacing Core/Cuda/cudafuncs.cu:109-762).  All are one-touch streaming kernels:
// a workgroup is 64 x 4 pixels so a wave64 reads/writes one contiguous 256-byte row segment
// per plane (the reference's 32 x 8 blocks are shaped for 32-wide warps).
#pragma once
#include "device_math.hpp"

namespace mmf {

constexpr int kTileX = 64, kTileY = 4;

#define MMF_PIXEL_XY()                                        \
    const int x = blockIdx.x * kTileX + threadIdx.x;          \
    const int y = blockIdx.y * kTileY + threadIdx.y

// cudafuncs.cu:109-134 (computeVmapKernel); the mask test is commented out there (:119)
__device__ __forceinline__ void create_vmap_px(int x, int y, const float* __restrict__ depth, int d_stride, int cols,
                                                          int rows, float* __restrict__ vmap, int v_stride,
                                                          float fx_inv, float fy_inv, float cx, float cy,
                                                          float cutoff) {
    if (x >= cols || y >= rows) return;
    const float z = depth[(size_t)y * d_stride + x];
    if (z != 0 && z < cutoff) {
        vmap[(size_t)y * v_stride + x] = z * (x - cx) * fx_inv;
        vmap[(size_t)(y + rows) * v_stride + x] = z * (y - cy) * fy_inv;
        vmap[(size_t)(y + 2 * rows) * v_stride + x] = z;
    } else {
        vmap[(size_t)y * v_stride + x] = qnan();
    }
}
// the vertex create_vmap_px stores for depth z at pixel (x, y); false: none (the x plane gets NaN)
__device__ __forceinline__ bool vmap_value(int x, int y, float z, float fx_inv, float fy_inv, float cx, float cy, float cutoff, f3& v) {
    if (!(z != 0 && z < cutoff)) return false;
    v.x = z * (x - cx) * fx_inv;
    v.y = z * (y - cy) * fy_inv;
    v.z = z;
    return true;
}
__global__ __launch_bounds__(256) void create_vmap_kernel(const float* __restrict__ depth, int d_stride, int cols,
                                                          int rows, float* __restrict__ vmap, int v_stride,
                                                          float fx_inv, float fy_inv, float cx, float cy,
                                                          float cutoff) {
    MMF_PIXEL_XY();
    create_vmap_px(x, y, depth, d_stride, cols, rows, vmap, v_stride, fx_inv, fy_inv, cx, cy, cutoff);
}

// cudafuncs.cu:152-189 (computeNmapKernel)
__device__ __forceinline__ void create_nmap_px(int x, int y, int rows, int cols, const float* __restrict__ vmap,
                                                          int v_stride, float* __restrict__ nmap, int n_stride) {
    if (x >= cols || y >= rows) return;
    if (x == cols - 1 || y == rows - 1) {
        nmap[(size_t)y * n_stride + x] = qnan();
        return;
    }
    f3 v00, v01, v10;
    v00.x = vmap[(size_t)y * v_stride + x];
    v01.x = vmap[(size_t)y * v_stride + x + 1];
    v10.x = vmap[(size_t)(y + 1) * v_stride + x];
    if (!(v00.x != v00.x) && !(v01.x != v01.x) && !(v10.x != v10.x)) {
        v00.y = vmap[(size_t)(y + rows) * v_stride + x];
        v01.y = vmap[(size_t)(y + rows) * v_stride + x + 1];
        v10.y = vmap[(size_t)(y + 1 + rows) * v_stride + x];
        v00.z = vmap[(size_t)(y + 2 * rows) * v_stride + x];
        v01.z = vmap[(size_t)(y + 2 * rows) * v_stride + x + 1];
        v10.z = vmap[(size_t)(y + 1 + 2 * rows) * v_stride + x];
        const f3 r = normalized(cross(v01 - v00, v10 - v00));
        nmap[(size_t)y * n_stride + x] = r.x;
        nmap[(size_t)(y + rows) * n_stride + x] = r.y;
        nmap[(size_t)(y + 2 * rows) * n_stride + x] = r.z;
    } else {
        nmap[(size_t)y * n_stride + x] = qnan();
    }
}
// create_vmap_px and create_nmap_px of one pixel in one pass over the depth image: the two neighbours' vertices are
// computed from their depths (the same expressions that give them their own vmap entries) instead of being read back
__device__ __forceinline__ void create_vmap_nmap_px(int x, int y, const float* __restrict__ depth, int cols, int rows,
                                                    float* __restrict__ vmap, float* __restrict__ nmap, float fx_inv, float fy_inv,
                                                    float cx, float cy, float cutoff) {
    if (x >= cols || y >= rows) return;
    const float z00 = depth[(size_t)y * cols + x];
    const float z01 = depth[(size_t)y * cols + min(x + 1, cols - 1)];
    const float z10 = depth[(size_t)min(y + 1, rows - 1) * cols + x];
    f3 v00, v01, v10;
    const bool ok00 = vmap_value(x, y, z00, fx_inv, fy_inv, cx, cy, cutoff, v00);
    if (ok00) {
        vmap[(size_t)y * cols + x] = v00.x;
        vmap[(size_t)(y + rows) * cols + x] = v00.y;
        vmap[(size_t)(y + 2 * rows) * cols + x] = v00.z;
    } else {
        vmap[(size_t)y * cols + x] = qnan();
    }
    if (x == cols - 1 || y == rows - 1) {
        nmap[(size_t)y * cols + x] = qnan();
        return;
    }
    const bool ok01 = vmap_value(x + 1, y, z01, fx_inv, fy_inv, cx, cy, cutoff, v01);
    const bool ok10 = vmap_value(x, y + 1, z10, fx_inv, fy_inv, cx, cy, cutoff, v10);
    if (ok00 && ok01 && ok10) {
        const f3 r = normalized(cross(v01 - v00, v10 - v00));
        nmap[(size_t)y * cols + x] = r.x;
        nmap[(size_t)(y + rows) * cols + x] = r.y;
        nmap[(size_t)(y + 2 * rows) * cols + x] = r.z;
    } else {
        nmap[(size_t)y * cols + x] = qnan();
    }
}
__global__ __launch_bounds__(256) void create_nmap_kernel(int rows, int cols, const float* __restrict__ vmap,
                                                          int v_stride, float* __restrict__ nmap, int n_stride) {
    MMF_PIXEL_XY();
    create_nmap_px(x, y, rows, cols, vmap, v_stride, nmap, n_stride);
}

// cudafuncs.cu:207-249 (tranformMapsKernel); in place is fine (one pixel per lane)
__device__ __forceinline__ void transform_maps_px(int x, int y, int rows, int cols, const float* vsrc, const float* nsrc,
                                                             int s_stride, m33 R, f3 t, float* vdst, float* ndst,
                                                             int d_stride) {
    if (x >= cols || y >= rows) return;
    f3 vs, vd = make_f3(qnan(), qnan(), qnan());
    vs.x = vsrc[(size_t)y * s_stride + x];
    if (!(vs.x != vs.x)) {
        vs.y = vsrc[(size_t)(y + rows) * s_stride + x];
        vs.z = vsrc[(size_t)(y + 2 * rows) * s_stride + x];
        vd = R * vs + t;
        vdst[(size_t)(y + rows) * d_stride + x] = vd.y;
        vdst[(size_t)(y + 2 * rows) * d_stride + x] = vd.z;
    }
    vdst[(size_t)y * d_stride + x] = vd.x;

    f3 ns, nd = make_f3(qnan(), qnan(), qnan());
    ns.x = nsrc[(size_t)y * s_stride + x];
    if (!(ns.x != ns.x)) {
        ns.y = nsrc[(size_t)(y + rows) * s_stride + x];
        ns.z = nsrc[(size_t)(y + 2 * rows) * s_stride + x];
        nd = R * ns;
        ndst[(size_t)(y + rows) * d_stride + x] = nd.y;
        ndst[(size_t)(y + 2 * rows) * d_stride + x] = nd.z;
    }
    ndst[(size_t)y * d_stride + x] = nd.x;
}
__global__ __launch_bounds__(256) void transform_maps_kernel(int rows, int cols, const float* vsrc, const float* nsrc,
                                                             int s_stride, m33 R, f3 t, float* vdst, float* ndst,
                                                             int d_stride) {
    MMF_PIXEL_XY();
    transform_maps_px(x, y, rows, cols, vsrc, nsrc, s_stride, R, t, vdst, ndst, d_stride);
}

// cudafuncs.cu:271-311 (copyMapsKernel): one float4 load per map per pixel (RGBA32F texel)
__device__ __forceinline__ void copy_maps_px(int x, int y, int rows, int cols, const float4* __restrict__ vsrc,
                                                        const float4* __restrict__ nsrc, float* __restrict__ vdst,
                                                        float* __restrict__ ndst, int d_stride) {
    if (x >= cols || y >= rows) return;
    const float4 v = vsrc[(size_t)y * cols + x];
    const float4 n = nsrc[(size_t)y * cols + x];
    f3 vd = make_f3(qnan(), qnan(), qnan()), nd = vd;
    if (!(v.z == 0)) {
        vd = make_f3(v.x, v.y, v.z);
        nd = make_f3(n.x, n.y, n.z);
    }
    vdst[(size_t)y * d_stride + x] = vd.x;
    vdst[(size_t)(y + rows) * d_stride + x] = vd.y;
    vdst[(size_t)(y + 2 * rows) * d_stride + x] = vd.z;
    ndst[(size_t)y * d_stride + x] = nd.x;
    ndst[(size_t)(y + rows) * d_stride + x] = nd.y;
    ndst[(size_t)(y + 2 * rows) * d_stride + x] = nd.z;
}
__global__ __launch_bounds__(256) void copy_maps_kernel(int rows, int cols, const float4* __restrict__ vsrc,
                                                        const float4* __restrict__ nsrc, float* __restrict__ vdst,
                                                        float* __restrict__ ndst, int d_stride) {
    MMF_PIXEL_XY();
    copy_maps_px(x, y, rows, cols, vsrc, nsrc, vdst, ndst, d_stride);
}

// cudafuncs.cu:366-417 (resizeMapKernel<normalize>): float2 loads cover the 2x2 footprint
// The value of destination pixel (x, y); `in` = the three source planes (srows rows each).  false: one of the four x taps is
// NaN -- the kernel then writes NaN to the x plane ONLY (the other two keep what they held), which every reader treats
// as "invalid" by looking at x first.
template <bool NORMALIZE>
__device__ __forceinline__ bool resize_map_value(int x, int y, int srows, const float* __restrict__ in, int i_stride, f3& n) {
    const int xs = x * 2, ys = y * 2;
    const float x00 = in[(size_t)(ys + 0) * i_stride + xs + 0];
    const float x01 = in[(size_t)(ys + 0) * i_stride + xs + 1];
    const float x10 = in[(size_t)(ys + 1) * i_stride + xs + 0];
    const float x11 = in[(size_t)(ys + 1) * i_stride + xs + 1];
    if ((x00 != x00) || (x01 != x01) || (x10 != x10) || (x11 != x11)) return false;
    n.x = (x00 + x01 + x10 + x11) / 4;
    const float y00 = in[(size_t)(ys + srows + 0) * i_stride + xs + 0];
    const float y01 = in[(size_t)(ys + srows + 0) * i_stride + xs + 1];
    const float y10 = in[(size_t)(ys + srows + 1) * i_stride + xs + 0];
    const float y11 = in[(size_t)(ys + srows + 1) * i_stride + xs + 1];
    n.y = (y00 + y01 + y10 + y11) / 4;
    const float z00 = in[(size_t)(ys + 2 * srows + 0) * i_stride + xs + 0];
    const float z01 = in[(size_t)(ys + 2 * srows + 0) * i_stride + xs + 1];
    const float z10 = in[(size_t)(ys + 2 * srows + 1) * i_stride + xs + 0];
    const float z11 = in[(size_t)(ys + 2 * srows + 1) * i_stride + xs + 1];
    n.z = (z00 + z01 + z10 + z11) / 4;
    if (NORMALIZE) n = normalized(n);
    return true;
}
template <bool NORMALIZE>
__device__ __forceinline__ void resize_map_px(int x, int y, int drows, int dcols, int srows, const float* __restrict__ in,
                                                         int i_stride, float* __restrict__ out, int o_stride) {
    if (x >= dcols || y >= drows) return;
    f3 n;
    if (!resize_map_value<NORMALIZE>(x, y, srows, in, i_stride, n)) {
        out[(size_t)y * o_stride + x] = qnan();
        return;
    }
    out[(size_t)y * o_stride + x] = n.x;
    out[(size_t)(y + drows) * o_stride + x] = n.y;
    out[(size_t)(y + 2 * drows) * o_stride + x] = n.z;
}
template <bool NORMALIZE>
__global__ __launch_bounds__(256) void resize_map_kernel(int drows, int dcols, int srows, const float* __restrict__ in,
                                                         int i_stride, float* __restrict__ out, int o_stride) {
    MMF_PIXEL_XY();
    resize_map_px<NORMALIZE>(x, y, drows, dcols, srows, in, i_stride, out, o_stride);
}

// 5x5 binomial weights (cudafuncs.cu:517-521) = outer product of {1,4,6,4,1}.  The reference
// cudaMallocs, uploads and frees this table on every call (:523-531); here the weight is
// computed (a table indexed per tap would live in scratch or constant memory).
__device__ __forceinline__ float binom5(int k) { return k == 2 ? 6.f : ((k == 1 || k == 3) ? 4.f : 1.f); }

// cudafuncs.cu:333-364 (pyrDownKernelGaussF), quirks kept: int `count`, the window is
// [max(0,2x-2), min(2x+3, cols-1)) and the weight index is mirrored from the clipped end (:358)
// `at(yy, xx)`: the source pixel (a plain image, or a value derived from another image in the same job: prep_batch.hpp)
template <typename At>
__device__ __forceinline__ float pyrdown_gauss_f_taps(int x, int y, int scols, int srows, At at) {
    const int tx = min(2 * x + 3, scols - 1);
    const int ty = min(2 * y + 3, srows - 1);
    // All 25 taps are loaded from clamped addresses BEFORE any is consumed, and the accumulation is
    // branch free: with the loads next to their data-dependent `if`, hipcc emitted 25 load / wait /
    // branch sequences -- 25 dependent round trips, 8-10 us for a 320x240 output.
    float tap[5][5];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) tap[dy][dx] = at(min(max(2 * y - 2 + dy, 0), srows - 1), min(max(2 * x - 2 + dx, 0), scols - 1));
    __builtin_amdgcn_sched_barrier(0);
    float sum = 0;
    int count = 0;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int cy = 2 * y - 2 + dy;
        const bool rowok = cy >= 0 && cy < ty;
        const float wy = binom5(ty - cy - 1);
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            const int cx = 2 * x - 2 + dx;
            const float s = tap[dy][dx];
            const bool use = rowok && cx >= 0 && cx < tx && !(s != s);
            const float w = wy * binom5(tx - cx - 1);
            sum = use ? sum + s * w : sum;
            count = use ? (int)((float)count + w) : count;
        }
    }
    return (float)(sum / (float)count);
}
__device__ __forceinline__ float pyrdown_gauss_f_value(int x, int y, const float* __restrict__ src, int s_stride, int scols,
                                                                int srows) {
    return pyrdown_gauss_f_taps(x, y, scols, srows, [&](int yy, int xx) { return src[(size_t)yy * s_stride + xx]; });
}
__device__ __forceinline__ void pyrdown_gauss_f_px(int x, int y, const float* __restrict__ src, int s_stride, int scols,
                                                              int srows, float* __restrict__ dst, int d_stride,
                                                              int dcols, int drows) {
    if (x >= dcols || y >= drows) return;
    dst[(size_t)y * d_stride + x] = pyrdown_gauss_f_value(x, y, src, s_stride, scols, srows);
}
__global__ __launch_bounds__(256) void pyrdown_gauss_f_kernel(const float* __restrict__ src, int s_stride, int scols,
                                                              int srows, float* __restrict__ dst, int d_stride,
                                                              int dcols, int drows) {
    MMF_PIXEL_XY();
    pyrdown_gauss_f_px(x, y, src, s_stride, scols, srows, dst, d_stride, dcols, drows);
}

// cudafuncs.cu:534-564 (pyrDownKernelIntensityGauss)
template <typename At>
__device__ __forceinline__ uint8_t pyrdown_uchar_gauss_taps(int x, int y, int scols, int srows, At at) {
    const int tx = min(2 * x + 3, scols - 1);
    const int ty = min(2 * y + 3, srows - 1);
    uint8_t tap[5][5];  // all taps in flight before the first use, see pyrdown_gauss_f_kernel
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) tap[dy][dx] = at(min(max(2 * y - 2 + dy, 0), srows - 1), min(max(2 * x - 2 + dx, 0), scols - 1));
    __builtin_amdgcn_sched_barrier(0);
    float sum = 0;
    int count = 0;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int cy = 2 * y - 2 + dy;
        const bool rowok = cy >= 0 && cy < ty;
        const float wy = binom5(ty - cy - 1);
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
            const int cx = 2 * x - 2 + dx;
            const uint8_t s = tap[dy][dx];
            const bool use = rowok && cx >= 0 && cx < tx && s > 0;
            const float w = wy * binom5(tx - cx - 1);
            sum = use ? sum + s * w : sum;
            count = use ? (int)((float)count + w) : count;
        }
    }
    const float q = sum / (float)count;
    return (q != q) ? (uint8_t)0 : (uint8_t)(unsigned)q;
}
__device__ __forceinline__ void pyrdown_uchar_gauss_px(int x, int y, const uint8_t* __restrict__ src, int s_stride,
                                                                  int scols, int srows, uint8_t* __restrict__ dst,
                                                                  int d_stride, int dcols, int drows) {
    if (x >= dcols || y >= drows) return;
    dst[(size_t)y * d_stride + x] =
        pyrdown_uchar_gauss_taps(x, y, scols, srows, [&](int yy, int xx) { return src[(size_t)yy * s_stride + xx]; });
}
__global__ __launch_bounds__(256) void pyrdown_uchar_gauss_kernel(const uint8_t* __restrict__ src, int s_stride,
                                                                  int scols, int srows, uint8_t* __restrict__ dst,
                                                                  int d_stride, int dcols, int drows) {
    MMF_PIXEL_XY();
    pyrdown_uchar_gauss_px(x, y, src, s_stride, scols, srows, dst, d_stride, dcols, drows);
}

// cudafuncs.cu:602-613 (verticesToDepthKernel)
__device__ __forceinline__ float vertex_depth_value(float z, float cutoff) { return (z > cutoff || z <= 0) ? qnan() : z; }
__device__ __forceinline__ void vertices_to_depth_px(int x, int y, const float4* __restrict__ vmap_rgba, int cols, int rows,
                                                                float* __restrict__ dst, int d_stride, float cutoff) {
    if (x >= cols || y >= rows) return;
    dst[(size_t)y * d_stride + x] = vertex_depth_value(vmap_rgba[(size_t)y * cols + x].z, cutoff);
}
__global__ __launch_bounds__(256) void vertices_to_depth_kernel(const float4* __restrict__ vmap_rgba, int cols, int rows,
                                                                float* __restrict__ dst, int d_stride, float cutoff) {
    MMF_PIXEL_XY();
    vertices_to_depth_px(x, y, vmap_rgba, cols, rows, dst, d_stride, cutoff);
}

// cudafuncs.cu:624-637 (bgr2IntensityKernel): channel order as uploaded
__device__ __forceinline__ uint8_t intensity_value(const uint8_t* __restrict__ p) {
    return (uint8_t)(int)((float)p[0] * 0.114f + (float)p[1] * 0.299f + (float)p[2] * 0.587f);
}
// the same from the three low bytes of one RGBA8 texel (one 4-byte load instead of three byte loads)
__device__ __forceinline__ uint8_t intensity_value_rgba(unsigned texel) {
    return (uint8_t)(int)((float)(texel & 0xFFu) * 0.114f + (float)((texel >> 8) & 0xFFu) * 0.299f + (float)((texel >> 16) & 0xFFu) * 0.587f);
}
__device__ __forceinline__ void image_to_intensity_px(int x, int y, const uint8_t* __restrict__ img, int i_stride,
                                                                 int channels, int cols, int rows,
                                                                 uint8_t* __restrict__ dst, int d_stride) {
    if (x >= cols || y >= rows) return;
    if (channels == 4 && (i_stride & 3) == 0)  // (wave uniform; hipMalloc'd images are 256-byte aligned)
        dst[(size_t)y * d_stride + x] = intensity_value_rgba(*reinterpret_cast<const unsigned*>(img + (size_t)y * i_stride + (size_t)x * 4));
    else
        dst[(size_t)y * d_stride + x] = intensity_value(img + (size_t)y * i_stride + (size_t)x * channels);
}
__global__ __launch_bounds__(256) void image_to_intensity_kernel(const uint8_t* __restrict__ img, int i_stride,
                                                                 int channels, int cols, int rows,
                                                                 uint8_t* __restrict__ dst, int d_stride) {
    MMF_PIXEL_XY();
    image_to_intensity_px(x, y, img, i_stride, channels, cols, rows, dst, d_stride);
}

// cudafuncs.cu:669-694 (applyKernel) with the tables of :702-708; border quirk kept
__device__ __forceinline__ void derivative_px(int x, int y, const uint8_t* __restrict__ src, int s_stride, int cols,
                                                         int rows, int16_t* __restrict__ dx, int dx_stride,
                                                         int16_t* __restrict__ dy, int dy_stride) {
    if (x >= cols || y >= rows) return;
    constexpr float gx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    constexpr float gy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
    // the 9 taps are loaded unconditionally (clamped), then the reference's running kernel index
    // (it only advances over the taps actually visited, cudafuncs.cu:681-690) is replayed
    float v[9];
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
        for (int di = -1; di <= 1; ++di)
            v[(dj + 1) * 3 + di + 1] = (float)src[(size_t)min(max(y + dj, 0), rows - 1) * s_stride + min(max(x + di, 0), cols - 1)];
    float dxv = 0, dyv = 0;
    int k = 8;
#pragma unroll
    for (int dj = -1; dj <= 1; ++dj)
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
            const int j = y + dj, i = x + di;
            if (j >= 0 && j <= rows - 1 && i >= 0 && i <= cols - 1) {
                const float gxk = gx[k], gyk = gy[k];
                dxv += v[(dj + 1) * 3 + di + 1] * gxk;
                dyv += v[(dj + 1) * 3 + di + 1] * gyk;
                --k;
            }
        }
    dx[(size_t)y * dx_stride + x] = (int16_t)dxv;
    dy[(size_t)y * dy_stride + x] = (int16_t)dyv;
}
__global__ __launch_bounds__(256) void derivative_kernel(const uint8_t* __restrict__ src, int s_stride, int cols,
                                                         int rows, int16_t* __restrict__ dx, int dx_stride,
                                                         int16_t* __restrict__ dy, int dy_stride) {
    MMF_PIXEL_XY();
    derivative_px(x, y, src, s_stride, cols, rows, dx, dx_stride, dy, dy_stride);
}

// cudafuncs.cu:729-747 (projectPointsKernel); AoS float3 output, dense
// cloud4 (optional): the same point as {X, Y, Z, 1 / Z} in one 16-byte record -- rgbStep divides by Z for every
// correspondence of every Gauss-Newton iteration (reduce.cu:523); the quotient is the same float each time, so the
// one-launch chain gathers it (one aligned 16-byte load) instead of dividing
// (the pixel's depth in a register: prep_batch.hpp computes it in the same job)
__device__ __forceinline__ void project_points_store(int x, int y, float z, int cols, float* __restrict__ cloud, float inv_fx,
                                                                float inv_fy, float cx, float cy, float4* __restrict__ cloud4) {
    const float X = (float)((x - cx) * z * inv_fx), Y = (float)((y - cy) * z * inv_fy);
    if (cloud) {  // (the AoS copy: three 12-byte-strided stores per pixel; the chains read the records)
        float* c = cloud + ((size_t)y * cols + x) * 3;
        c[0] = X;
        c[1] = Y;
        c[2] = z;
    }
    if (cloud4) cloud4[(size_t)y * cols + x] = make_float4(X, Y, z, 1.0f / z);
}
__device__ __forceinline__ void project_points_px(int x, int y, const float* __restrict__ depth, int d_stride, int cols,
                                                             int rows, float* __restrict__ cloud, float inv_fx,
                                                             float inv_fy, float cx, float cy, float4* __restrict__ cloud4 = nullptr) {
    if (x >= cols || y >= rows) return;
    project_points_store(x, y, depth[(size_t)y * d_stride + x], cols, cloud, inv_fx, inv_fy, cx, cy, cloud4);
}
__global__ __launch_bounds__(256) void project_points_kernel(const float* __restrict__ depth, int d_stride, int cols,
                                                             int rows, float* __restrict__ cloud, float inv_fx,
                                                             float inv_fy, float cx, float cy, float4* __restrict__ cloud4) {
    MMF_PIXEL_XY();
    project_points_px(x, y, depth, d_stride, cols, rows, cloud, inv_fx, inv_fy, cx, cy, cloud4);
}

}  // namespace mmf
