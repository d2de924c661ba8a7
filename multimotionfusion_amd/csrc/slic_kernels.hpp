// slic_kernels.hpp -- super-pixel resampling for the segmentation on the device:
// Slic::downsample<float>(image, channel), Slic::downsampleThresholded<float>, Slic::downsample() and
// Slic::upsample<unsigned char> (Core/Segmentation/Slic.h:48-146, Slic.cpp:72-112), which the reference
// runs on the CPU after downloading the per-model ICP-error and vertex-confidence textures
// (Segmentation.cpp:218-221: 6 MB per model per frame).  Here the maps stay where the tracker left them and
// only (W/S) x (H/S) values per map leave the GPU.
//
// The float sums of the reference run in pixel order with one `+=` per pixel, and float addition does not
// reassociate, so a wave owns one super-pixel and walks that label's bounding box in row-major order: 64
// labels per step, a ballot of the matches, then the matching lanes' values are added one by one in lane
// (= pixel) order on broadcast values.  Integer sums (pixel counts, RGB) are order free and use atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

namespace mmf {

struct SlicBox {
    int min_x, min_y, max_x, max_y;
};

__global__ __launch_bounds__(256) void slic_reset_kernel(int n, SlicBox* __restrict__ box, int* __restrict__ counts,
                                                         int* __restrict__ rgb_sums) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    box[s] = SlicBox{INT_MAX, INT_MAX, -1, -1};
    counts[s] = 0;
    rgb_sums[3 * s] = rgb_sums[3 * s + 1] = rgb_sums[3 * s + 2] = 0;
}

// spixelCounts (Slic.cpp:76-79) + the bounding box of every label (+ the RGB sums of Slic.cpp:93-94 when
// rgb != nullptr).  Labels outside [0, n) are ignored (the reference would write out of bounds).
// A wave covers 64 consecutive pixels, which belong to a handful of labels in runs: only the first lane of a
// run touches memory (5 atomics per run instead of per pixel -- the per-pixel form spent 200 us at 640x480
// queueing on 1200 addresses); integer run sums come from a wave prefix sum.
__device__ __forceinline__ int slic_wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(v, d);
        v += lane >= d ? up : 0;
    }
    return v;
}

__global__ __launch_bounds__(256) void slic_census_kernel(const int* __restrict__ labels, int W, int H, int n,
                                                          SlicBox* box, int* counts, const uint8_t* __restrict__ rgb,
                                                          int channels, int* rgb_sums) {
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool live = i < W * H;
    const int s = live ? labels[i] : -1;
    const int y = live ? i / W : -1, x = live ? i - y * W : -1;
    // run starts: the label or the image row changes (lane 0 always starts one)
    const int prev_s = __shfl_up(s, 1), prev_y = __shfl_up(y, 1);
    const bool start = lane == 0 || prev_s != s || prev_y != y;
    const unsigned long long starts = __ballot(start);
    const unsigned long long later = lane == 63 ? 0ull : (starts >> (lane + 1)) << (lane + 1);
    const int next_start = later ? __ffsll((long long)later) - 1 : 64;
    const int len = next_start - lane;  // meaningful in start lanes
    int c0 = 0, c1 = 0, c2 = 0;
    if (rgb != nullptr) {  // wave uniform
        if (live) {
            const uint8_t* px = rgb + (size_t)i * channels;
            c0 = px[2], c1 = px[1], c2 = px[0];
        }
        const int p0 = slic_wave_inclusive_scan(c0, lane), p1 = slic_wave_inclusive_scan(c1, lane);
        const int p2 = slic_wave_inclusive_scan(c2, lane);
        const int last = next_start - 1;  // last lane of this lane's run (for start lanes)
        c0 = __shfl(p0, last) - (p0 - c0), c1 = __shfl(p1, last) - (p1 - c1), c2 = __shfl(p2, last) - (p2 - c2);
    }
    if (!live || !start || s < 0 || s >= n) return;
    atomicAdd(&counts[s], len);
    atomicMin(&box[s].min_x, x), atomicMin(&box[s].min_y, y);
    atomicMax(&box[s].max_x, x + len - 1), atomicMax(&box[s].max_y, y);
    if (rgb != nullptr) {
        atomicAdd(&rgb_sums[3 * s], c0), atomicAdd(&rgb_sums[3 * s + 1], c1), atomicAdd(&rgb_sums[3 * s + 2], c2);
    }
}

// one wave per super-pixel: sums[s] = sum of image[.., channel] over the label's pixels in pixel order;
// dcounts[s] = number of pixels added (all of them, or those above the threshold)
__global__ __launch_bounds__(256) void slic_sum_kernel(const int* __restrict__ labels, int W, int n,
                                                       const float* __restrict__ image, int channels, int channel,
                                                       int thresholded, float min_threshold,
                                                       const SlicBox* __restrict__ box, float* __restrict__ sums,
                                                       int* __restrict__ dcounts) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n) return;  // wave uniform
    const SlicBox b = box[s];
    float sum = 0.f;
    int cnt = 0;
    for (int y = b.min_y; y <= b.max_y; ++y)
        for (int xb = b.min_x; xb <= b.max_x; xb += 64) {
            const int x = xb + lane;
            bool hit = x <= b.max_x && labels[y * W + x] == s;
            const float v = hit ? image[((size_t)y * W + x) * channels + channel] : 0.f;
            if (thresholded) hit = hit && v > min_threshold;
            unsigned long long m = __ballot(hit);
            cnt += __popcll(m);
            while (m) {  // wave uniform: lanes in ascending order = pixels in ascending order
                const int l = __ffsll((long long)m) - 1;
                sum = sum + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
                m &= m - 1;
            }
        }
    if (lane == 0) sums[s] = sum, dcounts[s] = cnt;
}

// Slic.h:191-209, literally (mapToHigh(index) divides by spixelY)
__device__ __forceinline__ int slic_resample_empty_index(const int* __restrict__ labels, int W, int H, int S, int spx,
                                                         int spy, int index) {
    const int hx = index % spx, hy = index / spy;
    int cx = (int)(hx * S + S * 0.5), cy = (int)(hy * S + S * 0.5);
    if (cy >= H) cy = H - 1;
    if (cx >= W) cx = W - 1;
    return labels[cx + cy * W];
}

// The reference finishes with `res[index] = res[readIndex] / cnt` IN PLACE over ascending index
// (Slic.h:72-81,113-122): an empty super-pixel whose substitute has a LOWER index reads that one's final
// value, otherwise its raw sum.  Each thread resolves its own chain of substitutes (strictly descending, so
// it ends) and applies the divisions from the innermost outwards -- the same values without the serial walk.
__global__ __launch_bounds__(256) void slic_finish_kernel(const int* __restrict__ labels, int W, int H, int S, int spx,
                                                          int spy, const int* __restrict__ counts,
                                                          const int* __restrict__ cnt_used,
                                                          const float* __restrict__ sums, float* __restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x, n = spx * spy;
    if (s >= n) return;
    int depth = 0, idx = s;
    bool raw = false;
    while (cnt_used[idx] == 0) {
        int r = slic_resample_empty_index(labels, W, H, S, spx, spy, idx);
        if (r < 0 || r >= n) r = idx;  // a label the census ignored: no substitute
        ++depth;
        const bool lower = r < idx;
        idx = r;
        if (!lower) {
            raw = true;
            break;
        }
    }
    float v = raw ? sums[idx] : sums[idx] / (float)cnt_used[idx];
    for (int k = depth; k >= 1; --k) {  // divisor of level k = counts[substitute reached after k steps]
        int j = s;
        for (int q = 0; q < k; ++q) {
            const int r = slic_resample_empty_index(labels, W, H, S, spx, spy, j);
            j = (r < 0 || r >= n) ? j : r;
        }
        v = v / (float)counts[j];
    }
    out[s] = v;
}

// Slic.cpp:96-106: integer means of the three channels, saturated to u8
__global__ __launch_bounds__(256) void slic_finish_rgb_kernel(const int* __restrict__ labels, int W, int H, int S, int spx,
                                                              int spy, const int* __restrict__ counts,
                                                              const int* __restrict__ rgb_sums, uint8_t* __restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x, n = spx * spy;
    if (s >= n) return;
    int cnt = counts[s], r = s;
    if (cnt == 0) {
        r = slic_resample_empty_index(labels, W, H, S, spx, spy, s);
        if (r < 0 || r >= n) r = s;
        cnt = counts[r];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v = cnt != 0 ? rgb_sums[3 * r + k] / cnt : 0;
        out[3 * s + k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

// Slic.h:133-146 with unsigned char
__global__ __launch_bounds__(256) void slic_upsample_u8_kernel(const int* __restrict__ labels, int npix, int n,
                                                               const uint8_t* __restrict__ map, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int s = labels[i];
    out[i] = (s >= 0 && s < n) ? map[s] : 0;
}

}  // namespace mmf
