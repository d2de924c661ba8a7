// superpoint_kernels.hpp -- the SuperPoint keypoint network and its post-processing on gfx950.
//
// What the reference gets from `SuperPoint::getFeatures` (Core/MultiMotionFusion.cpp:78,233; the network
// itself lives in the un-vendored super_point_inference package and runs through libtorch there):
//   VGG encoder (8 x conv3x3+ReLU, three 2x2 max pools), detector head (conv3x3+ReLU, conv1x1 -> 65),
//   descriptor head (conv3x3+ReLU, conv1x1 -> 256, L2 normalised), softmax + depth-to-space heat map,
//   greedy non-maximum suppression, bilinear descriptor sampling.
//
// The convolutions are the only dense contractions of the whole project and run on the matrix cores as an
// implicit GEMM in f32 (v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate -- the reference computes in
// fp32, so no reduced-precision operand is introduced):
//   * activations are channels-last [H][W][C], so the K dimension (tap, channel) is contiguous per pixel;
//   * one workgroup (4 waves) owns an 8 x 16 pixel tile and 32*NT output channels; wave w owns rows 2w, 2w+1.
//     The 32 MFMA rows of a wave are eight 2x2 pixel blocks (row = 4*block + 2*dy + dx), which puts the four
//     pixels of a max-pool window into ONE lane's accumulator registers: the pool is a 4-way max in the
//     epilogue and the un-pooled activation never exists in memory;
//   * K is walked in blocks of 32 input channels x 9 taps.  The 10 x 18 pixel halo of a block is staged once
//     in LDS (34.5 KB) and serves all nine taps by address offset.  The weights never touch LDS: they are
//     pre-packed on the host in exactly the order the lanes consume them, so every wave streams its B
//     operands as 1 KB coalesced loads from L2/L1 (all workgroups walk the same few hundred KB) two MFMA
//     groups ahead of their use -- the K loop has no barrier except the two around a halo refill;
//   * LDS images are laid out for conflict-free ds_read_b128: a pixel is 36 floats (32 + 4 pad), a halo row
//     24 pixels (18 + 6 pad) -- with these strides the 16 lanes of every b128 group hit 16 distinct 16-B
//     slots (MI355X_MICROARCH.md, LDS).  The channels of a pixel are stored even/odd de-interleaved in groups
//     of 8 so that one b128 read hands a lane its operand for four consecutive MFMAs;
//   * the accumulation order of one output is a single fmaf chain (block, tap, channel) -- see the oracle's
//     header -- which the MFMA reproduces exactly, so parity is bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmf_math.h"

namespace mmf {

typedef float sp_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kSpTileH = 8, kSpTileW = 16;  // output pixels per workgroup
constexpr int kSpRowPix = 24;               // LDS pixels per halo row (18 used; 24 = 8 mod 16 keeps b128 reads conflict free)
constexpr int kSpPixF4 = 9;                 // LDS float4 per pixel (32 channels + 4 floats pad)
constexpr int kSpKBlock = 32;               // input channels per K block

struct SpConvArgs {
    const float* in;     // [H][W][in_stride], first `cin` channels used
    const float* wpack;  // packed weights (sp_pack_weights on the host)
    const float* bias;   // [cout]
    float* out;          // [H or H/2][W or W/2][out_stride]
    int in_stride, out_stride;
    int H, W, cin, cout;
    int relu;
};

#ifdef SP_PROBE_STAMPS  // diagnostic builds only (tools/sp_conv_probe.hip): per-workgroup phase stamps
__device__ unsigned long long* g_sp_stamps = nullptr;
#define SP_STAMP(i)                                                                                              \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (g_sp_stamps && threadIdx.x == 0) {                                                                   \
            unsigned long long t_;                                                                               \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
            g_sp_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = t_;        \
        }                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
#else
#define SP_STAMP(i) \
    do {            \
    } while (0)
#endif

template <int NT, int TAPS, bool POOL>
__device__ __forceinline__ void sp_conv_mfma_body(const SpConvArgs& p, const int z) {
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int HR = kSpTileH + 2 * HALO, HC = kSpTileW + 2 * HALO;
    constexpr int UNITS = HR * HC * 4;  // (pixel, 8-channel group) staging units of one K block
    constexpr int UPT = (UNITS + 255) / 256;
    constexpr int GROUPS = TAPS * 4;  // MFMA groups (8 channels of one tap) per K block
    __shared__ float4 lds_a[HR * kSpRowPix * kSpPixF4];

    SP_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * kSpTileW, y0 = blockIdx.y * kSpTileH;
    const int chunks = p.cin / kSpKBlock;
    const int last_group = chunks * GROUPS - 1;
    // this lane's weight operands: one float4 per (group, output tile), 1 KB contiguous per wave and load
    const float4* __restrict__ wp =
        reinterpret_cast<const float4*>(p.wpack) + (size_t)z * chunks * GROUPS * NT * 64 + lane;

    // staging addresses of this thread's units (the same for every K block, which only shifts the channel)
    int a_src[UPT], a_dst[UPT];
#pragma unroll
    for (int j = 0; j < UPT; ++j) {
        const int u = tid + 256 * j;
        const int pix = u >> 2, q = u & 3;
        const int r = pix / HC, c = pix - r * HC;
        const int gy = y0 - HALO + r, gx = x0 - HALO + c;
        const bool inside = u < UNITS && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_src[j] = inside ? (gy * p.W + gx) * p.in_stride + 8 * q : -1;
        a_dst[j] = u < UNITS ? (r * kSpRowPix + c) * kSpPixF4 + 2 * q : -1;
    }
    float4 a_lo[UPT], a_hi[UPT];
#define SP_LOAD_A(chunk)                                                                                    \
    _Pragma("unroll") for (int j = 0; j < UPT; ++j) {                                                       \
        a_lo[j] = a_hi[j] = make_float4(0.f, 0.f, 0.f, 0.f);                                                \
        if (a_src[j] >= 0) {                                                                                \
            const float4* s_ = reinterpret_cast<const float4*>(p.in + a_src[j] + (chunk) * kSpKBlock);      \
            a_lo[j] = s_[0], a_hi[j] = s_[1];                                                               \
        }                                                                                                   \
    }
    // even channels of a group of 8 first, then the odd ones
#define SP_STORE_A()                                                                                        \
    _Pragma("unroll") for (int j = 0; j < UPT; ++j) if (a_dst[j] >= 0) {                                    \
        lds_a[a_dst[j]] = make_float4(a_lo[j].x, a_lo[j].z, a_hi[j].x, a_hi[j].z);                          \
        lds_a[a_dst[j] + 1] = make_float4(a_lo[j].y, a_lo[j].w, a_hi[j].y, a_hi[j].w);                      \
    }

    // MFMA row of this lane: i = 4*block + 2*dy + dx; the lane half kh supplies the odd k of each pair
    const int i = lane & 31, kh = lane >> 5;
    const int a_base = ((2 * wave + ((i >> 1) & 1)) * kSpRowPix + 2 * (i >> 2) + (i & 1)) * kSpPixF4 + kh;

    sp_f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    // weights: straight from L2/L1 into registers, two groups ahead of their use (every wave of the chip
    // walks the same few hundred KB, so they stay cache resident); no LDS copy and no barrier per step
    float4 b[3][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[0][t] = wp[(size_t)t * 64], b[1][t] = wp[(size_t)(min(1, last_group) * NT + t) * 64];
#ifndef SP_PROBE_NO_PROLOGUE
    SP_LOAD_A(0);
    SP_STORE_A();
    __syncthreads();
#endif
    SP_STAMP(1);

    for (int chunk = 0; chunk < chunks; ++chunk) {
#ifdef SP_PROBE_NO_STAGE  // diagnostic builds only (tools/sp_conv_probe.hip): results are wrong, only the time matters
        const bool more = false;
#else
        const bool more = chunk + 1 < chunks;
#endif
        if (more) SP_LOAD_A(chunk + 1);  // lands during this block's MFMAs
        float4 a[2];
        a[0] = lds_a[a_base];
#pragma unroll
        for (int q = 0; q < GROUPS; ++q) {
            const int gi = chunk * GROUPS + q;
            {  // operands of the next two groups
                const int gn = min(gi + 2, last_group);
#ifndef SP_PROBE_NO_B
#pragma unroll
                for (int t = 0; t < NT; ++t) b[(q + 2) % 3][t] = wp[(size_t)(gn * NT + t) * 64];
#endif
#ifndef SP_PROBE_NO_A
                if (q + 1 < GROUPS) {
                    const int tap = (q + 1) >> 2, g = (q + 1) & 3;
                    const int tap_off = TAPS == 9 ? ((tap / 3) * kSpRowPix + tap % 3) * kSpPixF4 : 0;
                    a[(q + 1) & 1] = lds_a[a_base + tap_off + 2 * g];
                }
#else
                a[(q + 1) & 1] = a[q & 1];
                (void)gn;
#endif
            }
            const float4 av = a[q & 1];
            // k ascending within every accumulator's chain; the NT independent chains interleave.  The
            // scheduling fences keep the operand loads above ahead of this group's MFMAs (left alone, the
            // compiler sinks them to just before their use and serialises the chains)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[q % 3][t].x, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[q % 3][t].y, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[q % 3][t].z, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[q % 3][t].w, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (GROUPS % 3 != 0) {  // keep the rotation of the three weight buffers aligned with q = 0
#pragma unroll
            for (int r = 0; r < (3 - GROUPS % 3) % 3; ++r)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 t0 = b[0][t];
                    b[0][t] = b[2][t], b[2][t] = b[1][t], b[1][t] = t0;
                }
        }
        if (more) {
            __syncthreads();  // every wave is done with this block's halo
            SP_STORE_A();
            __syncthreads();
        }
    }
#undef SP_LOAD_A
#undef SP_STORE_A
    SP_STAMP(2);

    // epilogue: accumulator register v of lane (n, kh) is MFMA row (v&3) + 8*(v>>2) + 4*kh = pixel block
    // 2*(v>>2) + kh, (dy, dx) = ((v>>1)&1, v&1); column n = output channel
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = (z * NT + t) * 32 + i;
        const bool co_ok = co < p.cout;
        const float bias = co_ok ? p.bias[co] : 0.f;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int bx = 2 * blk + kh;
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = acc[t][4 * blk + k] + bias;
                if (p.relu) o[k] = o[k] < 0.f ? 0.f : o[k];
            }
#ifdef SP_PROBE_NO_EPILOGUE
            if (o[0] + o[1] + o[2] + o[3] == 123.456f) p.out[co] = o[0];  // keeps the accumulators alive, never stores
            continue;
#endif
            if (POOL) {
                const int py = (y0 >> 1) + wave, px = (x0 >> 1) + bx;
                const float m0 = o[0] > o[1] ? o[0] : o[1], m1 = o[2] > o[3] ? o[2] : o[3];
                if (co_ok && py < (p.H >> 1) && px < (p.W >> 1))
                    p.out[((size_t)py * (p.W >> 1) + px) * p.out_stride + co] = m0 > m1 ? m0 : m1;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int oy = y0 + 2 * wave + (k >> 1), ox = x0 + 2 * bx + (k & 1);
                    if (co_ok && oy < p.H && ox < p.W) p.out[((size_t)oy * p.W + ox) * p.out_stride + co] = o[k];
                }
            }
        }
    }
    SP_STAMP(3);
}

template <int NT, int TAPS, bool POOL>
__global__ __launch_bounds__(256) void sp_conv_mfma_kernel(SpConvArgs p) {
    sp_conv_mfma_body<NT, TAPS, POOL>(p, blockIdx.z);
}

// two layers of the same shape class in one launch (the two 1x1 heads: independent of each other, each too
// small to fill the chip): the first `za` z-slices belong to layer a, the rest to layer b
struct SpConvPair {
    SpConvArgs a, b;
    int za;
};

template <int NT, int TAPS, bool POOL>
__global__ __launch_bounds__(256) void sp_conv_mfma_pair_kernel(SpConvPair q) {
    if ((int)blockIdx.z < q.za)  // workgroup uniform
        sp_conv_mfma_body<NT, TAPS, POOL>(q.a, blockIdx.z);
    else
        sp_conv_mfma_body<NT, TAPS, POOL>(q.b, blockIdx.z - q.za);
}

// ---- network input: [0,1] grey from 1-, 3- or 4-channel u8 (first three channels = R, G, B) ---------------
__device__ __forceinline__ float sp_grey(const uint8_t* __restrict__ img, size_t i, int channels) {
    if (channels == 1) return (float)img[i] / 255.0f;
    const uint8_t* px = img + i * channels;
    return ((0.299f * (float)px[0] + 0.587f * (float)px[1]) + 0.114f * (float)px[2]) / 255.0f;
}

// ---- conv1a: one input channel, 64 outputs.  K = 9 is no contraction to speak of: VALU, four lanes per pixel
// (16 output channels each) so that a wave stores 4 KB contiguous per instruction group.  The grey conversion
// of the nine taps happens here (a few flops per tap; no separate input plane, no extra launch) ---------------
__global__ __launch_bounds__(256) void sp_conv1a_kernel(const uint8_t* __restrict__ img, int channels, int H, int W,
                                                        const float* __restrict__ w_tap_co /* [9][64] */,
                                                        const float* __restrict__ bias, float* __restrict__ out) {
    __shared__ float4 wl[9 * 16];
    if (threadIdx.x < 9 * 16) wl[threadIdx.x] = reinterpret_cast<const float4*>(w_tap_co)[threadIdx.x];
    __syncthreads();
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int pix = gid >> 2, q = gid & 3;
    if (pix >= H * W) return;
    const int y = pix / W, x = pix - y * W;
    float v[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
        v[tap] = inside ? sp_grey(img, (size_t)yy * W + xx, channels) : 0.f;
    }
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int c = 0; c < 16; c += 4) {
            const float4 w = wl[tap * 16 + 4 * q + (c >> 2)];
            acc[c] = fmaf(v[tap], w.x, acc[c]), acc[c + 1] = fmaf(v[tap], w.y, acc[c + 1]);
            acc[c + 2] = fmaf(v[tap], w.z, acc[c + 2]), acc[c + 3] = fmaf(v[tap], w.w, acc[c + 3]);
        }
    float4* o = reinterpret_cast<float4*>(out + (size_t)pix * 64 + 16 * q);
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = acc[c + k] + bias[16 * q + c + k];
            r[k] = r[k] < 0.f ? 0.f : r[k];
        }
        o[c >> 2] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

__device__ __forceinline__ float sp_lane_value(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---- L2 normalisation over channels: one wave per pixel, lane L holds channels 4L..4L+3.  The chain of
// squares is sequential (as in the oracle); every lane walks it on broadcast values, so all lanes end with
// the same norm and no LDS or barrier is involved --------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void sp_l2_normalize_kernel(float* __restrict__ desc, int npix) {
    static_assert(C == 256, "one float4 per lane");
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= npix) return;  // wave uniform
    float4* d = reinterpret_cast<float4*>(desc + (size_t)p * C) + lane;
    float4 v = *d;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 64; ++l) {
        const float x = sp_lane_value(v.x, l), y = sp_lane_value(v.y, l), z = sp_lane_value(v.z, l), w = sp_lane_value(v.w, l);
        s = fmaf(x, x, s), s = fmaf(y, y, s), s = fmaf(z, z, s), s = fmaf(w, w, s);
    }
    const float n = sqrtf(s);
    v.x = v.x / n, v.y = v.y / n, v.z = v.z / n, v.w = v.w / n;
    *d = v;
}

// ---- heat map: softmax over the 65 logits of a cell (+1e-5 in the denominator), dustbin dropped, 8x8
// depth-to-space.  One wave per cell, lane c owns logit c (lane 0 also the dustbin); the sum runs in channel
// order on broadcast values ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_heatmap_kernel(const float* __restrict__ semi, int Hc, int Wc,
                                                        float* __restrict__ heat) {
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (cell >= Hc * Wc) return;  // wave uniform
    const int hc = cell / Wc, wc = cell - hc * Wc;
    const float* s = semi + (size_t)cell * 65;
    const float e = mmf_expf(s[lane]);
    const float dust = mmf_expf(s[64]);  // same address in every lane
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 64; ++c) sum = sum + sp_lane_value(e, c);
    sum = sum + dust;
    sum = sum + 0.00001f;
    heat[(size_t)(hc * 8 + (lane >> 3)) * (Wc * 8) + wc * 8 + (lane & 7)] = e / sum;
}

// ---- greedy non-maximum suppression as a fixed point --------------------------------------------------------
// nms_fast visits candidates strongest first and keeps one iff no KEPT stronger candidate lies within the
// Chebyshev radius.  That outcome is the unique fixed point of: "kept iff every stronger candidate in the
// window is suppressed; suppressed iff some stronger candidate in the window is kept".  Decisions are final
// once taken, so the passes may update in place; priority = (confidence desc, row-major index asc).
enum : uint8_t { SP_NONE = 0, SP_UNDECIDED = 1, SP_SUPPRESSED = 2, SP_KEPT = 3 };

__global__ __launch_bounds__(256) void sp_nms_init_kernel(const float* __restrict__ heat, int n, float conf_thresh,
                                                          uint8_t* __restrict__ state) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) state[i] = heat[i] >= conf_thresh ? SP_UNDECIDED : SP_NONE;
}

__global__ __launch_bounds__(256) void sp_nms_pass_kernel(const float* __restrict__ heat, int H, int W, int dist,
                                                          uint8_t* state, unsigned* __restrict__ undecided) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    if (__builtin_nontemporal_load(&state[i]) != SP_UNDECIDED) return;
    const int y = i / W, x = i - y * W;
    const float ci = heat[i];
    bool stronger_kept = false, stronger_open = false;
    for (int yy = max(y - dist, 0); yy <= min(y + dist, H - 1); ++yy)
        for (int xx = max(x - dist, 0); xx <= min(x + dist, W - 1); ++xx) {
            const int j = yy * W + xx;
            const uint8_t sj = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sj == SP_NONE || sj == SP_SUPPRESSED || j == i) continue;
            const float cj = heat[j];
            if (!(cj > ci || (cj == ci && j < i))) continue;
            stronger_kept |= sj == SP_KEPT;
            stronger_open |= sj == SP_UNDECIDED;
        }
    if (stronger_kept)
        __hip_atomic_store(&state[i], (uint8_t)SP_SUPPRESSED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (!stronger_open)
        __hip_atomic_store(&state[i], (uint8_t)SP_KEPT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        atomicAdd(undecided, 1u);
}

// kept and outside the border band -> flag (row-major order)
__global__ __launch_bounds__(256) void sp_keep_flag_kernel(const uint8_t* __restrict__ state, int H, int W, int border,
                                                           unsigned* __restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    flags[i] = (state[i] == SP_KEPT && x >= border && x < W - border && y >= border && y < H - border) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void sp_keep_scatter_kernel(const unsigned* __restrict__ flags,
                                                              const unsigned* __restrict__ prefix,
                                                              const float* __restrict__ heat, int n, int W, int max_out,
                                                              int* __restrict__ xy, float* __restrict__ conf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flags[i]) return;
    const unsigned k = prefix[i];
    if (k >= (unsigned)max_out) return;
    xy[2 * k] = i % W, xy[2 * k + 1] = i / W;
    conf[k] = heat[i];
}

// ---- descriptor sampling: one workgroup per keypoint, one thread per channel ------------------------------
__global__ __launch_bounds__(256) void sp_sample_kernel(const float* __restrict__ desc, int Hc, int Wc,
                                                        const int* __restrict__ xy, int H, int W,
                                                        float* __restrict__ out) {
    __shared__ float vals[256];
    __shared__ float norm;
    const int k = blockIdx.x, c = threadIdx.x;
    const float gx = (float)xy[2 * k] / ((float)W / 2.0f) - 1.0f, gy = (float)xy[2 * k + 1] / ((float)H / 2.0f) - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(Wc - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(Hc - 1);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.0f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.0f) - iy;
    const float wnw = wx0 * wy0, wne = wx1 * wy0, wsw = wx0 * wy1, wse = wx1 * wy1;
    const bool x0ok = x0 >= 0 && x0 < Wc, x1ok = x1 >= 0 && x1 < Wc, y0ok = y0 >= 0 && y0 < Hc, y1ok = y1 >= 0 && y1 < Hc;
    const float nw = (x0ok && y0ok) ? desc[((size_t)y0 * Wc + x0) * 256 + c] : 0.f;
    const float ne = (x1ok && y0ok) ? desc[((size_t)y0 * Wc + x1) * 256 + c] : 0.f;
    const float sw = (x0ok && y1ok) ? desc[((size_t)y1 * Wc + x0) * 256 + c] : 0.f;
    const float se = (x1ok && y1ok) ? desc[((size_t)y1 * Wc + x1) * 256 + c] : 0.f;
    const float v = ((nw * wnw + ne * wne) + sw * wsw) + se * wse;
    vals[c] = v;
    __syncthreads();
    if (c == 0) {
        float s = 0.f;
        for (int q = 0; q < 256; ++q) s = fmaf(vals[q], vals[q], s);
        norm = sqrtf(s);
    }
    __syncthreads();
    out[(size_t)k * 256 + c] = v / norm;
}

}  // namespace mmf
