// track_kernels.hpp -- gfx950 kernels of the dense-tracking reductions:
//   icp_kernel          <- icpKernel + reduceSum            (Core/Cuda/reduce.cu:231-473)
//   rgb_residual_kernel <- residualKernel + reduceSum(int2) (Core/Cuda/reduce.cu:722-945)
//   rgb_step_kernel     <- rgbKernel + reduceSum            (Core/Cuda/reduce.cu:477-661)
//   so3_kernel          <- so3Kernel + reduceSum            (Core/Cuda/reduce.cu:947-1150)
// plus the single-lane bookkeeping kernels of the device-resident Gauss-Newton loop.
//
// All are HBM/latency bound (about 110 flop per 48 bytes for ICP), so the design goals are:
// coalesced 16-byte loads of the planar maps, all gathers of a pixel group issued together,
// one launch per reduction (grid_reduce.hpp) and no host synchronisation between iterations.
#pragma once
#include "device_math.hpp"
#include "extent.hpp"
#include "grid_reduce.hpp"
#include "icp_kernels.hpp"
#include "odom_state.hpp"
#include "pose_algebra.hpp"
#include "prep_batch.hpp"
#include "frame_rider.hpp"

namespace mmf {

enum FinishMode { FINISH_RAW = 0, FINISH_GN = 1 };

// Diagnostic builds only (tools/rgb_step_probe.py): per-workgroup phase stamps of the 100 MHz
// constant clock into a side buffer nothing else reads.  Compiled out of the library.
#ifdef MMF_STAMPS
__device__ unsigned long long* g_mmf_dbg = nullptr;
#define MMF_STAMP(i)                                                                          \
    do {                                                                                      \
        if (g_mmf_dbg && threadIdx.x == 0) g_mmf_dbg[blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MMF_STAMP(i) \
    do {             \
    } while (0)
#endif

constexpr int kBlock = 256;  // threads per workgroup of the photometric / SO3 reductions

// accumulate the 27 upper-triangular products of a 7-vector + residual^2 + inlier flag
// in the member order of JtJJtrSE3 (types.cuh:101-112, reduce.cu:331-365).  The Jacobian row is
// computed without contraction (bit-exact with the oracle); only this running sum uses an
// explicit fused multiply-add -- sums are compared within a summation-order tolerance anyway,
// and it halves the accumulate instruction count.
__device__ __forceinline__ void accumulate_se3(float (&sum)[29], const float (&row)[7], float found) {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 7; ++j) {
            sum[k] = __builtin_fmaf(row[i], row[j], sum[k]);
            ++k;
        }
    sum[27] = __builtin_fmaf(row[6], row[6], sum[27]);
    sum[28] = sum[28] + found;
}

// the ICP producer (icp_kernels.hpp): W pixels per lane vector x NV vectors per lane.  MULTI: the grid does not cover
// the image (more pixels than kMaxIcpGrid workgroups take in one pass): every lane walks the image with the grid's stride.
template <int W, int NV, int BLOCK, bool PACKED, int MODE, bool MULTI = false>
__global__ __launch_bounds__(BLOCK) void icp_kernel2(const OdomState* __restrict__ st, IcpArgs a,
                                                     float* __restrict__ partials) {
    __shared__ GridReduceLds<float, BLOCK> lds;
    using T = typename std::conditional<W == 2, v2f, float>::type;
    if (a.err_map)
        icp_block2<T, NV, BLOCK, PACKED, true, MODE == FINISH_GN, MULTI>(st, a, partials, lds, blockIdx.x, gridDim.x);
    else
        icp_block2<T, NV, BLOCK, PACKED, false, MODE == FINISH_GN, MULTI>(st, a, partials, lds, blockIdx.x, gridDim.x);
}

// One workgroup: sums the ICP partial records of the preceding launch.  MODE RAW: totals ->
// st->out_f (stand-alone icpStep); MODE GN (ICP-only tracking): unpack, solve, update the pose.
template <int MODE>
__global__ __launch_bounds__(256) void icp_finish_kernel(OdomState* __restrict__ st, const float* __restrict__ partials,
                                                         unsigned nrecords, LevelIntr intr) {
    __shared__ GridReduceLds<float, 256> lds;
    if (MODE == FINISH_GN && st->level_break) return;
    sum_partial_records<256, false>(partials, nrecords, lds);
    if (threadIdx.x == 0) {
        if (MODE == FINISH_RAW) {
            for (int k = 0; k < 29; ++k) st->out_f[k] = lds.total[k];
        } else {
            solve_and_update(st, nullptr, lds.total, intr);
        }
    }
}

// ---- photometric correspondence pass ----------------------------------------------------
struct RgbResidualArgs {
    float min_scale, max_depth_delta;
    const int16_t *dIdx, *dIdy;
    int d_stride;  // in int16 elements
    const float *last_depth, *next_depth;
    int ld_stride, nd_stride;
    const uint8_t *last_image, *next_image;
    int li_stride, ni_stride;
    mmf_dataterm* corres;  // dense; holds CorresPk records inside the Gauss-Newton loop
    int cols, rows;
    unsigned cols_magic;  // floor(2^32 / cols) + 1 (see IcpArgs)
    float* err_map;
    int err_stride;
    LevelIntr intr;
    // non-null (object models): the extent of the model's own depth (extent.hpp) -- the 256-pixel blocks outside it
    // hold no pixel that can take part, and neither this pass nor rgb_step_kernel touches them
    const unsigned long long* extent;
    unsigned extent_gen;
    int extent_level;  // this pass's pyramid level (extent_of_level)
};

// Like the ICP kernel this one only produces partial records {count, sum diff^2}; they are summed
// by residual_finish_kernel (stand-alone) or by the prologue of rgb_step_kernel.
// publishes a workgroup's {count, sum diff^2}: a dense int2 record (stand-alone computeRgbResidual, summed
// by residual_finish_kernel) or, inside the Gauss-Newton loop, two integer atomics into the device state
template <bool ACC>
__device__ __forceinline__ void residual_publish(const OdomState* st, int2* __restrict__ partials, unsigned bid, int count,
                                                 int sigma) {
    if (threadIdx.x != 0) return;
    if (ACC) {
        unsigned long long* acc = const_cast<unsigned long long*>(st->res_acc) + kResStride * (bid % kResShards);
        const unsigned long long packed = ((unsigned long long)(unsigned)count << kResCountShift) | (unsigned long long)(unsigned)sigma;
        (void)__hip_atomic_fetch_add(acc, packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        partials[bid] = make_int2(count, sigma);
    }
}

// Compact correspondence record of the Gauss-Newton loop (8 bytes instead of the 16-byte DataTerm of
// the stand-alone computeRgbResidual): `one` is the pixel's own position (implicit in the record index)
// and diff is an integer in [-255, 255].  Halves the write traffic of this pass and the read traffic of
// rgb_step_kernel.
struct CorresPk {
    int16_t zero_x, zero_y, diff, valid;
};

// bit 7 of every byte of the result is set iff that byte of w is non-zero
__device__ __forceinline__ unsigned nonzero_bytes(unsigned w) {
    return (((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w) & 0x80808080u;
}
// the four byte flags (bit 7 of each byte) as bits 0..3
__device__ __forceinline__ unsigned byte_flags_to_bits(unsigned m) { return (((m >> 7) * 0x00204081u) >> 21) & 0xFu; }

// Four consecutive pixels of a row per lane: one 4-byte load of the intensities, one 8-byte load of
// each gradient image, one 16-byte load of the depth, 16-byte stores of the records; the 4x4 "all
// neighbours > 0" windows of the four pixels are evaluated from three aligned 32-bit words per image
// row (12 loads instead of 64) with byte-parallel bit arithmetic.  GN: inside the Gauss-Newton loop
// (atomic totals, break flag).
// ALIGNED: cols % 4 == 0 and 16-byte aligned rows; the records of the Gauss-Newton loop are then the compact ones.
// !ALIGNED: any width and pitch: a row is cut into groups of four pixels with a ragged last group, the same words are
// put together from single bytes (a byte outside the image counts as set: the reference's window loops skip it), the
// other loads and all stores go pixel by pixel, records are full DataTerms.  Same evaluation, same bits.
template <bool GN, bool ALIGNED = true>
__device__ __forceinline__ void residual_block4(const OdomState* __restrict__ st, const RgbResidualArgs& a,
                                                int2* __restrict__ partials, GridReduceLds<int, kBlock>& lds,
                                                unsigned bid, unsigned nblocks) {
    int sum[2] = {0, 0};
    MMF_STAMP(8);
    const int cols = a.cols, rows = a.rows;
    const int gpr = ALIGNED ? cols / 4 : (cols + 3) / 4;  // groups per row
    const int N4 = gpr * rows;
    const float* K = st->krkinv;
    const float ktx = st->kt[0], kty = st->kt[1], ktz = st->kt[2];
    const int level_break = GN ? st->level_break : 0;  // consumed after the image loads are in flight
    const bool cull = GN && ALIGNED && a.extent != nullptr;  // (uniform)
    ExtentBox box{0, 0, 0, 0};
    if (cull) box = extent_of_level(a.extent, a.extent_gen, a.extent_level);

    for (int g = bid * kBlock + threadIdx.x; g < N4; g += nblocks * kBlock) {
        int i, j0;
        if (ALIGNED) {
            i = (int)__umulhi((unsigned)(g * 4), a.cols_magic), j0 = g * 4 - i * cols;
        } else {
            i = g / gpr, j0 = (g - i * gpr) * 4;
        }
        const int k0 = i * cols + j0;  // index of the group's first record
        // the wave's 256 pixels all outside the model's own depth: every one of them fails `!(d1 != d1)` below
        if (cull && extent_misses(box, (unsigned)(g - (int)(threadIdx.x & 63u)) * 4u, 256u, cols, a.cols_magic)) {
            if (a.err_map) *reinterpret_cast<float4*>(a.err_map + (size_t)i * a.err_stride + j0) = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        // twelve unconditional word loads from clamped addresses (conditions applied afterwards), so
        // they and the four loads below are one round trip
        unsigned ww[4][3];
        bool has_l = j0 >= 4, has_r = j0 + 4 < cols;
        unsigned own;
        short4 gx, gy;
        float4 dv;
        if (ALIGNED) {
#pragma unroll
            for (int dr = -2; dr <= 1; ++dr) {
                const int u = min(max(i + dr, 0), rows - 1);
                const uint8_t* rowp = a.next_image + (size_t)u * a.ni_stride + j0;
                ww[dr + 2][1] = *reinterpret_cast<const unsigned*>(rowp);
                ww[dr + 2][0] = *reinterpret_cast<const unsigned*>(rowp - (has_l ? 4 : 0));
                ww[dr + 2][2] = *reinterpret_cast<const unsigned*>(rowp + (has_r ? 4 : 0));
            }
            own = *reinterpret_cast<const unsigned*>(a.next_image + (size_t)i * a.ni_stride + j0);
            gx = *reinterpret_cast<const short4*>(a.dIdx + (size_t)i * a.d_stride + j0);
            gy = *reinterpret_cast<const short4*>(a.dIdy + (size_t)i * a.d_stride + j0);
            dv = *reinterpret_cast<const float4*>(a.next_depth + (size_t)i * a.nd_stride + j0);
        } else {
            auto word = [&](const uint8_t* rowp, int c0) {  // columns c0 .. c0 + 3 of one image row
                unsigned w = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int c = c0 + b;
                    const unsigned v = (c >= 0 && c < cols) ? rowp[min(max(c, 0), cols - 1)] : 0xFFu;
                    w |= v << (8 * b);
                }
                return w;
            };
#pragma unroll
            for (int dr = -2; dr <= 1; ++dr) {
                const uint8_t* rowp = a.next_image + (size_t)min(max(i + dr, 0), rows - 1) * a.ni_stride;
                ww[dr + 2][0] = word(rowp, j0 - 4), ww[dr + 2][1] = word(rowp, j0), ww[dr + 2][2] = word(rowp, j0 + 4);
            }
            has_l = has_r = true;  // the outside columns are in the words, as set bytes
            own = word(a.next_image + (size_t)i * a.ni_stride, j0);
            int16_t gxs[4], gys[4];
            float dvs[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int x = min(j0 + p, cols - 1);  // (pixels past the row's end fail `x < cols - 5` below)
                gxs[p] = a.dIdx[(size_t)i * a.d_stride + x], gys[p] = a.dIdy[(size_t)i * a.d_stride + x];
                dvs[p] = a.next_depth[(size_t)i * a.nd_stride + x];
            }
            gx = make_short4(gxs[0], gxs[1], gxs[2], gxs[3]), gy = make_short4(gys[0], gys[1], gys[2], gys[3]);
            dv = make_float4(dvs[0], dvs[1], dvs[2], dvs[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (level_break) return;  // wave-uniform; the state read overlapped the loads above
        // nonzero masks of columns j0-4 .. j0+7 (bit b <-> column j0-4+b) ANDed over rows i-2 .. i+1;
        // columns / rows outside the image are skipped by the reference's loops => treated as set
        unsigned nz[3] = {0x80808080u, 0x80808080u, 0x80808080u};
#pragma unroll
        for (int dr = -2; dr <= 1; ++dr) {
            const bool rowin = (i + dr) >= 0 && (i + dr) < rows;
#pragma unroll
            for (int k = 0; k < 3; ++k) nz[k] &= rowin ? nonzero_bytes(ww[dr + 2][k]) : 0x80808080u;
        }
        const unsigned ok = (has_l ? byte_flags_to_bits(nz[0]) : 0xFu) | (byte_flags_to_bits(nz[1]) << 4) |
                            ((has_r ? byte_flags_to_bits(nz[2]) : 0xFu) << 8);
#ifdef MMF_STAMPS
        if (dv.x == 1234.5f && gx.x == 77 && gy.y == 78 && own == 0x1234567u && ok == 77u) sum[0] += 1;
        MMF_STAMP(9);
#endif
        const int valxs[4] = {gx.x, gx.y, gx.z, gx.w}, valys[4] = {gy.x, gy.y, gy.z, gy.w};
        const float d1s[4] = {dv.x, dv.y, dv.z, dv.w};
        // The warps of the four pixels are evaluated first (under their `if`: only ~30 % of the pixels
        // pass the gradient test, in clusters, so whole waves skip the divisions), with NO load inside
        // the branches; then the eight gathers from the last frame are issued TOGETHER from clamped
        // addresses and masked afterwards.  Nested `if`s around the gathers made them four dependent
        // round trips -- the long pole of the producer launch at the coarse levels.
        bool inb[4];
        int u0s[4], v0s[4];
        float td1s[4], d0s[4];
        uint8_t lis[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x = j0 + p, y = i;
            // window of pixel x: columns x-2 .. x+1 = bits p+2 .. p+5
            const bool valid = x < cols - 5 && y < rows - 1 && ((ok >> (p + 2)) & 0xFu) == 0xFu;
            const int valx = valxs[p], valy = valys[p];
            const float mTwo = (float)((valx * valx) + (valy * valy));
            const float d1 = d1s[p];
            inb[p] = false;
            u0s[p] = v0s[p] = 0;
            td1s[p] = 0.f;
            if (valid && mTwo >= a.min_scale && !(d1 != d1)) {
                td1s[p] = (float)(d1 * (K[6] * x + K[7] * y + K[8]) + ktz);
                u0s[p] = float2int_rn((d1 * (K[0] * x + K[1] * y + K[2]) + ktx) / td1s[p]);
                v0s[p] = float2int_rn((d1 * (K[3] * x + K[4] * y + K[5]) + kty) / td1s[p]);
                inb[p] = u0s[p] >= 0 && v0s[p] >= 0 && u0s[p] < cols && v0s[p] < rows;
            }
        }
#ifdef MMF_STAMPS
        if (u0s[0] == -12345 && v0s[3] == -4) sum[0] += 1;
        MMF_STAMP(10);
#endif
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int gu = inb[p] ? u0s[p] : 0, gv = inb[p] ? v0s[p] : 0;
            d0s[p] = a.last_depth[(size_t)gv * a.ld_stride + gu];
            lis[p] = a.last_image[(size_t)gv * a.li_stride + gu];
        }
#ifdef MMF_STAMPS
        if (d0s[0] == 1234.5f && lis[3] == 7 && d0s[3] == 3.f && lis[0] == 9) sum[0] += 1;
        MMF_STAMP(11);
#endif
        __builtin_amdgcn_sched_barrier(0);
        float errs[4];
        CorresPk pk[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x = j0 + p, y = i;
            const float d0 = d0s[p];
            const uint8_t li = lis[p];
            const bool hit = inb[p] && d0 > 0 && fabsf(td1s[p] - d0) <= a.max_depth_delta && li != 0;
            const int idiff = (int)((own >> (8 * p)) & 0xFFu) - (int)li;  // == (float)next - (float)last, exactly
            const int vy = hit ? idiff * idiff : 0;                        // == (int)(diff * diff)
            errs[p] = hit ? 0.001f * vy : 0.0f;
            sum[0] += hit ? 1 : 0;
            sum[1] += vy;
            if (GN && ALIGNED) {
                pk[p].zero_x = hit ? (int16_t)u0s[p] : (int16_t)0;
                pk[p].zero_y = hit ? (int16_t)v0s[p] : (int16_t)0;
                pk[p].diff = hit ? (int16_t)idiff : (int16_t)0;
                pk[p].valid = hit ? (int16_t)1 : (int16_t)0;
            } else if (ALIGNED || x < cols) {
                mmf_dataterm c;
                c.zero_x = hit ? (int16_t)u0s[p] : (int16_t)0;
                c.zero_y = hit ? (int16_t)v0s[p] : (int16_t)0;
                c.one_x = hit ? (int16_t)x : (int16_t)0;
                c.one_y = hit ? (int16_t)y : (int16_t)0;
                c.diff = hit ? (float)idiff : 0.f;
                c.valid = hit ? 1 : 0;
                c.pad_[0] = c.pad_[1] = c.pad_[2] = 0;
                *reinterpret_cast<int4*>(&a.corres[k0 + p]) = *reinterpret_cast<const int4*>(&c);
            }
        }
        if (GN && ALIGNED) {  // 4 x 8 bytes = two 16-byte stores
            int4* dst = reinterpret_cast<int4*>(reinterpret_cast<CorresPk*>(a.corres) + k0);
            dst[0] = *reinterpret_cast<const int4*>(&pk[0]);
            dst[1] = *reinterpret_cast<const int4*>(&pk[2]);
        }
        if (a.err_map) {
            if (ALIGNED) {
                *reinterpret_cast<float4*>(a.err_map + (size_t)i * a.err_stride + j0) = make_float4(errs[0], errs[1], errs[2], errs[3]);
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (j0 + p < cols) a.err_map[(size_t)i * a.err_stride + j0 + p] = errs[p];
            }
        }
    }

    MMF_STAMP(12);
    block_sum2<kBlock>(sum[0], sum[1], lds);
    residual_publish<GN>(st, partials, bid, sum[0], sum[1]);
    MMF_STAMP(13);
}

template <int MODE, int PX>
__global__ __launch_bounds__(kBlock) void rgb_residual_kernel(const OdomState* __restrict__ st, RgbResidualArgs a,
                                                              int2* __restrict__ partials) {
    __shared__ GridReduceLds<int, kBlock> lds;
    if (MODE == FINISH_GN && PX != 4 && st->level_break) return;
    if (PX == 4)
        residual_block4<MODE == FINISH_GN, true>(st, a, partials, lds, blockIdx.x, gridDim.x);
    else
        residual_block4<MODE == FINISH_GN, false>(st, a, partials, lds, blockIdx.x, gridDim.x);
}

// ---- several rigid-body models in ONE launch ----------------------------------------------------------------
// The Gauss-Newton chains of the models of a frame (MultiMotionFusion.cpp:312-387: one performTracking per model)
// are independent and have the same fixed schedule, so the orchestrator runs them as ONE chain of launches with
// gridDim.y = number of models instead of one chain per model (the launches are latency bound: ~20 us per
// iteration whatever the work).  Every RGBDOdometry keeps all its buffers -- state, model-side pyramids, records,
// reduction scratch, error images -- in one slab with the same layout, so model m's pointers are model 0's plus
// the byte distance between the two slabs; the sensor-side images are shared and take no offset.
constexpr int kMaxBatch = 8;
struct BatchDelta {
    long long d[kMaxBatch];  // slab(m) - slab(0) in bytes; d[0] = 0
};
template <typename T>
__device__ __forceinline__ T* batch_shift(T* p, long long d) {
    return p ? reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + d) : p;
}

// An OBJECT model's images are empty outside a box of a hundred pixels (extent.hpp), and what a workgroup of these launches
// costs is mostly fixed -- state, reduction, hand-over -- so such a model walks its image with a quarter of the workgroups
// in the two photometric passes (grid-stride passes, most of them skipped) whether it is tracked in a batch or alone: its launch geometry, and with it the
// order of its float sums, is a property of the model.  In a batched launch (gridDim.y = model, one gridDim.x for all) the
// surplus workgroups of such a model leave at once.
struct ChainGeom {
    unsigned res_f;   // track_producer_kernel: correspondence workgroups of the models behind the first (0: the launch's own)
    unsigned step_f;  // rgb_step_kernel: likewise
};

// Both producers of one Gauss-Newton iteration in ONE launch: workgroups [0, icp_blocks) run the
// ICP reduction, the rest the photometric correspondence pass.  The two passes are independent
// (RGBDOdometry.cpp:363-410), so running them side by side removes a launch boundary and lets
// their latency chains overlap.
template <int W, bool PACKED>
__global__ __launch_bounds__(kBlock) void track_producer_kernel(const OdomState* __restrict__ st, IcpArgs ia,
                                                                unsigned icp_blocks, RgbResidualArgs ra,
                                                                float* __restrict__ icp_partials,
                                                                int2* __restrict__ res_partials, BatchDelta bd, ChainGeom geom) {
    __shared__ GridReduceLds<float, kBlock> lds;
    // (the ICP workgroups keep the launch's count for every model: a grid-stride ICP pass keeps its 58 running sums live
    // across its loads and takes the whole kernel from 80 to 151 registers)
    const unsigned my_icp = icp_blocks;
    unsigned my_res = gridDim.x - icp_blocks;
    if (blockIdx.y > 0 && geom.res_f) {
        my_res = geom.res_f;
        if (blockIdx.x >= my_res + my_icp) return;
    }
    if (gridDim.y > 1) {  // model blockIdx.y: its state, model-side maps, records and error images (wave uniform)
        const long long d = bd.d[blockIdx.y];
        st = batch_shift(st, d);
        ia.vmap_g_prev.base = batch_shift(ia.vmap_g_prev.base, d), ia.nmap_g_prev.base = batch_shift(ia.nmap_g_prev.base, d);
        ia.prev_packed = batch_shift(ia.prev_packed, d), ia.err_map = batch_shift(ia.err_map, d);
        ra.last_depth = batch_shift(ra.last_depth, d), ra.next_depth = batch_shift(ra.next_depth, d);
        ra.last_image = batch_shift(ra.last_image, d), ra.corres = batch_shift(ra.corres, d);
        ra.err_map = batch_shift(ra.err_map, d);
        ra.extent = blockIdx.y ? batch_shift(ra.extent, d) : nullptr;  // (the batch's first model is the dense one)
        icp_partials = batch_shift(icp_partials, d), res_partials = batch_shift(res_partials, d);
    }
    // st->level_break is checked inside the blocks, after their state-independent loads are in flight
    using T = typename std::conditional<W == 2, v2f, float>::type;
    // the correspondence workgroups are the long pole of the launch (phase stamps): they take the
    // FIRST block indices so that they are dispatched first
    if (blockIdx.x >= my_res) {
        const unsigned bid = blockIdx.x - my_res;
        if (ia.err_map)
            icp_block2<T, 1, kBlock, PACKED, true, true>(st, ia, icp_partials, lds, bid, my_icp);
        else
            icp_block2<T, 1, kBlock, PACKED, false, true>(st, ia, icp_partials, lds, bid, my_icp);
    } else {
        residual_block4<true>(st, ra, res_partials, reinterpret_cast<GridReduceLds<int, kBlock>&>(lds), blockIdx.x, my_res);
    }
}

// {count, sigma} decision of RGBDOdometry.cpp:373-385 as a pure function of the two totals
struct ResidualDecision {
    float tmpError, sigmaVal;
    bool brk;
};
__device__ __forceinline__ ResidualDecision residual_decide(int count, int sigma, int rgb_only, float lastRGBError) {
    ResidualDecision d;
    d.tmpError = (float)(sqrt((double)sigma) / count);
    d.sigmaVal = (d.tmpError == 0) ? 1 : (float)count;
    d.brk = rgb_only && d.tmpError > lastRGBError;
    if (rgb_only) d.sigmaVal = -1;  // signals the Jacobian pass to weight evenly
    return d;
}

// stand-alone computeRgbResidual: one workgroup sums the records into st->out_i
__global__ __launch_bounds__(256) void residual_finish_kernel(OdomState* __restrict__ st, const int2* __restrict__ partials,
                                                              unsigned nrecords) {
    __shared__ GridReduceLds<int, 256> lds;
    int count, sigma;
    sum_int2_records<256>(partials, nrecords, count, sigma, lds);
    if (threadIdx.x == 0) {
        st->out_i[0] = count;
        st->out_i[1] = sigma;
    }
}

// ---- photometric Jacobian reduction -------------------------------------------------------
struct RgbStepArgs {
    const int2* residual_partials;  // GN mode: {count, sigma} records of the preceding rgb_residual_kernel
    unsigned residual_records;
    const float* icp_partials;     // GN mode, when the ICP term is on: records of icp_kernel
    unsigned icp_records;
    const mmf_dataterm* corres;
    const float* cloud;    // AoS float3, dense (the stand-alone entry's argument) ...
    const float4* cloud4;  // ... or, when non-null, the {X, Y, Z, 1/Z} records the preparation writes for the chains
    float fx, fy;
    const int16_t *dIdx, *dIdy;
    int d_stride;
    float sobel_scale;
    int cols, rows;
    unsigned cols_magic;  // floor(2^32 / cols) + 1
    LevelIntr intr;       // intrinsics the finishing lane prepares the NEXT correspondence pass with
    int next_level;       // 1: that pass belongs to the next pyramid level (gn_level_begin_kernel folded in)
    int final_step;       // 1: the very last step of the frame: the finishing lane also does odom_end
    const unsigned long long* extent;  // as RgbResidualArgs::extent: the blocks the correspondence pass skipped hold no record
    unsigned extent_gen;
    int extent_level;
};

// RGBDOdometry.cpp:464-467, 475-476
__device__ __forceinline__ void odom_end(OdomState* st) {
    if (st->rgb) {
        const float dx = st->tcurr[0] - st->tprev[0], dy = st->tcurr[1] - st->tprev[1], dz = st->tcurr[2] - st->tprev[2];
        if (sqrtf(dx * dx + dy * dy + dz * dz) > 0.3) {
            for (int k = 0; k < 9; ++k) st->Rcurr[k] = st->Rprev[k];
            for (int k = 0; k < 3; ++k) st->tcurr[k] = st->tprev[k];
        }
    }
    for (int k = 0; k < 3; ++k) st->trans_out[k] = st->tcurr[k];
    for (int k = 0; k < 9; ++k) st->rot_out[k] = st->Rcurr[k];
    // what the projection passes of this frame need, for the ones enqueued before the host has the pose
    float m[16], inv[16];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[r * 4 + c] = st->Rcurr[r * 3 + c];
        m[r * 4 + 3] = st->tcurr[r];
    }
    m[12] = m[13] = m[14] = 0.f, m[15] = 1.f;
    inverse4f(m, inv);
    for (int k = 0; k < 16; ++k) st->pose_inv[k] = inv[k], st->pose_out[k] = m[k];
}

// The two halves of one pass of rgbKernel (reduce.cu:504-535) over the PX records of a lane.
// rgb_gather: every gather issued before any is consumed.  Records without a correspondence carry
// zero coordinates, i.e. a valid address, and are masked afterwards (a per-record `if (valid)` made
// the PX gathers PX dependent round trips).
template <int PX>
struct RgbLane {
    mmf_dataterm c[PX];
    float X[PX], Y[PX], Z[PX];
    int gx[PX], gy[PX];
    float invz[PX];  // 1.0f / Z when the caller has it (gn_iter_kernel gathers it beside the point), see rgb_rows
};
// COMPACT: the lane's PX records are CorresPk (8 bytes, PX / 2 loads); i0 = index of its first record
template <int PX, bool COMPACT>
__device__ __forceinline__ void rgb_gather(const RgbStepArgs& a, const int4 (&raws)[COMPACT ? PX / 2 : PX], int i0,
                                           RgbLane<PX>& l) {
    int y0 = 0, x0 = 0;
    if (COMPACT) {
        y0 = (int)__umulhi((unsigned)i0, a.cols_magic);
        x0 = i0 - y0 * a.cols;
    }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        if (COMPACT) {
            const CorresPk pk = reinterpret_cast<const CorresPk*>(raws)[p];
            l.c[p].zero_x = pk.zero_x, l.c[p].zero_y = pk.zero_y;
            l.c[p].one_x = (int16_t)(pk.valid ? x0 + p : 0), l.c[p].one_y = (int16_t)(pk.valid ? y0 : 0);
            l.c[p].diff = (float)pk.diff;
            l.c[p].valid = (unsigned char)pk.valid;
        } else {
            *reinterpret_cast<int4*>(&l.c[p]) = raws[p];
        }
        if (a.cloud4) {  // (uniform; the same three floats either way)
            const float4 cp = a.cloud4[(size_t)(l.c[p].zero_y * a.cols + l.c[p].zero_x)];
            l.X[p] = cp.x, l.Y[p] = cp.y, l.Z[p] = cp.z;
        } else {
            const float* cp = a.cloud + (size_t)(l.c[p].zero_y * a.cols + l.c[p].zero_x) * 3;
            l.X[p] = cp[0], l.Y[p] = cp[1], l.Z[p] = cp[2];
        }
        l.gx[p] = a.dIdx[(size_t)l.c[p].one_y * a.d_stride + l.c[p].one_x];
        l.gy[p] = a.dIdy[(size_t)l.c[p].one_y * a.d_stride + l.c[p].one_x];
    }
}
// rgbStep's weight of a correspondence (reduce.cu:506-514): a function of sigma (one value per pass) and of |diff|, an integer
// in [0, 255] -- so a pass needs at most 256 different weights (rgb_weight_table: gn_iter_kernel keeps them in LDS, one
// division per entry instead of one per pixel)
__device__ __forceinline__ float rgb_weight(float sigma, float absdiff) {
    float w = sigma + absdiff;
    w = w > FLT_EPSILON ? 1.0f / w : 1.0f;
    if (sigma == -1) w = 1;
    return w;
}

template <int PX>
__device__ __forceinline__ void rgb_rows(float sobel_scale, float fx, float fy, float sigma, bool live, const RgbLane<PX>& l,
                                         float (&sum)[29], const float* wtab = nullptr, bool have_invz = false) {
#pragma unroll
    for (int p = 0; p < PX; ++p) {  // branch free
        const bool found = live && l.c[p].valid != 0;
        const float w = wtab ? wtab[(int)fabsf(l.c[p].diff)] : rgb_weight(sigma, fabsf(l.c[p].diff));
        const float X = l.X[p], Y = l.Y[p], Z = l.Z[p];
        const float invz = have_invz ? l.invz[p] : 1.0f / Z;  // (float)(1.0 / Z): double rounding is innocuous for division
        const float dI_dx = w * sobel_scale * l.gx[p];
        const float dI_dy = w * sobel_scale * l.gy[p];
        const float v0 = dI_dx * fx * invz;
        const float v1 = dI_dy * fy * invz;
        const float v2 = -(v0 * X + v1 * Y) * invz;
        float row[7] = {v0, v1, v2, -Z * v1 + Y * v2, Z * v0 - X * v2, -Y * v0 + X * v1, -w * l.c[p].diff};
#pragma unroll
        for (int k = 0; k < 7; ++k) row[k] = found ? row[k] : 0.f;
        accumulate_se3(sum, row, found ? 1.0f : 0.0f);
    }
}

// PX = 4: a lane takes four consecutive records (64 contiguous bytes), so the grid -- and with it
// the number of partial records the finishing workgroup has to re-read -- shrinks 4x.
//
// Order of the first pass (phase stamps, tools/rgb_step_probe.py): the device state was written by
// the previous kernel's finishing lane, so reading it is a cold ~1 us round trip -- as long as the
// record loads.  Neither the records nor the gathers depend on the state, so the kernel issues the
// record loads, then the (scalar) state loads, then the gathers, and only then consumes the state
// (break flags, sigma): three round trips overlap instead of queueing.
template <int MODE, int PX, bool COMPACT = false>
__global__ __launch_bounds__(kBlock) void rgb_step_kernel(OdomState* __restrict__ st, RgbStepArgs a,
                                                          float* __restrict__ partials,
                                                          unsigned* __restrict__ ticket, BatchDelta bd, ChainGeom geom) {
    __shared__ GridReduceLds<float, kBlock> lds;
    unsigned nblocks = gridDim.x;
    const unsigned icp_records = a.icp_records;
    if (blockIdx.y > 0 && geom.step_f) {  // (ChainGeom)
        nblocks = geom.step_f;
        if (blockIdx.x >= nblocks) return;
    }
    if (gridDim.y > 1) {  // model blockIdx.y (see BatchDelta)
        const long long d = bd.d[blockIdx.y];
        st = batch_shift(st, d);
        a.corres = batch_shift(a.corres, d), a.cloud = batch_shift(a.cloud, d), a.cloud4 = batch_shift(a.cloud4, d);
        a.icp_partials = batch_shift(a.icp_partials, d), a.residual_partials = batch_shift(a.residual_partials, d);
        partials = batch_shift(partials, d), ticket = batch_shift(ticket, d);
        a.extent = blockIdx.y ? batch_shift(a.extent, d) : nullptr;
    }
    MMF_STAMP(0);
    const int N = a.cols * a.rows;
    int i0 = (blockIdx.x * kBlock + threadIdx.x) * PX;
    const bool live = i0 < N;
    const bool cull = COMPACT && a.extent != nullptr;  // (uniform)
    ExtentBox box{0, 0, 0, 0};
    if (cull) box = extent_of_level(a.extent, a.extent_gen, a.extent_level);
    // this wave's records were never written (residual_block4 skipped the block): nothing to add
    const bool skip = cull && extent_misses(box, (unsigned)(i0 - (int)(threadIdx.x & 63u) * PX), 64u * PX, a.cols, a.cols_magic);
    static_assert(!COMPACT || PX % 2 == 0, "compact records are loaded in pairs");
    constexpr int NQ = COMPACT ? PX / 2 : PX;          // 16-byte loads per lane
    constexpr int REC = COMPACT ? (int)sizeof(CorresPk) : (int)sizeof(mmf_dataterm);
    const char* recs = reinterpret_cast<const char*>(a.corres);
    int4 raws[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) raws[q] = skip ? make_int4(0, 0, 0, 0) : reinterpret_cast<const int4*>(recs + (size_t)(live ? i0 : 0) * REC)[q];
    __builtin_amdgcn_sched_barrier(0);
    // state reads (wave-uniform scalar loads), not consumed before the gathers are in flight
    float sigma = st->sigmaVal;
    const int level_break = st->level_break, rgb_only = st->rgb_only;
    const float lastRGBError = st->st.lastRGBError;
    SolveIn si;
    unsigned cnt = 0, sg = 0;
    if (MODE == FINISH_GN) {
        si = load_solve_in(st);
        unsigned long long acc = 0;
#pragma unroll
        for (int k = 0; k < kResShards; ++k) acc += st->res_acc[kResStride * k];
        cnt = (unsigned)(acc >> kResCountShift), sg = (unsigned)(acc & ((1ull << kResCountShift) - 1ull));  // the sum wraps at 2^32 like the reference's int
    }
    __builtin_amdgcn_sched_barrier(0);
    RgbLane<PX> lane;
    if (!skip) rgb_gather<PX, COMPACT>(a, raws, live ? i0 : 0, lane);
    __builtin_amdgcn_sched_barrier(0);

    int res_count = 0, res_sigma = 0;
    ResidualDecision dec = {0.f, 0.f, false};
    if (MODE == FINISH_GN) {
        if (level_break) return;
        // {count, sigma} of the preceding correspondence pass (integer atomics, see res_acc):
        // RGBDOdometry.cpp:373-385
        res_count = (int)cnt;
        res_sigma = (int)sg;
        dec = residual_decide(res_count, res_sigma, rgb_only, lastRGBError);
        if (dec.brk) {  // rgbOnly divergence: the reference `break`s out of this level's loop
            if (blockIdx.x == 0 && threadIdx.x == 0) st->level_break = 1;
            return;
        }
        sigma = dec.sigmaVal;
    }
    float sum[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) sum[k] = 0.f;
#ifdef MMF_STAMPS
    if (sigma == 1234.5f && lane.X[0] == 1234.5f && lane.gx[0] == 77 && lane.gy[PX - 1] == 78) sum[0] = lane.Z[PX - 1];
    MMF_STAMP(3);
#endif
    if (!skip) rgb_rows<PX>(a.sobel_scale, a.fx, a.fy, sigma, live, lane, sum);
    // images beyond the grid's single pass
    for (i0 += nblocks * kBlock * PX; i0 < N; i0 += nblocks * kBlock * PX) {
        if (cull && extent_misses(box, (unsigned)(i0 - (int)(threadIdx.x & 63u) * PX), 64u * PX, a.cols, a.cols_magic)) continue;
#pragma unroll
        for (int q = 0; q < NQ; ++q) raws[q] = reinterpret_cast<const int4*>(recs + (size_t)i0 * REC)[q];
        rgb_gather<PX, COMPACT>(a, raws, i0, lane);
        rgb_rows<PX>(a.sobel_scale, a.fx, a.fy, sigma, true, lane, sum);
    }

#ifdef MMF_STAMPS
    if (sum[5] == 1234.5f) sum[0] += 1.f;
    MMF_STAMP(4);
#endif
    const bool is_last_wg = grid_arrive<29, kBlock>(sum, partials, ticket, lds, nblocks);
    MMF_STAMP(5);
    if (!is_last_wg) return;
    if (MODE == FINISH_RAW) {
        sum_partial_records<kBlock, true>(partials, nblocks, lds);
        if (threadIdx.x == 0)
            for (int k = 0; k < 29; ++k) st->out_f[k] = lds.total[k];
    }
    if (MODE == FINISH_GN) {
        const bool icp = st->icp != 0;  // wave-uniform
        // the photometric records of this launch and the ICP records of the preceding one (kernel
        // boundary => plain loads), all loads in flight together
        sum_partial_records2<kBlock>(partials, nblocks, a.icp_partials, icp ? icp_records : 0u, lds);
        MMF_STAMP(6);
        // the 36 + 6 elements of the combined system by 42 lanes (and their lastA / lastb stores), so that the
        // solving lane starts from the finished matrix
        __shared__ double sol[42];
        if (threadIdx.x < 42) sol[threadIdx.x] = combine_element(st, threadIdx.x, si.w, lds.total, icp ? lds.total2 : nullptr);
        if (threadIdx.x >= 64 && threadIdx.x < 64 + kResShards)  // the next correspondence pass starts from zero totals
            st->res_acc[kResStride * (threadIdx.x - 64)] = 0ull;
        __syncthreads();
        if (threadIdx.x == 0) {
            st->sigma = res_sigma;
            st->rgbCount = res_count;
            st->sigmaVal = dec.sigmaVal;
            st->st.lastRGBError = a.next_level ? FLT_MAX : dec.tmpError;  // RGBDOdometry.cpp:329 at a level start
            st->st.lastRGBCount = (float)res_count;
#ifndef MMF_SKIP_SOLVE
            solve_and_update(st, si, lds.total, icp ? lds.total2 : nullptr, a.intr, sol, sol + 36);
            if (a.final_step) odom_end(st);  // one launch less at the end of the frame
#endif
#ifdef MMF_STAMPS
            if (g_mmf_dbg) g_mmf_dbg[blockIdx.x * 16 + 7] = wall_clock64() + (st->Rcurr[0] == 1234.5f ? 1 : 0);
#endif
        }
    }
}

// ---- SO3 pre-alignment ----------------------------------------------------------------------
struct So3Args {
    const uint8_t *last_image, *next_image;
    int l_stride, n_stride;
    int cols, rows;
    unsigned cols_magic;  // floor(2^32 / cols) + 1
    LevelIntr intr;
};

// the five taps of one central-difference gradient (reduce.cu:963-979), loaded from clamped addresses
struct So3Taps {
    uint8_t actu, left, right, up, down;
};
__device__ __forceinline__ So3Taps so3_load_taps(const uint8_t* img, int stride, int cols, int rows, int x, int y) {
    const int xc = min(max(x, 1), cols - 2), yc = min(max(y, 1), rows - 2);  // found => 1 <= x < cols-1, same for y
    So3Taps t;
    t.actu = img[(size_t)yc * stride + xc];
    t.left = img[(size_t)yc * stride + xc - 1];
    t.right = img[(size_t)yc * stride + xc + 1];
    t.up = img[(size_t)(yc - 1) * stride + xc];
    t.down = img[(size_t)(yc + 1) * stride + xc];
    return t;
}
__device__ __forceinline__ void so3_gradient(const So3Taps& t, float& gx, float& gy) {
    const float actu = (float)t.actu;
    gx = (((float)t.left + actu) / 2.0f) - (((float)t.right + actu) / 2.0f);
    gy = (((float)t.up + actu) / 2.0f) - (((float)t.down + actu) / 2.0f);
}

// Load order as in rgb_step_kernel: the last-frame taps depend on nothing but the pixel, so they are
// issued before the (cold) device state is read; the warped taps follow as one group; no load sits
// under a data-dependent branch.
template <int MODE>
__global__ __launch_bounds__(kBlock) void so3_kernel(OdomState* __restrict__ st, So3Args a,
                                                     float* __restrict__ partials,
                                                     unsigned* __restrict__ ticket) {
    __shared__ GridReduceLds<float, kBlock> lds;
    float sum[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) sum[k] = 0.f;
    const int N = a.cols * a.rows, cols = a.cols, rows = a.rows;
    int k = blockIdx.x * kBlock + threadIdx.x;
    const bool live0 = k < N;
    // k / cols by multiply-high; cols_magic == 0 (image too large for it) selects the plain division
    auto row_of = [&](int i) { return a.cols_magic ? (int)__umulhi((unsigned)i, a.cols_magic) : i / cols; };
    int y = row_of(live0 ? k : 0), x = (live0 ? k : 0) - y * cols;
    So3Taps tl = so3_load_taps(a.last_image, a.l_stride, cols, rows, x, y);
    __builtin_amdgcn_sched_barrier(0);
    const int so3_done = MODE == FINISH_GN ? st->so3_done : 0;
    m33 B, kinv;
    float krlr[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) B.m[q] = st->imageBasis[q], kinv.m[q] = st->kinv[q], krlr[q] = st->krlr[q];
    if (so3_done) return;

    for (bool first = true; k < N; k += gridDim.x * kBlock, first = false) {
        if (!first) {  // images beyond the grid's single pass
            y = row_of(k), x = k - y * cols;
            tl = so3_load_taps(a.last_image, a.l_stride, cols, rows, x, y);
        }
        const f3 unwarped = make_f3((float)x, (float)y, 1.0f);
        const f3 warped = B * unwarped;
        const int wx = float2int_rn(warped.x / warped.z);
        const int wy = float2int_rn(warped.y / warped.z);
        const bool found = (wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 &&
                            x < cols - 1 && y >= 1 && y < rows - 1);
        const So3Taps tn = so3_load_taps(a.next_image, a.n_stride, cols, rows, found ? wx : 1, found ? wy : 1);
        __builtin_amdgcn_sched_barrier(0);
        float row[4];
        {  // reduce.cu:1011-1046, branch free
            float gnx, gny, glx, gly;
            so3_gradient(tn, gnx, gny);
            so3_gradient(tl, glx, gly);
            const float gx = (gnx + glx) / 2.0f;
            const float gy = (gny + gly) / 2.0f;
            const f3 point = kinv * unwarped;
            const float z2 = point.z * point.z;
            const float A = krlr[0], Bc = krlr[1], C = krlr[2];
            const float D = krlr[3], E = krlr[4], F = krlr[5];
            const float G = krlr[6], H = krlr[7], I = krlr[8];
            f3 left;
            left.x = ((point.z * (D * gy + A * gx)) - (gy * G * y) - (gx * G * x)) / z2;
            left.y = ((point.z * (E * gy + Bc * gx)) - (gy * H * y) - (gx * H * x)) / z2;
            left.z = ((point.z * (F * gy + C * gx)) - (gy * I * y) - (gx * I * x)) / z2;
            const f3 jac = cross(left, point);
            row[0] = found ? jac.x : 0.f;
            row[1] = found ? jac.y : 0.f;
            row[2] = found ? jac.z : 0.f;
            row[3] = found ? -((float)tn.actu - (float)tl.actu) : 0.f;
        }
        // member order of JtJJtrSO3 (types.cuh:154-162)
        sum[0] = sum[0] + row[0] * row[0];
        sum[1] = sum[1] + row[0] * row[1];
        sum[2] = sum[2] + row[0] * row[2];
        sum[3] = sum[3] + row[0] * row[3];
        sum[4] = sum[4] + row[1] * row[1];
        sum[5] = sum[5] + row[1] * row[2];
        sum[6] = sum[6] + row[1] * row[3];
        sum[7] = sum[7] + row[2] * row[2];
        sum[8] = sum[8] + row[2] * row[3];
        sum[9] = sum[9] + row[3] * row[3];
        sum[10] = sum[10] + (found ? 1.0f : 0.0f);
    }

    if (!grid_reduce<11, kBlock>(sum, partials, ticket, lds)) return;
    if (threadIdx.x == 0) {
        if (MODE == FINISH_RAW) {
            for (int k = 0; k < 11; ++k) st->out_f[k] = lds.total[k];
        } else {
            so3_finish(st, lds.total, a.intr);
        }
    }
}

// ---- single-lane bookkeeping of the device-resident loop ----------------------------------
struct BeginPoses {  // the pose each model's tracking starts from (batched launches)
    float trans[kMaxBatch][3], rot[kMaxBatch][9];
};
struct BeginArgs {
    float trans[3], rot[9];
    int rgb_only, icp, rgb, so3;
    float icp_weight;
    LevelIntr so3_intr;  // level 2
    int so3_prefetched;  // the SO3 pre-alignment of this frame already ran (so3_begin_kernel + the so3 launches)
    int fold_level_begin;  // nothing runs between this kernel and the first gn_level_begin: do it here
    LevelIntr first_intr;  // intrinsics of the first (coarsest) level
    // so3_prefetched: where the pre-alignment sits complete -- a state of its own that no chain touches (the orchestrator's
    // staging states: the next frame's pre-alignment runs while this frame's chain is still at work), or nullptr = in the
    // leader's state
    const OdomState* so3_stage;
};

// the SO3 part of the beginning (RGBDOdometry.cpp:239-255): depends on the two images only, so the orchestrator
// can run it -- and the pre-alignment itself -- ahead of the rest (mmf_fusion_prefetch_frame)
__device__ __forceinline__ void so3_begin(OdomState* st, const LevelIntr& so3_intr, int so3) {
    for (int k = 0; k < 9; ++k) {
        const double e = (k % 4 == 0) ? 1.0 : 0.0;
        st->resultR[k] = e;
        st->lastResultR[k] = e;
        st->R_lr[k] = (float)e;
    }
    st->so3_lastError = FLT_MAX / 2;
    st->so3_lastCount = FLT_MAX / 2;
    st->so3_done = so3 ? 0 : 1;
    st->st.so3_iterations_run = 0;
    if (so3) {
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        so3_prepare_store(st, I3, so3_intr);
    }
}

__global__ void so3_begin_kernel(OdomState* st, LevelIntr so3_intr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    so3_begin(st, so3_intr, 1);
}

// start of a pyramid level: RGBDOdometry.cpp:320-328 (first level only), :344, :348-358
__device__ __forceinline__ void gn_level_begin(OdomState* st, int first_level, const LevelIntr& intr) {
    double resultRt[16];
    if (first_level) {
        for (int k = 0; k < 16; ++k) resultRt[k] = (k % 5 == 0) ? 1.0 : 0.0;
        if (st->so3)
            for (int x = 0; x < 3; ++x)
                for (int y = 0; y < 3; ++y) resultRt[x * 4 + y] = st->resultR[x * 3 + y];
        for (int k = 0; k < 16; ++k) st->resultRt[k] = resultRt[k];
    } else {
        for (int k = 0; k < 16; ++k) resultRt[k] = st->resultRt[k];
    }
    float krkinv[9], kt[3];
    rgb_prepare(resultRt, intr, krkinv, kt);
    st->st.lastRGBError = FLT_MAX;
    st->level_break = 0;
    for (int k = 0; k < kResShards; ++k) st->res_acc[kResStride * k] = 0ull;
    for (int k = 0; k < 9; ++k) st->krkinv[k] = krkinv[k];
    for (int k = 0; k < 3; ++k) st->kt[k] = kt[k];
}

// the SO3 pre-alignment depends on the two sensor images only (RGBDOdometry.cpp:239-310): with the sensor side
// shared by all models of a frame it is computed once, in the leader's state, and copied to the others
__device__ __forceinline__ void so3_share(OdomState* dst, const OdomState* src) {
    const So3State s = so3_load(src);
    so3_store(dst, s);
}

// RGBDOdometry.cpp:221-228, 237, 252-255, 316-328 for one model (64 lanes; lane 0 does the bookkeeping).
// leader: the state a prefetched pre-alignment sits in when there is no staging state (the first model's); follower: not that model
__device__ __forceinline__ void odom_begin_model(OdomState* st, const OdomState* leader, const BeginArgs& a, bool follower, int lane) {
    if (lane < 64)  // the first gn_iter_kernel launch adds its sums here
        for (int x = 0; x < kGnSumShards; ++x) st->gn_sum[0][x][lane] = 0ll;
    if (lane != 0) return;
    for (int k = 0; k < 9; ++k) st->Rprev[k] = st->Rcurr[k] = a.rot[k];
    for (int k = 0; k < 3; ++k) st->tprev[k] = st->tcurr[k] = a.trans[k];
    inverse3f(st->Rprev, st->Rprev_inv);
    st->rgb_only = a.rgb_only;
    st->icp = a.icp;
    st->rgb = a.rgb;
    st->so3 = a.so3;
    st->icp_weight = a.icp_weight;
    st->level_break = 0;
    st->st.iterations_run = 0;
    st->gn_fault = 0;
    st->gn_dbg_outside = 0u;
    for (int k = 0; k < 6; ++k) st->gn_dbg_rect[k] = 0;
    for (int k = 0; k < kResShards; ++k) st->gn_acc[0][kResStride * k] = 0ull;  // the first gn_iter_kernel launch adds here
    if (!a.so3_prefetched)
        so3_begin(st, a.so3_intr, a.so3);
    else if (a.so3_stage)
        so3_share(st, a.so3_stage);
    else if (follower)
        so3_share(st, leader);  // the prefetched pre-alignment sits complete in the leader's state (nobody writes it here)
    if (a.fold_level_begin) gn_level_begin(st, 1, a.first_intr);
}
// Batched: block m = model m, poses from `poses`.
__global__ void odom_begin_kernel(OdomState* st, BeginArgs a, BatchDelta bd, BeginPoses poses) {
    OdomState* sm = st;
    if (gridDim.x > 1) {
        sm = batch_shift(st, bd.d[blockIdx.x]);
        for (int k = 0; k < 9; ++k) a.rot[k] = poses.rot[blockIdx.x][k];
        for (int k = 0; k < 3; ++k) a.trans[k] = poses.trans[blockIdx.x][k];
    }
    odom_begin_model(sm, st, a, blockIdx.x > 0, (int)threadIdx.x);
}
// The last launch of a frame's model-side preparation (prep_batch.hpp) with the beginning of the tracking it prepares on one
// more workgroup: when a frame is prepared at the end of the call before it (fusion_orchestrator.hpp: FusionModel::spec_valid)
// the pose it starts from, the mode and the staged pre-alignment are known as well, and odom_begin_kernel -- 4-5 us of one
// lane's bookkeeping between the preparation and the first Gauss-Newton launch on the stream a frame waits for -- runs beside
// the preparation's last stage instead.  odom_enqueue_tracking skips its own launch when the arguments it would pass are
// these, bit for bit.
struct BeginRider {
    OdomState* st;
    int prep_blocks;  // workgroups of the preparation in this launch; the one after them is the rider's
    BeginArgs a;
};
__global__ __launch_bounds__(256) void prep_batch_begin_kernel(PrepBatch b, BeginRider r) {
    if ((int)blockIdx.x < r.prep_blocks) {
        prep_batch_body(b, (int)blockIdx.x);
        return;
    }
    if (threadIdx.y == 0) odom_begin_model(r.st, r.st, r.a, false, (int)threadIdx.x);
}

// share_so3: the SO3 loop has just run in the leader's state (block 0's)
__global__ void gn_level_begin_kernel(OdomState* st, int first_level, LevelIntr intr, BatchDelta bd, int share_so3) {
    if (threadIdx.x != 0) return;
    const OdomState* leader = st;
    if (gridDim.x > 1) st = batch_shift(st, bd.d[blockIdx.x]);
    if (share_so3 && blockIdx.x > 0) so3_share(st, leader);
    gn_level_begin(st, first_level, intr);
}

// The result of a chain goes to the host WITHOUT a runtime copy.  hipMemcpyAsync(D2H) + hipEventRecord cost a blit
// kernel and a marker on the stream, and -- what matters -- the next launches the host makes on that stream stall inside
// hipLaunchKernel until the copy has run (60 us each in an API trace), so nothing could be enqueued behind the copy
// while the chain was still running.  One wave per model copies the state words into the host's pinned, device-visible
// OdomState, fences at system scope and then stores the sequence number the host is polling for.
struct PublishTargets {
    OdomState* host[kMaxBatch];
};
// A second wave evaluates Model::computeFusionWeight for a fuse pass enqueued before the host has the pose (frame_rider.hpp):
// a few microseconds on one lane, beside the copy.
__global__ __launch_bounds__(128) void odom_publish_kernel(OdomState* st, PublishTargets to, unsigned seq, BatchDelta bd) {
    if (gridDim.x > 1) st = batch_shift(st, bd.d[blockIdx.x]);
    if (threadIdx.x >= 64) {
        if (threadIdx.x == 64) odom_fusion_weight(st);
        return;
    }
    odom_publish_wave(st, to.host[blockIdx.x], seq, threadIdx.x);
}

__global__ void odom_end_kernel(OdomState* st, BatchDelta bd) {
    if (threadIdx.x != 0) return;
    if (gridDim.x > 1) st = batch_shift(st, bd.d[blockIdx.x]);
    odom_end(st);
}

}  // namespace mmf
