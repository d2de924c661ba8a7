// match_kernels.hpp -- brute-force keypoint descriptor matching on the matrix cores (gfx950).
//
// Replaces the cv::BFMatcher(cv::NORM_L2, crossCheck = true).match(...) call of
// PointTracker::addKeypoints (Core/Utils/PointTracker.cpp:100-114): for every query descriptor the
// train descriptor at the smallest Euclidean distance, kept when the choice is mutual and the distance
// passes the gate.  SuperPoint descriptors are 256-dimensional unit vectors; a frame has a few hundred
// to a thousand of them.
//
// This is the one dense contraction next to the tracking path: the nq x nt Gram matrix G = Q T^T.  It
// runs on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate), whose accumulation is exactly an fmaf chain
// in k order, so d2(i, j) = (|q_i|^2 + |t_j|^2) - 2 G(i, j) is bit-identical to the CPU restatement
// (oracle/mmf_oracle_match.c) and so are the arg-mins, ties included.  One wave owns one 32 x 32 tile
// of the distance matrix: 128 MFMAs for dim = 256, operands straight from global memory (each lane
// reads 32 contiguous bytes per four MFMAs; a descriptor row is re-read by the 32 or 64 tiles of its
// tile row / column out of L2 -- the whole problem is ~2 MB).  Row and column minima leave the tile as
// 64-bit atomicMin keys (ordered distance bits << 32 | index: the smallest index wins a tie, like the
// first minimum of a sequential scan).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr unsigned long long kNoMatchKey = ~0ull;

// monotone map float -> uint32 (total order of the finite floats and infinities; -0 < +0)
__device__ __forceinline__ unsigned ordered_bits(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// 32 x 32 tile of A B^T on the f32 matrix cores: rows `ra`, `rb` of this lane's operands (lane & 31 picks
// the row, lane >> 5 the parity of k).  A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]:
// one MFMA consumes k, k + 1, so the accumulation is the fmaf chain over k = 0, 1, 2, ... of the oracle.
// The operands of the next 16 k are loaded (64 bytes per lane and operand) before the eight MFMAs of the
// current 16 are issued: with one wave per SIMD nothing else would hide the load latency.
__device__ __forceinline__ f32x16 gram_tile(const float* __restrict__ ra, const float* __restrict__ rb, int dim, int h) {
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    float4 a[4], b[4], an[4], bn[4];
    auto load = [&](int k, float4 (&x)[4], float4 (&y)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // dim % 8 == 0: a 16-float block may run 8 floats past the row's end -> clamp (those MFMAs are skipped)
            const int kk = min(k + 4 * u, dim - 4);
            x[u] = *reinterpret_cast<const float4*>(ra + kk);
            y[u] = *reinterpret_cast<const float4*>(rb + kk);
        }
    };
    load(0, a, b);
    for (int k = 0; k < dim; k += 16) {
        if (k + 16 < dim) load(k + 16, an, bn);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k + 4 * u < dim) {  // wave uniform
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[u].y : a[u].x, h ? b[u].y : b[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a[u].w : a[u].z, h ? b[u].w : b[u].z, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = an[u], b[u] = bn[u];
    }
    return acc;
}

// |x_r|^2 of every descriptor row = the diagonal of X X^T, taken from diagonal 32 x 32 MFMA tiles so that it
// is the same fmaf chain as the oracle's (a lane-per-row scalar loop took 30 us per set for 1024 rows).
// One wave per 32 rows of the concatenation [query ; train]; out = [qn ; tn].
// also resets the arg-min keys of its rows (row_best for query blocks, col_best for train blocks) to `empty`
__global__ __launch_bounds__(64) void row_norms_kernel(const float* __restrict__ q, int nq, const float* __restrict__ t, int nt,
                                                       int dim, float* __restrict__ qn, float* __restrict__ tn,
                                                       unsigned long long* __restrict__ row_best,
                                                       unsigned long long* __restrict__ col_best, unsigned long long empty) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int qblocks = (nq + 31) / 32;
    const bool is_q = (int)blockIdx.x < qblocks;
    const float* x = is_q ? q : t;
    const int n = is_q ? nq : nt, i0 = (is_q ? blockIdx.x : blockIdx.x - qblocks) * 32;
    float* out = is_q ? qn : tn;
    if (h == 0 && i0 + r < n) (is_q ? row_best : col_best)[i0 + r] = empty;
    const float* row = x + (size_t)min(i0 + r, n - 1) * dim;
    const f32x16 acc = gram_tile(row, row, dim, h);
    // C/D layout: col = lane & 31, row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5): the diagonal element of
    // column r sits in the lane whose half h holds row r
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int lr = (v & 3) + 8 * (v >> 2) + 4 * h;
        if (lr == r && i0 + r < n) out[i0 + r] = acc[v];
    }
}

// one wave (64 threads) per 32 x 32 tile; grid = (ceil(nt / 32), ceil(nq / 32)); dim % 8 == 0
__global__ __launch_bounds__(64) void match_tile_kernel(const float* __restrict__ q, const float* __restrict__ t,
                                                        const float* __restrict__ qn, const float* __restrict__ tn,
                                                        int nq, int nt, int dim,
                                                        unsigned long long* __restrict__ row_best,
                                                        unsigned long long* __restrict__ col_best) {
    __shared__ float tile[32][33];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    // operand rows of this lane (clamped: rows past the end are computed and masked afterwards)
    const f32x16 acc = gram_tile(q + (size_t)min(i0 + r, nq - 1) * dim, t + (size_t)min(j0 + r, nt - 1) * dim, dim, h);
    // C/D layout: col = lane & 31, row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)
    const int col = j0 + r;
    const float tnc = tn[min(col, nt - 1)];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int lr = (v & 3) + 8 * (v >> 2) + 4 * h, row = i0 + lr;
        const float d2 = (qn[min(row, nq - 1)] + tnc) - 2.0f * acc[v];
        tile[lr][r] = (row < nq && col < nt) ? d2 : __builtin_inff();
    }
    __syncthreads();
    // lanes 0..31: minimum of tile row `r` over the columns; lanes 32..63: of tile column `r` over the rows
    float best = __builtin_inff();
    int arg = -1;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
        const float d = h ? tile[s][r] : tile[r][s];
        if (d < best) best = d, arg = s;  // first minimum
    }
    if (arg < 0) return;  // nothing valid in this row / column of the tile
    if (!h) {
        if (i0 + r < nq) atomicMin(&row_best[i0 + r], ((unsigned long long)ordered_bits(best) << 32) | (unsigned)(j0 + arg));
    } else {
        if (j0 + r < nt) atomicMin(&col_best[j0 + r], ((unsigned long long)ordered_bits(best) << 32) | (unsigned)(i0 + arg));
    }
}

// The same tile arithmetic with operand sharing: a 256-thread workgroup owns a 64 x 64 block of the
// distance matrix (wave w the 32 x 32 quadrant (w >> 1, w & 1)) and stages 32-wide k slabs of its 64 query
// and 64 train rows through LDS, double buffered, so every descriptor element is fetched from L2 once per
// workgroup instead of once per tile (4x less traffic: the one-wave kernel re-read 134 MB for 2 MB of
// descriptors at 1024 x 1024 x 256 and was bound by it).  Slabs and the k inside them are consumed in
// ascending order: the accumulation is the same fmaf chain, the results stay bit-identical.  dim % 32 == 0.
constexpr int kMatchSlab = 32, kMatchPitch = kMatchSlab + 1;  // +1 float: rows of a slab fall into different banks

__global__ __launch_bounds__(256) void match_tile64_kernel(const float* __restrict__ q, const float* __restrict__ t,
                                                           const float* __restrict__ qn, const float* __restrict__ tn,
                                                           int nq, int nt, int dim,
                                                           unsigned long long* __restrict__ row_best,
                                                           unsigned long long* __restrict__ col_best) {
    __shared__ float As[2][64][kMatchPitch], Bs[2][64][kMatchPitch];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    const int I0 = blockIdx.y * 64, J0 = blockIdx.x * 64;
    // loader role: thread -> (row = tid >> 2, two float4 columns (tid & 3) and (tid & 3) + 4) of both slabs
    const int lrow = tid >> 2, lc = (tid & 3) * 4;
    const float* qa = q + (size_t)min(I0 + lrow, nq - 1) * dim + lc;
    const float* tb = t + (size_t)min(J0 + lrow, nt - 1) * dim + lc;
    float4 ga[2], gb[2];
    auto fetch = [&](int k) {
        ga[0] = *reinterpret_cast<const float4*>(qa + k), ga[1] = *reinterpret_cast<const float4*>(qa + k + 16);
        gb[0] = *reinterpret_cast<const float4*>(tb + k), gb[1] = *reinterpret_cast<const float4*>(tb + k + 16);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float* a = &As[buf][lrow][lc + 16 * u];
            float* b = &Bs[buf][lrow][lc + 16 * u];
            a[0] = ga[u].x, a[1] = ga[u].y, a[2] = ga[u].z, a[3] = ga[u].w;
            b[0] = gb[u].x, b[1] = gb[u].y, b[2] = gb[u].z, b[3] = gb[u].w;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    fetch(0);
    stash(0);
    __syncthreads();
    const int nslabs = dim / kMatchSlab;
    for (int s = 0; s < nslabs; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslabs) fetch((s + 1) * kMatchSlab);  // in flight during the 16 MFMAs below
        const float* ar = &As[buf][wr * 32 + r][h];
        const float* br = &Bs[buf][wc * 32 + r][h];
#pragma unroll
        for (int kk = 0; kk < kMatchSlab; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk], br[kk], acc, 0, 0, 0);
        if (s + 1 < nslabs) stash(buf ^ 1);
        __syncthreads();
    }
    // epilogue per quadrant, as in match_tile_kernel; the operand slabs are dead: reuse As as 4 x [32][33]
    float(*tile)[kMatchPitch] = reinterpret_cast<float(*)[kMatchPitch]>(&As[0][0][0]) + w * 32;
    const int i0 = I0 + wr * 32, j0 = J0 + wc * 32;
    const int col = j0 + r;
    const float tnc = tn[min(col, nt - 1)];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int lr = (v & 3) + 8 * (v >> 2) + 4 * h, row = i0 + lr;
        const float d2 = (qn[min(row, nq - 1)] + tnc) - 2.0f * acc[v];
        tile[lr][r] = (row < nq && col < nt) ? d2 : __builtin_inff();
    }
    __syncthreads();
    float best = __builtin_inff();
    int arg = -1;
#pragma unroll 8
    for (int sidx = 0; sidx < 32; ++sidx) {
        const float d = h ? tile[sidx][r] : tile[r][sidx];
        if (d < best) best = d, arg = sidx;  // first minimum
    }
    if (arg < 0) return;
    if (!h) {
        if (i0 + r < nq) atomicMin(&row_best[i0 + r], ((unsigned long long)ordered_bits(best) << 32) | (unsigned)(j0 + arg));
    } else {
        if (j0 + r < nt) atomicMin(&col_best[j0 + r], ((unsigned long long)ordered_bits(best) << 32) | (unsigned)(i0 + arg));
    }
}

// crossCheck + distance gate (PointTracker.cpp:108)
__global__ __launch_bounds__(256) void match_cross_check_kernel(unsigned long long* __restrict__ row_best,
                                                                const unsigned long long* __restrict__ col_best, int nq,
                                                                float max_distance, int* __restrict__ train_idx,
                                                                float* __restrict__ distance) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    const unsigned long long key = row_best[i];
    int j = -1;
    float d = 0.f;
    if (key != kNoMatchKey) {
        const int cand = (int)(unsigned)key;
        if ((int)(unsigned)col_best[cand] == i) {
            const float d2 = from_ordered_bits((unsigned)(key >> 32));
            d = sqrtf(d2 > 0.f ? d2 : 0.f);
            if (max_distance < 1.1920929e-7f || d <= max_distance) j = cand;
        }
    }
    train_idx[i] = j;
    distance[i] = j >= 0 ? d : 0.f;
}

}  // namespace mmf
