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

// |x_r|^2 of every descriptor row as an fmaf chain in index order (one lane per row)
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ x, int n, int dim, float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float* p = x + (size_t)r * dim;
    float s = 0.f;
    for (int k = 0; k < dim; ++k) s = __builtin_fmaf(p[k], p[k], s);
    out[r] = s;
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
    const float* qa = q + (size_t)min(i0 + r, nq - 1) * dim;
    const float* tb = t + (size_t)min(j0 + r, nt - 1) * dim;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]: one MFMA consumes k, k + 1
    for (int k = 0; k < dim; k += 8) {
        const float4 a0 = *reinterpret_cast<const float4*>(qa + k), a1 = *reinterpret_cast<const float4*>(qa + k + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(tb + k), b1 = *reinterpret_cast<const float4*>(tb + k + 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.y : a0.x, h ? b0.y : b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.w : a0.z, h ? b0.w : b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.y : a1.x, h ? b1.y : b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.w : a1.z, h ? b1.w : b1.z, acc, 0, 0, 0);
    }
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

// crossCheck + distance gate (PointTracker.cpp:108); also hands the key arrays back empty
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
