// grid_reduce.hpp -- deterministic wave64 -> workgroup -> grid sum for the JtJ reductions.
//
// Replaces the reference's warpReduceSum / blockReduceSum / second-launch reduceSum
// (Core/Cuda/reduce.cu:64-229, 663-720), which assume 32-wide warps and cost a second kernel
// launch plus a device synchronise per Gauss-Newton step.  Here:
//   1. each lane keeps NV running sums in registers;
//   2. a wave64 reduces them with DPP row shifts / row broadcasts (no LDS traffic);
//   3. the workgroup's waves combine through LDS in wave order;
//   4. the workgroup publishes its NV partials with write-through (sc1) stores, drains them and
//      takes a ticket from one device-scope counter; the workgroup whose ticket is the last
//      re-reads every partial with sc1 loads and sums them in a fixed order.
// The summation order depends only on the launch geometry, so results are bit-reproducible
// from run to run (float atomics would not be).  The hand-off follows the write-through form
// of the CDNA4 guide (sc1 payload stores, every storing wave drains with s_waitcnt vmcnt(0),
// ONE lane signals with an agent-scope atomic, the last arriver loads with sc1 loads after
// its add has returned and the other waves after a workgroup barrier).
#pragma once
#include <hip/hip_runtime.h>

namespace mmf {

constexpr int kBlock = 256;           // threads per workgroup for every reduction kernel
constexpr int kWaves = kBlock / 64;   // wave64
constexpr int kPartialStride = 32;    // floats per workgroup partial record (NV <= 32)

// One DPP move: lanes without a source (row edge, masked row/bank) read 0, the sum's identity.
template <int CTRL, int ROW_MASK, int BANK_MASK, typename T>
__device__ __forceinline__ T dpp_mov0(T x) {
    static_assert(sizeof(T) == 4, "32-bit values only");
    return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL,
                                                             ROW_MASK, BANK_MASK, false));
}

// Sum over the 64 lanes of a wave; the total ends in lane 63.
template <typename T>
__device__ __forceinline__ T wave_sum_to_lane63(T v) {
    v = v + dpp_mov0<0x111, 0xf, 0xf>(v);  // row_shr:1
    v = v + dpp_mov0<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = v + dpp_mov0<0x114, 0xf, 0xe>(v);  // row_shr:4
    v = v + dpp_mov0<0x118, 0xf, 0xc>(v);  // row_shr:8  -> lane 15 of each row holds the row sum
    v = v + dpp_mov0<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
    v = v + dpp_mov0<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return v;
}

__device__ __forceinline__ void store_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_sc1(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int load_sc1(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS scratch one reduction needs.
template <typename T>
struct GridReduceLds {
    T wave[kWaves][kPartialStride];
    T group[8][kPartialStride];
    T total[kPartialStride];
    int is_last;
};

// Reduce v[0..NV) over the whole grid.  Returns true in every thread of the LAST workgroup to
// arrive; there lds.total[0..NV) holds the grid totals (valid after the function returns).
// partials: gridDim.x * kPartialStride elements; ticket: one zero-initialised counter, reset to
// zero by the last workgroup so the same buffer serves the next launch on the stream.
template <int NV, typename T>
__device__ __forceinline__ bool grid_reduce(T (&v)[NV], T* __restrict__ partials,
                                            unsigned* __restrict__ ticket, GridReduceLds<T>& lds) {
    static_assert(NV <= kPartialStride, "too many values");
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const T s = wave_sum_to_lane63(v[k]);
        if (lane == 63) lds.wave[wave][k] = s;
    }
    __syncthreads();

    if (wave == 0) {
        if (lane < NV) {
            T s = lds.wave[0][lane];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) s = s + lds.wave[w][lane];
            store_sc1(&partials[blockIdx.x * kPartialStride + lane], s);
        }
        // every storing wave drains its write-through stores before the signal
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned t =
                __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds.is_last = (t == gridDim.x - 1) ? 1 : 0;
        }
    }
    __syncthreads();
    if (!lds.is_last) return false;

    // last workgroup: fixed-order sum of all partials (sc1 loads bypass this CU's L1)
    {
        const int c = tid & 31, g = tid >> 5;  // 8 groups of 32
        T s = T(0);
        if (c < NV)
            for (unsigned b = g; b < gridDim.x; b += 8) s = s + load_sc1(&partials[b * kPartialStride + c]);
        lds.group[g][c] = s;
    }
    __syncthreads();
    if (tid < NV) {
        T s = lds.group[0][tid];
#pragma unroll
        for (int g = 1; g < 8; ++g) s = s + lds.group[g][tid];
        lds.total[tid] = s;
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return true;
}

}  // namespace mmf
