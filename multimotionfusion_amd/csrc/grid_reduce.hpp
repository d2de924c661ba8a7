// grid_reduce.hpp -- deterministic wave64 -> workgroup -> grid sum for the JtJ reductions.
//
// Replaces the reference's warpReduceSum / blockReduceSum / second-launch reduceSum
// (Core/Cuda/reduce.cu:64-229, 663-720), which assume 32-wide warps and cost a second kernel
// launch plus a device synchronise per Gauss-Newton step.  Here, in ONE launch:
//   1. each lane keeps NV running sums in registers;
//   2. a wave64 reduces them in registers (no LDS traffic): float sums with the transposed butterfly
//      below, integer sums with DPP row shifts / row broadcasts;
//   3. the workgroup's waves combine through LDS in wave order;
//   4. the workgroup publishes its partial record (32 values = one 128-byte line) with 16-byte
//      write-through (sc1) stores, drains them and takes a ticket.  Tickets are SHARDED over
//      kShards counters (one 128-byte line each) plus a top counter, because one device-scope
//      counter serialises at ~12 ns per arrival (a 1200-workgroup grid would spend ~14 us in
//      its ticket alone); the last arriver of a shard arrives at the top counter;
//   5. the workgroup that completes the top counter re-reads every record with 16-byte sc1
//      buffer loads, ALL issued before the first use (one round trip instead of one per
//      record), and sums them in a fixed order.
// The summation order depends only on the launch geometry, so results are bit-reproducible
// from run to run (float atomics would not be).  The hand-off is the write-through form of the
// CDNA4 guide (sc1 payload stores, the storing wave drains with s_waitcnt vmcnt(0), ONE lane
// signals with an agent-scope atomic, the last arriver loads with sc1 loads after its add has
// returned and the other waves after a workgroup barrier).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace mmf {

constexpr int kPartialStride = 32;  // values per workgroup partial record (NV <= 32), 128 bytes
constexpr int kShards = 16;         // ticket shards; counters live 32 words (128 B) apart
constexpr int kTicketWords = (kShards + 1) * 32;

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

// ---- transposed wave reduction ----------------------------------------------------------------
// Sums N = 32 or 16 values per lane over the 64 lanes of a wave with a HALVING butterfly: each
// stage pairs lanes across one lane-index bit and halves the values a lane carries (the lane with the
// bit clear keeps the lower half, its partner the upper half): 2 instructions per surviving value
// instead of 6 per value for the plain shift tree (~65 instead of 174 for the 29 sums of a JtJ
// reduction).  Lane L returns the total of value L >> (6 - log2 N).  All 64 lanes must be active.
// v_permlane{32,16}_swap go through inline asm (the builtin of this compiler drops its second result);
// eight independent swaps per asm statement so the leading wait state is paid once.
#define MMF_SWAP8(op, v, i, h)                                                                               \
    asm volatile("s_nop 1\n\t" op " %0, %8\n\t" op " %1, %9\n\t" op " %2, %10\n\t" op " %3, %11\n\t" op      \
                 " %4, %12\n\t" op " %5, %13\n\t" op " %6, %14\n\t" op " %7, %15"                            \
                 : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]), "+v"(v[i + 4]), "+v"(v[i + 5]), \
                   "+v"(v[i + 6]), "+v"(v[i + 7]), "+v"(v[i + h]), "+v"(v[i + h + 1]), "+v"(v[i + h + 2]),     \
                   "+v"(v[i + h + 3]), "+v"(v[i + h + 4]), "+v"(v[i + h + 5]), "+v"(v[i + h + 6]), "+v"(v[i + h + 7]))
#define MMF_SWAP4(op, v, i, h)                                                                     \
    asm volatile("s_nop 1\n\t" op " %0, %4\n\t" op " %1, %5\n\t" op " %2, %6\n\t" op " %3, %7"        \
                 : "+v"(v[i]), "+v"(v[i + 1]), "+v"(v[i + 2]), "+v"(v[i + 3]), "+v"(v[i + h]),      \
                   "+v"(v[i + h + 1]), "+v"(v[i + h + 2]), "+v"(v[i + h + 3]))
// r = a + a[partner] on the lanes of the banks in `bank` (4-lane groups of a row); other lanes keep r
#define MMF_ADD_DPP(r, a, ctrl, bank) \
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:" bank : "+v"(r) : "v"(a))

__device__ __forceinline__ float wave_sum_transposed(float (&v)[32]) {
    MMF_SWAP8("v_permlane32_swap_b32", v, 0, 16);  // lane bit 5
    MMF_SWAP8("v_permlane32_swap_b32", v, 8, 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = v[i] + v[i + 16];
    MMF_SWAP8("v_permlane16_swap_b32", v, 0, 8);  // lane bit 4
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] + v[i + 8];
    float r[4], q[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // lane bit 3: partner = lane ^ 8 = rotate the 16-lane row by 8
        MMF_ADD_DPP(r[i], v[i], "row_ror:8", "0x3");
        MMF_ADD_DPP(r[i], v[i + 4], "row_ror:8", "0xc");
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // lane bit 2: partner = lane + 4 (banks 0, 2) or lane - 4 (banks 1, 3)
        MMF_ADD_DPP(q[i], r[i], "row_shl:4", "0x5");
        MMF_ADD_DPP(q[i], r[i + 2], "row_shr:4", "0xa");
    }
    // lane bit 1 (inside a quad, where bank masks cannot select): selects + quad_perm [2,3,0,1]
    const bool hi = (threadIdx.x & 2) != 0;
    const float keep = hi ? q[1] : q[0], send = hi ? q[0] : q[1];
    float t = keep + dpp_mov0<0x4E, 0xf, 0xf>(send);
    t = t + dpp_mov0<0xB1, 0xf, 0xf>(t);  // lane bit 0: quad_perm [1,0,3,2], both lanes keep the sum
    return t;
}

// 16 values: lane L returns the total of value L >> 2
__device__ __forceinline__ float wave_sum_transposed(float (&v)[16]) {
    MMF_SWAP8("v_permlane32_swap_b32", v, 0, 8);  // lane bit 5
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] + v[i + 8];
    MMF_SWAP4("v_permlane16_swap_b32", v, 0, 4);  // lane bit 4
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] + v[i + 4];
    float r[2], q;
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // lane bit 3
        MMF_ADD_DPP(r[i], v[i], "row_ror:8", "0x3");
        MMF_ADD_DPP(r[i], v[i + 2], "row_ror:8", "0xc");
    }
    MMF_ADD_DPP(q, r[0], "row_shl:4", "0x5");  // lane bit 2
    MMF_ADD_DPP(q, r[1], "row_shr:4", "0xa");
    q = q + dpp_mov0<0x4E, 0xf, 0xf>(q);  // bits 1 and 0: plain butterfly, every lane of the quad keeps the sum
    q = q + dpp_mov0<0xB1, 0xf, 0xf>(q);
    return q;
}

// the exact vector type of __builtin_amdgcn_raw_buffer_{load,store}_b128 (a converted ext_vector
// made hipcc splat element 0 -- checked in the ISA)
typedef unsigned int v4u __attribute__((__vector_size__(4 * sizeof(unsigned int))));

// LDS scratch of one reduction.
template <typename T, int BLOCK>
struct GridReduceLds {
    T wave[BLOCK / 64][kPartialStride];
    T group[32][kPartialStride + 4];  // +4: keeps 16-byte alignment, staggers banks between rows
    T total[kPartialStride];
    T group2[32][kPartialStride + 4];  // second record set of sum_partial_records2
    T total2[kPartialStride];
    int is_last;
};

// Descriptor over a partial-record array; inputs go through readfirstlane so hipcc can PROVE them
// wave-uniform (otherwise every buffer op is wrapped in a serialising "waterfall" loop).
template <typename T>
__device__ __forceinline__ auto partials_rsrc(const T* partials, unsigned nrecords) {
    const unsigned long long pbits = reinterpret_cast<unsigned long long>(partials);
    const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pbits);
    const unsigned phi = __builtin_amdgcn_readfirstlane((unsigned)(pbits >> 32));
    void* pbase = reinterpret_cast<void*>(((unsigned long long)phi << 32) | plo);
    const int pbytes = __builtin_amdgcn_readfirstlane((int)(nrecords * kPartialStride * sizeof(T)));
    return __builtin_amdgcn_make_buffer_rsrc(pbase, 0, pbytes, 0x00020000);
}

// Steps 2-4a: reduce v[] over the workgroup and store its 128-byte record (record index =
// blockIdx.x).  WRITE_THROUGH selects sc1 stores (needed when another workgroup of the SAME launch
// reads the record); plain stores are enough when the consumer is a later kernel on the stream.
// Ends with a workgroup barrier; wave 0 has issued (not drained) the stores.
template <int NV, int BLOCK, bool WRITE_THROUGH, typename T>
__device__ __forceinline__ void block_reduce_store(T (&v)[NV], T* __restrict__ partials,
                                                   GridReduceLds<T, BLOCK>& lds, unsigned record,
                                                   unsigned nrecords) {
    static_assert(NV <= kPartialStride, "too many values");
    static_assert(BLOCK % 64 == 0, "BLOCK must be a multiple of 64");
    constexpr int kWaves = BLOCK / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (std::is_same<T, float>::value) {
        constexpr int N = NV <= 16 ? 16 : 32, SH = NV <= 16 ? 2 : 1;
        float w[N];
#pragma unroll
        for (int k = 0; k < N; ++k) w[k] = k < NV ? v[k] : 0.f;
        const float t = wave_sum_transposed(w);
        if ((lane & ((1 << SH) - 1)) == 0) lds.wave[wave][lane >> SH] = t;
        if (N == 16 && (lane & 3) == 1) lds.wave[wave][16 + (lane >> 2)] = 0.f;
    } else {
    // integer sums: step-major DPP reduction: the NV chains advance together, so consecutive instructions are
    // independent and the DPP read-after-write wait states are hidden (value-major order ran one
    // 6-deep dependent chain after the other: ~2 us per workgroup in phase stamps)
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v[k] + dpp_mov0<0x111, 0xf, 0xf>(v[k]);  // row_shr:1
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v[k] + dpp_mov0<0x112, 0xf, 0xf>(v[k]);  // row_shr:2
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v[k] + dpp_mov0<0x114, 0xf, 0xe>(v[k]);  // row_shr:4
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v[k] + dpp_mov0<0x118, 0xf, 0xc>(v[k]);  // row_shr:8
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v[k] + dpp_mov0<0x142, 0xa, 0xf>(v[k]);  // row_bcast:15
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = v[k] + dpp_mov0<0x143, 0xc, 0xf>(v[k]);  // row_bcast:31
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds.wave[wave][k] = v[k];
        for (int k = NV; k < kPartialStride; ++k) lds.wave[wave][k] = T(0);
    }
    }
    __syncthreads();
    if (wave == 0 && lane < 8) {  // 8 lanes x 16 bytes = the workgroup's 128-byte record
        const auto rsrc = partials_rsrc(partials, nrecords);
        T s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = lds.wave[0][lane * 4 + j];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) s[j] = s[j] + lds.wave[w][lane * 4 + j];
        }
        v4u pk;
        pk[0] = __builtin_bit_cast(unsigned, s[0]);
        pk[1] = __builtin_bit_cast(unsigned, s[1]);
        pk[2] = __builtin_bit_cast(unsigned, s[2]);
        pk[3] = __builtin_bit_cast(unsigned, s[3]);
        __builtin_amdgcn_raw_buffer_store_b128(pk, rsrc, (int)((record * kPartialStride + lane * 4) * sizeof(T)), 0,
                                               WRITE_THROUGH ? 16 /* sc1 */ : 0);
    }
}

// Step 5: fixed-order sum of `nrecords` partial records by the first 256 threads of a workgroup
// (all BLOCK threads must call); totals land in lds.total[0..32).  All 16-byte loads of a pass are
// issued before the first use; reads beyond the last record return zero (buffer range check).
// SC1 selects loads that bypass this CU's L1 (records written during the SAME launch).
template <int BLOCK, bool SC1, typename T>
__device__ __forceinline__ void sum_partial_records(const T* __restrict__ partials, unsigned nrecords,
                                                    GridReduceLds<T, BLOCK>& lds) {
    const int tid = threadIdx.x;
    if (tid < 256) {
        const auto rsrc = partials_rsrc(partials, nrecords);
        const int rec0 = tid >> 3, q = tid & 7;
        T acc[4] = {T(0), T(0), T(0), T(0)};
        constexpr int U = 8;
        for (unsigned base = 0; base < nrecords; base += 32 * U) {
            v4u r[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                r[u] = __builtin_amdgcn_raw_buffer_load_b128(
                    rsrc, (int)(((base + u * 32 + rec0) * kPartialStride + q * 4) * sizeof(T)), 0, SC1 ? 16 : 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[0] = acc[0] + __builtin_bit_cast(T, (unsigned)r[u][0]);
                acc[1] = acc[1] + __builtin_bit_cast(T, (unsigned)r[u][1]);
                acc[2] = acc[2] + __builtin_bit_cast(T, (unsigned)r[u][2]);
                acc[3] = acc[3] + __builtin_bit_cast(T, (unsigned)r[u][3]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) lds.group[rec0][q * 4 + j] = acc[j];
    }
    __syncthreads();
    if (tid < kPartialStride) {
        T s = lds.group[0][tid];
#pragma unroll
        for (int g = 1; g < 32; ++g) s = s + lds.group[g][tid];
        lds.total[tid] = s;
    }
    __syncthreads();
}

// Two record sets at once (the finishing workgroup of a Gauss-Newton step needs the photometric
// records of THIS launch -- sc1 loads -- and the ICP records of the previous one): the first pass of
// both sets is in flight before anything is consumed, and one pair of barriers serves both.
// Totals: lds.total (set a), lds.total2 (set b).
template <int BLOCK, typename T>
__device__ __forceinline__ void sum_partial_records2(const T* __restrict__ pa, unsigned na, const T* __restrict__ pb,
                                                     unsigned nb, GridReduceLds<T, BLOCK>& lds) {
    const int tid = threadIdx.x;
    if (tid < 256) {
        const auto ra = partials_rsrc(pa, na), rb = partials_rsrc(pb, nb);
        const int rec0 = tid >> 3, q = tid & 7;
        T acc_a[4] = {T(0), T(0), T(0), T(0)}, acc_b[4] = {T(0), T(0), T(0), T(0)};
        // UA + UB loads per thread per pass = 320 + 640 records: the 300 + 600 records of a 640x480
        // step arrive in ONE round trip (three dependent passes cost 2.6 us in the phase stamps)
        constexpr int UA = 10, UB = 20;
        for (unsigned ba = 0, bb = 0; ba < na || bb < nb; ba += 32 * UA, bb += 32 * UB) {
            v4u xa[UA], xb[UB];
#pragma unroll
            for (int u = 0; u < UA; ++u)
                xa[u] = __builtin_amdgcn_raw_buffer_load_b128(
                    ra, (int)(((ba + u * 32 + rec0) * kPartialStride + q * 4) * sizeof(T)), 0, 16 /* sc1 */);
#pragma unroll
            for (int u = 0; u < UB; ++u)
                xb[u] = __builtin_amdgcn_raw_buffer_load_b128(
                    rb, (int)(((bb + u * 32 + rec0) * kPartialStride + q * 4) * sizeof(T)), 0, 0);
#pragma unroll
            for (int u = 0; u < UA; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc_a[j] = acc_a[j] + __builtin_bit_cast(T, (unsigned)xa[u][j]);
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc_b[j] = acc_b[j] + __builtin_bit_cast(T, (unsigned)xb[u][j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lds.group[rec0][q * 4 + j] = acc_a[j];
            lds.group2[rec0][q * 4 + j] = acc_b[j];
        }
    }
    __syncthreads();
    if (tid < 2 * kPartialStride) {
        const int k = tid & (kPartialStride - 1);
        T s;
        if (tid < kPartialStride) {
            s = lds.group[0][k];
#pragma unroll
            for (int g = 1; g < 32; ++g) s = s + lds.group[g][k];
            lds.total[k] = s;
        } else {
            s = lds.group2[0][k];
#pragma unroll
            for (int g = 1; g < 32; ++g) s = s + lds.group2[g][k];
            lds.total2[k] = s;
        }
    }
    __syncthreads();
}

// Workgroup-wide sum of two ints, result broadcast to every thread (the photometric pass only
// reduces {count, sum diff^2}; a full 128-byte record per workgroup would waste the consumer's
// bandwidth, so these travel as dense int2 records instead).
template <int BLOCK>
__device__ __forceinline__ void block_sum2(int& a, int& b, GridReduceLds<int, BLOCK>& lds) {
    constexpr int kWaves = BLOCK / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    a = wave_sum_to_lane63(a);
    b = wave_sum_to_lane63(b);
    __syncthreads();  // protects lds.wave against a previous use
    if (lane == 63) {
        lds.wave[wave][0] = a;
        lds.wave[wave][1] = b;
    }
    __syncthreads();
    a = lds.wave[0][0];
    b = lds.wave[0][1];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
        a += lds.wave[w][0];
        b += lds.wave[w][1];
    }
}

// Sum `n` dense int2 records (written by an EARLIER launch) over the workgroup; result in every thread.
template <int BLOCK>
__device__ __forceinline__ void sum_int2_records(const int2* __restrict__ rec, unsigned n, int& a, int& b,
                                                 GridReduceLds<int, BLOCK>& lds) {
    a = 0;
    b = 0;
    for (unsigned i = threadIdx.x; i < n; i += BLOCK) {
        const int2 r = rec[i];
        a += r.x;
        b += r.y;
    }
    block_sum2<BLOCK>(a, b, lds);
}

// Reduce v[0..NV) over the whole grid in one launch.  Returns true in every thread of the LAST
// workgroup to arrive; there lds.total[0..NV) holds the grid totals.  partials: gridDim.x records;
// tickets: kTicketWords zero-initialised words, left zeroed for the next launch.
// grid_arrive: steps 1-4 (returns true in every thread of the last workgroup to arrive; the caller
// then sums the records); grid_reduce = grid_arrive + the plain record sum.
template <int NV, int BLOCK, typename T>
__device__ __forceinline__ bool grid_arrive(T (&v)[NV], T* __restrict__ partials,
                                            unsigned* __restrict__ tickets, GridReduceLds<T, BLOCK>& lds, unsigned nblocks = 0u) {
    // nblocks (optional): the workgroups [0, nblocks) of the launch take part (the others have left: a model of a batched launch
    // that walks its image with fewer workgroups than the launch has per model)
    const int tid = threadIdx.x;
    if (nblocks == 0u) nblocks = gridDim.x;
    block_reduce_store<NV, BLOCK, true>(v, partials, lds, blockIdx.x, nblocks);
    if (tid < 64) {
        // the storing wave drains its write-through stores before the signal
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && nblocks <= 2 * (unsigned)kShards) {
            // small grid: one counter (<= 32 arrivals x ~12 ns) beats two dependent atomic round trips
            const unsigned t = __hip_atomic_fetch_add(&tickets[kShards * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == nblocks - 1;
            if (last) __hip_atomic_store(&tickets[kShards * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds.is_last = last;
        } else if (tid == 0) {
            int last = 0;
            const unsigned shard = blockIdx.x % kShards;
            const unsigned in_shard = (nblocks - shard + kShards - 1) / kShards;
            const unsigned t = __hip_atomic_fetch_add(&tickets[shard * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == in_shard - 1) {
                __hip_atomic_store(&tickets[shard * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned nshards = nblocks < (unsigned)kShards ? nblocks : (unsigned)kShards;
                const unsigned t2 = __hip_atomic_fetch_add(&tickets[kShards * 32], 1u, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                if (t2 == nshards - 1) {
                    __hip_atomic_store(&tickets[kShards * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = 1;
                }
            }
            lds.is_last = last;
        }
    }
    __syncthreads();
    return lds.is_last != 0;
}

template <int NV, int BLOCK, typename T>
__device__ __forceinline__ bool grid_reduce(T (&v)[NV], T* __restrict__ partials,
                                            unsigned* __restrict__ tickets, GridReduceLds<T, BLOCK>& lds) {
    if (!grid_arrive<NV, BLOCK>(v, partials, tickets, lds)) return false;
    sum_partial_records<BLOCK, true>(partials, gridDim.x, lds);
    return true;
}

}  // namespace mmf
