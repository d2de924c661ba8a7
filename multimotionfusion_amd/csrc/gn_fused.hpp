// gn_fused.hpp -- ONE launch per Gauss-Newton iteration of RGBDOdometry::getIncrementalTransformation
// (Core/Utils/RGBDOdometry.cpp:331-462) when both terms are on (icp && rgb, the default tracking mode).
//
// The reference runs an iteration as computeRgbResidual -> icpStep -> rgbStep -> host solve, each a launch pair plus a
// device synchronise.  The two-launch chain of track_kernels.hpp (producer launch, then rgb_step_kernel whose last
// workgroup sums the records and solves) cost ~16 us per iteration at 640x480, two thirds of it latency that is not
// memory traffic: the second kernel boundary, the re-read of the correspondence records, the write-through drain +
// ticket of the in-launch finish and the cold read of the state the finishing lane had just written.  Here:
//
//   * prologue, in EVERY workgroup: sum the partial records of the previous launch (plain loads, a kernel boundary lies
//     between), combine A = A_rgb + w^2 A_icp, solve the 6x6 system in double and update the pose -- redundantly, on
//     identical inputs, so every workgroup holds the new pose without any hand-off inside the launch.  The
//     pose-independent image loads of the pixel phase are issued BEFORE the prologue and land while it runs;
//   * photometric correspondence search (reduce.cu:722-865) for four pixels per lane, correspondences kept in
//     registers (no DataTerm records are written or re-read), the point-cloud gather of rgbStep issued with the
//     depth / intensity gathers of the search (same target pixel);
//   * the weight of rgbStep's rows needs sigma = the number of correspondences of THIS pass over the whole image
//     (RGBDOdometry.cpp:373-385, reduce.cu:506-514): every workgroup adds {1 arrival, count, sum diff^2} to a sharded
//     64-bit counter (integer atomics: exact, order independent) and carries on with the ICP rows of its pixels
//     (reduce.cu:231-397) -- by the time those are done the other workgroups have arrived, so the one poll that follows
//     finds the totals complete.  The grid is at most kGnMaxGroups workgroups per model, all co-resident;
//   * rgbStep's Jacobian rows (reduce.cu:504-535) from the registers, both 29-sum sets reduced over the workgroup and
//     stored as ONE 256-byte partial record (plain stores): the next launch's prologue -- or gn_final_kernel -- sums them.
//
// Per-pixel arithmetic is the very code of the two-launch chain (icp_project_v / icp_rows_v, rgb_rows), so Jacobian rows,
// counts and error images keep their bits; only the order in which workgroup partials are summed differs (fixed by the
// launch geometry, so results stay reproducible run to run).
#pragma once
#include "track_kernels.hpp"

namespace mmf {

constexpr int kGnRec = 64;          // floats per workgroup record: [0, 32) ICP sums, [32, 64) photometric sums
constexpr int kGnMaxGroups = 512;   // workgroups per model per launch: all resident at once, <= 32 arrivals per shard
constexpr int kGnArriveShift = 58;  // counter word: arrivals << 58 | count << 40 | sum diff^2 (per shard: < 2^6, 2^18, 2^40)
constexpr int kGnMaxPolls = 1 << 17;
constexpr int kGnMaxWaves = 8;      // a workgroup is 256 .. 512 threads: the host sizes it so that a launch has <= one workgroup per CU

struct GnIterArgs {
    IcpArgs ia;
    RgbResidualArgs ra;  // the correspondence pass's images; `corres` is not used
    const float4* cloud4;  // {X, Y, Z, 1/Z} of the model's depth (rgbStep's point and the quotient it divides by)
    float fx, fy, sobel_scale;
    LevelIntr intr;      // this level's intrinsics: the prologue prepares THIS launch's K R K^-1
    int it;              // index of the launch in the chain; 0: nothing to solve yet
    unsigned prev_groups;  // partial records the previous launch wrote
    const float* rec_in;
    float* rec_out;
    double ifx, ify;  // 1.0 / (double)intr.fx, 1.0 / (double)intr.fy
    int poll_sleep;  // s_sleep(1) repetitions between two polls of the count barrier
    int lanes;       // lanes of a workgroup that take pixels (<= blockDim.x, the rest only help with the reductions)
};

struct GnLds {
    float group[16][kGnRec + 4];
    float total[kGnRec];
    float wave[kGnMaxWaves][kGnRec];
    double sol[42];
    float pose[24];  // Rcurr[9], tcurr[3], krkinv[9], kt[3]
    double sd[16];   // the running transform the solve starts from
    float sf[13];    // Rprev[9], tprev[3], icp_weight
    int wsum[kGnMaxWaves][2];
    unsigned bar[4];
    float wtab[256];  // rgbStep's weight by |diff| for this pass's sigma
};

__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// {count, sum diff^2} of launch `it` (all its workgroups have finished)
__device__ __forceinline__ void gn_read_counts(const OdomState* st, int it, int& count, int& sigma) {
    unsigned c = 0, s = 0;
    for (int k = 0; k < kResShards; ++k) {
        const unsigned long long v = st->gn_acc[it % 3][kResStride * k];
        c += (unsigned)(v >> kResCountShift) & 0x3FFFFu;
        s += (unsigned)v;  // the sum wraps at 2^32 like the reference's int
    }
    count = (int)c, sigma = (int)s;
}

// Sums the previous launch's records, solves, leaves the new pose in lds.pose.  Two halves so that the caller can place
// pose-independent work between the issue of the loads and their first use.  All threads call both; the second ends with a
// barrier.
struct GnRecLoads {
    static constexpr int U = 20;  // 16 x 20 = 320 records per pass: the 300 of a 640x480 launch in one round trip
    v4u r[U];
    double sd;  // lane t < 16 of the first wave: element t of the running transform
    float sf;   // lane 16 + k: Rprev[k], tprev[k - 9], icp_weight (k = 12)
};
// The loads the critical path of the launch starts with: issued before anything else.  The solve's state is fetched by
// 29 lanes, one word each (lane-dependent addresses: a uniform load hipcc sinks into the one lane's branch that uses it,
// where it is a cold ~1 us round trip in the middle of the solve).
__device__ __forceinline__ void gn_records_issue(const OdomState* st, const GnIterArgs& a, GnRecLoads& rl) {
    if (a.it == 0 || threadIdx.x >= 256) return;  // uniform per wave; the first four waves fetch the records
    const auto rsrc = partials_rsrc(a.rec_in, a.prev_groups * (kGnRec / kPartialStride));
    const int q = threadIdx.x & 15, r0 = threadIdx.x >> 4;
#pragma unroll
    for (int u = 0; u < GnRecLoads::U; ++u)  // reads past the last record return zero
        rl.r[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((u * 16 + r0) * kGnRec + q * 4) * sizeof(float)), 0, 0);
    const int t = threadIdx.x;
    rl.sd = st->gn_rt[a.it & 1][t & 15];
    const int k = t < 16 ? 0 : (t > 28 ? 12 : t - 16);
    const float* fsrc = k < 9 ? &st->Rprev[k] : (k < 12 ? &st->tprev[k - 9] : &st->icp_weight);
    rl.sf = *fsrc;
}

// lead: this workgroup also stores the running transform for the next launch.  FINAL: the chain's last solve
// (gn_final_kernel): everything the host reads goes to the state, then odom_end.
// Part 1 ends with the workgroup's first barrier (the record sums are in LDS); part 2 is the one wave's work up to the pose
// and ends with the second.  Between the two the other waves are free: the caller gives them pose-independent work.
template <bool FINAL>
__device__ __forceinline__ void gn_prologue_sums(OdomState* st, const GnIterArgs& a, GnRecLoads& rl, GnLds& lds, bool lead) {
    const int tid = threadIdx.x;
    if (a.it == 0) {  // nothing to solve yet: the pose the beginning left in the state
        if (tid < 9)
            lds.pose[tid] = st->Rcurr[tid];
        else if (tid < 12)
            lds.pose[tid] = st->tcurr[tid - 9];
        else if (tid < 21)
            lds.pose[tid] = st->krkinv[tid - 12];
        else if (tid < 24)
            lds.pose[tid] = st->kt[tid - 21];
        if (lead && tid >= 64 && tid < 80) st->gn_rt[1][tid - 64] = st->resultRt[tid - 64];
        return;
    }
    if (tid < 256) {
        const auto rsrc = partials_rsrc(a.rec_in, a.prev_groups * (kGnRec / kPartialStride));
        const int q = tid & 15, r0 = tid >> 4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < GnRecLoads::U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + __builtin_bit_cast(float, (unsigned)rl.r[u][j]);
        for (unsigned base = 16 * GnRecLoads::U; base < a.prev_groups; base += 16 * GnRecLoads::U) {  // more than 320 records
#pragma unroll
            for (int u = 0; u < GnRecLoads::U; ++u)
                rl.r[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((base + u * 16 + r0) * kGnRec + q * 4) * sizeof(float)), 0, 0);
#pragma unroll
            for (int u = 0; u < GnRecLoads::U; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = acc[j] + __builtin_bit_cast(float, (unsigned)rl.r[u][j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) lds.group[r0][q * 4 + j] = acc[j];
        if (tid < 16) lds.sd[tid] = rl.sd;
        else if (tid < 29) lds.sf[tid - 16] = rl.sf;
    }
    __syncthreads();
    if (!FINAL) MMF_STAMP(3);
}

template <bool FINAL>
__device__ __forceinline__ void gn_prologue_solve(OdomState* st, const GnIterArgs& a, GnLds& lds, bool lead) {
    const int tid = threadIdx.x;
    if (a.it == 0) {
        __syncthreads();
        return;
    }
    // From here to the pose one wave works alone (its LDS accesses execute in program order: no workgroup barriers):
    // the 64 totals, RGBDOdometry.cpp:431-435 with one element of the combined system per lane, the solve on lane 0.
    if (tid < 64) {
        float s = lds.group[0][tid];
#pragma unroll
        for (int g = 1; g < 16; ++g) s = s + lds.group[g][tid];
        lds.total[tid] = s;
        __builtin_amdgcn_wave_barrier();
        const double w = (double)lds.sf[12];
        if (tid < 42) lds.sol[tid] = combine_element(FINAL ? st : nullptr, tid, w, lds.total + 32, lds.total);
        __builtin_amdgcn_wave_barrier();
        if (!FINAL) MMF_STAMP(4);
        if (tid == 0) {
            double A[36], b[6], rt[16];
            float Rprev[9], tprev[3];
            for (int k = 0; k < 36; ++k) A[k] = lds.sol[k];
            for (int k = 0; k < 6; ++k) b[k] = lds.sol[36 + k];
            for (int k = 0; k < 16; ++k) rt[k] = lds.sd[k];
            for (int k = 0; k < 9; ++k) Rprev[k] = lds.sf[k];
            for (int k = 0; k < 3; ++k) tprev[k] = lds.sf[9 + k];
            GnPose np;
            gn_solve_core(A, b, rt, Rprev, tprev, a.intr, np, a.ifx, a.ify);
            for (int k = 0; k < 9; ++k) lds.pose[k] = np.Rcurr[k], lds.pose[12 + k] = np.krkinv[k];
            for (int k = 0; k < 3; ++k) lds.pose[9 + k] = np.tcurr[k], lds.pose[21 + k] = np.kt[k];
            if (FINAL) {
                int count, sigma;
                gn_read_counts(st, a.it - 1, count, sigma);
                const ResidualDecision dec = residual_decide(count, sigma, 0, 0.f);
                st->sigma = sigma;
                st->rgbCount = count;
                st->sigmaVal = dec.sigmaVal;
                st->st.lastRGBError = dec.tmpError;
                st->st.lastRGBCount = (float)count;
                st->st.lastICPError = sqrtf(lds.total[27]) / lds.total[28];  // RGBDOdometry.cpp:412-413
                st->st.lastICPCount = lds.total[28];
                st->st.iterations_run = a.it;
                for (int k = 0; k < 16; ++k) st->resultRt[k] = rt[k];
                for (int k = 0; k < 9; ++k) st->Rcurr[k] = np.Rcurr[k], st->krkinv[k] = np.krkinv[k];
                for (int k = 0; k < 3; ++k) st->tcurr[k] = np.tcurr[k], st->kt[k] = np.kt[k];
                odom_end(st);
            } else if (lead) {
                for (int k = 0; k < 16; ++k) st->gn_rt[(a.it + 1) & 1][k] = rt[k];
            }
        }
    }
    __syncthreads();
}

// model blockIdx.y of a batched launch: everything model-side moves by the distance between the slabs
__device__ __forceinline__ void gn_batch_shift(OdomState*& st, GnIterArgs& a, const BatchDelta& bd) {
    const long long d = bd.d[blockIdx.y];
    st = batch_shift(st, d);
    a.ia.vmap_g_prev.base = batch_shift(a.ia.vmap_g_prev.base, d), a.ia.nmap_g_prev.base = batch_shift(a.ia.nmap_g_prev.base, d);
    a.ia.prev_packed = batch_shift(a.ia.prev_packed, d), a.ia.err_map = batch_shift(a.ia.err_map, d);
    a.ra.last_depth = batch_shift(a.ra.last_depth, d), a.ra.next_depth = batch_shift(a.ra.next_depth, d);
    a.ra.last_image = batch_shift(a.ra.last_image, d), a.ra.err_map = batch_shift(a.ra.err_map, d);
    a.cloud4 = batch_shift(a.cloud4, d);
    a.rec_in = batch_shift(a.rec_in, d), a.rec_out = batch_shift(a.rec_out, d);
}

// PX consecutive int16 of one row as one load
template <int PX>
__device__ __forceinline__ void load_i16(const int16_t* __restrict__ p, int (&out)[PX]) {
    if constexpr (PX == 4) {
        const short4 t = *reinterpret_cast<const short4*>(p);
        out[0] = t.x, out[1] = t.y, out[2] = t.z, out[3] = t.w;
    } else if constexpr (PX == 2) {
        const short2 t = *reinterpret_cast<const short2*>(p);
        out[0] = t.x, out[1] = t.y;
    } else {
        out[0] = *p;
    }
}

// PX pixels per lane (4, 2 or 1: the host picks it per level so that a launch has a few hundred workgroups whatever the
// level's size).  ERR: the launch also writes the two error images (the last level-0 iteration, RGBDOdometry.cpp:367,408).
// Needs cols % 4 == 0, 16-byte aligned rows, the packed model maps; blockDim.x = 256 .. 512 (a multiple of 64) >= a.lanes.
template <int PX, bool ERR>
__global__ __launch_bounds__(64 * kGnMaxWaves) void gn_iter_kernel(OdomState* st, GnIterArgs a, BatchDelta bd) {
    __shared__ GnLds lds;
    if (gridDim.y > 1) gn_batch_shift(st, a, bd);
    const OdomState* __restrict__ stc = st;  // what this launch only reads: scalar loads
    MMF_STAMP(5);
    using T = typename std::conditional<PX == 1, float, v2f>::type;
    using L = lanevec<T>;
    constexpr int W = L::W, NV = PX / W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cols = a.ra.cols, rows = a.ra.rows, N = cols * rows;
    const int nwaves = blockDim.x >> 6;
    int k0 = (blockIdx.x * a.lanes + tid) * PX;
    const bool live = tid < a.lanes && k0 < N;  // the other lanes stay active for the reductions: pixel 0, masked
    k0 = live ? k0 : 0;
    const int i = (int)__umulhi((unsigned)k0, a.ra.cols_magic), j0 = k0 - i * cols;
    const int jq = j0 & ~3, p0 = j0 & 3;  // the lane's pixels are [p0, p0 + PX) of the 4-pixel group at column jq

    // ---- the launch's critical path starts with the previous launch's records and the solve's state ----
    GnRecLoads rl;
    gn_records_issue(st, a, rl);
    __builtin_amdgcn_sched_barrier(0);

    // ---- loads that do not depend on the pose: the photometric pass's images, the current vertex / normal maps ----
    unsigned ww[4][3];
    const bool has_l = jq >= 4, has_r = jq + 4 < cols;
#pragma unroll
    for (int dr = -2; dr <= 1; ++dr) {
        const int u = min(max(i + dr, 0), rows - 1);
        const uint8_t* rowp = a.ra.next_image + (size_t)u * a.ra.ni_stride + jq;
        ww[dr + 2][1] = *reinterpret_cast<const unsigned*>(rowp);
        ww[dr + 2][0] = *reinterpret_cast<const unsigned*>(rowp - (has_l ? 4 : 0));
        ww[dr + 2][2] = *reinterpret_cast<const unsigned*>(rowp + (has_r ? 4 : 0));
    }
    const unsigned own = *reinterpret_cast<const unsigned*>(a.ra.next_image + (size_t)i * a.ra.ni_stride + jq);
    int valxs[PX], valys[PX];
    float d1s[PX];
    load_i16<PX>(a.ra.dIdx + (size_t)i * a.ra.d_stride + j0, valxs);
    load_i16<PX>(a.ra.dIdy + (size_t)i * a.ra.d_stride + j0, valys);
    load_px<PX>(a.ra.next_depth + (size_t)i * a.ra.nd_stride + j0, d1s);
    float cur[6][PX];
    {
        const float* pv = a.ia.vmap_curr.base + (size_t)i * a.ia.vmap_curr.stride + j0;
        const float* pn = a.ia.nmap_curr.base + (size_t)i * a.ia.nmap_curr.stride + j0;
        const size_t sv = (size_t)rows * a.ia.vmap_curr.stride, sn = (size_t)rows * a.ia.nmap_curr.stride;
        load_px<PX>(pv, cur[0]);
        load_px<PX>(pv + sv, cur[1]);
        load_px<PX>(pv + 2 * sv, cur[2]);
        load_px<PX>(pn, cur[3]);
        load_px<PX>(pn + sn, cur[4]);
        load_px<PX>(pn + 2 * sn, cur[5]);
    }
    // the model pose of the frame (constant over the chain)
    IcpPose P;
#pragma unroll
    for (int k = 0; k < 9; ++k) P.Rprev_inv[k] = stc->Rprev_inv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) P.tprev[k] = stc->tprev[k];
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP(0);

    // the counters of the launch after this one start from zero (nobody else touches that buffer during this launch)
    if (blockIdx.x == 0 && tid >= 64 && tid < 64 + kResShards) st->gn_acc[(a.it + 1) % 3][kResStride * (tid - 64)] = 0ull;

    // what the correspondence search can decide without the pose (reduce.cu:773-797): the 4x4 "all neighbours > 0"
    // windows from the twelve words, the gradient test -- done by the waves that wait for the solve, beside it
    bool cand[PX];
    auto pose_free_work = [&]() {
    unsigned nz[3] = {0x80808080u, 0x80808080u, 0x80808080u};
#pragma unroll
    for (int dr = -2; dr <= 1; ++dr) {
        const bool rowin = (i + dr) >= 0 && (i + dr) < rows;
#pragma unroll
        for (int k = 0; k < 3; ++k) nz[k] &= rowin ? nonzero_bytes(ww[dr + 2][k]) : 0x80808080u;
    }
    const unsigned okw = (has_l ? byte_flags_to_bits(nz[0]) : 0xFu) | (byte_flags_to_bits(nz[1]) << 4) |
                         ((has_r ? byte_flags_to_bits(nz[2]) : 0xFu) << 8);
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int x = j0 + p;
        const bool valid = live && x < cols - 5 && i < rows - 1 && ((okw >> (p0 + p + 2)) & 0xFu) == 0xFu;
        const float mTwo = (float)((valxs[p] * valxs[p]) + (valys[p] * valys[p]));
        cand[p] = valid && mTwo >= a.ra.min_scale && !(d1s[p] != d1s[p]);
    }
    __builtin_amdgcn_sched_barrier(0);
    };
    gn_prologue_sums<false>(st, a, rl, lds, blockIdx.x == 0);
    __builtin_amdgcn_sched_barrier(0);
    if (wave != 0) pose_free_work();  // beside the one wave that solves
    __builtin_amdgcn_sched_barrier(0);
    gn_prologue_solve<false>(st, a, lds, blockIdx.x == 0);
    if (wave == 0) pose_free_work();
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP(8);
    float K[9], kt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) P.Rcurr[k] = uniform_f(lds.pose[k]), K[k] = uniform_f(lds.pose[12 + k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) P.tcurr[k] = uniform_f(lds.pose[9 + k]), kt[k] = uniform_f(lds.pose[21 + k]);

    // ---- photometric correspondence search: warp (reduce.cu:799-812), then its gathers at once ----
    bool inb[PX];
    int u0s[PX], v0s[PX];
    float td1s[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int x = j0 + p, y = i;
        const float d1 = d1s[p];
        inb[p] = false;
        u0s[p] = v0s[p] = 0;
        td1s[p] = 0.f;
        if (cand[p]) {  // kept under its `if`: whole waves skip the divisions
            td1s[p] = (float)(d1 * (K[6] * x + K[7] * y + K[8]) + kt[2]);
            u0s[p] = float2int_rn((d1 * (K[0] * x + K[1] * y + K[2]) + kt[0]) / td1s[p]);
            v0s[p] = float2int_rn((d1 * (K[3] * x + K[4] * y + K[5]) + kt[1]) / td1s[p]);
            inb[p] = u0s[p] >= 0 && v0s[p] >= 0 && u0s[p] < cols && v0s[p] < rows;
        }
    }
    struct f3pk {
        float x, y, z;
    };
    float d0s[PX];
    uint8_t lis[PX];
    f3pk gv[PX], gn[PX];
    float4 cl[PX];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PX; ++p) {  // addresses clamped, masks applied afterwards
        const int gu = inb[p] ? u0s[p] : 0, gvv = inb[p] ? v0s[p] : 0;
        d0s[p] = a.ra.last_depth[(size_t)gvv * a.ra.ld_stride + gu];
        lis[p] = a.ra.last_image[(size_t)gvv * a.ra.li_stride + gu];
        cl[p] = a.cloud4[(size_t)gvv * cols + gu];  // rgbStep's point (reduce.cu:522) and 1 / Z
    }
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP(9);

    // ---- ICP: projection into the model's camera (reduce.cu:257-273) while those are in flight, then its gathers ----
    IcpProj<T> pr[NV];
#pragma unroll
    for (int h = 0; h < NV; ++h) {
        f3t<T> v;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            L::set(v.x, e, cur[0][h * W + e]);
            L::set(v.y, e, cur[1][h * W + e]);
            L::set(v.z, e, cur[2][h * W + e]);
        }
        pr[h] = icp_project_v<T>(P, a.ia, v);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < PX; ++q) {
        const f3pk* src = reinterpret_cast<const f3pk*>(a.ia.prev_packed) + 2 * ((size_t)pr[q / W].uy[q % W] * cols + pr[q / W].ux[q % W]);
        gv[q] = src[0];
        gn[q] = src[1];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- photometric: accept / reject (reduce.cu:813-836), the workgroup's {count, sum diff^2}, arrival ----
    RgbLane<PX> ph;
    int cnt = 0, sq = 0;
    float perr[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const bool hit = inb[p] && d0s[p] > 0 && fabsf(td1s[p] - d0s[p]) <= a.ra.max_depth_delta && lis[p] != 0;
        const int idiff = (int)((own >> (8 * (p0 + p))) & 0xFFu) - (int)lis[p];  // == (float)next - (float)last, exactly
        const int vy = hit ? idiff * idiff : 0;                                   // == (int)(diff * diff)
        perr[p] = hit ? 0.001f * vy : 0.0f;
        cnt += hit ? 1 : 0;
        sq += vy;
        ph.c[p].diff = hit ? (float)idiff : 0.f;
        ph.c[p].valid = hit ? 1 : 0;
        ph.X[p] = cl[p].x, ph.Y[p] = cl[p].y, ph.Z[p] = cl[p].z, ph.invz[p] = cl[p].w;
        ph.gx[p] = valxs[p], ph.gy[p] = valys[p];
    }
    if (ERR && a.ra.err_map && live) store_px<PX>(a.ra.err_map + (size_t)i * a.ra.err_stride + j0, perr);
#ifdef MMF_STAMPS
    MMF_STAMP(6);
#endif
    cnt = wave_sum_to_lane63(cnt);
    sq = wave_sum_to_lane63(sq);
    if (lane == 63) lds.wsum[wave][0] = cnt, lds.wsum[wave][1] = sq;
    __syncthreads();
    if (tid == 0) {  // one arrival per workgroup
        unsigned c = 0, s2 = 0;
        for (int wv = 0; wv < nwaves; ++wv) c += (unsigned)lds.wsum[wv][0], s2 += (unsigned)lds.wsum[wv][1];
        const unsigned long long word = (1ull << kGnArriveShift) | ((unsigned long long)c << kResCountShift) | (unsigned long long)s2;
        (void)__hip_atomic_fetch_add(&st->gn_acc[a.it % 3][kResStride * (blockIdx.x % kResShards)], word, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP(10);

    // ---- ICP: Jacobian rows (reduce.cu:275-368) while the other workgroups arrive ----
    T isum[29];
    float ierr[PX];
#pragma unroll
    for (int h = 0; h < NV; ++h) {
        f3t<T> n, vp, np;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            L::set(n.x, e, cur[3][h * W + e]), L::set(n.y, e, cur[4][h * W + e]), L::set(n.z, e, cur[5][h * W + e]);
            L::set(vp.x, e, gv[h * W + e].x), L::set(vp.y, e, gv[h * W + e].y), L::set(vp.z, e, gv[h * W + e].z);
            L::set(np.x, e, gn[h * W + e].x), L::set(np.y, e, gn[h * W + e].y), L::set(np.z, e, gn[h * W + e].z);
        }
        float er[W];
        if (h == 0)
            icp_rows_v<ERR, true, T>(P, a.ia, pr[h], live, n, vp, np, isum, er);
        else
            icp_rows_v<ERR, false, T>(P, a.ia, pr[h], live, n, vp, np, isum, er);
        if (ERR) {
#pragma unroll
            for (int e = 0; e < W; ++e) ierr[h * W + e] = er[e];
        }
    }
    if (ERR && a.ia.err_map && live) store_px<PX>(a.ia.err_map + (size_t)i * a.ia.err_stride + j0, ierr);
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP(11);

    // the ICP sums are complete: their reduction over the wave goes here, into the wait for the other workgroups
    {
        float s32[32];
#pragma unroll
        for (int k = 0; k < 29; ++k) s32[k] = L::hsum(isum[k]);
        s32[29] = s32[30] = s32[31] = 0.f;
        const float t = wave_sum_transposed(s32);
        if ((lane & 1) == 0) lds.wave[wave][lane >> 1] = t;
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the count barrier: wave 0 polls the 16 shards until every workgroup of this model has arrived, then decides
    //      sigma (RGBDOdometry.cpp:373-385) and lays out the pass's 256 weights ----
    if (wave == 0) {
        const unsigned long long* acc = st->gn_acc[a.it % 3];
        unsigned c = 0, s2 = 0, ok = 0;
        for (int poll = 0; poll < kGnMaxPolls; ++poll) {
            const unsigned long long v = lane < kResShards ? __hip_atomic_load(acc + kResStride * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            unsigned arr = wave_sum_to_lane63((unsigned)(v >> kGnArriveShift));
            c = wave_sum_to_lane63((unsigned)(v >> kResCountShift) & 0x3FFFFu);
            s2 = wave_sum_to_lane63((unsigned)v);  // wraps at 2^32 like the reference's int
            arr = (unsigned)__builtin_amdgcn_readlane((int)arr, 63);
            if (arr == gridDim.x) {
                ok = 1;
                break;
            }
            for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
        }
        c = (unsigned)__builtin_amdgcn_readlane((int)c, 63), s2 = (unsigned)__builtin_amdgcn_readlane((int)s2, 63);
        // sigmaVal of residual_decide without its double-precision square root and division: tmpError = sqrt(sum) / count is
        // zero exactly when sum is (count <= 2^19: no underflow), and NaN -- not zero -- for 0 / 0
        const float sigma_val = ((int)s2 == 0 && (int)c != 0) ? 1.0f : (float)(int)c;
#pragma unroll
        for (int q = 0; q < 4; ++q) lds.wtab[lane * 4 + q] = rgb_weight(sigma_val, (float)(lane * 4 + q));
        if (lane == 63) lds.bar[2] = ok, lds.bar[3] = __builtin_bit_cast(unsigned, sigma_val);
    }
    __syncthreads();
    MMF_STAMP(12);
    if (!lds.bar[2] && blockIdx.x == 0 && tid == 0) st->gn_fault = 1;  // a workgroup of this launch never arrived
    const float sigmaVal = __builtin_bit_cast(float, lds.bar[3]);

    // ---- photometric: rgbStep's rows (reduce.cu:504-535) ----
    float psum[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) psum[k] = 0.f;
    rgb_rows<PX>(a.sobel_scale, a.fx, a.fy, sigmaVal, live, ph, psum, lds.wtab, true);

    // ---- their sums over the wave; both sum sets over the workgroup -> one 256-byte record ----
    {
        float s32[32];
#pragma unroll
        for (int k = 0; k < 29; ++k) s32[k] = psum[k];
        s32[29] = s32[30] = s32[31] = 0.f;
        const float t2 = wave_sum_transposed(s32);
        if ((lane & 1) == 0) lds.wave[wave][32 + (lane >> 1)] = t2;
    }
    __syncthreads();
    if (tid < 16) {
        float4 s = *reinterpret_cast<const float4*>(&lds.wave[0][tid * 4]);
        for (int wv = 1; wv < nwaves; ++wv) {  // fixed order: the launch geometry decides the sum, not the timing
            const float4 t = *reinterpret_cast<const float4*>(&lds.wave[wv][tid * 4]);
            s.x = s.x + t.x, s.y = s.y + t.y, s.z = s.z + t.z, s.w = s.w + t.w;
        }
        *reinterpret_cast<float4*>(a.rec_out + (size_t)blockIdx.x * kGnRec + tid * 4) = s;
    }
    MMF_STAMP(13);
}

// the chain's last solve + RGBDOdometry.cpp:464-467, 475-476: one workgroup per model
__global__ __launch_bounds__(kBlock) void gn_final_kernel(OdomState* st, GnIterArgs a, BatchDelta bd) {
    __shared__ GnLds lds;
    if (gridDim.x > 1) {
        const long long d = bd.d[blockIdx.x];
        st = batch_shift(st, d);
        a.rec_in = batch_shift(a.rec_in, d);
    }
    GnRecLoads rl;
    gn_records_issue(st, a, rl);
    gn_prologue_sums<true>(st, a, rl, lds, true);
    gn_prologue_solve<true>(st, a, lds, true);
}

// the same, and the result goes to the host in the same launch (odom_publish_kernel's two waves behind the solve: the copy
// into the host's pinned state + sequence number, Model::computeFusionWeight of the new pose for an early fuse pass)
__global__ __launch_bounds__(kBlock) void gn_final_publish_kernel(OdomState* st, GnIterArgs a, BatchDelta bd, PublishTargets to,
                                                                  unsigned seq) {
    __shared__ GnLds lds;
    if (gridDim.x > 1) {
        const long long d = bd.d[blockIdx.x];
        st = batch_shift(st, d);
        a.rec_in = batch_shift(a.rec_in, d);
    }
    GnRecLoads rl;
    gn_records_issue(st, a, rl);
    gn_prologue_sums<true>(st, a, rl, lds, true);
    gn_prologue_solve<true>(st, a, lds, true);  // ends with a workgroup barrier: the state lane 0 stored is visible to the workgroup
    if (threadIdx.x >= 64) {
        if (threadIdx.x == 64) odom_fusion_weight(st);
        return;
    }
    odom_publish_wave(st, to.host[blockIdx.x], seq, threadIdx.x);
}

}  // namespace mmf
