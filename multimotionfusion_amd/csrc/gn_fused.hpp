// gn_fused.hpp -- ONE launch per Gauss-Newton iteration of RGBDOdometry::getIncrementalTransformation
// (Core/Utils/RGBDOdometry.cpp:331-462) when both terms are on (icp && rgb, the default tracking mode).
//
// The reference runs an iteration as computeRgbResidual -> icpStep -> rgbStep -> host solve, each a launch pair plus a
// device synchronise.  The two-launch chain of track_kernels.hpp (producer launch, then rgb_step_kernel whose last
// workgroup sums the records and solves) cost ~16 us per iteration at 640x480, two thirds of it latency that is not
// memory traffic.  Here a workgroup is ONE SOLVER WAVE plus four (or more) PIXEL WAVES, one workgroup per CU:
//
//   * solver wave, in EVERY workgroup: reads the previous launch's 58 sums (64-bit fixed-point
//     integers in sixteen shards, OdomState::gn_sum: plain loads, a kernel boundary lies between), combines A = A_rgb + w^2 A_icp, solves the
//     6x6 system in double and updates the pose -- redundantly, on identical inputs, so every workgroup holds the new pose
//     without any hand-off inside the launch.  It has no pixels: while it solves, the pixel waves' image loads land and
//     they decide what the correspondence search can decide without the pose; once the pose is out, nothing of the pixel
//     phase waits for the solver's share of it (round 3 gave the solver wave pixels as well: the workgroup then waited
//     for that wave's pose-free work, and at 640x480 a fifth wave shared its SIMD);
//   * pixel waves: photometric correspondence search (reduce.cu:722-865) for PX pixels per lane, correspondences kept in
//     registers (no DataTerm records are written or re-read), the point-cloud gather of rgbStep issued with the
//     depth / intensity gathers of the search (same target pixel); ICP projection and gathers (reduce.cu:231-397);
//   * the weight of rgbStep's rows needs sigma = the number of correspondences of THIS pass over the whole image
//     (RGBDOdometry.cpp:373-385, reduce.cu:506-514): the solver wave adds the workgroup's {1 arrival, count, sum diff^2} to
//     a sharded 64-bit counter (integer atomics: exact, order independent), polls until every workgroup of the launch has
//     arrived, decides sigma and lays out the pass's 256 possible weights in LDS, while the pixel waves compute their ICP
//     rows and reduce them.  The grid is at most kGnMaxGroups workgroups per model, all co-resident (the host checks the
//     occupancy: odom_fused_chain_ok);
//   * rgbStep's Jacobian rows (reduce.cu:504-535) from the registers, both 29-sum sets reduced over the workgroup in float
//     and added to the launch's sums as fixed-point integers by atomics (exact, order independent): the next
//     launch -- or gn_final_kernel -- reads 7.4 KB of totals instead of a 256-byte record per workgroup (round 3: 61 KB
//     fetched by every workgroup, 0.8 us of a 640x480 launch).
//
// Several models of a frame share ONE launch per iteration (gn_iter_mixed_kernel, further down): the camera model walks the
// image as described here, an object model walks the box of its own depth and the rectangle of sensor pixels its prediction
// can reach under the iteration's pose (gn_sparse_icp_box, gn_pixel_waves_sparse), with 4-wave workgroups for everybody.
//
// PX = 1, 2, 4 or 5 pixels per lane, chosen by the host per level (gn_geometry) so that a launch has at most one
// workgroup per CU with exactly four pixel waves wherever the level allows: 640x480 = 240 workgroups x 256 lanes x 5.
// Per-pixel arithmetic is the very code of the two-launch chain (icp_project_v / icp_rows_v, rgb_rows), so Jacobian rows,
// counts and error images keep their bits; only the order in which partials are summed differs (fixed by the launch
// geometry, so results stay reproducible run to run).
#pragma once
#include "track_kernels.hpp"

#ifndef MMF_ABL
#define MMF_ABL 0  // diagnostic builds only (tools/gn_floor_probe.sh): 64 = the dependent phases alone, no pixel work; + 4096 = no solve either
#endif

namespace mmf {

constexpr int kGnRec = 64;          // floats per workgroup record: [0, 32) ICP sums, [32, 64) photometric sums
constexpr int kGnMaxGroups = 512;   // workgroups per model per launch: all resident at once, <= 32 arrivals per shard
constexpr int kGnArriveShift = 58;  // counter word: arrivals << 58 | count << 40 | sum diff^2 (per shard: < 2^6, 2^18, 2^40)
constexpr int kGnMaxPolls = 1 << 17;
constexpr int kGnMaxWaves = 8;      // the solver wave + up to seven pixel waves (the host sizes it: one workgroup per CU)
constexpr int kGnSparseLanes = 256; // pixel lanes of a workgroup that walks an object model by its extents (gn_pixel_waves_sparse's LDS)
constexpr int kGnStashWords = 5;    // ... and the words per pixel it keeps there

struct GnIterArgs {
    IcpArgs ia;
    RgbResidualArgs ra;  // the correspondence pass's images; `corres` is not used
    const float4* cloud4;  // {X, Y, Z, 1/Z} of the model's depth (rgbStep's point and the quotient it divides by)
    float fx, fy, sobel_scale;
    LevelIntr intr;      // this level's intrinsics: the prologue prepares THIS launch's K R K^-1
    int it;              // index of the launch in the chain; 0: nothing to solve yet
    double ifx, ify;  // 1.0 / (double)intr.fx, 1.0 / (double)intr.fy
    int poll_sleep;  // s_sleep(1) repetitions between two polls of the count barrier
    int lanes;       // pixel lanes of a workgroup (<= blockDim.x - 64; the rest only help with the reductions)
    int max_polls;   // kGnMaxPolls (a test forces a time-out with 0: mmf_debug_force_gn_fault)
    int check_sparse;  // test mode (mmf_debug_set_sparse_check): object models walk the WHOLE image and count what icpStep
                       // accepts outside the rectangle they would have walked (OdomState::gn_dbg_outside)
};

struct GnLds {
    double dtot[kGnRec];  // the previous launch's totals: [0, 29) ICP, [29, 58) photometric
    float wave[kGnMaxWaves][kGnRec];
    double sol[42];
    double xch[12];  // gn_solve_rows: the rows of the new running transform
    float pose[24];  // Rcurr[9], tcurr[3], krkinv[9], kt[3]
    double sd[16];   // the running transform the solve starts from
    float sf[13];    // Rprev[9], tprev[3], icp_weight
    int wsum[kGnMaxWaves][2];
    unsigned bar[4];
    float wtab[256];  // rgbStep's weight by |diff| for this pass's sigma
    int sbox[10];     // an object model's ICP walk of this launch (gn_sparse_icp_box): x0, y0, lanes per row, rows, passes
};

// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of the
// wave (s_waitcnt vmcnt(0)): gathers in flight across a barrier would be waited for in front of it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

#ifdef MMF_STAMPS  // the pixel phases are stamped by the first lane of the first pixel wave
#define MMF_STAMP_PX(i)                                                                        \
    do {                                                                                       \
        if (g_mmf_dbg && threadIdx.x == 64) g_mmf_dbg[blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MMF_STAMP_PX(i) \
    do {                \
    } while (0)
#endif

// ---- the previous launch's sums and the solve's state (one wave) ----
// sigma of a pass as rgbStep uses it (residual_decide's sigmaVal without its double-precision square root and division:
// tmpError = sqrt(sum) / count is zero exactly when sum is (count <= 2^19: no underflow), and NaN -- not zero -- for 0 / 0)
__device__ __forceinline__ float gn_sigma_val(unsigned count, unsigned sumsq) {
    return ((int)sumsq == 0 && (int)count != 0) ? 1.0f : (float)(int)count;
}
// binary exponent of the photometric sums' fixed-point scale: their rows carry w = 1 / (sigma + |d|), so the scale follows
// sigma^2; a function of the pass's sigma alone, which every workgroup of the launch and the next launch's readers know
__device__ __forceinline__ int gn_rgb_exp(float sigma_val) {
    const float s = sigma_val > 1.0f ? sigma_val : 1.0f;  // (also NaN -> 1)
    return kGnSumRgbExp0 + 2 * ((int)((__builtin_bit_cast(unsigned, s) >> 23) & 0xFFu) - 127);
}
struct GnSumLoads {
    long long q[kGnSumShards];    // lane k < 58: sum k of every shard
    unsigned long long cnt;  // lane t < 16: shard t of the previous launch's {arrivals, count, sum diff^2}
    double sd;               // lane t < 16: element t of the running transform
    float sf;                // lane 16 + k: Rprev[k], tprev[k - 9], icp_weight (k = 12)
};
// every lane loads (lane-dependent addresses: a uniform load hipcc sinks into the one lane's branch that uses it, where it
// is a cold ~1 us round trip in the middle of the solve)
__device__ __forceinline__ void gn_sums_issue(const OdomState* st, const GnIterArgs& a, GnSumLoads& sl, int lane) {
    const int prev = (a.it + 2) % 3, k = lane < 58 ? lane : 57;
#pragma unroll
    for (int x = 0; x < kGnSumShards; ++x) sl.q[x] = st->gn_sum[prev][x][k];
    sl.cnt = st->gn_acc[prev][kResStride * (lane & 15)];
    sl.sd = st->gn_rt[a.it & 1][lane & 15];
    const int j = lane < 16 ? 0 : (lane > 28 ? 12 : lane - 16);
    const float* fsrc = j < 9 ? &st->Rprev[j] : (j < 12 ? &st->tprev[j - 9] : &st->icp_weight);
    sl.sf = *fsrc;
}
// the totals as doubles -> lds.dtot, the state -> lds.sd / lds.sf; returns the previous pass's {count, sum diff^2}
__device__ __forceinline__ void gn_totals_to_lds(const GnSumLoads& sl, GnLds& lds, int lane, unsigned& count, unsigned& sumsq) {
    unsigned c = lane < kResShards ? (unsigned)(sl.cnt >> kResCountShift) & 0x3FFFFu : 0u;
    unsigned s2 = lane < kResShards ? (unsigned)sl.cnt : 0u;  // wraps at 2^32 like the reference's int
    c = wave_sum_to_lane63(c), s2 = wave_sum_to_lane63(s2);
    count = (unsigned)__builtin_amdgcn_readlane((int)c, 63), sumsq = (unsigned)__builtin_amdgcn_readlane((int)s2, 63);
    const int e = lane < 29 ? kGnSumIcpExp : gn_rgb_exp(gn_sigma_val(count, sumsq));
    long long q = sl.q[0];
#pragma unroll
    for (int x = 1; x < kGnSumShards; ++x) q += sl.q[x];
    if (lane < 58) lds.dtot[lane] = __builtin_ldexp((double)q, -e);
    if (lane < 16) lds.sd[lane] = sl.sd;
    else if (lane < 29) lds.sf[lane - 16] = sl.sf;
}
// lane k < 58 adds sum k of this workgroup (float) to the launch's sums; false: the value does not fit the fixed-point range
__device__ __forceinline__ bool gn_sum_add(OdomState* st, int it, int lane, float v, int rgb_exp, unsigned bid) {
    const double d = __builtin_ldexp((double)v, lane < 29 ? kGnSumIcpExp : rgb_exp);  // exact
    const bool ok = __builtin_fabs(d) < 9007199254740992.0;  // 2^53 per workgroup: 512 of them fit 2^62 (a NaN fails too)
    const long long q = ok ? (long long)__builtin_rint(d) : 0ll;
    if (lane < 58)
        (void)__hip_atomic_fetch_add(&st->gn_sum[it % 3][bid % kGnSumShards][lane], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (same global_atomic_add_x2 as workgroup scope on gfx950; agent is what the memory model asks for)
    return ok || lane >= 58;
}

// it == 0: nothing to solve yet, the pose the beginning left in the state.  One wave.
__device__ __forceinline__ void gn_first_pose(OdomState* st, GnLds& lds, bool lead, int lane) {
    if (lane < 9)
        lds.pose[lane] = st->Rcurr[lane];
    else if (lane < 12)
        lds.pose[lane] = st->tcurr[lane - 9];
    else if (lane < 21)
        lds.pose[lane] = st->krkinv[lane - 12];
    else if (lane < 24)
        lds.pose[lane] = st->kt[lane - 21];
    if (lead && lane >= 32 && lane < 48) st->gn_rt[1][lane - 32] = st->resultRt[lane - 32];
}

// The solve and pose update of gn_solve_core (RGBDOdometry.cpp:435-460 + OdometryProvider.h:69-89, RGBDOdometry.cpp:348-358)
// for a whole wave instead of one lane.  On one lane it is ~330 dependent double-precision instructions at ~12 cycles each:
// 1.8 us of every launch with nothing beside it (tools/gn_iter_probe.py: barrier A -> barrier B 2.4 us).  Here EVERY lane
// runs the 6x6 LDL^T (same instructions on the same operands: no divergence, the solution sits in every lane's registers),
// then lanes 0..2 take one ROW each of what follows -- the Rodrigues rotation, the 3x4 product with the running transform,
// the isometry inverse and Rprev product, the 3x3 inverse, K R K^-1 and K t -- and exchange rows through LDS (in-order
// within a wave) or v_readlane.  Same formulas in the same order per element; contraction may differ (~1e-16 relative
// before the casts to float).  A, b: the combined system (LDS, 42 doubles); rt_old: the running transform (LDS, 16);
// sf: Rprev[9], tprev[3] (LDS); xch: 12 doubles of LDS for the exchange; pose: Rcurr[9], tcurr[3], krkinv[9], kt[3] (LDS).
// Returns this lane's row of the new running transform in nr[4] (lanes >= 3 repeat row 0).
__device__ __forceinline__ void gn_solve_rows(const double* A_lds, const double* rt_old, const float* sf, const LevelIntr& in,
                                              double ifx, double ify, int lane, double* xch, float* pose, double (&nr)[4]) {
#pragma clang fp contract(fast)
    if (MMF_ABL & 4096) {
        if (lane < 24) pose[lane] = (lane % 4 == 0 ? 1.f : 0.f) + (float)A_lds[lane] * 1e-30f;
        for (int c = 0; c < 4; ++c) nr[c] = rt_old[c];
        return;
    }
    double A[36], b[6], result[6];
    for (int k = 0; k < 36; ++k) A[k] = A_lds[k];
    for (int k = 0; k < 6; ++k) b[k] = A_lds[36 + k];
    MMF_SOLVE_STAMP(1);
    ldlt_solve_recip<6>(A, b, result);
    MMF_SOLVE_STAMP(2);
    const int r = lane < 3 ? lane : 0;
    // R = cc I + cb v v^T + ca [v]x (OdometryProvider.h:32-67; small increments: rodrigues_increment's series in y = |r|^2)
    double vx = result[3], vy = result[4], vz = result[5];
    const double y = vx * vx + vy * vy + vz * vz;
    double ca = 0.0, cb = 0.0, cc = 1.0;
    if (!(y < 0.015625)) {  // |r| >= 1/8 or not a number (wave-uniform): the literal form
        const double theta = sqrt(y);
        if (theta >= DBL_EPSILON) {
            const double itheta = theta ? 1. / theta : 0.;
            vx *= itheta, vy *= itheta, vz *= itheta;
            cc = cos(theta), ca = sin(theta), cb = 1. - cc;
        }
    } else if (y >= DBL_EPSILON * DBL_EPSILON) {
        ca = 1.0 + y * (-1.0 / 6 + y * (1.0 / 120 + y * (-1.0 / 5040 + y * (1.0 / 362880 + y * (-1.0 / 39916800 + y * (1.0 / 6227020800.0))))));
        cb = 0.5 + y * (-1.0 / 24 + y * (1.0 / 720 + y * (-1.0 / 40320 + y * (1.0 / 3628800 + y * (-1.0 / 479001600 + y * (1.0 / 87178291200.0))))));
        cc = 1.0 - cb * y;
    }
    {
        const double vr = r == 0 ? vx : (r == 1 ? vy : vz);
        // row r of [v]x: (0, -vz, vy), (vz, 0, -vx), (-vy, vx, 0)
        const double e0 = r == 0 ? 0.0 : (r == 1 ? vz : -vy), e1 = r == 0 ? -vz : (r == 1 ? 0.0 : vx), e2 = r == 0 ? vy : (r == 1 ? -vx : 0.0);
        const double u0 = (r == 0 ? cc : 0.0) + cb * (vr * vx) + ca * e0;
        const double u1 = (r == 1 ? cc : 0.0) + cb * (vr * vy) + ca * e1;
        const double u2 = (r == 2 ? cc : 0.0) + cb * (vr * vz) + ca * e2;
        const double res = r == 0 ? result[0] : (r == 1 ? result[1] : result[2]);
        // resultRt <- [Rup | result(0..2); 0 0 0 1] * resultRt: the translation column multiplies the last row as in the
        // reference's full 4x4 product (a non-finite solution poisons the whole matrix, gn_solve_core)
        for (int c = 0; c < 4; ++c) nr[c] = u0 * rt_old[c] + u1 * rt_old[4 + c] + u2 * rt_old[8 + c] + res * rt_old[12 + c];
    }
    if (lane < 3)
        for (int c = 0; c < 4; ++c) xch[r * 4 + c] = nr[c];
    __builtin_amdgcn_wave_barrier();  // (a wave's LDS accesses execute in program order)
    MMF_SOLVE_STAMP(14);
    double N[12];
    for (int k = 0; k < 12; ++k) N[k] = xch[k];
    {  // currentT = [Rprev|tprev] * rgbOdom.inverse(); isometry inverse = (R^T, -R^T t): row r of Rcurr, tcurr[r]
#pragma clang fp contract(off)
        float RoT[9], to[3], ti[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) RoT[j * 3 + i] = (float)N[i * 4 + j];
            to[i] = (float)N[i * 4 + 3];
        }
        for (int i = 0; i < 3; ++i) ti[i] = -RoT[i * 3 + 0] * to[0] + -RoT[i * 3 + 1] * to[1] + -RoT[i * 3 + 2] * to[2];
        const float p0 = sf[r * 3 + 0], p1 = sf[r * 3 + 1], p2 = sf[r * 3 + 2];
        float rc[3], tc = 0.f;
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;  // matmul<3, float>: s = 0; s += a * b
            acc += p0 * RoT[0 * 3 + j];
            acc += p1 * RoT[1 * 3 + j];
            acc += p2 * RoT[2 * 3 + j];
            rc[j] = acc;
        }
        tc += p0 * ti[0];
        tc += p1 * ti[1];
        tc += p2 * ti[2];
        tc = tc + sf[9 + r];
        if (lane < 3) {
            for (int j = 0; j < 3; ++j) pose[r * 3 + j] = rc[j];
            pose[9 + r] = tc;
        }
    }
    {  // rgb_prepare: row r of (resultRt^-1)'s rotation, K R, (K R) K^-1, K t
        const int ca1 = r == 2 ? 0 : r + 1, cb2 = r == 0 ? 2 : r - 1;  // columns (r + 1) % 3 and (r + 2) % 3 of the 3x3 block
        const double a0 = xch[0 * 4 + ca1], a1 = xch[1 * 4 + ca1], a2 = xch[2 * 4 + ca1];
        const double b0 = xch[0 * 4 + cb2], b1 = xch[1 * 4 + cb2], b2 = xch[2 * 4 + cb2];
        const double c00 = N[5] * N[10] - N[6] * N[9], c01 = N[6] * N[8] - N[4] * N[10], c02 = N[4] * N[9] - N[5] * N[8];
        const double det = N[0] * c00 + N[1] * c01 + N[2] * c02;
        const double id = tail_rcp(det);
        // inverse[r][j] = (m[(j+1)%3][(r+1)%3] m[(j+2)%3][(r+2)%3] - m[(j+1)%3][(r+2)%3] m[(j+2)%3][(r+1)%3]) / det
        const double i0 = (a1 * b2 - b1 * a2) * id, i1 = (a2 * b0 - b2 * a0) * id, i2 = (a0 * b1 - b0 * a1) * id;
        const double t3 = -(i0 * N[3] + i1 * N[7] + i2 * N[11]);
        auto from_lane2 = [](double v) {
            const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 2), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 2);
            return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        };
        const double r20 = from_lane2(i0), r21 = from_lane2(i1), r22 = from_lane2(i2), t32 = from_lane2(t3);
        const double fx = in.fx, fy = in.fy, cx = in.cx, cy = in.cy;
        const double fr = r == 0 ? fx : (r == 1 ? fy : 1.0), cr = r == 0 ? cx : (r == 1 ? cy : 0.0);
        const double k0 = fr * i0 + cr * r20, k1 = fr * i1 + cr * r21, k2 = fr * i2 + cr * r22;
        const double ia = ifx != 0.0 ? ifx : 1.0 / fx, ib = ify != 0.0 ? ify : 1.0 / fy;
        const double qa = k0 * ia, qb = k1 * ib;
        if (lane < 3) {
            pose[12 + r * 3 + 0] = (float)qa;
            pose[12 + r * 3 + 1] = (float)qb;
            pose[12 + r * 3 + 2] = (float)(k2 - qa * cx - qb * cy);
            pose[21 + r] = (float)(fr * t3 + cr * t32);
        }
    }
    MMF_SOLVE_STAMP(15);
}

// One wave, lds.dtot / lds.sd / lds.sf complete (gn_totals_to_lds by the same wave): RGBDOdometry.cpp:431-435 with one
// element of the combined system per lane, the solve; leaves the new pose in lds.pose (the caller's next barrier hands it to
// the workgroup).  lead: this workgroup also stores the running transform for the next launch.  FINAL: the chain's last
// solve (gn_final_kernel), on one lane: everything the host reads goes to the state, then odom_end.
template <bool FINAL>
__device__ __forceinline__ void gn_solve_wave(OdomState* st, const GnIterArgs& a, GnLds& lds, bool lead, int lane, unsigned prev_count,
                                              unsigned prev_sumsq) {
    __builtin_amdgcn_wave_barrier();
    const double w = (double)lds.sf[12];
    if (lane < 42) lds.sol[lane] = combine_element_d(FINAL ? st : nullptr, lane, w, lds.dtot + 29, lds.dtot);
    __builtin_amdgcn_wave_barrier();
    if (!FINAL) MMF_STAMP(4);
    if (!FINAL) {
        double nr[4];
        gn_solve_rows(lds.sol, lds.sd, lds.sf, a.intr, a.ifx, a.ify, lane, lds.xch, lds.pose, nr);
        if (lead && lane < 4) {  // the running transform for the next launch: rows 0..2 new, the last row as it was
            double* dst = st->gn_rt[(a.it + 1) & 1] + lane * 4;
            for (int c = 0; c < 4; ++c) dst[c] = lane < 3 ? nr[c] : lds.sd[12 + c];
        }
        return;
    }
    if (lane == 0) {
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
        const int count = (int)prev_count, sigma = (int)prev_sumsq;
        const ResidualDecision dec = residual_decide(count, sigma, 0, 0.f);
        st->sigma = sigma;
        st->rgbCount = count;
        st->sigmaVal = dec.sigmaVal;
        st->st.lastRGBError = dec.tmpError;
        st->st.lastRGBCount = (float)count;
        st->st.lastICPError = sqrtf((float)lds.dtot[27]) / (float)lds.dtot[28];  // RGBDOdometry.cpp:412-413
        st->st.lastICPCount = (float)lds.dtot[28];
        st->st.iterations_run = a.it;
        for (int k = 0; k < 16; ++k) st->resultRt[k] = rt[k];
        for (int k = 0; k < 9; ++k) st->Rcurr[k] = np.Rcurr[k], st->krkinv[k] = np.krkinv[k];
        for (int k = 0; k < 3; ++k) st->tcurr[k] = np.tcurr[k], st->kt[k] = np.kt[k];
        odom_end(st);
    }
}

// model blockIdx.y of a batched launch: everything model-side moves by the distance between the slabs
__device__ __forceinline__ void gn_batch_shift(OdomState*& st, GnIterArgs& a, const BatchDelta& bd, unsigned model) {
    const long long d = bd.d[model];
    st = batch_shift(st, d);
    a.ia.vmap_g_prev.base = batch_shift(a.ia.vmap_g_prev.base, d), a.ia.nmap_g_prev.base = batch_shift(a.ia.nmap_g_prev.base, d);
    a.ia.prev_packed = batch_shift(a.ia.prev_packed, d), a.ia.err_map = batch_shift(a.ia.err_map, d);
    a.ra.last_depth = batch_shift(a.ra.last_depth, d), a.ra.next_depth = batch_shift(a.ra.next_depth, d);
    a.ra.last_image = batch_shift(a.ra.last_image, d), a.ra.err_map = batch_shift(a.ra.err_map, d);
    a.cloud4 = batch_shift(a.cloud4, d);
}

// ---- PX consecutive values of one row.  4 and 2: one aligned load (the host checks the rows' alignment); 5: a 16-byte
//      load at 4-byte alignment plus one dword (a wave still reads one contiguous 1280-byte run); else element by element ----
typedef float gn_f4 __attribute__((ext_vector_type(4)));
typedef gn_f4 gn_f4u __attribute__((aligned(4)));
template <int PX>
__device__ __forceinline__ void gn_load_f32(const float* __restrict__ p, float (&out)[PX]) {
    if constexpr (PX == 4 || PX == 2 || PX == 1) {
        load_px<PX>(p, out);
    } else {
        static_assert(PX == 5, "pixels per lane");
        const gn_f4u t = *reinterpret_cast<const gn_f4u*>(p);
        out[0] = t.x, out[1] = t.y, out[2] = t.z, out[3] = t.w;
        out[4] = p[4];
    }
}
template <int PX>
__device__ __forceinline__ void gn_store_f32(float* __restrict__ p, const float (&v)[PX]) {
    if constexpr (PX == 4 || PX == 2 || PX == 1) {
        store_px<PX>(p, v);
    } else {
        const gn_f4u t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<gn_f4u*>(p) = t;
        p[4] = v[4];
    }
}
template <int PX>
__device__ __forceinline__ void gn_load_i16(const int16_t* __restrict__ p, int (&out)[PX]) {
    if constexpr (PX == 4) {
        const short4 t = *reinterpret_cast<const short4*>(p);
        out[0] = t.x, out[1] = t.y, out[2] = t.z, out[3] = t.w;
    } else if constexpr (PX == 2) {
        const short2 t = *reinterpret_cast<const short2*>(p);
        out[0] = t.x, out[1] = t.y;
    } else {
#pragma unroll
        for (int k = 0; k < PX; ++k) out[k] = p[k];
    }
}

// ---- an OBJECT model in the one-launch chain: which sensor pixels can take part in its ICP term at all ----
// icpStep (reduce.cu:257-299) moves a sensor pixel's point into the model's camera, q = Rprev^-1 (Rcurr v + tcurr - tprev),
// projects it to a pixel of the model's maps and accepts the correspondence only if the vertex stored there is a number and
// lies within distThresh of the point (RGBDOdometry.h:35).  Hence for every accepted correspondence: q projects into the pixel
// box B of the model's valid vertices (extent.hpp: noted at level 0, shifted for the coarser levels), and q.z lies within
// distThresh of a model depth, i.e. in [zlo - d, zhi + d].  That set -- a slab of the cone over B -- is convex with eight
// corners; v = Rcurr^T (Rprev q + tprev - tcurr) maps it into the sensor camera, where a vertex lies on the viewing ray of
// its own pixel (createVMap, cudafuncs.cu:109-130): that pixel is inside the bounding rectangle of the eight projected
// corners (all in front of the camera: projection keeps convex hulls).  Every other sensor pixel adds exact zeros to the 29
// sums; the model's workgroups walk the rectangle only (gn_pixel_waves_sparse), in as many passes as it takes.
// Margins: B grows by one pixel (the projection is rounded to the nearest), d by 1 % + 1 mm, the rectangle by two pixels.
// The launch that writes the error image records distances WITHOUT the threshold (reduce.cu:275): there q is anywhere on the cone
// over B, and what bounds it is the sensor's own depth range (extent.hpp: sensor_zmin_*; a sensor vertex has zs_min <= v.z <
// cutoff): the cone's four edges, taken into the sensor camera, cut by the planes v.z = zs_min and v.z = cutoff.
// A corner that is not in front of the camera, a pose that is not a number, a depth range that is not known: the whole image.
struct GnSparseCtx {
    bool box_ok, err;    // the preparation noted vertices for this frame; this launch writes the error images
    float lo[3], hi[3];  // pixel x, pixel y (level 0), camera-frame z of the model's valid vertices
    float Rprev[9], tprev[3];
    int px, level;       // pixels per lane of this launch, its pyramid level
    bool zs_ok;          // the sensor frame's depth range is known: [zs_min, zs_max)
    float zs_min, zs_max;
};
// one wave; lds.pose complete (written by lanes of this very wave)
__device__ __forceinline__ void gn_sparse_icp_box(OdomState* st, const GnIterArgs& a, GnLds& lds, const GnSparseCtx& sp, int lane, int groups,
                                                  bool lead) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const int cols = a.ia.cols, rows = a.ia.rows;
    int x0 = 0, y0 = 0, x1 = cols - 1, y1 = rows - 1;
    bool derived = false;
    if (!sp.box_ok) {
        x1 = -1;  // the model predicted nothing: no pass
    } else {
        const float d = a.ia.dist_thres * 1.01f + 1e-3f;
        const int c = lane & 7;
        // corner c of B at this level (grown by a pixel): the direction of a cone edge in the model's camera
        const float bu = (c & 1) ? (float)((int)sp.hi[0] >> sp.level) + 1.f : (float)((int)sp.lo[0] >> sp.level) - 1.f;
        const float bv = (c & 2) ? (float)((int)sp.hi[1] >> sp.level) + 1.f : (float)((int)sp.lo[1] >> sp.level) - 1.f;
        const float ex = (bu - a.ia.intr.cx) / a.ia.intr.fx, ey = (bv - a.ia.intr.cy) / a.ia.intr.fy;
        // v(s) = Rcurr^T (Rprev (s e) + tprev - tcurr) = o + s w: the edge in the sensor camera
        const float hx = sp.Rprev[0] * ex + sp.Rprev[1] * ey + sp.Rprev[2], hy = sp.Rprev[3] * ex + sp.Rprev[4] * ey + sp.Rprev[5],
                    hz = sp.Rprev[6] * ex + sp.Rprev[7] * ey + sp.Rprev[8];
        const float wx = lds.pose[0] * hx + lds.pose[3] * hy + lds.pose[6] * hz, wy = lds.pose[1] * hx + lds.pose[4] * hy + lds.pose[7] * hz,
                    wz = lds.pose[2] * hx + lds.pose[5] * hy + lds.pose[8] * hz;
        const float tx = sp.tprev[0] - lds.pose[9], ty = sp.tprev[1] - lds.pose[10], tz = sp.tprev[2] - lds.pose[11];
        const float ox = lds.pose[0] * tx + lds.pose[3] * ty + lds.pose[6] * tz, oy = lds.pose[1] * tx + lds.pose[4] * ty + lds.pose[7] * tz,
                    oz = lds.pose[2] * tx + lds.pose[5] * ty + lds.pose[8] * tz;
        float s;
        bool ok;
        if (!sp.err) {  // the slab of model depths: s is the depth in the model's camera
            const float za = sp.lo[2] - d, zb = sp.hi[2] + d;
            s = (c & 4) ? zb : za;
            ok = za > 0.05f;
        } else {  // the sensor's depth range: s where the edge crosses the plane v.z = zs_min (a little in front of it) or cutoff
            const float zp = (c & 4) ? sp.zs_max : sp.zs_min * 0.999f - 1e-3f;
            s = (zp - oz) / wz;
            ok = sp.zs_ok && sp.zs_min > 0.06f && wz > 1e-3f && s > 0.f;
        }
        const float vx = ox + s * wx, vy = oy + s * wy, vz = oz + s * wz;
        ok = ok && vz > 0.05f && vz < 1e6f && fabsf(vx) < 1e6f && fabsf(vy) < 1e6f;  // (a NaN fails)
        const float u = ok ? vx * a.ia.intr.fx / vz + a.ia.intr.cx : 0.f, v = ok ? vy * a.ia.intr.fy / vz + a.ia.intr.cy : 0.f;
        float umin = fminf(fmaxf(u, -1e6f), 1e6f), vmin = fminf(fmaxf(v, -1e6f), 1e6f), umax = umin, vmax = vmin;
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            umin = fminf(umin, __shfl_xor(umin, m)), umax = fmaxf(umax, __shfl_xor(umax, m));
            vmin = fminf(vmin, __shfl_xor(vmin, m)), vmax = fmaxf(vmax, __shfl_xor(vmax, m));
        }
        if ((__builtin_amdgcn_ballot_w64(ok) & 0xFFull) == 0xFFull) {
            x0 = max(0, (int)floorf(umin) - 2), x1 = min(cols - 1, (int)ceilf(umax) + 2);
            y0 = max(0, (int)floorf(vmin) - 2), y1 = min(rows - 1, (int)ceilf(vmax) + 2);
            derived = true;
        }
    }
    const bool none = x1 < x0 || y1 < y0;
    const int x0a = x0 - x0 % sp.px;  // a lane's run of px pixels starts on a multiple of px (cols is one: the run stays in its row)
    int lpr = none ? 0 : (x1 - x0a + sp.px) / sp.px, nr = none ? 0 : y1 - y0 + 1;
    const int per_pass = groups * a.lanes;
    if (lane == 0) {
        lds.sbox[6] = x0a, lds.sbox[7] = y0, lds.sbox[8] = lpr, lds.sbox[9] = nr;  // (read in checking mode only)
        const bool whole = a.check_sparse != 0;
        const int rect_passes = (lpr * nr + per_pass - 1) / per_pass;
        if (whole) lpr = cols / sp.px, nr = rows;
        lds.sbox[0] = whole ? 0 : x0a, lds.sbox[1] = whole ? 0 : y0, lds.sbox[2] = lpr, lds.sbox[3] = nr;
        lds.sbox[4] = (lpr * nr + per_pass - 1) / per_pass;
        lds.sbox[5] = groups;
        if (lead && sp.level == 0) {  // what a test or a tool looks at (mmf_debug_odom_sparse_outside)
            st->gn_dbg_rect[0] = x0a, st->gn_dbg_rect[1] = y0, st->gn_dbg_rect[2] = lpr * sp.px, st->gn_dbg_rect[3] = nr;
            st->gn_dbg_rect[4] = max(st->gn_dbg_rect[4], rect_passes), st->gn_dbg_rect[5] = derived ? 1 : 0;
        }
    }
}

// ---- the solver wave (wave 0 of every workgroup).  bid / groups: this workgroup's index among its model's and their number
//      (the launch's blockIdx.x / gridDim.x unless several models share a one-dimensional grid: gn_iter_mixed_kernel);
//      sp: non-null for an object model walked by its extents ----
__device__ __forceinline__ void gn_solver_wave(OdomState* st, const GnIterArgs& a, GnLds& lds, unsigned bid, unsigned groups, int lane, int npw,
                                               const GnSparseCtx* sp = nullptr) {
    const bool lead = bid == 0;
    const int fault = a.it > 0 ? __hip_atomic_load(&st->gn_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    GnSumLoads sl;
    if (a.it > 0) gn_sums_issue(st, a, sl, lane);
    if (lead) {  // what the launch after this one adds to starts from zero (nobody else touches those buffers during this launch)
        if (lane < kResShards) st->gn_acc[(a.it + 1) % 3][kResStride * lane] = 0ull;
#pragma unroll
        for (int x = 0; x < kGnSumShards; ++x) st->gn_sum[(a.it + 1) % 3][x][lane] = 0ll;
    }
    MMF_STAMP(0);
    // an earlier launch of this chain gave up (gn_fault): the chain's result is void, the host re-runs it on the two-launch
    // chain; every wave of every workgroup reads the same word (a kernel boundary lies behind its writer) and leaves here
    if (fault) return;
    if (a.it > 0) {
        unsigned pc, ps;
        gn_totals_to_lds(sl, lds, lane, pc, ps);
        MMF_STAMP(3);
        gn_solve_wave<false>(st, a, lds, lead, lane, pc, ps);
    } else {
        gn_first_pose(st, lds, lead, lane);
    }
    if (sp != nullptr) gn_sparse_icp_box(st, a, lds, *sp, lane, (int)groups, lead);
    lds_barrier();  // B: the pose is in LDS
    MMF_STAMP(8);
    lds_barrier();  // C: every pixel wave's {count, sum diff^2} is in LDS

    // one arrival per workgroup
    {
        unsigned c = (lane >= 1 && lane <= npw) ? (unsigned)lds.wsum[lane][0] : 0u;
        unsigned s2 = (lane >= 1 && lane <= npw) ? (unsigned)lds.wsum[lane][1] : 0u;
        c = wave_sum_to_lane63(c), s2 = wave_sum_to_lane63(s2);
        if (lane == 63) {
            const unsigned long long word = (1ull << kGnArriveShift) | ((unsigned long long)c << kResCountShift) | (unsigned long long)s2;
            (void)__hip_atomic_fetch_add(&st->gn_acc[a.it % 3][kResStride * (bid % kResShards)], word, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    MMF_STAMP(10);
    // The pixel waves compute and reduce their ICP rows now; polling starts when they are done (barrier C2).  By then the other
    // workgroups have arrived and the first poll finds the totals complete: 240 workgroups polling 16 lines from the moment
    // they arrive queue in front of the very atomics they wait for (measured: level 0 +2.3 us).
    lds_barrier();  // C2
    // the count barrier: poll the 16 shards until every workgroup of this model has arrived, then decide sigma
    // (RGBDOdometry.cpp:373-385) and lay out the pass's 256 weights
    float sigma_val;
    unsigned ok = 0;
    {
        const unsigned long long* acc = st->gn_acc[a.it % 3];
        unsigned c = 0, s2 = 0;
        for (int poll = 0; poll < a.max_polls; ++poll) {
            const unsigned long long v = lane < kResShards ? __hip_atomic_load(acc + kResStride * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            unsigned arr = wave_sum_to_lane63((unsigned)(v >> kGnArriveShift));
            c = wave_sum_to_lane63((unsigned)(v >> kResCountShift) & 0x3FFFFu);
            s2 = wave_sum_to_lane63((unsigned)v);  // wraps at 2^32 like the reference's int
            arr = (unsigned)__builtin_amdgcn_readlane((int)arr, 63);
            if (arr == groups) {
                ok = 1;
                break;
            }
            for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
        }
        c = (unsigned)__builtin_amdgcn_readlane((int)c, 63), s2 = (unsigned)__builtin_amdgcn_readlane((int)s2, 63);
        sigma_val = gn_sigma_val(c, s2);
#pragma unroll
        for (int q = 0; q < 4; ++q) lds.wtab[lane * 4 + q] = rgb_weight(sigma_val, (float)(lane * 4 + q));
        if (lane == 63) lds.bar[3] = __builtin_bit_cast(unsigned, sigma_val);
    }
    lds_barrier();  // D: sigma and the weights; the pixel waves' ICP sums
    MMF_STAMP(12);
    lds_barrier();  // E: the pixel waves' photometric sums
    {  // the workgroup's 58 sums -> the launch's (lane k: ICP sum k, lane 29 + k: photometric sum k)
        const int col = lane < 29 ? lane : (lane < 58 ? 32 + (lane - 29) : 63);
        float v = lds.wave[1][col];
        for (int wv = 2; wv <= npw; ++wv) v = v + lds.wave[wv][col];  // fixed order: the launch geometry decides the sum, not the timing
        const bool fits = gn_sum_add(st, a.it, lane, v, gn_rgb_exp(sigma_val), bid);
        // 1: a workgroup of this launch never arrived (ANY workgroup that gives up says so: the one that arrives last sees a
        // full count although the early ones built their rows from a partial sigma); 2: a sum left the fixed-point range.
        // Either way the host re-runs the frame's tracking on the two-launch chain (odom_finish_tracking).
        const bool all_fit = __builtin_amdgcn_ballot_w64(!fits) == 0ull;
        if ((!ok || !all_fit) && lane == 0) __hip_atomic_store(&st->gn_fault, ok ? 2 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    MMF_STAMP(13);
}

// ---- the pixel waves (waves 1 .. npw).  PX pixels per lane; needs cols % PX == 0 and cols % 4 == 0, 16-byte aligned rows
//      (PX = 4, 2), the packed model maps.  ERR: the launch also writes the two error images (the last level-0 iteration,
//      RGBDOdometry.cpp:367,408). ----
template <int PX, bool ERR>
__device__ __forceinline__ void gn_pixel_waves(OdomState* st, const GnIterArgs& a, GnLds& lds, int ptid, int lane, int wave, unsigned bid) {
    const OdomState* __restrict__ stc = st;  // what this launch only reads: scalar loads
    const int fault = a.it > 0 ? __hip_atomic_load(&st->gn_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;  // (see gn_solver_wave)
    using T = typename std::conditional<PX % 2 == 0, v2f, float>::type;
    using L = lanevec<T>;
    constexpr int W = L::W, NV = PX / W;
    const int cols = a.ra.cols, rows = a.ra.rows, N = cols * rows;
    int k0 = ((int)bid * a.lanes + ptid) * PX;
    const bool live = ptid < a.lanes && k0 < N;  // the other lanes stay active for the reductions: pixel 0, masked
    k0 = live ? k0 : 0;
    const int i = (int)__umulhi((unsigned)k0, a.ra.cols_magic), j0 = k0 - i * cols;
    // the 4x4 windows of the lane's pixels cover columns [j0 - 2, j0 + PX + 1]: three aligned words from column jb
    const int jb = (j0 - 2) & ~3, sb = j0 - 2 - jb;  // sb = 0 .. 3: byte of column j0 - 2 in the first word

    // ---- loads that do not depend on the pose: the photometric pass's images, the current vertex / normal maps ----
    unsigned ww[4][3];
    bool win[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) win[k] = jb + 4 * k >= 0 && jb + 4 * k < cols;
#pragma unroll
    for (int dr = -2; dr <= 1; ++dr) {
        const int u = min(max(i + dr, 0), rows - 1);
        const uint8_t* rowp = a.ra.next_image + (size_t)u * a.ra.ni_stride;
#pragma unroll
        for (int k = 0; k < 3; ++k) ww[dr + 2][k] = *reinterpret_cast<const unsigned*>(rowp + (win[k] ? jb + 4 * k : (j0 & ~3)));
    }
    int valxs[PX], valys[PX];
    float d1s[PX];
    gn_load_i16<PX>(a.ra.dIdx + (size_t)i * a.ra.d_stride + j0, valxs);
    gn_load_i16<PX>(a.ra.dIdy + (size_t)i * a.ra.d_stride + j0, valys);
    gn_load_f32<PX>(a.ra.next_depth + (size_t)i * a.ra.nd_stride + j0, d1s);
    float cur[6][PX];
    {
        const float* pv = a.ia.vmap_curr.base + (size_t)i * a.ia.vmap_curr.stride + j0;
        const float* pn = a.ia.nmap_curr.base + (size_t)i * a.ia.nmap_curr.stride + j0;
        const size_t sv = (size_t)rows * a.ia.vmap_curr.stride, sn = (size_t)rows * a.ia.nmap_curr.stride;
        gn_load_f32<PX>(pv, cur[0]);
        gn_load_f32<PX>(pv + sv, cur[1]);
        gn_load_f32<PX>(pv + 2 * sv, cur[2]);
        gn_load_f32<PX>(pn, cur[3]);
        gn_load_f32<PX>(pn + sn, cur[4]);
        gn_load_f32<PX>(pn + 2 * sn, cur[5]);
    }
    // the model pose of the frame (constant over the chain)
    IcpPose P;
#pragma unroll
    for (int k = 0; k < 9; ++k) P.Rprev_inv[k] = stc->Rprev_inv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) P.tprev[k] = stc->tprev[k];
    __builtin_amdgcn_sched_barrier(0);

    // ---- beside the solve: what the correspondence search can decide without the pose (reduce.cu:773-797): the 4x4
    //      "all neighbours > 0" windows from the twelve words, the gradient test; the lane's own intensities ----
    bool cand[PX];
    int own[PX];
    {
        unsigned nz[3] = {0x80808080u, 0x80808080u, 0x80808080u};
#pragma unroll
        for (int dr = -2; dr <= 1; ++dr) {
            const bool rowin = (i + dr) >= 0 && (i + dr) < rows;
#pragma unroll
            for (int k = 0; k < 3; ++k) nz[k] &= rowin ? nonzero_bytes(ww[dr + 2][k]) : 0x80808080u;
        }
        // bit b: column jb + b is non-zero in all four rows (columns outside the image count as set: reduce.cu:777-787 clips its loops)
        const unsigned okw = (win[0] ? byte_flags_to_bits(nz[0]) : 0xFu) | ((win[1] ? byte_flags_to_bits(nz[1]) : 0xFu) << 4) |
                             ((win[2] ? byte_flags_to_bits(nz[2]) : 0xFu) << 8);
        const unsigned long long lo = (unsigned long long)ww[2][0] | ((unsigned long long)ww[2][1] << 32);
        const unsigned long long hi = (unsigned long long)ww[2][1] | ((unsigned long long)ww[2][2] << 32);
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int x = j0 + p;
            const bool valid = live && x < cols - 5 && i < rows - 1 && ((okw >> (sb + p)) & 0xFu) == 0xFu;
            const float mTwo = (float)((valxs[p] * valxs[p]) + (valys[p] * valys[p]));
            cand[p] = valid && mTwo >= a.ra.min_scale && !(d1s[p] != d1s[p]);
            const int b = sb + 2 + p;  // byte of column x in the row's three words (2 .. 9)
            own[p] = (int)(((b < 8 ? lo : hi) >> (8 * (b < 8 ? b : b - 4))) & 0xFFull);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (fault) return;
    lds_barrier();  // B: the pose is in LDS
    float K[9], kt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) P.Rcurr[k] = uniform_f(lds.pose[k]), K[k] = uniform_f(lds.pose[12 + k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) P.tcurr[k] = uniform_f(lds.pose[9 + k]), kt[k] = uniform_f(lds.pose[21 + k]);

    if (MMF_ABL & 64) {  // the dependent phases alone: no pixel work between the pose and the record
        if (lane == 63) lds.wsum[wave][0] = (int)(P.Rcurr[0] == 123.f), lds.wsum[wave][1] = (int)(K[0] == 123.f) + (int)cand[0] + own[0];
        lds_barrier();  // C
        if ((lane & 1) == 0) lds.wave[wave][lane >> 1] = 0.f, lds.wave[wave][32 + (lane >> 1)] = 0.f;
        lds_barrier();  // C2
        lds_barrier();  // D
        lds_barrier();  // E
        return;
    }

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
    uint8_t lis[PX];
    f3pk gv[PX], gn[PX];
    float4 cl[PX];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PX; ++p) {  // addresses clamped, masks applied afterwards
        const int gu = inb[p] ? u0s[p] : 0, gvv = inb[p] ? v0s[p] : 0;
        lis[p] = a.ra.last_image[(size_t)gvv * a.ra.li_stride + gu];
        cl[p] = a.cloud4[(size_t)gvv * cols + gu];  // rgbStep's point (reduce.cu:522) and 1 / Z
    }
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP_PX(9);

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

    // ---- photometric: accept / reject (reduce.cu:813-836), the wave's {count, sum diff^2} ----
    RgbLane<PX> ph;
    int cnt = 0, sq = 0;
    float perr[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        // lastDepth at the target (reduce.cu:813): the point's Z is that very float (projectPointsKernel, cudafuncs.cu:745:
        // cloud.z = z), so the record's gather brings it along -- one gather per pixel less, in the phase the CU's address unit bounds
        const float d0 = cl[p].z;
        const bool hit = inb[p] && d0 > 0 && fabsf(td1s[p] - d0) <= a.ra.max_depth_delta && lis[p] != 0;
        const int idiff = own[p] - (int)lis[p];  // == (float)next - (float)last, exactly
        const int vy = hit ? idiff * idiff : 0;   // == (int)(diff * diff)
        perr[p] = hit ? 0.001f * vy : 0.0f;
        cnt += hit ? 1 : 0;
        sq += vy;
        ph.c[p].diff = hit ? (float)idiff : 0.f;
        ph.c[p].valid = hit ? 1 : 0;
        ph.X[p] = cl[p].x, ph.Y[p] = cl[p].y, ph.Z[p] = cl[p].z, ph.invz[p] = cl[p].w;
        ph.gx[p] = valxs[p], ph.gy[p] = valys[p];
    }
    if (ERR && a.ra.err_map && live) gn_store_f32<PX>(a.ra.err_map + (size_t)i * a.ra.err_stride + j0, perr);
    MMF_STAMP_PX(6);
    cnt = wave_sum_to_lane63(cnt);
    sq = wave_sum_to_lane63(sq);
    if (lane == 63) lds.wsum[wave][0] = cnt, lds.wsum[wave][1] = sq;
    lds_barrier();  // C: the solver wave adds the workgroup's counts to the launch's and waits for the others
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP_PX(7);

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
    if (ERR && a.ia.err_map && live) gn_store_f32<PX>(a.ia.err_map + (size_t)i * a.ia.err_stride + j0, ierr);
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP_PX(11);
    {  // their reduction over the wave goes here, into the wait for the other workgroups
        float s32[32];
#pragma unroll
        for (int k = 0; k < 29; ++k) s32[k] = L::hsum(isum[k]);
        s32[29] = s32[30] = s32[31] = 0.f;
        const float t = wave_sum_transposed(s32);
        if ((lane & 1) == 0) lds.wave[wave][lane >> 1] = t;
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();  // C2: the solver wave starts polling
    lds_barrier();  // D: sigma and the weight table are in LDS
    const float sigmaVal = __builtin_bit_cast(float, lds.bar[3]);

    // ---- photometric: rgbStep's rows (reduce.cu:504-535) ----
    float psum[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) psum[k] = 0.f;
    rgb_rows<PX>(a.sobel_scale, a.fx, a.fy, sigmaVal, live, ph, psum, lds.wtab, true);
    {
        float s32[32];
#pragma unroll
        for (int k = 0; k < 29; ++k) s32[k] = psum[k];
        s32[29] = s32[30] = s32[31] = 0.f;
        const float t2 = wave_sum_transposed(s32);
        if ((lane & 1) == 0) lds.wave[wave][32 + (lane >> 1)] = t2;
    }
    lds_barrier();  // E: the solver wave adds the waves' sums up and stores the workgroup's record
}

// ---- the pixel waves of an OBJECT model (gn_iter_mixed_kernel).  The same per-pixel code as gn_pixel_waves on two pixel sets of
//      the model's own: the photometric term over the box of its own depth (x0a, y0, lpr lanes per row, nr rows: one pass,
//      the kernel has checked that it fits), the ICP term over the rectangle lds.sbox describes (gn_sparse_icp_box: known
//      with the pose), in sbox[4] passes.  A pixel outside the first box has no depth of its own (reduce.cu:600), one
//      outside the second cannot pass icpStep's distance test: both add exact zeros in the dense walk. ----
template <int PX, bool ERR>
__device__ __forceinline__ void gn_pixel_waves_sparse(OdomState* st, const GnIterArgs& a, GnLds& lds, int ptid, int lane, int wave, unsigned bid,
                                                      int x0a, int y0, int lpr, int nr) {
    const OdomState* __restrict__ stc = st;
    const int fault = a.it > 0 ? __hip_atomic_load(&st->gn_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    using T = typename std::conditional<PX % 2 == 0, v2f, float>::type;
    using L = lanevec<T>;
    constexpr int W = L::W, NV = PX / W;
    const int cols = a.ra.cols, rows = a.ra.rows;
    const unsigned q = bid * (unsigned)a.lanes + (unsigned)ptid;
    const unsigned prow = lpr ? q / (unsigned)lpr : 0u;
    const bool live = ptid < a.lanes && lpr != 0 && prow < (unsigned)nr;  // the other lanes stay active for the reductions: pixel 0, masked
    const int i = live ? y0 + (int)prow : 0, j0 = live ? x0a + (int)(q - prow * (unsigned)lpr) * PX : 0;
    const int jb = (j0 - 2) & ~3, sb = j0 - 2 - jb;

    // ---- loads that do not depend on the pose: the photometric pass's images ----
    unsigned ww[4][3];
    bool win[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) win[k] = jb + 4 * k >= 0 && jb + 4 * k < cols;
#pragma unroll
    for (int dr = -2; dr <= 1; ++dr) {
        const int u = min(max(i + dr, 0), rows - 1);
        const uint8_t* rowp = a.ra.next_image + (size_t)u * a.ra.ni_stride;
#pragma unroll
        for (int k = 0; k < 3; ++k) ww[dr + 2][k] = *reinterpret_cast<const unsigned*>(rowp + (win[k] ? jb + 4 * k : (j0 & ~3)));
    }
    int valxs[PX], valys[PX];
    float d1s[PX];
    gn_load_i16<PX>(a.ra.dIdx + (size_t)i * a.ra.d_stride + j0, valxs);
    gn_load_i16<PX>(a.ra.dIdy + (size_t)i * a.ra.d_stride + j0, valys);
    gn_load_f32<PX>(a.ra.next_depth + (size_t)i * a.ra.nd_stride + j0, d1s);
    IcpPose P;
#pragma unroll
    for (int k = 0; k < 9; ++k) P.Rprev_inv[k] = stc->Rprev_inv[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) P.tprev[k] = stc->tprev[k];
    __builtin_amdgcn_sched_barrier(0);
    bool cand[PX];
    int own[PX];
    {
        unsigned nz[3] = {0x80808080u, 0x80808080u, 0x80808080u};
#pragma unroll
        for (int dr = -2; dr <= 1; ++dr) {
            const bool rowin = (i + dr) >= 0 && (i + dr) < rows;
#pragma unroll
            for (int k = 0; k < 3; ++k) nz[k] &= rowin ? nonzero_bytes(ww[dr + 2][k]) : 0x80808080u;
        }
        const unsigned okw = (win[0] ? byte_flags_to_bits(nz[0]) : 0xFu) | ((win[1] ? byte_flags_to_bits(nz[1]) : 0xFu) << 4) |
                             ((win[2] ? byte_flags_to_bits(nz[2]) : 0xFu) << 8);
        const unsigned long long lo = (unsigned long long)ww[2][0] | ((unsigned long long)ww[2][1] << 32);
        const unsigned long long hi = (unsigned long long)ww[2][1] | ((unsigned long long)ww[2][2] << 32);
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int x = j0 + p;
            const bool valid = live && x < cols - 5 && i < rows - 1 && ((okw >> (sb + p)) & 0xFu) == 0xFu;
            const float mTwo = (float)((valxs[p] * valxs[p]) + (valys[p] * valys[p]));
            cand[p] = valid && mTwo >= a.ra.min_scale && !(d1s[p] != d1s[p]);
            const int b = sb + 2 + p;
            own[p] = (int)(((b < 8 ? lo : hi) >> (8 * (b < 8 ? b : b - 4))) & 0xFFull);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (fault) return;
    lds_barrier();  // B: the pose and the ICP rectangle are in LDS
    float K[9], kt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) P.Rcurr[k] = uniform_f(lds.pose[k]), K[k] = uniform_f(lds.pose[12 + k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) P.tcurr[k] = uniform_f(lds.pose[9 + k]), kt[k] = uniform_f(lds.pose[21 + k]);
    const int sx0 = __builtin_amdgcn_readfirstlane(lds.sbox[0]), sy0 = __builtin_amdgcn_readfirstlane(lds.sbox[1]);
    const unsigned slpr = (unsigned)__builtin_amdgcn_readfirstlane(lds.sbox[2]), snr = (unsigned)__builtin_amdgcn_readfirstlane(lds.sbox[3]);
    const int npass = __builtin_amdgcn_readfirstlane(lds.sbox[4]);
    const unsigned sgroups = (unsigned)__builtin_amdgcn_readfirstlane(lds.sbox[5]);

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
        if (cand[p]) {
            td1s[p] = (float)(d1 * (K[6] * x + K[7] * y + K[8]) + kt[2]);
            u0s[p] = float2int_rn((d1 * (K[0] * x + K[1] * y + K[2]) + kt[0]) / td1s[p]);
            v0s[p] = float2int_rn((d1 * (K[3] * x + K[4] * y + K[5]) + kt[1]) / td1s[p]);
            inb[p] = u0s[p] >= 0 && v0s[p] >= 0 && u0s[p] < cols && v0s[p] < rows;
        }
    }
    struct f3pk {
        float x, y, z;
    };
    uint8_t lis[PX];
    float4 cl[PX];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int gu = inb[p] ? u0s[p] : 0, gvv = inb[p] ? v0s[p] : 0;
        lis[p] = a.ra.last_image[(size_t)gvv * a.ra.li_stride + gu];
        cl[p] = a.cloud4[(size_t)gvv * cols + gu];
    }
    MMF_STAMP_PX(9);
    // ---- ICP: the sensor's vertices and normals of this lane's run in a pass ----
    float cur[6][PX];
    int si = 0, sj = 0;
    bool slive = false;
    auto icp_pass_load = [&](int pass) {
        const unsigned sq = ((unsigned)pass * sgroups + bid) * (unsigned)a.lanes + (unsigned)ptid;
        const unsigned srow = slpr ? sq / slpr : 0u;
        slive = ptid < a.lanes && slpr != 0 && srow < snr;
        si = slive ? sy0 + (int)srow : 0, sj = slive ? sx0 + (int)(sq - srow * slpr) * PX : 0;
        const float* pv = a.ia.vmap_curr.base + (size_t)si * a.ia.vmap_curr.stride + sj;
        const float* pn = a.ia.nmap_curr.base + (size_t)si * a.ia.nmap_curr.stride + sj;
        const size_t sv = (size_t)rows * a.ia.vmap_curr.stride, sn = (size_t)rows * a.ia.nmap_curr.stride;
        gn_load_f32<PX>(pv, cur[0]);
        gn_load_f32<PX>(pv + sv, cur[1]);
        gn_load_f32<PX>(pv + 2 * sv, cur[2]);
        gn_load_f32<PX>(pn, cur[3]);
        gn_load_f32<PX>(pn + sn, cur[4]);
        gn_load_f32<PX>(pn + 2 * sn, cur[5]);
    };
    __builtin_amdgcn_sched_barrier(0);

    // ---- photometric: accept / reject (reduce.cu:813-836), the wave's {count, sum diff^2} ----
    // The correspondences wait in LDS while the ICP passes run (five words per pixel: with them in registers across the pass
    // loop the kernel needs 246 registers -- one workgroup per CU instead of two, and the launch's workgroups must all be resident)
    __shared__ unsigned stash[kGnStashWords * PX][kGnSparseLanes];
    int cnt = 0, sq2 = 0;
    float perr[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const float d0 = cl[p].z;
        const bool hit = inb[p] && d0 > 0 && fabsf(td1s[p] - d0) <= a.ra.max_depth_delta && lis[p] != 0;
        const int idiff = own[p] - (int)lis[p];
        const int vy = hit ? idiff * idiff : 0;
        perr[p] = hit ? 0.001f * vy : 0.0f;
        cnt += hit ? 1 : 0;
        sq2 += vy;
        // (1 / Z is not kept: cloud4.w IS 1.0f / z, projectPointsKernel's record -- rgb_rows divides again)
        stash[5 * p + 0][ptid] = __builtin_bit_cast(unsigned, cl[p].x), stash[5 * p + 1][ptid] = __builtin_bit_cast(unsigned, cl[p].y);
        stash[5 * p + 2][ptid] = __builtin_bit_cast(unsigned, cl[p].z);
        stash[5 * p + 3][ptid] = ((unsigned)valxs[p] & 0xFFFFu) | ((unsigned)valys[p] << 16);  // (Sobel sums of bytes: 16 bits each)
        stash[5 * p + 4][ptid] = hit ? (unsigned)(idiff + 256) : 0u;                          // 0: no correspondence
    }
    if (ERR && a.ra.err_map && live) gn_store_f32<PX>(a.ra.err_map + (size_t)i * a.ra.err_stride + j0, perr);
    cnt = wave_sum_to_lane63(cnt);
    sq2 = wave_sum_to_lane63(sq2);
    MMF_STAMP_PX(6);
    if (lane == 63) lds.wsum[wave][0] = cnt, lds.wsum[wave][1] = sq2;
    lds_barrier();  // C: the solver wave adds the workgroup's counts to the launch's and waits for the others
    __builtin_amdgcn_sched_barrier(0);
    MMF_STAMP_PX(7);

    if (ERR) {  // the error images outside what the passes and the photometric lanes write: zeros, as the dense walk leaves there
        const unsigned rpr = (unsigned)(cols / PX), total = rpr * (unsigned)rows, per = sgroups * (unsigned)a.lanes;
        float zeros[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) zeros[p] = 0.f;
        for (unsigned r = bid * (unsigned)a.lanes + (unsigned)ptid; r < total && ptid < a.lanes; r += per) {
            const unsigned y = r / rpr;
            const int x = (int)(r - y * rpr) * PX, yi = (int)y;
            const bool in_rect = slpr != 0 && yi >= sy0 && yi < sy0 + (int)snr && x >= sx0 && x < sx0 + (int)slpr * PX;
            const bool in_box = lpr != 0 && yi >= y0 && yi < y0 + nr && x >= x0a && x < x0a + lpr * PX;
            if (a.ia.err_map && !in_rect) gn_store_f32<PX>(a.ia.err_map + (size_t)yi * a.ia.err_stride + x, zeros);
            if (a.ra.err_map && !in_box) gn_store_f32<PX>(a.ra.err_map + (size_t)yi * a.ra.err_stride + x, zeros);
        }
    }
    // ---- ICP: projection, gathers, Jacobian rows (reduce.cu:257-368), pass by pass ----
    T isum[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) isum[k] = L::splat(0.f);
    for (int pass = 0; pass < npass; ++pass) {  // (uniform)
        icp_pass_load(pass);
        __builtin_amdgcn_sched_barrier(0);
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
        f3pk gv[PX], gn[PX];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 0; qq < PX; ++qq) {
            const f3pk* src = reinterpret_cast<const f3pk*>(a.ia.prev_packed) + 2 * ((size_t)pr[qq / W].uy[qq % W] * cols + pr[qq / W].ux[qq % W]);
            gv[qq] = src[0];
            gn[qq] = src[1];
        }
        __builtin_amdgcn_sched_barrier(0);
        float ierr[PX];
        const float inl0 = L::hsum(isum[28]);
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
            icp_rows_v<ERR, false, T>(P, a.ia, pr[h], slive, n, vp, np, isum, er);
            if (ERR) {
#pragma unroll
                for (int e = 0; e < W; ++e) ierr[h * W + e] = er[e];
            }
        }
        if (a.check_sparse && !ERR) {  // (uniform) what this run added although the rectangle does not hold it
            const int rx = lds.sbox[6], ry = lds.sbox[7], rl = lds.sbox[8], rn = lds.sbox[9];
            const bool in_rect = rl != 0 && si >= ry && si < ry + rn && sj >= rx && sj < rx + rl * PX;
            const float added = L::hsum(isum[28]) - inl0;
            if (slive && !in_rect && added > 0.f) atomicAdd(&st->gn_dbg_outside, (unsigned)added);
        }
        if (ERR && slive && a.ia.err_map) gn_store_f32<PX>(a.ia.err_map + (size_t)si * a.ia.err_stride + sj, ierr);
    }
    MMF_STAMP_PX(11);
    {
        float s32[32];
#pragma unroll
        for (int k = 0; k < 29; ++k) s32[k] = L::hsum(isum[k]);
        s32[29] = s32[30] = s32[31] = 0.f;
        const float t = wave_sum_transposed(s32);
        if ((lane & 1) == 0) lds.wave[wave][lane >> 1] = t;
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();  // C2: the solver wave starts polling
    lds_barrier();  // D: sigma and the weight table are in LDS
    const float sigmaVal = __builtin_bit_cast(float, lds.bar[3]);

    // ---- photometric: rgbStep's rows (reduce.cu:504-535) ----
    RgbLane<PX> ph;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        ph.X[p] = __builtin_bit_cast(float, stash[5 * p + 0][ptid]), ph.Y[p] = __builtin_bit_cast(float, stash[5 * p + 1][ptid]);
        ph.Z[p] = __builtin_bit_cast(float, stash[5 * p + 2][ptid]), ph.invz[p] = 0.f;
        const unsigned gxy = stash[5 * p + 3][ptid], dv = stash[5 * p + 4][ptid];
        ph.gx[p] = (int)(short)(gxy & 0xFFFFu), ph.gy[p] = (int)(short)(gxy >> 16);
        ph.c[p].diff = dv ? (float)((int)dv - 256) : 0.f;
        ph.c[p].valid = dv ? 1 : 0;
    }
    float psum[29];
#pragma unroll
    for (int k = 0; k < 29; ++k) psum[k] = 0.f;
    rgb_rows<PX>(a.sobel_scale, a.fx, a.fy, sigmaVal, live, ph, psum, lds.wtab, false);
    {
        float s32[32];
#pragma unroll
        for (int k = 0; k < 29; ++k) s32[k] = psum[k];
        s32[29] = s32[30] = s32[31] = 0.f;
        const float t2 = wave_sum_transposed(s32);
        if ((lane & 1) == 0) lds.wave[wave][32 + (lane >> 1)] = t2;
    }
    lds_barrier();  // E: the solver wave adds the waves' sums up
}

// blockDim.x = 64 (solver wave) + the pixel waves (>= 4, a.lanes <= their lanes)
template <int PX, bool ERR>
__global__ __launch_bounds__(64 * kGnMaxWaves) void gn_iter_kernel(OdomState* st, GnIterArgs a, BatchDelta bd) {
    __shared__ GnLds lds;
    if (gridDim.y > 1) gn_batch_shift(st, a, bd, blockIdx.y);
    MMF_STAMP(5);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 0)
        gn_solver_wave(st, a, lds, blockIdx.x, gridDim.x, lane, (int)(blockDim.x >> 6) - 1);
    else
        gn_pixel_waves<PX, ERR>(st, a, lds, tid - 64, lane, wave, blockIdx.x);
}

// Several models in ONE launch with a geometry PER MODEL (MultiMotionFusion.cpp:312-387: the models' trackings are independent):
// a one-dimensional grid, workgroups [start[m], start[m + 1]) belong to model m.  The camera model (and any model without
// extents) is walked as in gn_iter_kernel; an OBJECT model -- a few thousand pixels of prediction in an otherwise empty image
// -- by a fraction of the workgroups (gn_pixel_waves_sparse): its photometric term over the box of its own depth, its ICP
// term over the rectangle its prediction can reach under this iteration's pose.  The count barrier stays per model, and
// ALL workgroups of the launch must be resident together (the host budgets start[n] against the occupancy).
struct GnBatchGeom {
    int start[kMaxBatch + 1];
    unsigned sparse_mask;  // bit m: model m is an object model with extents noted for this frame
    unsigned ext_gen;      // ... the frame number they carry
    int level;             // this launch's pyramid level (extent_of_level)
    const unsigned long long* extent;  // the first model's extent words (in its slab: the others' by BatchDelta)
    const unsigned long long* sensor;  // the extent words that hold the sensor frame's smallest depth (the leader's: shared, not shifted)
    unsigned sensor_gen;               // ... the number it was noted under (0: unknown)
    float sensor_cutoff;               // ... and the cut-off no sensor depth reaches (createVMap)
    int rotate;  // the grid's first workgroup is number `rotate` of the list start[] describes (tunables: gn_obj_first: the object
                 // models' workgroups, which have the longer way to go, are dispatched before the camera model's)
};
constexpr int kGnFaultExtent = 3;  // OdomState::gn_fault: an object model's extent does not fit its workgroups (the host walks it densely from then on)

#ifndef MMF_MIXED_ATTR
#define MMF_MIXED_ATTR
#endif
template <int PX, bool ERR>
__global__ __launch_bounds__(64 * kGnMaxWaves) MMF_MIXED_ATTR void gn_iter_mixed_kernel(OdomState* st, GnIterArgs a, BatchDelta bd, GnBatchGeom g) {
    __shared__ GnLds lds;
    unsigned model = 0;
    int vb = (int)blockIdx.x + g.rotate;
    vb = vb >= g.start[kMaxBatch] ? vb - g.start[kMaxBatch] : vb;
#pragma unroll
    for (int m = 1; m < kMaxBatch; ++m) model = (vb >= g.start[m] && g.start[m] < g.start[m + 1]) ? (unsigned)m : model;  // (uniform)
    const unsigned bid = (unsigned)(vb - g.start[model]), groups = (unsigned)(g.start[model + 1] - g.start[model]);
    gn_batch_shift(st, a, bd, model);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!((g.sparse_mask >> model) & 1u)) {
        if (wave == 0)
            gn_solver_wave(st, a, lds, bid, groups, lane, (int)(blockDim.x >> 6) - 1);
        else
            gn_pixel_waves<PX, ERR>(st, a, lds, tid - 64, lane, wave, bid);
        return;
    }
    const unsigned long long* ext = batch_shift(g.extent, bd.d[model]);
    // the box of the model's own depth at this level (pose independent: the photometric term's pixels), PX-aligned runs
    const ExtentBox pb = extent_of_level(ext, g.ext_gen, g.level);
    const bool none = pb.x1 < pb.x0 || pb.y1 < pb.y0;
    const int x0a = none ? 0 : pb.x0 - pb.x0 % PX, lpr = none ? 0 : (pb.x1 - x0a + PX) / PX, nr = none ? 0 : pb.y1 - pb.y0 + 1;
    if (bid == 0 && tid == 0) st->gn_need[g.level] = lpr * nr;
    if ((long long)lpr * nr > (long long)groups * a.lanes) {  // (uniform over the model's workgroups: all leave, none waits)
        if (bid == 0 && tid == 0 && (a.it == 0 || !__hip_atomic_load(&st->gn_fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            __hip_atomic_store(&st->gn_fault, kGnFaultExtent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (wave == 0) {
        GnSparseCtx sp;
        sp.box_ok = aabb_load(ext, g.ext_gen, sp.lo, sp.hi);
        sp.err = ERR;
        sp.px = PX, sp.level = g.level;
        sp.zs_ok = ERR && sensor_zmin_load(g.sensor, g.sensor_gen, sp.zs_min);
        sp.zs_max = g.sensor_cutoff;
#pragma unroll
        for (int k = 0; k < 9; ++k) sp.Rprev[k] = st->Rprev[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) sp.tprev[k] = st->tprev[k];
        gn_solver_wave(st, a, lds, bid, groups, lane, (int)(blockDim.x >> 6) - 1, &sp);
    } else {
        gn_pixel_waves_sparse<PX, ERR>(st, a, lds, tid - 64, lane, wave, bid, x0a, pb.y0, lpr, nr);
    }
}

// the chain's last solve + RGBDOdometry.cpp:464-467, 475-476: one workgroup (kBlock threads) per model, its first wave works
__device__ __forceinline__ void gn_final_solve(OdomState* st, const GnIterArgs& a, GnLds& lds) {
    const int tid = threadIdx.x;
    if (st->gn_fault) return;  // (uniform) the chain's result is void: the state goes to the host as it is, the fault word with it
    if (tid < 64) {
        GnSumLoads sl;
        gn_sums_issue(st, a, sl, tid);
        unsigned pc, ps;
        gn_totals_to_lds(sl, lds, tid, pc, ps);
        gn_solve_wave<true>(st, a, lds, true, tid, pc, ps);
    }
    __syncthreads();  // the state lane 0 stored is visible to the workgroup
}
__global__ __launch_bounds__(kBlock) void gn_final_kernel(OdomState* st, GnIterArgs a, BatchDelta bd) {
    __shared__ GnLds lds;
    if (gridDim.x > 1) {
        st = batch_shift(st, bd.d[blockIdx.x]);
    }
    gn_final_solve(st, a, lds);
}

// the same, and the result goes to the host in the same launch (odom_publish_kernel's two waves behind the solve: the copy
// into the host's pinned state + sequence number, Model::computeFusionWeight of the new pose for an early fuse pass)
__global__ __launch_bounds__(kBlock) void gn_final_publish_kernel(OdomState* st, GnIterArgs a, BatchDelta bd, PublishTargets to,
                                                                  unsigned seq) {
    __shared__ GnLds lds;
    if (gridDim.x > 1) {
        st = batch_shift(st, bd.d[blockIdx.x]);
    }
    gn_final_solve(st, a, lds);
    if (threadIdx.x >= 64) {
        if (threadIdx.x == 64) odom_fusion_weight(st);
        return;
    }
    odom_publish_wave(st, to.host[blockIdx.x], seq, threadIdx.x);
}

}  // namespace mmf
