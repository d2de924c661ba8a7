// odom_state.hpp -- device-resident Gauss-Newton state of one RGBDOdometry object and the
// double-precision pose algebra the reference runs on the host between kernel launches.
//
// The reference returns to the host after every reduction (2 launches + cudaDeviceSynchronize +
// a small D2H copy + an Eigen solve, >= 3 x 19 + 10 times per model per frame;
// Core/Utils/RGBDOdometry.cpp:257-310, 331-462).  Here the state lives in HBM, the workgroup
// that finishes a reduction also runs the 6x6 solve / SE3 update in double on one lane, and
// the next kernel on the stream reads the new pose from this struct -- no host round trip
// inside getIncrementalTransformation.
//
// The algebra restates Eigen operations the reference calls (ldlt().solve, inverse(),
// Isometry3f products; Eigen itself is an unpinned system dependency that is absent here):
// same formulas as oracle/mmf_oracle.c so both agree to rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

namespace mmf {

struct OdomStats {  // mirrors mmf_odom_stats (include/mmf_hip.h)
    float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
    double lastA[36];
    double lastb[6];
    int iterations_run;
    int so3_iterations_run;
};

struct OdomState {
    // pose being optimised (RGBDOdometry.cpp:224-228)
    float Rprev[9], tprev[3], Rprev_inv[9];
    float Rcurr[9], tcurr[3];
    double resultRt[16];
    // SO3 pre-alignment (RGBDOdometry.cpp:237-310)
    double resultR[9], lastResultR[9];
    float R_lr[9];
    float so3_lastError, so3_lastCount;
    int so3_done;
    float imageBasis[9], kinv[9], krlr[9];
    // photometric term
    float krkinv[9], kt[3];
    int sigma, rgbCount;
    float sigmaVal;
    int level_break;  // rgbOnly divergence `break` of the current level (RGBDOdometry.cpp:376-378)
    // configuration of this call
    int rgb_only, icp, rgb, so3;
    float icp_weight;
    // last reductions
    float A_icp[36], b_icp[6], residual_icp[2];
    float A_rgb[36], b_rgb[6];
    // raw totals for the stand-alone *Step entry points
    float out_f[32];
    int out_i[2];
    // outputs
    float trans_out[3], rot_out[9];
    float pose_inv[16];  // inverse of [rot_out | trans_out; 0 0 0 1], as the host computes it for the projection passes
    float pose_out[16];  // [rot_out | trans_out; 0 0 0 1]
    float fusion_weight; // Model::computeFusionWeight(1) of this pose against the pose the chain started from (odom_fusion_weight_kernel)
    int gn_fault;  // the count barrier of a gn_iter_kernel launch timed out (a workgroup of the launch never arrived)
    // checking mode of an object model's one-launch chain (GnIterArgs::check_sparse): correspondences icpStep accepted OUTSIDE
    // the rectangle the chain would have walked, over all launches of the chain -- zero, or the rectangle's argument is wrong
    unsigned gn_dbg_outside;
    // ... and, in every mode, the level-0 rectangle of the chain's last launch {x0, y0, width, rows}, the largest number of
    // passes a level-0 rectangle took, whether it was derived (1) or the whole image for want of a usable pose / box (0)
    int gn_dbg_rect[6];
    // lanes the box of the model's own depth needs at each pyramid level (gn_iter_mixed_kernel: runs of PX pixels x rows), noted by
    // every launch that walks the model by its extents -- also by the one that finds the box does not fit (kGnFaultExtent): the
    // host sizes the model's NEXT chain by it
    int gn_need[3];
    OdomStats st;
    // everything above travels to the host's (pinned, device-visible) copy of this struct at the end of a chain
    // (odom_publish_kernel); this word follows it there, after a system-scope fence: the host polls it
    unsigned publish_seq;
    // {count, sum diff^2} of the photometric correspondence pass of the current Gauss-Newton step,
    // accumulated with integer atomics (exact, order independent) over kResShards address pairs;
    // zeroed by gn_level_begin_kernel and by the finishing lane of every step
    // over kResShards address pairs, each pair on ITS OWN 128-byte line: device-scope atomics on one line serialise
    // (~12 ns per arrival); with all 16 counters in one line the 600 atomics of a 640x480 correspondence pass
    // cost ~7 us -- more than the pass itself (and 4x that at 1280x960)
    // Count and sum travel in ONE 64-bit atomic: count << 40 | sum (sum < 2^40 per shard: 255^2 x 16 M pixels).
    alignas(128) unsigned long long res_acc[16 * 16];
    // ---- one-launch-per-iteration chain (gn_fused.hpp) ----
    // Launch `it` accumulates {arrivals, count, sum diff^2} into gn_acc[it % 3] (the same sharding as res_acc), the
    // prologue of launch it + 1 reads it for the statistics, and launch it + 1 zeroes gn_acc[(it + 2) % 3]: a
    // buffer is never read, added to and zeroed by workgroups of the same launch.
    alignas(128) unsigned long long gn_acc[3][16 * 16];
    // the running transform: launch `it` reads gn_rt[it & 1] (it = 0: resultRt) and its workgroup 0 stores the updated
    // one to gn_rt[(it + 1) & 1], so no workgroup reads what another workgroup of the same launch writes
    double gn_rt[2][16];
    // The 58 sums of a launch (29 ICP, 29 photometric: gn_fused.hpp) as 64-bit fixed-point integers: every workgroup adds its
    // float partials, scaled by a power of two, with integer atomics -- exact and order independent, so the totals are the
    // same bits however the launch is laid out, and the next launch reads 16 x 464 bytes instead of one 256-byte record per
    // workgroup (61 KB at 640x480, fetched by EVERY workgroup: 0.8 us of a level-0 launch).  Atomics execute at the memory
    // side on this chip whatever their scope (MI355X_MICROARCH.md, "Global float atomics"), so the sharding (by block index,
    // kGnSumShards lines sets) is there against contention on a line only, never for correctness.  Rotation like gn_acc:
    // launch `it` adds to gn_sum[it % 3], reads gn_sum[(it + 2) % 3] and zeroes gn_sum[(it + 1) % 3]; a kernel boundary lies
    // between a buffer's writers and its readers.
    alignas(128) long long gn_sum[3][16][64];
};
constexpr int kGnSumIcpExp = 30;  // ICP sums: value x 2^30 (per-workgroup partials < 2^23: 1280 pixels x |v|^2)
constexpr int kGnSumRgbExp0 = 4;  // photometric sums: value x 2^(4 + 2 floor(log2 sigma)) -- the rows carry w = 1 / (sigma + |d|)
constexpr int kGnSumShards = 16;
constexpr int kResShards = 16;
constexpr int kResStride = 16;  // 64-bit words between two shards (128 B)
constexpr int kResCountShift = 40;

// ---- small dense algebra (double unless suffixed f) -----------------------------------------

__device__ inline void inverse3f(const float* m, float* inv) {  // RGBDOdometry.cpp:316
    const float c00 = m[4] * m[8] - m[5] * m[7];
    const float c01 = m[5] * m[6] - m[3] * m[8];
    const float c02 = m[3] * m[7] - m[4] * m[6];
    const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const float id = 1.0f / det;
    inv[0] = c00 * id;
    inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c01 * id;
    inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c02 * id;
    inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// Eigen `pose.inverse()` of the projection passes (ModelProjection.cpp:108): the host's inverse4f_host, term by term
__host__ __device__ inline void inverse4f(const float* m, float* inv) {
    const float s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2];
    const float s2 = m[0] * m[7] - m[4] * m[3], s3 = m[1] * m[6] - m[5] * m[2];
    const float s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11];
    const float c3 = m[9] * m[14] - m[13] * m[10], c2 = m[8] * m[15] - m[12] * m[11];
    const float c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const float id = 1.0f / det;
    inv[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    inv[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    inv[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    inv[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    inv[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    inv[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    inv[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    inv[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    inv[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    inv[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    inv[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    inv[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    inv[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    inv[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    inv[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    inv[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
}

// 1 / d for the pivots and determinants of the one-lane tail: the hardware's reciprocal estimate and two Newton steps
// (full double precision for normal numbers; 0 and non-finite d end in NaN or infinity like a division) instead of the
// twelve-instruction IEEE division sequence -- the lane that runs it is the critical path of a whole launch.
__device__ __forceinline__ double tail_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    return r;
}

__device__ inline void inverse3d(const double* m, double* inv) {
#pragma clang fp contract(fast)
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c01 = m[5] * m[6] - m[3] * m[8];
    const double c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = tail_rcp(det);
    inv[0] = c00 * id;
    inv[1] = (m[2] * m[7] - m[1] * m[8]) * id;
    inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    inv[3] = c01 * id;
    inv[4] = (m[0] * m[8] - m[2] * m[6]) * id;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = c02 * id;
    inv[7] = (m[1] * m[6] - m[0] * m[7]) * id;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

__device__ inline void inverse4d(const double* m, double* inv) {  // RGBDOdometry.cpp:348
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2];
    const double s2 = m[0] * m[7] - m[4] * m[3], s3 = m[1] * m[6] - m[5] * m[2];
    const double s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11];
    const double c3 = m[9] * m[14] - m[13] * m[10], c2 = m[8] * m[15] - m[12] * m[11];
    const double c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double id = 1.0 / det;
    inv[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    inv[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    inv[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    inv[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    inv[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    inv[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    inv[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    inv[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    inv[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    inv[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    inv[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    inv[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    inv[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    inv[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    inv[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    inv[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
}

template <int N, typename T>
__device__ inline void matmul(const T* a, const T* b, T* c) {  // c may alias a or b
    T r[N * N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            T s = 0;
            for (int k = 0; k < N; ++k) s += a[i * N + k] * b[k * N + j];
            r[i * N + j] = s;
        }
    for (int i = 0; i < N * N; ++i) c[i] = r[i];
}

__device__ inline void rodrigues(const double* src, double* R) {  // OdometryProvider.h:32-67
#pragma clang fp contract(fast)
    double rx = src[0], ry = src[1], rz = src[2];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (theta >= DBL_EPSILON) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double c = cos(theta), s = sin(theta), c1 = 1. - c;
        const double itheta = theta ? 1. / theta : 0.;
        rx *= itheta;
        ry *= itheta;
        rz *= itheta;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        for (int k = 0; k < 9; ++k) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    }
}

// The same rotation for the small increments of a Gauss-Newton step, without square root, division, sine or cosine:
// with y = |r|^2, R = cos(t) I + ((1 - cos t) / t^2) r r^T + (sin(t) / t) [r]x, and both quotients are short series in y
// (|r| < 1/8: the first omitted terms are < 2e-25).  Mathematically the reference's formula (OdometryProvider.h:32-67
// normalises r first); the results differ by rounding (~1e-16), the chain of dependent double-precision instructions on
// the one lane that runs it is a quarter as long.  Larger angles take the literal form.
__device__ inline void rodrigues_increment(const double* src, double* R) {
#pragma clang fp contract(fast)
    const double rx = src[0], ry = src[1], rz = src[2];
    const double y = rx * rx + ry * ry + rz * rz;
    if (!(y < 0.015625)) {  // |r| >= 1/8 (or not a number)
        rodrigues(src, R);
        return;
    }
    for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
    if (y >= DBL_EPSILON * DBL_EPSILON) {  // theta >= DBL_EPSILON (:38)
        // sin(t) / t and (1 - cos t) / t^2
        const double a = 1.0 + y * (-1.0 / 6 + y * (1.0 / 120 + y * (-1.0 / 5040 + y * (1.0 / 362880 + y * (-1.0 / 39916800 + y * (1.0 / 6227020800.0))))));
        const double b = 0.5 + y * (-1.0 / 24 + y * (1.0 / 720 + y * (-1.0 / 40320 + y * (1.0 / 3628800 + y * (-1.0 / 479001600 + y * (1.0 / 87178291200.0))))));
        const double c = 1.0 - b * y;
        const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
        const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; ++k) R[k] = c * I[k] + b * rrt[k] + a * r_x[k];
    }
}

// LDL^T solve of a symmetric N x N system (what ldlt().solve computes for an SPD matrix up to
// rounding; RGBDOdometry.cpp:298, 435-443)
template <int N, typename T>
__device__ inline void ldlt_solve(const T* A, const T* b, T* x) {
    T L[N * N], D[N], y[N];
    for (int j = 0; j < N; ++j) {
        T d = A[j * N + j];
        for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k] * D[k];
        D[j] = d;
        for (int i = j + 1; i < N; ++i) {
            T s = A[i * N + j];
            for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k] * D[k];
            L[i * N + j] = s / d;
        }
    }
    for (int i = 0; i < N; ++i) {
        T s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k];
        y[i] = s;
    }
    for (int i = 0; i < N; ++i) y[i] /= D[i];
    for (int i = N - 1; i >= 0; --i) {
        T s = y[i];
        for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
        x[i] = s;
    }
}

// Same LDL^T solve with ONE reciprocal per pivot instead of a division per element (21 -> 6
// double divisions on the critical path; each result changes by at most an ulp or two).
template <int N>
__device__ inline void ldlt_solve_recip(const double* A, const double* b, double* x) {
    // double-precision host algebra of the reference: contraction moves a result by ~1e-16 relative and halves the
    // instruction count of the one lane that runs it (the translation unit is built with -ffp-contract=off)
#pragma clang fp contract(fast)
    double L[N * N], D[N], iD[N], y[N];
    for (int j = 0; j < N; ++j) {
        double d = A[j * N + j];
        for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k] * D[k];
        D[j] = d;
        iD[j] = tail_rcp(d);
        for (int i = j + 1; i < N; ++i) {
            double s = A[i * N + j];
            for (int k = 0; k < j; ++k) s -= L[i * N + k] * L[j * N + k] * D[k];
            L[i * N + j] = s * iD[j];
        }
    }
    for (int i = 0; i < N; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * N + k] * y[k];
        y[i] = s;
    }
    for (int i = 0; i < N; ++i) y[i] *= iD[i];
    for (int i = N - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < N; ++k) s -= L[k * N + i] * x[k];
        x[i] = s;
    }
}

struct LevelIntr {  // CameraModel::operator()(level), types.cuh:94-98 (computed on the host)
    float fx, fy, cx, cy;
};

__device__ inline void k_matrix(const LevelIntr& in, double* K) {  // RGBDOdometry.cpp:336-342
    for (int i = 0; i < 9; ++i) K[i] = 0;
    K[0] = in.fx;
    K[4] = in.fy;
    K[2] = in.cx;
    K[5] = in.cy;
    K[8] = 1;
}

// NOTE on structure: everything below works on LOCAL copies (registers) and touches the state in
// HBM exactly twice -- one batch of independent loads at entry, one batch of stores at exit.  A
// first version read and wrote `st->...` in place; every dependent global round trip costs
// 0.5-1 us on one lane and the finishing step took ~10 us.

// SO3 kernel parameters for the next iteration (RGBDOdometry.cpp:261-272)
__device__ inline void so3_prepare(const double* resultR, const LevelIntr& in, float* imageBasis, float* kinv,
                                   float* krlr) {
    double K[9], K_inv[9], tmp[9], H[9];
    k_matrix(in, K);
    inverse3d(K, K_inv);
    matmul<3>(K, resultR, tmp);
    for (int k = 0; k < 9; ++k) krlr[k] = (float)tmp[k];
    matmul<3>(tmp, K_inv, H);
    for (int k = 0; k < 9; ++k) {
        imageBasis[k] = (float)H[k];
        kinv[k] = (float)K_inv[k];
    }
}

__device__ inline void so3_prepare_store(OdomState* st, const double* resultR, const LevelIntr& in) {
    float B[9], ki[9], kr[9];
    so3_prepare(resultR, in, B, ki, kr);
    for (int k = 0; k < 9; ++k) {
        st->imageBasis[k] = B[k];
        st->kinv[k] = ki[k];
        st->krlr[k] = kr[k];
    }
}

// The SO3 pre-alignment's solver state (what RGBDOdometry.cpp:237-310 keeps in locals): it lives in the
// device state between the launches of the per-iteration kernel and in ONE lane's registers for the whole
// loop of the persistent kernel.
struct So3State {
    float lastError, lastCount;  // so3_lastError / so3_lastCount
    int iterations, done;
    float statError, statCount;  // lastSO3Error / lastSO3Count (public statistics)
    double resultR[9], lastResultR[9];
    float R_lr[9];
    float imageBasis[9], kinv[9], krlr[9];  // kernel parameters of the next iteration
};
__device__ inline So3State so3_load(const OdomState* st) {
    So3State s;
    s.lastError = st->so3_lastError, s.lastCount = st->so3_lastCount;
    s.iterations = st->st.so3_iterations_run, s.done = st->so3_done;
    s.statError = st->st.lastSO3Error, s.statCount = st->st.lastSO3Count;
    for (int k = 0; k < 9; ++k) {
        s.resultR[k] = st->resultR[k], s.lastResultR[k] = st->lastResultR[k], s.R_lr[k] = st->R_lr[k];
        s.imageBasis[k] = st->imageBasis[k], s.kinv[k] = st->kinv[k], s.krlr[k] = st->krlr[k];
    }
    return s;
}
__device__ inline void so3_store(OdomState* st, const So3State& s) {
    st->so3_lastError = s.lastError, st->so3_lastCount = s.lastCount;
    st->st.so3_iterations_run = s.iterations, st->so3_done = s.done;
    st->st.lastSO3Error = s.statError, st->st.lastSO3Count = s.statCount;
    for (int k = 0; k < 9; ++k) {
        st->resultR[k] = s.resultR[k], st->lastResultR[k] = s.lastResultR[k], st->R_lr[k] = s.R_lr[k];
        st->imageBasis[k] = s.imageBasis[k], st->kinv[k] = s.kinv[k], st->krlr[k] = s.krlr[k];
    }
}

// after a so3 reduction: RGBDOdometry.cpp:281-308
__device__ inline void so3_step(So3State& s, const float* tot, const LevelIntr& in) {
    float jtj[9], jtr[3];
    int shift = 0;  // reduce.cu:1135-1146
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            const float value = tot[shift++];
            if (j == 3)
                jtr[i] = value;
            else
                jtj[j * 3 + i] = jtj[i * 3 + j] = value;
        }
    const float r0 = tot[9], r1 = tot[10];
    const float err = sqrtf(r0) / r1;
    s.iterations += 1;
    if (err < s.lastError && fabsf(s.lastError - r1) < 0.001) {
        // "converged" (the reference compares the error with the COUNT, :285)
        s.statError = err;
        s.statCount = r1;
        s.done = 1;
        return;
    } else if (err > s.lastError + 0.001) {  // diverging: roll back
        s.statError = s.lastError;
        s.statCount = s.lastCount;
        for (int k = 0; k < 9; ++k) s.resultR[k] = s.lastResultR[k];
        s.done = 1;
        return;
    }
    float delta[3];
    ldlt_solve<3, float>(jtj, jtr, delta);
    const double dd[3] = {delta[0], delta[1], delta[2]};
    double rotUpdate[9];
    float rotUpdatef[9];
    rodrigues(dd, rotUpdate);
    for (int k = 0; k < 9; ++k) rotUpdatef[k] = (float)rotUpdate[k];
    matmul<3, float>(rotUpdatef, s.R_lr, s.R_lr);
    double newR[9];
    for (int k = 0; k < 9; ++k) newR[k] = s.R_lr[k];
    so3_prepare(newR, in, s.imageBasis, s.kinv, s.krlr);
    s.statError = err;
    s.statCount = r1;
    s.lastError = err;
    s.lastCount = r1;
    for (int k = 0; k < 9; ++k) {
        s.lastResultR[k] = s.resultR[k];
        s.resultR[k] = newR[k];
    }
}

__device__ inline void so3_finish(OdomState* st, const float* tot, const LevelIntr& in) {
    So3State s = so3_load(st);
    so3_step(s, tot, in);
    so3_store(st, s);
}

// KRK^-1 and K t of the inverse running transform (RGBDOdometry.cpp:348-358)
//
// The reference evaluates this with a general 4x4 inverse and two dense 3x3 products in double,
// then casts to float.  This runs on ONE lane on the critical path of every Gauss-Newton
// iteration (double-precision issue is ~8 cycles per op there), so the same quantities are
// formed from the structure of the operands: resultRt = [A t; 0 1], so its inverse is
// [A^-1 | -A^-1 t] with the exact 3x3 inverse of A (NOT A^T: with SO3 seeding A is a product of
// float-cast rotations and orthogonal only to ~1e-7, which the reference's general inverse does
// not assume either), and K = [fx 0 cx; 0 fy cy; 0 0 1].  The results differ from the literal
// 4x4 formulas by O(1e-16) relative before the cast to float, with ~5x fewer double operations.
// ifx, ify: 1.0 / (double)in.fx and 1.0 / (double)in.fy when the caller has them (the host computes the same IEEE quotient
// once per launch instead of one lane per workgroup per iteration); 0 = divide here
__device__ inline void rgb_prepare(const double* resultRt, const LevelIntr& in, float* krkinv, float* kt, double ifx_in = 0.0,
                                   double ify_in = 0.0) {
#pragma clang fp contract(fast)
    const double fx = in.fx, fy = in.fy, cx = in.cx, cy = in.cy;
    // Rt = resultRt^-1
    const double A3[9] = {resultRt[0], resultRt[1], resultRt[2], resultRt[4], resultRt[5],
                          resultRt[6], resultRt[8], resultRt[9], resultRt[10]};
    double R[9];
    inverse3d(A3, R);
    const double tx = resultRt[3], ty = resultRt[7], tz = resultRt[11];
    const double t3[3] = {-(R[0] * tx + R[1] * ty + R[2] * tz), -(R[3] * tx + R[4] * ty + R[5] * tz),
                          -(R[6] * tx + R[7] * ty + R[8] * tz)};
    // KR = K * R
    double KR[9];
    for (int c = 0; c < 3; ++c) {
        KR[c] = fx * R[c] + cx * R[6 + c];
        KR[3 + c] = fy * R[3 + c] + cy * R[6 + c];
        KR[6 + c] = R[6 + c];
    }
    // (K R) K^-1 with K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1]
    const double ifx = ifx_in != 0.0 ? ifx_in : 1.0 / fx, ify = ify_in != 0.0 ? ify_in : 1.0 / fy;
    for (int r = 0; r < 3; ++r) {
        const double a = KR[r * 3 + 0] * ifx, b = KR[r * 3 + 1] * ify;
        krkinv[r * 3 + 0] = (float)a;
        krkinv[r * 3 + 1] = (float)b;
        krkinv[r * 3 + 2] = (float)(KR[r * 3 + 2] - a * cx - b * cy);
    }
    kt[0] = (float)(fx * t3[0] + cx * t3[2]);
    kt[1] = (float)(fy * t3[1] + cy * t3[2]);
    kt[2] = (float)t3[2];
}

// unpack 29 sums into the symmetric A and b (reduce.cu:458-472)
__device__ inline void unpack_se3(const float* tot, float* A, float* b) {
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = tot[shift++];
            if (j == 6)
                b[i] = value;
            else
                A[j * 6 + i] = A[i * 6 + j] = value;
        }
}

// combine, solve, update the pose: RGBDOdometry.cpp:412-460 + OdometryProvider.h:69-89.
// tot_rgb / tot_icp: the 29 grid totals of the photometric / ICP reduction (nullptr = term off).
// The state the solve reads.  A reduction kernel loads it at its START (wave-uniform scalar loads that
// cost nothing there), so the finishing lane does not begin its serial tail with a global round trip.
struct SolveIn {
    double w;
    int iters;
    double resultRt[16];
    float Rprev[9], tprev[3];
};
__device__ inline SolveIn load_solve_in(const OdomState* st) {
    SolveIn s;
    s.w = st->icp_weight;
    s.iters = st->st.iterations_run;
    for (int k = 0; k < 16; ++k) s.resultRt[k] = st->resultRt[k];
    for (int k = 0; k < 9; ++k) s.Rprev[k] = st->Rprev[k];
    for (int k = 0; k < 3; ++k) s.tprev[k] = st->tprev[k];
    return s;
}

#ifdef MMF_STAMPS  // diagnostic builds only (tools/rgb_step_probe.py): stamps inside the solve
__device__ unsigned long long* g_mmf_dbg_solve = nullptr;
#define MMF_SOLVE_STAMP(i)                                                                                         \
    do {                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if (g_mmf_dbg_solve && (threadIdx.x & 63) == 0) g_mmf_dbg_solve[blockIdx.x * 16 + (i)] = wall_clock64();    \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
#else
#define MMF_SOLVE_STAMP(i) \
    do {                   \
    } while (0)
#endif

// index of element (i, j), i <= j < 7, in the 29-sum record (the order unpack_se3 walks)
__device__ __forceinline__ int se3_packed_index(int i, int j) { return i * 7 - (i * (i - 1)) / 2 + (j - i); }

// Lane l < 42 of the finishing workgroup: element l of the combined system (l < 36: A[l], else b[l - 36]),
// RGBDOdometry.cpp:431-435 -- the same expression per element as the serial combine.  Also stores it to
// lastA / lastb, so the solving lane neither unpacks, combines nor issues those 42 stores.
__device__ __forceinline__ double combine_element(OdomState* st, int l, double w, const float* tot_rgb, const float* tot_icp) {
    int i, j;
    if (l < 36) {
        const int r = l / 6, c = l - r * 6;
        i = r < c ? r : c, j = r < c ? c : r;
    } else {
        i = l - 36, j = 6;
    }
    const int idx = se3_packed_index(i, j);
    double v;
    if (tot_icp && tot_rgb)
        v = l < 36 ? (double)tot_rgb[idx] + w * w * (double)tot_icp[idx] : (double)tot_rgb[idx] + w * (double)tot_icp[idx];
    else
        v = tot_icp ? (double)tot_icp[idx] : (double)tot_rgb[idx];
    if (st) {  // nullptr: no store (gn_iter_kernel: only the chain's last solve is visible to the host)
        if (l < 36)
            st->st.lastA[l] = v;
        else
            st->st.lastb[l - 36] = v;
    }
    return v;
}

// the same from totals held in double (the fixed-point sums of gn_fused.hpp): A = A_rgb + w^2 A_icp, b = b_rgb + w b_icp
__device__ __forceinline__ double combine_element_d(OdomState* st, int l, double w, const double* tot_rgb, const double* tot_icp) {
    int i, j;
    if (l < 36) {
        const int r = l / 6, c = l - r * 6;
        i = r < c ? r : c, j = r < c ? c : r;
    } else {
        i = l - 36, j = 6;
    }
    const int idx = se3_packed_index(i, j);
    const double v = l < 36 ? tot_rgb[idx] + w * w * tot_icp[idx] : tot_rgb[idx] + w * tot_icp[idx];
    if (st) {
        if (l < 36)
            st->st.lastA[l] = v;
        else
            st->st.lastb[l - 36] = v;
    }
    return v;
}

// What one Gauss-Newton step leaves for the next pass: the pose for the ICP reduction and K R K^-1, K t for the
// photometric correspondence pass (RGBDOdometry.cpp:348-358, 450-460).
struct GnPose {
    float Rcurr[9], tcurr[3], krkinv[9], kt[3];
};

// solve the combined system and update the running transform (RGBDOdometry.cpp:435-460 + OdometryProvider.h:69-89):
// pure function of its arguments, resultRt in / out.  Shared by the finishing lane of rgb_step_kernel / icp_finish_kernel
// and by the prologue of gn_iter_kernel (every workgroup runs it there, on identical inputs).
__device__ inline void gn_solve_core(const double* A, const double* b, double* resultRt, const float* Rprev, const float* tprev,
                                     const LevelIntr& in, GnPose& out, double ifx = 0.0, double ify = 0.0) {
    MMF_SOLVE_STAMP(1);
    double result[6];
    ldlt_solve_recip<6>(A, b, result);
    MMF_SOLVE_STAMP(2);

    double Rup[9];
    const double rvec[3] = {result[3], result[4], result[5]};
    rodrigues_increment(rvec, Rup);
    // resultRt <- [Rup | result(0..2); 0 0 0 1] * resultRt, both with a (0 0 0 1) last row.  The
    // translation column multiplies that last row as in the reference's full 4x4 product
    // (OdometryProvider.h:81-88): for finite numbers it adds an exact 0 or result[r], but a non-finite
    // solution (singular system, e.g. a frame without any correspondence) must poison the whole
    // matrix the way it does there, not only its translation.
    {
#pragma clang fp contract(fast)
        double nr[12];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c)
                nr[r * 4 + c] = Rup[r * 3 + 0] * resultRt[c] + Rup[r * 3 + 1] * resultRt[4 + c] +
                                Rup[r * 3 + 2] * resultRt[8 + c] + result[r] * resultRt[12 + c];
        for (int k = 0; k < 12; ++k) resultRt[k] = nr[k];
    }

    MMF_SOLVE_STAMP(14);
    float Ro[9], to[3], RoT[9], ti[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Ro[r * 3 + c] = (float)resultRt[r * 4 + c];
        to[r] = (float)resultRt[r * 4 + 3];
    }
    // currentT = [Rprev|tprev] * rgbOdom.inverse(); isometry inverse = (R^T, -R^T t)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) RoT[r * 3 + c] = Ro[c * 3 + r];
    for (int r = 0; r < 3; ++r)
        ti[r] = -RoT[r * 3 + 0] * to[0] + -RoT[r * 3 + 1] * to[1] + -RoT[r * 3 + 2] * to[2];
    matmul<3, float>(Rprev, RoT, out.Rcurr);
    for (int r = 0; r < 3; ++r) {
        float s = 0;
        for (int k = 0; k < 3; ++k) s += Rprev[r * 3 + k] * ti[k];
        out.tcurr[r] = s + tprev[r];
    }
    rgb_prepare(resultRt, in, out.krkinv, out.kt, ifx, ify);  // parameters of the next iteration's correspondence pass
    MMF_SOLVE_STAMP(15);
}

// preA / preb (optional): the combined system, already stored to lastA / lastb (combine_element)
__device__ inline void solve_and_update(OdomState* st, const SolveIn& si, const float* tot_rgb, const float* tot_icp,
                                        const LevelIntr& in, const double* preA = nullptr, const double* preb = nullptr) {
    const double w = si.w;
    const int iters = si.iters;
    double resultRt[16];
    float Rprev[9], tprev[3];
    for (int k = 0; k < 16; ++k) resultRt[k] = si.resultRt[k];
    for (int k = 0; k < 9; ++k) Rprev[k] = si.Rprev[k];
    for (int k = 0; k < 3; ++k) tprev[k] = si.tprev[k];

    float A_rgb[36], b_rgb[6], A_icp[36], b_icp[6];
    double A[36], b[6];
    if (preA) {
        for (int k = 0; k < 36; ++k) A[k] = preA[k];
        for (int k = 0; k < 6; ++k) b[k] = preb[k];
    } else if (tot_rgb) unpack_se3(tot_rgb, A_rgb, b_rgb);
    if (!preA && tot_icp) unpack_se3(tot_icp, A_icp, b_icp);
    if (preA) {
    } else if (tot_icp && tot_rgb) {  // :431-435
        for (int k = 0; k < 36; ++k) A[k] = (double)A_rgb[k] + w * w * (double)A_icp[k];
        for (int k = 0; k < 6; ++k) b[k] = (double)b_rgb[k] + w * (double)b_icp[k];
    } else if (tot_icp) {
        for (int k = 0; k < 36; ++k) A[k] = A_icp[k];
        for (int k = 0; k < 6; ++k) b[k] = b_icp[k];
    } else {
        for (int k = 0; k < 36; ++k) A[k] = A_rgb[k];
        for (int k = 0; k < 6; ++k) b[k] = b_rgb[k];
    }
    GnPose np;
    gn_solve_core(A, b, resultRt, Rprev, tprev, in, np);

    // exit stores
    st->st.iterations_run = iters + 1;
    if (!preA) {
        for (int k = 0; k < 36; ++k) st->st.lastA[k] = A[k];
        for (int k = 0; k < 6; ++k) st->st.lastb[k] = b[k];
    }
    for (int k = 0; k < 16; ++k) st->resultRt[k] = resultRt[k];
    for (int k = 0; k < 9; ++k) {
        st->Rcurr[k] = np.Rcurr[k];
        st->krkinv[k] = np.krkinv[k];
    }
    for (int k = 0; k < 3; ++k) {
        st->tcurr[k] = np.tcurr[k];
        st->kt[k] = np.kt[k];
    }
    if (tot_icp) {  // RGBDOdometry.cpp:412-413
        st->st.lastICPError = sqrtf(tot_icp[27]) / tot_icp[28];
        st->st.lastICPCount = tot_icp[28];
    }
}

__device__ inline void solve_and_update(OdomState* st, const float* tot_rgb, const float* tot_icp,
                                        const LevelIntr& in) {
    const SolveIn si = load_solve_in(st);
    solve_and_update(st, si, tot_rgb, tot_icp, in);
}

}  // namespace mmf
