// rigid_ransac.hpp -- host-side restatement of Core/Utils/RigidRANSAC.{h,cpp} (keypoint-based pose
// initialisation: Model::getLastTrackTransform, Model.cpp:739-779, called from MultiMotionFusion.cpp:322
// ahead of the dense tracker).  Plain C++ without Eigen; N is a few dozen to a few hundred keypoints, so this
// stays on the host like the reference's.
//
// Kept from the reference: the hash-sorted correspondence order (RigidRANSAC.cpp:10-58, std::hash<float> of
// libstdc++), std::shuffle on a std::default_random_engine that lives in the object (so successive
// estimate() calls continue its sequence), the candidate test `Ninliers > max(rint(fraction * N), 3)`, the
// refit on the inliers and the mean inlier error as the score.  fit() is the least-squares rigid transform
// T_01 with p0 ~ R p1 + t (Umeyama 1991 / Kabsch): R = U diag(1, 1, det U det V) V^T of the 3x3 correlation
// matrix; the 3x3 SVD is a Jacobi eigen-decomposition of A^T A in double (Eigen::JacobiSVD<Matrix3f> in the
// reference: same rotation up to float rounding wherever it is unique).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <functional>
#include <limits>
#include <random>
#include <utility>
#include <vector>

namespace mmf {

struct Isometry3f {  // row-major 3x3 rotation + translation: x -> R x + t
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float t[3] = {0, 0, 0};
};

namespace ransac_detail {

inline void jacobi_eigen_sym3(double S[9], double V[9]) {  // S symmetric -> eigenvalues on its diagonal, S = V D V^T
    for (int k = 0; k < 9; ++k) V[k] = (k % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = S[1] * S[1] + S[2] * S[2] + S[5] * S[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = S[p * 3 + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (S[q * 3 + q] - S[p * 3 + p]) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 3; ++k) {  // S <- J^T S J, V <- V J
                    const double skp = S[k * 3 + p], skq = S[k * 3 + q];
                    S[k * 3 + p] = c * skp - s * skq;
                    S[k * 3 + q] = s * skp + c * skq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double spk = S[p * 3 + k], sqk = S[q * 3 + k];
                    S[p * 3 + k] = c * spk - s * sqk;
                    S[q * 3 + k] = s * spk + c * sqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
}

inline double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// A = U diag(s) V^T with s descending, U and V orthogonal (full SVD of a 3x3)
inline void svd3(const double A[9], double U[9], double s[3], double V[9]) {
    double AtA[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) AtA[r * 3 + c] = A[0 * 3 + r] * A[0 * 3 + c] + A[1 * 3 + r] * A[1 * 3 + c] + A[2 * 3 + r] * A[2 * 3 + c];
    double Vt[9];
    jacobi_eigen_sym3(AtA, Vt);
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) { return AtA[a * 3 + a] > AtA[b * 3 + b]; });
    for (int c = 0; c < 3; ++c) {
        s[c] = std::sqrt(std::max(0.0, AtA[order[c] * 3 + order[c]]));
        for (int r = 0; r < 3; ++r) V[r * 3 + c] = Vt[r * 3 + order[c]];
    }
    // U columns: A v_c / s_c; rank-deficient columns completed to an orthonormal basis
    double u[3][3];
    int good = 0;
    for (int c = 0; c < 3; ++c) {
        double col[3] = {0, 0, 0};
        for (int r = 0; r < 3; ++r) col[r] = A[r * 3 + 0] * V[0 * 3 + c] + A[r * 3 + 1] * V[1 * 3 + c] + A[r * 3 + 2] * V[2 * 3 + c];
        if (s[c] > 1e-12 * std::max(s[0], 1e-300)) {
            for (int r = 0; r < 3; ++r) u[c][r] = col[r] / s[c];
            good = c + 1;
        }
    }
    auto cross = [](const double* a, const double* b, double* o) {
        o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
    };
    auto normalise = [](double* a) {
        const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (n > 0) a[0] /= n, a[1] /= n, a[2] /= n;
    };
    if (good == 0) u[0][0] = 1, u[0][1] = 0, u[0][2] = 0, good = 1;
    if (good == 1) {  // any unit vector orthogonal to u0
        const double* a = u[0];
        double e[3] = {0, 0, 0};
        e[std::fabs(a[0]) < std::fabs(a[1]) ? (std::fabs(a[0]) < std::fabs(a[2]) ? 0 : 2) : (std::fabs(a[1]) < std::fabs(a[2]) ? 1 : 2)] = 1;
        cross(a, e, u[1]);
        normalise(u[1]);
        good = 2;
    }
    if (good == 2) {
        cross(u[0], u[1], u[2]);
        normalise(u[2]);
    }
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) U[r * 3 + c] = u[c][r];
}

}  // namespace ransac_detail

// RigidRANSAC.cpp:73-120: least-squares T_01 over the rows selected by mask (all rows when mask is null)
inline Isometry3f rigid_fit(const float* p0, const float* p1, int n, const unsigned char* mask = nullptr) {
    using namespace ransac_detail;
    double m0[3] = {0, 0, 0}, m1[3] = {0, 0, 0};
    int cnt = 0;
    for (int i = 0; i < n; ++i)
        if (!mask || mask[i]) {
            for (int k = 0; k < 3; ++k) m0[k] += p0[3 * i + k], m1[k] += p1[3 * i + k];
            ++cnt;
        }
    Isometry3f T;
    if (cnt == 0) return T;
    for (int k = 0; k < 3; ++k) m0[k] /= cnt, m1[k] /= cnt;
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // A[r][c] = sum_i (p0_i - m0)[r] (p1_i - m1)[c]
    for (int i = 0; i < n; ++i)
        if (!mask || mask[i])
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) A[r * 3 + c] += (p0[3 * i + r] - m0[r]) * (p1[3 * i + c] - m1[c]);
    double U[9], s[3], V[9];
    svd3(A, U, s, V);
    const double d = det3(U) * det3(V);  // guarantee det R = +1 (RigidRANSAC.cpp:111)
    double R[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) R[r * 3 + c] = U[r * 3 + 0] * V[c * 3 + 0] + U[r * 3 + 1] * V[c * 3 + 1] + d * U[r * 3 + 2] * V[c * 3 + 2];
    for (int k = 0; k < 9; ++k) T.R[k] = (float)R[k];
    for (int r = 0; r < 3; ++r) T.t[r] = (float)(m0[r] - (R[r * 3 + 0] * m1[0] + R[r * 3 + 1] * m1[1] + R[r * 3 + 2] * m1[2]));
    return T;
}

// RigidRANSAC.cpp:122-126: per-row distance || p0 - T p1 ||
inline void rigid_apply(const Isometry3f& T, const float* p0, const float* p1, int n, float* dist) {
    for (int i = 0; i < n; ++i) {
        float d2 = 0;
        for (int r = 0; r < 3; ++r) {
            const float x = T.R[r * 3 + 0] * p1[3 * i + 0] + T.R[r * 3 + 1] * p1[3 * i + 1] + T.R[r * 3 + 2] * p1[3 * i + 2] + T.t[r];
            const float e = p0[3 * i + r] - x;
            d2 += e * e;
        }
        dist[i] = std::sqrt(d2);
    }
}

class RigidRANSAC {
   public:
    struct Config {
        int iterations;
        float inlier_threshold;
        float inlier_fraction;
    };
    struct Result {
        Isometry3f transformation;
        float error = std::numeric_limits<float>::infinity();
        std::vector<unsigned char> inlier;  // over the HASH-SORTED rows, like the reference's (empty: no model beat the initial fit)
    };

    RigidRANSAC(int iterations, float inlier_threshold, float inlier_fraction) : cfg{iterations, inlier_threshold, inlier_fraction} {}
    explicit RigidRANSAC(const Config& config) : cfg(config) {}

    // RigidRANSAC.cpp:128-180; needs n >= 3 (and >= 3 masked rows when a mask is given)
    Result estimate(const float* p0, const float* p1, int N, const unsigned char* mask = nullptr) {
        Result result;
        std::vector<float> p0s(3 * (size_t)N), p1s(3 * (size_t)N);
        sort_by_hash(p0, p1, N, p0s.data(), p1s.data());
        result.transformation = rigid_fit(p0s.data(), p1s.data(), N, mask);
        std::vector<float> distance(N);
        std::vector<unsigned char> weights(N), inliers(N);
        const int Nparams = 3;
        for (int it = 0; it < cfg.iterations; ++it) {
            std::vector<std::ptrdiff_t> idx(N);
            for (int i = 0; i < N; ++i) idx[i] = i;
            std::shuffle(idx.begin(), idx.end(), generator);
            std::fill(weights.begin(), weights.end(), 0);
            int chosen = 0;
            for (size_t i = 0; i < idx.size() && chosen < Nparams; ++i) {
                const std::ptrdiff_t id = idx[i];
                const unsigned char w = mask ? mask[id] : 1;
                chosen += (w && !weights[id]) ? 1 : 0;
                weights[id] = w;
            }
            if (chosen < Nparams) break;  // the reference asserts here
            const Isometry3f transform = rigid_fit(p0s.data(), p1s.data(), N, weights.data());
            rigid_apply(transform, p0s.data(), p1s.data(), N, distance.data());
            int Ninliers = 0;
            for (int i = 0; i < N; ++i) {
                inliers[i] = (distance[i] < cfg.inlier_threshold) && (!mask || mask[i]);
                Ninliers += inliers[i];
            }
            if (Ninliers > std::max<int>((int)std::rint(cfg.inlier_fraction * N), Nparams)) {
                const Isometry3f Tall = rigid_fit(p0s.data(), p1s.data(), N, inliers.data());
                rigid_apply(Tall, p0s.data(), p1s.data(), N, distance.data());
                float sum = 0;
                for (int i = 0; i < N; ++i) sum += inliers[i] ? distance[i] : 0.f;
                const float error = sum / Ninliers;
                if (error < result.error) {
                    result.error = error;
                    result.transformation = Tall;
                    result.inlier = inliers;
                }
            }
        }
        return result;
    }

   private:
    // RigidRANSAC.cpp:10-58: correspondences ordered by a hash of their six floats
    static void sort_by_hash(const float* p0, const float* p1, int N, float* p0s, float* p1s) {
        auto hash3 = [](const float* v) {
            std::size_t seed = 0;
            for (int i = 0; i < 3; ++i) seed ^= std::hash<float>()(v[i]) + 0xBADEAFFE + (seed << 6) + (seed >> 2);
            return seed;
        };
        std::vector<std::pair<std::size_t, std::size_t>> hash(N);
        for (int i = 0; i < N; ++i) {
            std::size_t seed = 0;
            seed ^= hash3(p0 + 3 * i) + 0xCAFED00D + (seed << 6) + (seed >> 2);
            seed ^= hash3(p1 + 3 * i) + 0xCAFED00D + (seed << 6) + (seed >> 2);
            hash[i] = {seed, (std::size_t)i};
        }
        std::sort(hash.begin(), hash.end());
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 3; ++k) {
                p0s[3 * i + k] = p0[3 * hash[i].second + k];
                p1s[3 * i + k] = p1[3 * hash[i].second + k];
            }
    }

    std::default_random_engine generator;
    const Config cfg;
};

}  // namespace mmf
