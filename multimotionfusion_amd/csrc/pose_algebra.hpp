// pose_algebra.hpp -- pose algebra of the orchestrator: Model::computeFusionWeight and
// Model::rodrigues2 (Core/Model/Model.cpp:876-891, 1301-1342) with the Eigen operations they call.
// Host code in the reference; here callable on both sides (MMF_HD): the host uses it wherever it has the pose, and the
// last step of a tracking chain evaluates the fusion weight on the device so that the fusion passes of the frame can be
// enqueued before the pose has reached the host (same float operations in the same order; the one double-precision
// library call, acos, differs between the two C libraries by at most an ulp of a double, far below the float the
// result is rounded to).
//
// Eigen (an unpinned system package of the reference, absent here) is restated from its published
// algorithm: JacobiSVD of a real square matrix = two-sided Jacobi sweeps over the pairs (1,0), (2,0),
// (2,1) of the matrix scaled by its largest magnitude, each pair handled by real_2x2_jacobi_svd
// (a symmetrising rotation followed by JacobiRotation::makeJacobi), until every off-diagonal pair is below
// max(FLT_MIN, 2 eps * largest diagonal entry); then negative diagonal entries flip their column of U and
// the singular values are sorted in descending order.  Everything in float32, no contraction
// (the library is built with -ffp-contract=off).
#pragma once
#include <cfloat>
#include <cmath>

#if defined(__HIPCC__)
#define MMF_HD __host__ __device__
#else
#define MMF_HD
#endif

namespace mmf {
namespace host {

struct Givens {  // Eigen::JacobiRotation<float>
    float c = 1.f, s = 0.f;
    MMF_HD Givens transpose() const { return Givens{c, -s}; }
    MMF_HD Givens operator*(const Givens& o) const { return Givens{c * o.c - s * o.s, c * o.s + s * o.c}; }
    MMF_HD bool identity() const { return c == 1.f && s == 0.f; }
};

struct Mat3 {
    float m[3][3];
    MMF_HD static Mat3 eye() {
        Mat3 r;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r.m[i][j] = i == j ? 1.f : 0.f;
        return r;
    }
    // rows p, q <- j applied on the left
    MMF_HD void rotate_rows(int p, int q, const Givens& j) {
        if (j.identity()) return;
        for (int k = 0; k < 3; ++k) {
            const float x = m[p][k], y = m[q][k];
            m[p][k] = j.c * x + j.s * y;
            m[q][k] = -j.s * x + j.c * y;
        }
    }
    // columns p, q <- j applied on the right (Eigen rotates them by j.transpose())
    MMF_HD void rotate_cols(int p, int q, const Givens& j) {
        const Givens t = j.transpose();
        if (t.identity()) return;
        for (int k = 0; k < 3; ++k) {
            const float x = m[k][p], y = m[k][q];
            m[k][p] = t.c * x + t.s * y;
            m[k][q] = -t.s * x + t.c * y;
        }
    }
};

// JacobiRotation::makeJacobi for the symmetric 2x2 block [x y; y z]
MMF_HD inline Givens jacobi_of_symmetric(float x, float y, float z) {
    const float deno = 2.f * std::fabs(y);
    if (deno < FLT_MIN) return Givens{};
    const float tau = (x - z) / deno;
    const float w = std::sqrt(tau * tau + 1.f);
    const float t = tau > 0.f ? 1.f / (tau + w) : 1.f / (tau - w);
    const float sign_t = t > 0.f ? 1.f : -1.f;
    const float n = 1.f / std::sqrt(t * t + 1.f);
    return Givens{n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}

// internal::real_2x2_jacobi_svd on the (p, q) block of w
MMF_HD inline void svd_2x2(const Mat3& w, int p, int q, Givens& left, Givens& right) {
    float a = w.m[p][p], b = w.m[p][q], c = w.m[q][p], d = w.m[q][q];
    Givens rot1;
    const float t = a + d, diff = c - b;
    if (std::fabs(diff) < FLT_MIN) {
        rot1 = Givens{1.f, 0.f};
    } else {
        const float u = t / diff;
        const float tmp = std::sqrt(1.f + u * u);
        rot1 = Givens{u / tmp, 1.f / tmp};
    }
    if (!rot1.identity()) {
        const float a2 = rot1.c * a + rot1.s * c, b2 = rot1.c * b + rot1.s * d;
        const float c2 = -rot1.s * a + rot1.c * c, d2 = -rot1.s * b + rot1.c * d;
        a = a2, b = b2, c = c2, d = d2;
    }
    right = jacobi_of_symmetric(a, b, d);
    left = rot1 * right.transpose();
}

// Eigen::JacobiSVD<Matrix3f>(a, ComputeFullU | ComputeFullV)
MMF_HD inline void jacobi_svd3(const Mat3& a, Mat3& U, float sv[3], Mat3& V) {
    float scale = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) scale = std::fabs(a.m[i][j]) > scale ? std::fabs(a.m[i][j]) : scale;
    if (scale == 0.f) scale = 1.f;
    Mat3 w;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) w.m[i][j] = a.m[i][j] / scale;
    U = Mat3::eye(), V = Mat3::eye();
    float max_diag = 0.f;
    for (int i = 0; i < 3; ++i) max_diag = std::fabs(w.m[i][i]) > max_diag ? std::fabs(w.m[i][i]) : max_diag;
    for (bool finished = false; !finished;) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                float threshold = 2.f * FLT_EPSILON * max_diag;
                if (threshold < FLT_MIN) threshold = FLT_MIN;
                if (!(std::fabs(w.m[p][q]) > threshold || std::fabs(w.m[q][p]) > threshold)) continue;
                finished = false;
                Givens jl, jr;
                svd_2x2(w, p, q, jl, jr);
                w.rotate_rows(p, q, jl);
                U.rotate_cols(p, q, jl.transpose());
                w.rotate_cols(p, q, jr);
                V.rotate_cols(p, q, jr);
                const float dp = std::fabs(w.m[p][p]), dq = std::fabs(w.m[q][q]);
                const float dm = dp > dq ? dp : dq;
                if (dm > max_diag) max_diag = dm;
            }
    }
    for (int i = 0; i < 3; ++i) {
        sv[i] = std::fabs(w.m[i][i]);
        if (w.m[i][i] < 0.f)
            for (int r = 0; r < 3; ++r) U.m[r][i] = -U.m[r][i];
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    for (int i = 0; i < 3; ++i) {
        int pos = i;
        for (int k = i + 1; k < 3; ++k)
            if (sv[k] > sv[pos]) pos = k;
        if (sv[pos] == 0.f) break;
        if (pos == i) continue;
        const float tsv = sv[i];
        sv[i] = sv[pos], sv[pos] = tsv;
        for (int r = 0; r < 3; ++r) {
            const float tu = U.m[r][i], tv = V.m[r][i];
            U.m[r][i] = U.m[r][pos], U.m[r][pos] = tu;
            V.m[r][i] = V.m[r][pos], V.m[r][pos] = tv;
        }
    }
}

// Model::rodrigues2 (Model.cpp:1301-1342)
MMF_HD inline void rodrigues2(const Mat3& matrix, float out[3]) {
    Mat3 U, V, R;
    float sv[3];
    jacobi_svd3(matrix, U, sv, V);
    for (int i = 0; i < 3; ++i)  // R = U V^T (:1303)
        for (int j = 0; j < 3; ++j) {
            float acc = U.m[i][0] * V.m[j][0];
            acc = acc + U.m[i][1] * V.m[j][1];
            acc = acc + U.m[i][2] * V.m[j][2];
            R.m[i][j] = acc;
        }
    double rx = R.m[2][1] - R.m[1][2];
    double ry = R.m[0][2] - R.m[2][0];
    double rz = R.m[1][0] - R.m[0][1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = ((R.m[0][0] + R.m[1][1] + R.m[2][2]) - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            rx = ry = rz = 0;
        } else {
            double t = (R.m[0][0] + 1) * 0.5;
            rx = std::sqrt(t > 0.0 ? t : 0.0);
            t = (R.m[1][1] + 1) * 0.5;
            ry = std::sqrt(t > 0.0 ? t : 0.0) * (R.m[0][1] < 0 ? -1.0 : 1.0);
            t = (R.m[2][2] + 1) * 0.5;
            rz = std::sqrt(t > 0.0 ? t : 0.0) * (R.m[0][2] < 0 ? -1.0 : 1.0);
            if (std::fabs(rx) < std::fabs(ry) && std::fabs(rx) < std::fabs(rz) && (R.m[1][2] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= std::sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta, ry *= theta, rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth, ry *= vth, rz *= vth;
    }
    out[0] = (float)rx, out[1] = (float)ry, out[2] = (float)rz;
}

// row-major 4x4 float product, terms summed in k order
MMF_HD inline void matmul4(const float* a, const float* b, float* out) {
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = a[4 * i] * b[j];
            for (int k = 1; k < 4; ++k) acc = acc + a[4 * i + k] * b[4 * k + j];
            r[4 * i + j] = acc;
        }
    for (int k = 0; k < 16; ++k) out[k] = r[k];
}

MMF_HD inline float norm3(float x, float y, float z) { return std::sqrt((x * x + y * y) + z * z); }

// Model::computeFusionWeight (Model.cpp:876-891); pose_inv = getPose().inverse()
MMF_HD inline float compute_fusion_weight(const float* pose_inv, const float* last_pose, float weight_multiplier) {
    float diff[16], rv[3];
    matmul4(pose_inv, last_pose, diff);  // getLastTransform() (Model.h:305)
    Mat3 rot;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rot.m[i][j] = diff[4 * i + j];
    rodrigues2(rot, rv);
    const float tn = norm3(diff[3], diff[7], diff[11]), rn = norm3(rv[0], rv[1], rv[2]);
    float weighting = tn < rn ? rn : tn;
    const float largest = 0.01f, minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    const float w = 1.0f - (weighting / largest);
    return (w < minWeight ? minWeight : w) * weight_multiplier;
}

}  // namespace host
}  // namespace mmf
