/*
 * mmf_oracle_pose.c -- CPU restatement of the host pose algebra around the surfel passes:
 * Model::computeFusionWeight with Model::rodrigues2 and the Eigen operations they call.
 * TEST INFRASTRUCTURE ONLY (see mmf_oracle.h).
 *
 * PARITY UNPINNED.  Eigen is a system package of the reference (find_package(Eigen3 REQUIRED NO_MODULE),
 * CMakeLists.txt:69; version unpinned) and is absent from this image.  What is restated here is its
 * PUBLISHED algorithm for a real square matrix:
 *   JacobiSVD<Matrix3f>::compute      two-sided Jacobi sweeps over (p, q) = (1,0), (2,0), (2,1) on the matrix divided
 *                                     by its largest |coefficient|, threshold max(FLT_MIN, 2 eps * maxDiagEntry),
 *                                     sign fix of U's columns, descending sort of the singular values
 *   internal::real_2x2_jacobi_svd     rot1 symmetrises the 2x2 block, makeJacobi diagonalises it,
 *                                     j_left = rot1 * j_right^T
 *   JacobiRotation::makeJacobi        tau = (x - z) / (2 |y|), t = 1 / (tau +- sqrt(tau^2 + 1))
 *   apply_rotation_in_the_plane       x' = c x + s y, y' = -s x + c y, skipped when (c, s) == (1, 0)
 * all in float32, no contraction.  Matrix4f::inverse() is the cofactor inverse of orc_inverse4f
 * (mmf_oracle_surfel.c); 4x4 / 3x3 products sum left to right.
 */
#include <float.h>
#include <math.h>
#include <string.h>

#include "mmf_oracle.h"

typedef struct {
    float c, s;
} jrot;

/* JacobiRotation::makeJacobi(x, y, z) for the real symmetric block [x y; y z] */
static jrot make_jacobi(float x, float y, float z) {
    jrot j;
    const float deno = 2.0f * fabsf(y);
    if (deno < FLT_MIN) {
        j.c = 1.0f, j.s = 0.0f;
        return j;
    }
    const float tau = (x - z) / deno;
    const float w = sqrtf(tau * tau + 1.0f);
    float t;
    if (tau > 0.0f)
        t = 1.0f / (tau + w);
    else
        t = 1.0f / (tau - w);
    const float sign_t = t > 0.0f ? 1.0f : -1.0f;
    const float n = 1.0f / sqrtf(t * t + 1.0f);
    j.s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    j.c = n;
    return j;
}

/* apply_rotation_in_the_plane on two strided 3-vectors */
static void rot_plane(float *x, float *y, int stride, int n, jrot j) {
    if (j.c == 1.0f && j.s == 0.0f) return;
    for (int i = 0; i < n; ++i) {
        const float xi = x[i * stride], yi = y[i * stride];
        x[i * stride] = j.c * xi + j.s * yi;
        y[i * stride] = -j.s * xi + j.c * yi;
    }
}
static void apply_left(float *m, int p, int q, jrot j) { rot_plane(m + 3 * p, m + 3 * q, 1, 3, j); }
static void apply_right(float *m, int p, int q, jrot j) { /* columns p, q with j.transpose() = (c, -s) */
    jrot t = {j.c, -j.s};
    rot_plane(m + p, m + q, 3, 3, t);
}

/* internal::real_2x2_jacobi_svd */
static void real_2x2_jacobi_svd(const float *w, int p, int q, jrot *j_left, jrot *j_right) {
    float m[4] = {w[3 * p + p], w[3 * p + q], w[3 * q + p], w[3 * q + q]};
    jrot rot1;
    const float t = m[0] + m[3];
    const float d = m[2] - m[1];
    if (fabsf(d) < FLT_MIN) {
        rot1.s = 0.0f, rot1.c = 1.0f;
    } else {
        const float u = t / d;
        const float tmp = sqrtf(1.0f + u * u);
        rot1.s = 1.0f / tmp;
        rot1.c = u / tmp;
    }
    /* m.applyOnTheLeft(0, 1, rot1) */
    if (!(rot1.c == 1.0f && rot1.s == 0.0f))
        for (int i = 0; i < 2; ++i) {
            const float xi = m[i], yi = m[2 + i];
            m[i] = rot1.c * xi + rot1.s * yi;
            m[2 + i] = -rot1.s * xi + rot1.c * yi;
        }
    *j_right = make_jacobi(m[0], m[1], m[3]);
    /* *j_left = rot1 * j_right->transpose() */
    const jrot rt = {j_right->c, -j_right->s};
    j_left->c = rot1.c * rt.c - rot1.s * rt.s;
    j_left->s = rot1.c * rt.s + rot1.s * rt.c;
}

/* JacobiSVD<Matrix3f>(a, ComputeFullU | ComputeFullV): row-major U, V (a = U diag(sv) V^T) */
void orc_jacobi_svd3f(const float a[9], float U[9], float sv[3], float V[9]) {
    const float precision = 2.0f * FLT_EPSILON;
    const float consider_zero = FLT_MIN;
    float scale = 0.0f;
    for (int i = 0; i < 9; ++i)
        if (fabsf(a[i]) > scale) scale = fabsf(a[i]);
    if (scale == 0.0f) scale = 1.0f;
    float w[9];
    for (int i = 0; i < 9; ++i) w[i] = a[i] / scale;
    for (int i = 0; i < 9; ++i) U[i] = V[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    float max_diag = fabsf(w[0]);
    if (fabsf(w[4]) > max_diag) max_diag = fabsf(w[4]);
    if (fabsf(w[8]) > max_diag) max_diag = fabsf(w[8]);
    int finished = 0;
    while (!finished) {
        finished = 1;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const float thr0 = precision * max_diag;
                const float threshold = consider_zero > thr0 ? consider_zero : thr0;
                if (fabsf(w[3 * p + q]) > threshold || fabsf(w[3 * q + p]) > threshold) {
                    finished = 0;
                    jrot jl, jr;
                    real_2x2_jacobi_svd(w, p, q, &jl, &jr);
                    apply_left(w, p, q, jl);
                    jrot jlt = {jl.c, -jl.s};
                    apply_right(U, p, q, jlt); /* m_matrixU.applyOnTheRight(p, q, j_left.transpose()) */
                    apply_right(w, p, q, jr);
                    apply_right(V, p, q, jr);
                    const float dp = fabsf(w[3 * p + p]), dq = fabsf(w[3 * q + q]);
                    const float dm = dp > dq ? dp : dq;
                    if (dm > max_diag) max_diag = dm;
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        const float d = w[4 * i];
        sv[i] = fabsf(d);
        if (d < 0.0f)
            for (int r = 0; r < 3; ++r) U[3 * r + i] = -U[3 * r + i];
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    for (int i = 0; i < 3; ++i) { /* descending sort, columns of U and V follow */
        int pos = i;
        for (int k = i + 1; k < 3; ++k)
            if (sv[k] > sv[pos]) pos = k;
        if (sv[pos] == 0.0f) break;
        if (pos != i) {
            float t = sv[i];
            sv[i] = sv[pos], sv[pos] = t;
            for (int r = 0; r < 3; ++r) {
                t = U[3 * r + i], U[3 * r + i] = U[3 * r + pos], U[3 * r + pos] = t;
                t = V[3 * r + i], V[3 * r + i] = V[3 * r + pos], V[3 * r + pos] = t;
            }
        }
    }
}

/* Model::rodrigues2 (Core/Model/Model.cpp:1301-1342): rotation vector of the orthonormalised matrix */
void orc_rodrigues2(const float matrix[9], float out[3]) {
    float U[9], sv[3], V[9], R[9];
    orc_jacobi_svd3f(matrix, U, sv, V); /* :1302 */
    for (int i = 0; i < 3; ++i)         /* :1303 R = U * V^T */
        for (int j = 0; j < 3; ++j) {
            float s = U[3 * i] * V[3 * j];
            s = s + U[3 * i + 1] * V[3 * j + 1];
            s = s + U[3 * i + 2] * V[3 * j + 2];
            R[3 * i + j] = s;
        }
    double rx = R[7] - R[5]; /* :1305-1307, float differences */
    double ry = R[2] - R[6];
    double rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = ((R[0] + R[4] + R[8]) - 1) * 0.5; /* float trace, float - 1, then double */
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) { /* :1315-1333 */
        double t;
        if (c > 0)
            rx = ry = rz = 0;
        else {
            t = (R[0] + 1) * 0.5;
            rx = sqrt(t > 0.0 ? t : 0.0);
            t = (R[4] + 1) * 0.5;
            ry = sqrt(t > 0.0 ? t : 0.0) * (R[1] < 0 ? -1.0 : 1.0);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(t > 0.0 ? t : 0.0) * (R[2] < 0 ? -1.0 : 1.0);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta, ry *= theta, rz *= theta;
        }
    } else { /* :1334-1340 */
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth, ry *= vth, rz *= vth;
    }
    out[0] = (float)rx, out[1] = (float)ry, out[2] = (float)rz; /* :1341 */
}

/* row-major 4x4 float product (Eigen Matrix4f * Matrix4f), terms summed in k order */
void orc_matmul4f(const float a[16], const float b[16], float out[16]) {
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = a[4 * i] * b[j];
            for (int k = 1; k < 4; ++k) s = s + a[4 * i + k] * b[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(out, r, sizeof(r));
}

static float norm3f(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); } /* Vector3f::norm() */

/* Model::computeFusionWeight (Core/Model/Model.cpp:876-891) with getLastTransform() = pose^-1 * lastPose
 * (Core/Model/Model.h:305) */
float orc_compute_fusion_weight(const float pose[16], const float last_pose[16], float weight_multiplier) {
    float inv[16], diff[16], rv[3];
    orc_inverse4f(pose, inv);
    orc_matmul4f(inv, last_pose, diff); /* :877 */
    const float rot[9] = {diff[0], diff[1], diff[2], diff[4], diff[5], diff[6], diff[8], diff[9], diff[10]};
    orc_rodrigues2(rot, rv);
    const float tn = norm3f(diff[3], diff[7], diff[11]);
    const float rn = norm3f(rv[0], rv[1], rv[2]);
    float weighting = (tn < rn) ? rn : tn; /* :881 std::max(a, b) = (a < b) ? b : a */
    const float largest = 0.01f;
    const float minWeight = 0.5f;
    if (weighting > largest) weighting = largest;
    const float w = 1.0f - (weighting / largest);
    weighting = ((w < minWeight) ? minWeight : w) * weight_multiplier; /* :888 */
    return weighting;
}
