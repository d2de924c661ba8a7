/*
 * mmf_oracle_match.c -- CPU restatement of the keypoint descriptor matcher.
 *
 * TEST INFRASTRUCTURE ONLY (see mmf_oracle.h).  PARITY UNPINNED: the reference delegates to OpenCV, which
 * is not vendored (system package, version unpinned, CMakeLists.txt:56-69) and cannot be built here.
 *
 * What is restated: PointTracker::addKeypoints (Core/Utils/PointTracker.cpp:100-114)
 *     cv::BFMatcher(cv::NORM_L2, true).match(current, previous, matches);      // "query", "train"
 *     keep a match when min_feature_distance < epsilon || match.distance <= min_feature_distance
 * i.e. OpenCV's published brute-force matcher: for every query descriptor the train descriptor at the
 * smallest Euclidean distance (first one on ties), kept only when that train descriptor's nearest query
 * is this query again (crossCheck), reported in query order.
 *
 * Arithmetic (ours to define; OpenCV's float summation order is an implementation detail of its SIMD
 * build): d2(i, j) = (|q_i|^2 + |t_j|^2) - 2 <q_i, t_j> with every sum an fmaf chain in index order,
 * distance = sqrtf(max(d2, 0)).  The f32 MFMA of gfx950 accumulates exactly such a chain, so the HIP
 * kernel (csrc/match_kernels.hpp) agrees with this file bit for bit, ties included.
 */
#include <math.h>
#include <stdlib.h>

#include "mmf_oracle.h"

static float dot_chain(const float *a, const float *b, int n) {
    float s = 0.f;
    for (int k = 0; k < n; ++k) s = fmaf(a[k], b[k], s);
    return s;
}

/* train_idx[i] = matched train row or -1, distance[i] = its distance (0 when unmatched); returns the
 * number of matches */
int orc_match_descriptors(const float *query, int nq, const float *train, int nt, int dim, float max_distance,
                          int *train_idx, float *distance) {
    for (int i = 0; i < nq; ++i) train_idx[i] = -1, distance[i] = 0.f;
    if (nq <= 0 || nt <= 0) return 0;
    float *qn = (float *)malloc(sizeof(float) * (size_t)nq), *tn = (float *)malloc(sizeof(float) * (size_t)nt);
    float *row_d = (float *)malloc(sizeof(float) * (size_t)nq), *col_d = (float *)malloc(sizeof(float) * (size_t)nt);
    int *row_j = (int *)malloc(sizeof(int) * (size_t)nq), *col_i = (int *)malloc(sizeof(int) * (size_t)nt);
    for (int i = 0; i < nq; ++i) qn[i] = dot_chain(query + (size_t)i * dim, query + (size_t)i * dim, dim), row_j[i] = -1;
    for (int j = 0; j < nt; ++j) tn[j] = dot_chain(train + (size_t)j * dim, train + (size_t)j * dim, dim), col_i[j] = -1;
    for (int i = 0; i < nq; ++i)
        for (int j = 0; j < nt; ++j) {
            const float g = dot_chain(query + (size_t)i * dim, train + (size_t)j * dim, dim);
            const float d2 = (qn[i] + tn[j]) - 2.0f * g;
            if (row_j[i] < 0 || d2 < row_d[i]) row_d[i] = d2, row_j[i] = j; /* first minimum: j ascending */
            if (col_i[j] < 0 || d2 < col_d[j]) col_d[j] = d2, col_i[j] = i; /* i ascending */
        }
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        const int j = row_j[i];
        if (j < 0 || col_i[j] != i) continue; /* crossCheck */
        const float d = sqrtf(row_d[i] > 0.f ? row_d[i] : 0.f);
        if (!(max_distance < 1.1920929e-7f || d <= max_distance)) continue; /* PointTracker.cpp:108 */
        train_idx[i] = j;
        distance[i] = d;
        ++n;
    }
    free(qn), free(tn), free(row_d), free(col_d), free(row_j), free(col_i);
    return n;
}
