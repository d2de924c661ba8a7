/*
 * mmf_oracle.c -- CPU restatement of the reference's dense-tracking kernels.
 * TEST INFRASTRUCTURE ONLY; see mmf_oracle.h for the contract.  PARITY UNPINNED.
 *
 * Every function cites the reference lines it follows (paths relative to the reference
 * tree).  Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math).
 */
#include "mmf_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------- */
/* float3 helpers in the reference's operation order (Core/Cuda/operators.cuh:56-91)      */
/* ------------------------------------------------------------------------------------- */
typedef struct {
    float x, y, z;
} f3;

static inline f3 f3_make(float x, float y, float z) {
    f3 r = {x, y, z};
    return r;
}
static inline f3 f3_sub(f3 a, f3 b) { return f3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_add(f3 a, f3 b) { return f3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_cross(f3 a, f3 b) {
    return f3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float f3_norm(f3 a) { return sqrtf(f3_dot(a, a)); }
/* operators.cuh:80-84 uses rsqrtf (approximate on CUDA); restated with a correctly rounded
 * reciprocal square root so the HIP kernels can match bit-for-bit. */
static inline f3 f3_normalized(f3 a) {
    const float rn = 1.0f / sqrtf(f3_dot(a, a));
    return f3_make(a.x * rn, a.y * rn, a.z * rn);
}
/* mat33 * float3 (operators.cuh:86-89); m is row major */
static inline f3 m33_mul(const float m[9], f3 a) {
    return f3_make(f3_dot(f3_make(m[0], m[1], m[2]), a), f3_dot(f3_make(m[3], m[4], m[5]), a),
                   f3_dot(f3_make(m[6], m[7], m[8]), a));
}

/* CUDA __float2int_rn: round to nearest even, NaN -> 0, saturating (reduce.cu:269-270 relies
 * on the NaN -> 0 behaviour for invalid vertices). */
static inline int float2int_rn(float x) {
    if (isnan(x)) return 0;
    if (x >= 2147483648.0f) return INT_MAX;
    if (x <= -2147483648.0f) return INT_MIN;
    return (int)rintf(x);
}

static inline float qnan_f(void) {
    union {
        uint32_t u;
        float f;
    } v;
    v.u = 0x7fffffffu; /* cudafuncs.cu:131 */
    return v.f;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------- */
/* Map kernels                                                                            */
/* ------------------------------------------------------------------------------------- */

/* Core/Cuda/cudafuncs.cu:109-134 (computeVmapKernel) + :136-150 (createVMap).
 * The mask test is commented out in the reference (:119) so no mask is taken. */
void orc_create_vmap(const float *depth, int cols, int rows, float fx, float fy, float cx, float cy,
                     float depth_cutoff, float *vmap) {
    const float fx_inv = 1.f / fx, fy_inv = 1.f / fy;
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            const float z = depth[v * cols + u];
            if (z != 0 && z < depth_cutoff) {
                vmap[v * cols + u] = z * (u - cx) * fx_inv;
                vmap[(v + rows) * cols + u] = z * (v - cy) * fy_inv;
                vmap[(v + 2 * rows) * cols + u] = z;
            } else {
                vmap[v * cols + u] = qnan_f(); /* only the x plane is written */
            }
        }
}

/* Core/Cuda/cudafuncs.cu:152-189 (computeNmapKernel) */
void orc_create_nmap(const float *vmap, int cols, int rows, float *nmap) {
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            if (u == cols - 1 || v == rows - 1) {
                nmap[v * cols + u] = qnan_f();
                continue;
            }
            f3 v00, v01, v10;
            v00.x = vmap[v * cols + u];
            v01.x = vmap[v * cols + u + 1];
            v10.x = vmap[(v + 1) * cols + u];
            if (!isnan(v00.x) && !isnan(v01.x) && !isnan(v10.x)) {
                v00.y = vmap[(v + rows) * cols + u];
                v01.y = vmap[(v + rows) * cols + u + 1];
                v10.y = vmap[(v + 1 + rows) * cols + u];
                v00.z = vmap[(v + 2 * rows) * cols + u];
                v01.z = vmap[(v + 2 * rows) * cols + u + 1];
                v10.z = vmap[(v + 1 + 2 * rows) * cols + u];
                const f3 r = f3_normalized(f3_cross(f3_sub(v01, v00), f3_sub(v10, v00)));
                nmap[v * cols + u] = r.x;
                nmap[(v + rows) * cols + u] = r.y;
                nmap[(v + 2 * rows) * cols + u] = r.z;
            } else {
                nmap[v * cols + u] = qnan_f();
            }
        }
}

/* Core/Cuda/cudafuncs.cu:207-249 (tranformMapsKernel); src may alias dst (the reference calls
 * it in place, RGBDOdometry.cpp:171). */
void orc_transform_maps(const float *vsrc, const float *nsrc, int cols, int rows, const float R[9],
                        const float t[3], float *vdst, float *ndst) {
    const f3 tv = f3_make(t[0], t[1], t[2]);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            f3 s, d = f3_make(qnan_f(), qnan_f(), qnan_f());
            s.x = vsrc[y * cols + x];
            if (!isnan(s.x)) {
                s.y = vsrc[(y + rows) * cols + x];
                s.z = vsrc[(y + 2 * rows) * cols + x];
                d = f3_add(m33_mul(R, s), tv);
                vdst[(y + rows) * cols + x] = d.y;
                vdst[(y + 2 * rows) * cols + x] = d.z;
            }
            vdst[y * cols + x] = d.x;

            f3 n, nd = f3_make(qnan_f(), qnan_f(), qnan_f());
            n.x = nsrc[y * cols + x];
            if (!isnan(n.x)) {
                n.y = nsrc[(y + rows) * cols + x];
                n.z = nsrc[(y + 2 * rows) * cols + x];
                nd = m33_mul(R, n);
                ndst[(y + rows) * cols + x] = nd.y;
                ndst[(y + 2 * rows) * cols + x] = nd.z;
            }
            ndst[y * cols + x] = nd.x;
        }
}

/* Core/Cuda/cudafuncs.cu:271-311 (copyMapsKernel): RGBA32F interleaved -> planar; z==0 => NaN.
 * Note the normal is also gated on the VERTEX z (:302). */
void orc_copy_maps(const float *vsrc, const float *nsrc, int cols, int rows, float *vdst,
                   float *ndst) {
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float *v = vsrc + (size_t)(y * cols + x) * 4;
            const float *n = nsrc + (size_t)(y * cols + x) * 4;
            f3 vd = f3_make(qnan_f(), qnan_f(), qnan_f()), nd = vd;
            if (!(v[2] == 0)) {
                vd = f3_make(v[0], v[1], v[2]);
                nd = f3_make(n[0], n[1], n[2]);
            }
            vdst[y * cols + x] = vd.x;
            vdst[(y + rows) * cols + x] = vd.y;
            vdst[(y + 2 * rows) * cols + x] = vd.z;
            ndst[y * cols + x] = nd.x;
            ndst[(y + rows) * cols + x] = nd.y;
            ndst[(y + 2 * rows) * cols + x] = nd.z;
        }
}

/* Core/Cuda/cudafuncs.cu:366-417 (resizeMapKernel<normalize>) */
void orc_resize_map(const float *in, int in_cols, int in_rows, int normalize, float *out) {
    const int dcols = in_cols / 2, drows = in_rows / 2, srows = in_rows;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            const int xs = x * 2, ys = y * 2;
            const float x00 = in[(ys + 0) * in_cols + xs + 0];
            const float x01 = in[(ys + 0) * in_cols + xs + 1];
            const float x10 = in[(ys + 1) * in_cols + xs + 0];
            const float x11 = in[(ys + 1) * in_cols + xs + 1];
            if (isnan(x00) || isnan(x01) || isnan(x10) || isnan(x11)) {
                out[y * dcols + x] = qnan_f();
                continue;
            }
            f3 n;
            n.x = (x00 + x01 + x10 + x11) / 4;
            const float y00 = in[(ys + srows + 0) * in_cols + xs + 0];
            const float y01 = in[(ys + srows + 0) * in_cols + xs + 1];
            const float y10 = in[(ys + srows + 1) * in_cols + xs + 0];
            const float y11 = in[(ys + srows + 1) * in_cols + xs + 1];
            n.y = (y00 + y01 + y10 + y11) / 4;
            const float z00 = in[(ys + 2 * srows + 0) * in_cols + xs + 0];
            const float z01 = in[(ys + 2 * srows + 0) * in_cols + xs + 1];
            const float z10 = in[(ys + 2 * srows + 1) * in_cols + xs + 0];
            const float z11 = in[(ys + 2 * srows + 1) * in_cols + xs + 1];
            n.z = (z00 + z01 + z10 + z11) / 4;
            if (normalize) n = f3_normalized(n);
            out[y * dcols + x] = n.x;
            out[(y + drows) * dcols + x] = n.y;
            out[(y + 2 * drows) * dcols + x] = n.z;
        }
}

static const float k_gauss5[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36,
                                   24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1}; /* cudafuncs.cu:517-521 */

/* Core/Cuda/cudafuncs.cu:333-364 (pyrDownKernelGaussF).  Quirks kept: `count` is an int that
 * accumulates float weights (:350,359), the window is [max(0,2x-2), min(2x+3,cols-1)) and the
 * weight index is mirrored from the clipped end (:358). */
void orc_pyrdown_gauss_f(const float *src, int src_cols, int src_rows, float *dst) {
    const int dcols = src_cols / 2, drows = src_rows / 2, D = 5;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            const int tx = imin(2 * x - D / 2 + D, src_cols - 1);
            const int ty = imin(2 * y - D / 2 + D, src_rows - 1);
            float sum = 0;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    const float s = src[cy * src_cols + cx];
                    if (!isnan(s)) {
                        const float w = k_gauss5[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += s * w;
                        count = (int)((float)count + w);
                    }
                }
            dst[y * dcols + x] = (float)(sum / (float)count);
        }
}

/* Core/Cuda/cudafuncs.cu:534-564 (pyrDownKernelIntensityGauss): zeros are skipped; the float
 * quotient is truncated to uchar (NaN -> 0 as the CUDA conversion does). */
void orc_pyrdown_uchar_gauss(const uint8_t *src, int src_cols, int src_rows, uint8_t *dst) {
    const int dcols = src_cols / 2, drows = src_rows / 2, D = 5;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            const int tx = imin(2 * x - D / 2 + D, src_cols - 1);
            const int ty = imin(2 * y - D / 2 + D, src_rows - 1);
            float sum = 0;
            int count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    const uint8_t s = src[cy * src_cols + cx];
                    if (s > 0) {
                        const float w = k_gauss5[(ty - cy - 1) * 5 + (tx - cx - 1)];
                        sum += s * w;
                        count = (int)((float)count + w);
                    }
                }
            const float q = sum / (float)count;
            dst[y * dcols + x] = isnan(q) ? 0 : (uint8_t)(unsigned)q;
        }
}

/* Core/Cuda/cudafuncs.cu:602-613 (verticesToDepthKernel) */
void orc_vertices_to_depth(const float *vmap_rgba, int cols, int rows, float cutoff, float *dst) {
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float z = vmap_rgba[(size_t)(y * cols + x) * 4 + 2];
            dst[y * cols + x] = (z > cutoff || z <= 0) ? qnan_f() : z;
        }
}

/* Core/Cuda/cudafuncs.cu:624-637 (bgr2IntensityKernel): channel 0/1/2 of the texture AS
 * UPLOADED weighted 0.114/0.299/0.587 and truncated to int. */
void orc_image_to_intensity(const uint8_t *img, int channels, int cols, int rows, uint8_t *dst) {
    for (int i = 0; i < cols * rows; ++i) {
        const uint8_t *p = img + (size_t)i * channels;
        const int value = (int)((float)p[0] * 0.114f + (float)p[1] * 0.299f + (float)p[2] * 0.587f);
        dst[i] = (uint8_t)value;
    }
}

/* Core/Cuda/cudafuncs.cu:669-694 (applyKernel) with the coefficient tables of :702-708.
 * Quirk kept: the kernel index runs down from 8 over the taps actually visited, so at the
 * image border the shrunken window uses misaligned coefficients. */
void orc_derivative_images(const uint8_t *src, int cols, int rows, int16_t *dx, int16_t *dy) {
    const float gx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f,
                         -0.79451f, 0.52201f, 0.00000f, -0.52201f};
    const float gy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f,
                         0.00000f, -0.52201f, -0.79451f, -0.52201f};
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float dxv = 0, dyv = 0;
            int k = 8;
            for (int j = imax(y - 1, 0); j <= imin(y + 1, rows - 1); ++j)
                for (int i = imax(x - 1, 0); i <= imin(x + 1, cols - 1); ++i) {
                    dxv += (float)src[j * cols + i] * gx[k];
                    dyv += (float)src[j * cols + i] * gy[k];
                    --k;
                }
            dx[y * cols + x] = (int16_t)dxv;
            dy[y * cols + x] = (int16_t)dyv;
        }
}

/* Core/Cuda/cudafuncs.cu:729-747 (projectPointsKernel); cloud is AoS float3 */
void orc_project_to_cloud(const float *depth, int cols, int rows, float fx, float fy, float cx,
                          float cy, float *cloud) {
    const float inv_fx = 1.0f / fx, inv_fy = 1.0f / fy;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float z = depth[y * cols + x];
            float *c = cloud + (size_t)(y * cols + x) * 3;
            c[0] = (float)((x - cx) * z * inv_fx);
            c[1] = (float)((y - cy) * z * inv_fy);
            c[2] = z;
        }
}

/* ------------------------------------------------------------------------------------- */
/* ICP reduction                                                                          */
/* ------------------------------------------------------------------------------------- */

typedef struct {
    const float *Rcurr, *tcurr, *vmap_curr, *nmap_curr, *Rprev_inv, *tprev;
    float fx, fy, cx, cy;
    const float *vmap_g_prev, *nmap_g_prev;
    float dist_thres, angle_thres;
    int cols, rows;
} icp_args;

/* Core/Cuda/reduce.cu:257-368 (ICPReduction::search + getProducts) for one pixel.
 * Returns found; row[7] is the Jacobian row (zeros when not found). */
static inline int icp_row(const icp_args *a, int x, int y, float row[7], float *err_map) {
    const int cols = a->cols, rows = a->rows;
    const f3 tcurr = f3_make(a->tcurr[0], a->tcurr[1], a->tcurr[2]);
    const f3 tprev = f3_make(a->tprev[0], a->tprev[1], a->tprev[2]);
    for (int k = 0; k < 7; ++k) row[k] = 0;

    f3 vcurr;
    vcurr.x = a->vmap_curr[y * cols + x];
    vcurr.y = a->vmap_curr[(y + rows) * cols + x];
    vcurr.z = a->vmap_curr[(y + 2 * rows) * cols + x];

    const f3 vcurr_g = f3_add(m33_mul(a->Rcurr, vcurr), tcurr);
    const f3 vcurr_cp = m33_mul(a->Rprev_inv, f3_sub(vcurr_g, tprev));

    const int ux = float2int_rn(vcurr_cp.x * a->fx / vcurr_cp.z + a->cx);
    const int uy = float2int_rn(vcurr_cp.y * a->fy / vcurr_cp.z + a->cy);

    if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp.z < 0) {
        if (err_map) err_map[y * cols + x] = 0.0f; /* reduce.cu:275 */
        return 0;
    }

    f3 vprev_g, ncurr, nprev_g;
    vprev_g.x = a->vmap_g_prev[uy * cols + ux];
    vprev_g.y = a->vmap_g_prev[(uy + rows) * cols + ux];
    vprev_g.z = a->vmap_g_prev[(uy + 2 * rows) * cols + ux];
    ncurr.x = a->nmap_curr[y * cols + x];
    ncurr.y = a->nmap_curr[(y + rows) * cols + x];
    ncurr.z = a->nmap_curr[(y + 2 * rows) * cols + x];
    const f3 ncurr_g = m33_mul(a->Rcurr, ncurr);
    nprev_g.x = a->nmap_g_prev[uy * cols + ux];
    nprev_g.y = a->nmap_g_prev[(uy + rows) * cols + ux];
    nprev_g.z = a->nmap_g_prev[(uy + 2 * rows) * cols + ux];

    const float dist = f3_norm(f3_sub(vprev_g, vcurr_g));
    const float sine = f3_norm(f3_cross(ncurr_g, nprev_g));

    if (err_map) err_map[y * cols + x] = isfinite(dist) ? dist : 0.0f; /* reduce.cu:299 */

    const int found =
        (sine < a->angle_thres && dist <= a->dist_thres && !isnan(ncurr.x) && !isnan(nprev_g.x));
    if (found) {
        /* reduce.cu:320-329 */
        const f3 s_cp = m33_mul(a->Rprev_inv, f3_sub(vcurr_g, tprev));
        const f3 d_cp = m33_mul(a->Rprev_inv, f3_sub(vprev_g, tprev));
        const f3 n_cp = m33_mul(a->Rprev_inv, nprev_g);
        const f3 c = f3_cross(s_cp, n_cp);
        row[0] = n_cp.x;
        row[1] = n_cp.y;
        row[2] = n_cp.z;
        row[3] = c.x;
        row[4] = c.y;
        row[5] = c.z;
        row[6] = f3_dot(n_cp, f3_sub(s_cp, d_cp));
    }
    return found;
}

/* the 27 + 2 products in the member order of JtJJtrSE3 (types.cuh:101-112, reduce.cu:331-365) */
static inline void se3_products(const float row[7], int found, float p[29]) {
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) p[k++] = row[i] * row[j];
    p[27] = row[6] * row[6];
    p[28] = (float)found;
}

/* Core/Cuda/reduce.cu:370-391 (grid-stride sum) + :399-473; sums kept in double */
void orc_icp_step(const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                  const float *nmap_curr, const float Rprev_inv[9], const float tprev[3], float fx,
                  float fy, float cx, float cy, const float *vmap_g_prev, const float *nmap_g_prev,
                  float dist_thres, float angle_thres, int cols, int rows, double out29[29],
                  float *err_map) {
    icp_args a = {Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, fx, fy, cx, cy,
                  vmap_g_prev, nmap_g_prev, dist_thres, angle_thres, cols, rows};
    for (int k = 0; k < 29; ++k) out29[k] = 0;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float row[7], p[29];
            const int found = icp_row(&a, x, y, row, err_map);
            se3_products(row, found, p);
            for (int k = 0; k < 29; ++k) out29[k] += (double)p[k];
        }
}

int orc_omp_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* "#pragma omp parallel for over image rows with a 29-float
 * reduction, float accumulators" -- the naive OpenMP CPU run reported beside the GPU number. */
void orc_icp_step_omp_f32(const float Rcurr[9], const float tcurr[3], const float *vmap_curr,
                          const float *nmap_curr, const float Rprev_inv[9], const float tprev[3],
                          float fx, float fy, float cx, float cy, const float *vmap_g_prev,
                          const float *nmap_g_prev, float dist_thres, float angle_thres, int cols,
                          int rows, float out29[29]) {
    icp_args a = {Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, fx, fy, cx, cy,
                  vmap_g_prev, nmap_g_prev, dist_thres, angle_thres, cols, rows};
    float acc[29];
    for (int k = 0; k < 29; ++k) acc[k] = 0;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : acc[:29]) schedule(static)
#endif
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float row[7], p[29];
            const int found = icp_row(&a, x, y, row, NULL);
            se3_products(row, found, p);
            for (int k = 0; k < 29; ++k) acc[k] += p[k];
        }
    for (int k = 0; k < 29; ++k) out29[k] = acc[k];
}

/* ------------------------------------------------------------------------------------- */
/* RGB residual / step                                                                    */
/* ------------------------------------------------------------------------------------- */

/* Core/Cuda/reduce.cu:759-839 (RGBResidual::getProducts) + :867-945.  The mask test is
 * compiled out in the reference (MASK_RGB_RESIDUAL undefined).  The int2 sum wraps like the
 * device int does.  Invalid records are written as zeros (the reference leaves their other
 * fields uninitialised). */
void orc_rgb_residual(float min_scale, const int16_t *dIdx, const int16_t *dIdy,
                      const float *last_depth, const float *next_depth, const uint8_t *last_image,
                      const uint8_t *next_image, orc_dataterm *corres, float max_depth_delta,
                      const float kt[3], const float K[9], int cols, int rows, int *sigma_sum,
                      int *count, float *err_map) {
    uint32_t sum_x = 0, sum_y = 0;
    for (int i = 0; i < rows; ++i)
        for (int j0 = 0; j0 < cols; ++j0) {
            orc_dataterm c;
            memset(&c, 0, sizeof(c));
            int vx = 0, vy = 0;
            if (j0 < cols - 5 && i < rows - 1) {
                int valid = 1;
                for (int u = imax(i - 2, 0); u < imin(i + 2, rows); u++)
                    for (int v = imax(j0 - 2, 0); v < imin(j0 + 2, cols); v++)
                        valid = valid && (next_image[u * cols + v] > 0);
                if (valid) {
                    const int16_t valx = dIdx[i * cols + j0], valy = dIdy[i * cols + j0];
                    const float mTwo = (float)((valx * valx) + (valy * valy));
                    if (mTwo >= min_scale) {
                        const int y = i, x = j0;
                        const float d1 = next_depth[y * cols + x];
                        if (!isnan(d1)) {
                            const float td1 = (float)(d1 * (K[6] * x + K[7] * y + K[8]) + kt[2]);
                            const int u0 = float2int_rn(
                                (d1 * (K[0] * x + K[1] * y + K[2]) + kt[0]) / td1);
                            const int v0 = float2int_rn(
                                (d1 * (K[3] * x + K[4] * y + K[5]) + kt[1]) / td1);
                            if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
                                const float d0 = last_depth[v0 * cols + u0];
                                if (d0 > 0 && fabsf(td1 - d0) <= max_depth_delta &&
                                    last_image[v0 * cols + u0] != 0) {
                                    c.zero_x = (int16_t)u0;
                                    c.zero_y = (int16_t)v0;
                                    c.one_x = (int16_t)x;
                                    c.one_y = (int16_t)y;
                                    c.diff = (float)next_image[y * cols + x] -
                                             (float)last_image[v0 * cols + u0];
                                    c.valid = 1;
                                    vx = 1;
                                    vy = (int)(c.diff * c.diff);
                                    if (err_map) err_map[y * cols + x] = 0.001f * vy;
                                }
                            }
                        }
                    }
                }
            }
            if (!c.valid && err_map) err_map[i * cols + j0] = 0.0f;
            corres[i * cols + j0] = c;
            sum_x += (uint32_t)vx;
            sum_y += (uint32_t)vy;
        }
    *count = (int)sum_x;
    *sigma_sum = (int)sum_y;
}

/* Core/Cuda/reduce.cu:495-578 (RGBReduction::getProducts) + :609-661 */
void orc_rgb_step(const orc_dataterm *corres, float sigma, const float *cloud, float fx, float fy,
                  const int16_t *dIdx, const int16_t *dIdy, float sobel_scale, int cols, int rows,
                  double out29[29]) {
    for (int k = 0; k < 29; ++k) out29[k] = 0;
    for (int i = 0; i < cols * rows; ++i) {
        const orc_dataterm *c = &corres[i];
        float row[7] = {0, 0, 0, 0, 0, 0, 0}, p[29];
        const int found = c->valid;
        if (found) {
            float w = sigma + fabsf(c->diff);
            w = w > FLT_EPSILON ? 1.0f / w : 1.0f;
            if (sigma == -1) w = 1;
            row[6] = -w * c->diff;
            const float *cp = cloud + (size_t)(c->zero_y * cols + c->zero_x) * 3;
            const float X = cp[0], Y = cp[1], Z = cp[2];
            const float invz = (float)(1.0 / Z);
            const float dI_dx = w * sobel_scale * dIdx[c->one_y * cols + c->one_x];
            const float dI_dy = w * sobel_scale * dIdy[c->one_y * cols + c->one_x];
            const float v0 = dI_dx * fx * invz;
            const float v1 = dI_dy * fy * invz;
            const float v2 = -(v0 * X + v1 * Y) * invz;
            row[0] = v0;
            row[1] = v1;
            row[2] = v2;
            row[3] = -Z * v1 + Y * v2;
            row[4] = Z * v0 - X * v2;
            row[5] = -Y * v0 + X * v1;
        }
        se3_products(row, found, p);
        for (int k = 0; k < 29; ++k) out29[k] += (double)p[k];
    }
}

/* ------------------------------------------------------------------------------------- */
/* SO3 pre-alignment                                                                      */
/* ------------------------------------------------------------------------------------- */

/* Core/Cuda/reduce.cu:963-979 (getGradient) */
static inline void so3_gradient(const uint8_t *img, int cols, int x, int y, float *gx, float *gy) {
    const float actu = (float)img[y * cols + x];
    float back = (float)img[y * cols + x - 1];
    float fore = (float)img[y * cols + x + 1];
    *gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(y - 1) * cols + x];
    fore = (float)img[(y + 1) * cols + x];
    *gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}

/* Core/Cuda/reduce.cu:981-1064 (SO3Reduction::getProducts) + :1092-1150 */
void orc_so3_step(const uint8_t *last_image, const uint8_t *next_image, const float B[9],
                  const float kinv[9], const float krlr[9], int cols, int rows, double out11[11]) {
    for (int k = 0; k < 11; ++k) out11[k] = 0;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const f3 unwarped = f3_make((float)x, (float)y, 1.0f);
            const f3 warped = m33_mul(B, unwarped);
            const int wx = float2int_rn(warped.x / warped.z);
            const int wy = float2int_rn(warped.y / warped.z);
            const int found = (wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 &&
                               x < cols - 1 && y >= 1 && y < rows - 1);
            float row[4] = {0, 0, 0, 0};
            if (found) {
                float gnx, gny, glx, gly;
                so3_gradient(next_image, cols, wx, wy, &gnx, &gny);
                so3_gradient(last_image, cols, x, y, &glx, &gly);
                const float gx = (gnx + glx) / 2.0f;
                const float gy = (gny + gly) / 2.0f;
                const f3 point = m33_mul(kinv, unwarped);
                const float z2 = point.z * point.z;
                const float a = krlr[0], b = krlr[1], c = krlr[2];
                const float d = krlr[3], e = krlr[4], f = krlr[5];
                const float g = krlr[6], h = krlr[7], i = krlr[8];
                f3 left;
                left.x = ((point.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2;
                left.y = ((point.z * (e * gy + b * gx)) - (gy * h * y) - (gx * h * x)) / z2;
                left.z = ((point.z * (f * gy + c * gx)) - (gy * i * y) - (gx * i * x)) / z2;
                const f3 jac = f3_cross(left, point);
                row[0] = jac.x;
                row[1] = jac.y;
                row[2] = jac.z;
                row[3] = -((float)next_image[wy * cols + wx] - (float)last_image[y * cols + x]);
            }
            /* member order of JtJJtrSO3 (types.cuh:154-162) */
            const float p[11] = {row[0] * row[0], row[0] * row[1], row[0] * row[2], row[0] * row[3],
                                 row[1] * row[1], row[1] * row[2], row[1] * row[3], row[2] * row[2],
                                 row[2] * row[3], row[3] * row[3], (float)found};
            for (int k = 0; k < 11; ++k) out11[k] += (double)p[k];
        }
}

/* Core/Cuda/reduce.cu:455-472: the device sums are float; cast, then fill the symmetric A */
void orc_unpack_se3(const double out29[29], float A[36], float b[6], float residual[2]) {
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float value = (float)out29[shift++];
            if (j == 6)
                b[i] = value;
            else
                A[j * 6 + i] = A[i * 6 + j] = value;
        }
    residual[0] = (float)out29[27];
    residual[1] = (float)out29[28];
}

/* Core/Cuda/reduce.cu:1132-1149 */
void orc_unpack_so3(const double out11[11], float A[9], float b[3], float residual[2]) {
    int shift = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            const float value = (float)out11[shift++];
            if (j == 3)
                b[i] = value;
            else
                A[j * 3 + i] = A[i * 3 + j] = value;
        }
    residual[0] = (float)out11[9];
    residual[1] = (float)out11[10];
}

/* ------------------------------------------------------------------------------------- */
/* Host algebra.  The reference uses Eigen (system package, version unpinned, absent here): */
/* ldlt().solve, .inverse(), Isometry3f products.  Restated from the published algorithms;  */
/* parity with Eigen's exact rounding is unpinned.                                          */
/* ------------------------------------------------------------------------------------- */

/* Core/Utils/OdometryProvider.h:32-67 */
void orc_rodrigues(const double src[3], double R[9]) {
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

/* Symmetric solve by LDL^T without pivoting (what `A.ldlt().solve(b)` computes for a positive
 * definite A up to rounding; RGBDOdometry.cpp:298,435-443).  A is n x n row major, n <= 6.
 * Returns 0 on success, 1 when a pivot vanished (x then holds non-finite values exactly as a
 * division by zero would produce). */
int orc_ldlt_solve(int n, const double *A, const double *b, double *x) {
    double L[36], D[6], y[6];
    int bad = 0;
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k] * D[k];
        D[j] = d;
        if (d == 0.0) bad = 1;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k] * D[k];
            L[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
        y[i] = s;
    }
    for (int i = 0; i < n; ++i) y[i] /= D[i];
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s;
    }
    return bad;
}

static void ldlt_solve3f(const float A[9], const float b[3], float x[3]) {
    /* float 3x3 variant used by the SO3 loop (RGBDOdometry.cpp:258-259,298) */
    float L[9], D[3], y[3];
    for (int j = 0; j < 3; ++j) {
        float d = A[j * 3 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 3 + k] * L[j * 3 + k] * D[k];
        D[j] = d;
        for (int i = j + 1; i < 3; ++i) {
            float s = A[i * 3 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 3 + k] * L[j * 3 + k] * D[k];
            L[i * 3 + j] = s / d;
        }
    }
    for (int i = 0; i < 3; ++i) {
        float s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * 3 + k] * y[k];
        y[i] = s;
    }
    for (int i = 0; i < 3; ++i) y[i] /= D[i];
    for (int i = 2; i >= 0; --i) {
        float s = y[i];
        for (int k = i + 1; k < 3; ++k) s -= L[k * 3 + i] * x[k];
        x[i] = s;
    }
}

/* 3x3 inverse by cofactors times 1/det (the fixed-size path behind `.inverse()`);
 * RGBDOdometry.cpp:316 (float), :261,266,352 (double) */
void orc_inverse3f(const float m[9], float inv[9]) {
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

static void inverse3d(const double m[9], double inv[9]) {
    const double c00 = m[4] * m[8] - m[5] * m[7];
    const double c01 = m[5] * m[6] - m[3] * m[8];
    const double c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
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

/* general 4x4 inverse by cofactors (RGBDOdometry.cpp:348 `resultRt.inverse()`) */
void orc_inverse4d(const double m[16], double inv[16]) {
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

static void mul3d(const double a[9], const double b[9], double c[9]) {
    double r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
            r[i * 3 + j] = s;
        }
    memcpy(c, r, sizeof(r));
}

static void mul4d(const double a[16], const double b[16], double c[16]) {
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            r[i * 4 + j] = s;
        }
    memcpy(c, r, sizeof(r));
}

static void mul3f(const float a[9], const float b[9], float c[9]) {
    float r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0;
            for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
            r[i * 3 + j] = s;
        }
    memcpy(c, r, sizeof(r));
}

/* ------------------------------------------------------------------------------------- */
/* RGBDOdometry object                                                                    */
/* ------------------------------------------------------------------------------------- */
#define NUM_PYRS 3 /* RGBDOdometry.h:72 */

struct orc_odometry {
    int width, height;
    float cx, cy, fx, fy;
    float dist_thres, angle_thres;
    float sobel_scale, max_depth_delta_rgb, max_depth_rgb;
    float min_grad[NUM_PYRS];

    float *vmaps_tmp, *nmaps_tmp; /* RGBA32F, 4*W*H */
    float *vmaps_g_prev[NUM_PYRS], *nmaps_g_prev[NUM_PYRS];
    float *vmaps_curr[NUM_PYRS], *nmaps_curr[NUM_PYRS];
    float *last_depth[NUM_PYRS], *next_depth[NUM_PYRS];
    uint8_t *last_image[NUM_PYRS], *next_image[NUM_PYRS], *last_next_image[NUM_PYRS];
    int16_t *dIdx[NUM_PYRS], *dIdy[NUM_PYRS];
    float *cloud[NUM_PYRS];
    orc_dataterm *corres[NUM_PYRS];
    float *depth_pyr[NUM_PYRS];

    orc_odom_stats st;
};

static void level_intr(const orc_odometry *o, int level, float *fx, float *fy, float *cx, float *cy) {
    /* types.cuh:94-98 */
    const int div = 1 << level;
    *fx = o->fx / div;
    *fy = o->fy / div;
    *cx = o->cx / div;
    *cy = o->cy / div;
}

/* RGBDOdometry.cpp:21-106 */
orc_odometry *orc_odom_create(int width, int height, float cx, float cy, float fx, float fy,
                              float dist_thresh, float angle_thresh) {
    orc_odometry *o = (orc_odometry *)calloc(1, sizeof(*o));
    o->width = width;
    o->height = height;
    o->cx = cx;
    o->cy = cy;
    o->fx = fx;
    o->fy = fy;
    o->dist_thres = dist_thresh;
    o->angle_thres = angle_thresh;
    o->sobel_scale = (float)(1.0 / pow(2.0, 3)); /* :31-32 */
    o->max_depth_delta_rgb = 0.07f;              /* :33 */
    o->max_depth_rgb = 6.0f;                     /* :34 */
    o->min_grad[0] = 5;                          /* :103-105 */
    o->min_grad[1] = 3;
    o->min_grad[2] = 1;
    o->st.lastICPCount = o->st.lastRGBCount = o->st.lastSO3Count = (float)(width * height);
    o->vmaps_tmp = (float *)calloc((size_t)4 * width * height, sizeof(float));
    o->nmaps_tmp = (float *)calloc((size_t)4 * width * height, sizeof(float));
    for (int i = 0; i < NUM_PYRS; ++i) {
        const size_t n = (size_t)(width >> i) * (height >> i);
        o->vmaps_g_prev[i] = (float *)calloc(3 * n, sizeof(float));
        o->nmaps_g_prev[i] = (float *)calloc(3 * n, sizeof(float));
        o->vmaps_curr[i] = (float *)calloc(3 * n, sizeof(float));
        o->nmaps_curr[i] = (float *)calloc(3 * n, sizeof(float));
        o->last_depth[i] = (float *)calloc(n, sizeof(float));
        o->next_depth[i] = (float *)calloc(n, sizeof(float));
        o->depth_pyr[i] = (float *)calloc(n, sizeof(float));
        o->last_image[i] = (uint8_t *)calloc(n, 1);
        o->next_image[i] = (uint8_t *)calloc(n, 1);
        o->last_next_image[i] = (uint8_t *)calloc(n, 1);
        o->dIdx[i] = (int16_t *)calloc(n, sizeof(int16_t));
        o->dIdy[i] = (int16_t *)calloc(n, sizeof(int16_t));
        o->cloud[i] = (float *)calloc(3 * n, sizeof(float));
        o->corres[i] = (orc_dataterm *)calloc(n, sizeof(orc_dataterm));
    }
    return o;
}

void orc_odom_destroy(orc_odometry *o) {
    if (!o) return;
    free(o->vmaps_tmp);
    free(o->nmaps_tmp);
    for (int i = 0; i < NUM_PYRS; ++i) {
        free(o->vmaps_g_prev[i]);
        free(o->nmaps_g_prev[i]);
        free(o->vmaps_curr[i]);
        free(o->nmaps_curr[i]);
        free(o->last_depth[i]);
        free(o->next_depth[i]);
        free(o->depth_pyr[i]);
        free(o->last_image[i]);
        free(o->next_image[i]);
        free(o->last_next_image[i]);
        free(o->dIdx[i]);
        free(o->dIdy[i]);
        free(o->cloud[i]);
        free(o->corres[i]);
    }
    free(o);
}

/* Model::generateCUDATextures (Core/Model/Model.cpp:359-388: depth pyramid by pyrDownGaussF)
 * followed by RGBDOdometry::initICP (RGBDOdometry.cpp:110-118). */
void orc_odom_init_icp(orc_odometry *o, const float *depth_l0, float depth_cutoff) {
    memcpy(o->depth_pyr[0], depth_l0, sizeof(float) * o->width * o->height);
    for (int i = 1; i < NUM_PYRS; ++i)
        orc_pyrdown_gauss_f(o->depth_pyr[i - 1], o->width >> (i - 1), o->height >> (i - 1),
                            o->depth_pyr[i]);
    for (int i = 0; i < NUM_PYRS; ++i) {
        float fx, fy, cx, cy;
        level_intr(o, i, &fx, &fy, &cx, &cy);
        orc_create_vmap(o->depth_pyr[i], o->width >> i, o->height >> i, fx, fy, cx, cy, depth_cutoff,
                        o->vmaps_curr[i]);
        orc_create_nmap(o->vmaps_curr[i], o->width >> i, o->height >> i, o->nmaps_curr[i]);
    }
}

/* RGBDOdometry.cpp:120-141 */
void orc_odom_init_icp_from_prediction(orc_odometry *o, const float *vert_rgba,
                                       const float *norm_rgba) {
    const size_t n4 = (size_t)4 * o->width * o->height;
    memcpy(o->vmaps_tmp, vert_rgba, n4 * sizeof(float));
    memcpy(o->nmaps_tmp, norm_rgba, n4 * sizeof(float));
    orc_copy_maps(o->vmaps_tmp, o->nmaps_tmp, o->width, o->height, o->vmaps_curr[0], o->nmaps_curr[0]);
    for (int i = 1; i < NUM_PYRS; ++i) {
        orc_resize_map(o->vmaps_curr[i - 1], o->width >> (i - 1), o->height >> (i - 1), 0,
                       o->vmaps_curr[i]);
        orc_resize_map(o->nmaps_curr[i - 1], o->width >> (i - 1), o->height >> (i - 1), 1,
                       o->nmaps_curr[i]);
    }
}

/* RGBDOdometry.cpp:143-175 */
void orc_odom_init_icp_model(orc_odometry *o, const float *vert_rgba, const float *norm_rgba,
                             const float pose[16]) {
    const size_t n4 = (size_t)4 * o->width * o->height;
    memcpy(o->vmaps_tmp, vert_rgba, n4 * sizeof(float));
    memcpy(o->nmaps_tmp, norm_rgba, n4 * sizeof(float));
    orc_copy_maps(o->vmaps_tmp, o->nmaps_tmp, o->width, o->height, o->vmaps_g_prev[0],
                  o->nmaps_g_prev[0]);
    for (int i = 1; i < NUM_PYRS; ++i) {
        orc_resize_map(o->vmaps_g_prev[i - 1], o->width >> (i - 1), o->height >> (i - 1), 0,
                       o->vmaps_g_prev[i]);
        orc_resize_map(o->nmaps_g_prev[i - 1], o->width >> (i - 1), o->height >> (i - 1), 1,
                       o->nmaps_g_prev[i]);
    }
    const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    const float t[3] = {pose[3], pose[7], pose[11]};
    for (int i = 0; i < NUM_PYRS; ++i)
        orc_transform_maps(o->vmaps_g_prev[i], o->nmaps_g_prev[i], o->width >> i, o->height >> i, R, t,
                           o->vmaps_g_prev[i], o->nmaps_g_prev[i]);
}

/* RGBDOdometry.cpp:177-194 (populateRGBDData).  Quirk kept: the depth comes from vmaps_tmp,
 * i.e. from whatever initICPModel / initICP(prediction) copied last (:197,202 NOTE).  The mask
 * pyramids of the reference are never read (MASK_RGB_RESIDUAL undefined) and are omitted. */
static void populate_rgbd(orc_odometry *o, const uint8_t *rgb, int channels, float **depths,
                          uint8_t **images) {
    orc_vertices_to_depth(o->vmaps_tmp, o->width, o->height, o->max_depth_rgb, depths[0]);
    for (int i = 0; i + 1 < NUM_PYRS; ++i)
        orc_pyrdown_gauss_f(depths[i], o->width >> i, o->height >> i, depths[i + 1]);
    orc_image_to_intensity(rgb, channels, o->width, o->height, images[0]);
    for (int i = 0; i + 1 < NUM_PYRS; ++i)
        orc_pyrdown_uchar_gauss(images[i], o->width >> i, o->height >> i, images[i + 1]);
}

void orc_odom_init_rgb(orc_odometry *o, const uint8_t *rgb, int channels) {
    populate_rgbd(o, rgb, channels, o->next_depth, o->next_image); /* :201-204 */
}
void orc_odom_init_rgb_model(orc_odometry *o, const uint8_t *rgb, int channels) {
    populate_rgbd(o, rgb, channels, o->last_depth, o->last_image); /* :196-199 */
}
/* RGBDOdometry.cpp:206-215 */
void orc_odom_init_first_rgb(orc_odometry *o, const uint8_t *rgb, int channels) {
    orc_image_to_intensity(rgb, channels, o->width, o->height, o->last_next_image[0]);
    for (int i = 0; i + 1 < NUM_PYRS; ++i)
        orc_pyrdown_uchar_gauss(o->last_next_image[i], o->width >> i, o->height >> i,
                                o->last_next_image[i + 1]);
}

static void k_matrix(const orc_odometry *o, int level, double K[9]) {
    float fx, fy, cx, cy;
    level_intr(o, level, &fx, &fy, &cx, &cy);
    memset(K, 0, 9 * sizeof(double));
    K[0] = fx;
    K[4] = fy;
    K[2] = cx;
    K[5] = cy;
    K[8] = 1;
}

static void d9_to_f9(const double *d, float *f) {
    for (int k = 0; k < 9; ++k) f[k] = (float)d[k];
}

/* RGBDOdometry.cpp:217-477 */
void orc_odom_get_incremental_transformation(orc_odometry *o, float trans[3], float rot[9],
                                             int rgb_only, float icp_weight, int pyramid,
                                             int fast_odom, int so3, float *icp_err,
                                             float *rgb_err) {
    const int icp = !rgb_only && icp_weight > 0;
    const int rgb = rgb_only || icp_weight < 100;

    float Rprev[9], tprev[3], Rcurr[9], tcurr[3];
    memcpy(Rprev, rot, sizeof(Rprev));
    memcpy(tprev, trans, sizeof(tprev));
    memcpy(Rcurr, Rprev, sizeof(Rcurr));
    memcpy(tcurr, tprev, sizeof(tcurr));
    o->st.iterations_run = 0;
    o->st.so3_iterations_run = 0;

    if (rgb)
        for (int i = 0; i < NUM_PYRS; ++i) /* :230-235 */
            orc_derivative_images(o->next_image[i], o->width >> i, o->height >> i, o->dIdx[i],
                                  o->dIdy[i]);

    double resultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

    if (so3) { /* :239-310 */
        const int lvl = 2;
        float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double K[9], K_inv[9];
        k_matrix(o, lvl, K);
        inverse3d(K, K_inv);
        float lastError = FLT_MAX / 2, lastCount = FLT_MAX / 2;
        double lastResultR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int i = 0; i < 10; ++i) {
            double tmp[9], H[9], KR[9];
            float Hf[9], Kinvf[9], KRf[9];
            mul3d(K, resultR, tmp);
            mul3d(tmp, K_inv, H);
            d9_to_f9(H, Hf);
            d9_to_f9(K_inv, Kinvf);
            mul3d(K, resultR, KR);
            d9_to_f9(KR, KRf);
            double out11[11];
            float jtj[9], jtr[3], residual[2];
            orc_so3_step(o->last_next_image[lvl], o->next_image[lvl], Hf, Kinvf, KRf, o->width >> lvl,
                         o->height >> lvl, out11);
            orc_unpack_so3(out11, jtj, jtr, residual);
            o->st.so3_iterations_run++;
            o->st.lastSO3Error = sqrtf(residual[0]) / residual[1];
            o->st.lastSO3Count = residual[1];
            if (o->st.lastSO3Error < lastError && fabsf(lastError - o->st.lastSO3Count) < 0.001) {
                break; /* :285 (compares against the COUNT, as the reference does) */
            } else if (o->st.lastSO3Error > lastError + 0.001) {
                o->st.lastSO3Error = lastError;
                o->st.lastSO3Count = lastCount;
                memcpy(resultR, lastResultR, sizeof(resultR));
                break;
            }
            lastError = o->st.lastSO3Error;
            lastCount = o->st.lastSO3Count;
            memcpy(lastResultR, resultR, sizeof(resultR));
            float delta[3];
            ldlt_solve3f(jtj, jtr, delta);
            const double dd[3] = {delta[0], delta[1], delta[2]};
            double rotUpdate[9];
            float rotUpdatef[9];
            orc_rodrigues(dd, rotUpdate);
            d9_to_f9(rotUpdate, rotUpdatef);
            mul3f(rotUpdatef, R_lr, R_lr);
            for (int k = 0; k < 9; ++k) resultR[k] = R_lr[k];
        }
    }

    const int iterations[NUM_PYRS] = {fast_odom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0};

    float Rprev_inv[9];
    orc_inverse3f(Rprev, Rprev_inv);

    double resultRt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    if (so3)
        for (int x = 0; x < 3; ++x)
            for (int y = 0; y < 3; ++y) resultRt[x * 4 + y] = resultR[x * 3 + y];

    /* diagnostics (tests / tools only): ORC_TRACE_GN=1 prints every iteration's residual statistics,
     * ORC_MAX_GN_ITERS=n stops after n solved iterations */
    const int trace_gn = getenv("ORC_TRACE_GN") != NULL;
    const int max_gn_iters = getenv("ORC_MAX_GN_ITERS") ? atoi(getenv("ORC_MAX_GN_ITERS")) : -1;
    for (int i = NUM_PYRS - 1; i >= 0; --i) {
        const int cols = o->width >> i, rows = o->height >> i;
        float fx, fy, cx, cy;
        level_intr(o, i, &fx, &fy, &cx, &cy);
        if (rgb) orc_project_to_cloud(o->last_depth[i], cols, rows, fx, fy, cx, cy, o->cloud[i]);
        double K[9], K_inv[9];
        k_matrix(o, i, K);
        inverse3d(K, K_inv);
        o->st.lastRGBError = FLT_MAX;

        for (int j = 0; j < iterations[i]; ++j) {
            double Rt[16];
            orc_inverse4d(resultRt, Rt);
            const double R[9] = {Rt[0], Rt[1], Rt[2], Rt[4], Rt[5], Rt[6], Rt[8], Rt[9], Rt[10]};
            double tmp[9], KRK_inv[9];
            mul3d(K, R, tmp);
            mul3d(tmp, K_inv, KRK_inv);
            float krkInv[9];
            d9_to_f9(KRK_inv, krkInv);
            const double t3[3] = {Rt[3], Rt[7], Rt[11]};
            float kt[3];
            for (int r = 0; r < 3; ++r)
                kt[r] = (float)(K[r * 3 + 0] * t3[0] + K[r * 3 + 1] * t3[1] + K[r * 3 + 2] * t3[2]);

            int sigma = 0, rgbSize = 0;
            if (rgb) {
                const float min_scale =
                    (float)(pow(o->min_grad[i], 2.0) / pow(o->sobel_scale, 2.0)); /* :365 */
                orc_rgb_residual(min_scale, o->dIdx[i], o->dIdy[i], o->last_depth[i], o->next_depth[i],
                                 o->last_image[i], o->next_image[i], o->corres[i],
                                 o->max_depth_delta_rgb, kt, krkInv, cols, rows, &sigma, &rgbSize,
                                 (i == 0 && j == iterations[i] - 1) ? rgb_err : NULL);
            }
            const float tmpError = (float)(sqrt((double)sigma) / rgbSize); /* :373 */
            float sigmaVal = (tmpError == 0) ? 1 : (float)rgbSize;
            if (trace_gn)
                fprintf(stderr, "[orc gn] level %d iteration %d: count %d sigma %d error %.9g last %.9g%s\n", i, j, rgbSize, sigma,
                        (double)tmpError, (double)o->st.lastRGBError, (rgb_only && tmpError > o->st.lastRGBError) ? " -> break" : "");
            if (max_gn_iters >= 0 && o->st.iterations_run >= max_gn_iters) goto done_levels;
            if (rgb_only && tmpError > o->st.lastRGBError) break;
            o->st.lastRGBError = tmpError;
            o->st.lastRGBCount = (float)rgbSize;
            if (rgb_only) sigmaVal = -1;

            float A_icp[36], b_icp[6], residual[2] = {0, 0};
            memset(A_icp, 0, sizeof(A_icp));
            memset(b_icp, 0, sizeof(b_icp));
            if (icp) {
                double out29[29];
                orc_icp_step(Rcurr, tcurr, o->vmaps_curr[i], o->nmaps_curr[i], Rprev_inv, tprev, fx, fy,
                             cx, cy, o->vmaps_g_prev[i], o->nmaps_g_prev[i], o->dist_thres,
                             o->angle_thres, cols, rows, out29,
                             (i == 0 && j == iterations[i] - 1) ? icp_err : NULL);
                orc_unpack_se3(out29, A_icp, b_icp, residual);
                /* :412-413; when !icp the reference reads an uninitialised residual[] here --
                 * the members are left untouched in that case. */
                o->st.lastICPError = sqrtf(residual[0]) / residual[1];
                o->st.lastICPCount = residual[1];
            }

            float A_rgb[36], b_rgb[6], r2[2];
            memset(A_rgb, 0, sizeof(A_rgb));
            memset(b_rgb, 0, sizeof(b_rgb));
            if (rgb) {
                double out29[29];
                orc_rgb_step(o->corres[i], sigmaVal, o->cloud[i], fx, fy, o->dIdx[i], o->dIdy[i],
                             o->sobel_scale, cols, rows, out29);
                orc_unpack_se3(out29, A_rgb, b_rgb, r2);
            }

            double result[6];
            if (icp && rgb) { /* :431-435 */
                const double w = icp_weight;
                for (int k = 0; k < 36; ++k) o->st.lastA[k] = (double)A_rgb[k] + w * w * (double)A_icp[k];
                for (int k = 0; k < 6; ++k) o->st.lastb[k] = (double)b_rgb[k] + w * (double)b_icp[k];
            } else if (icp) {
                for (int k = 0; k < 36; ++k) o->st.lastA[k] = A_icp[k];
                for (int k = 0; k < 6; ++k) o->st.lastb[k] = b_icp[k];
            } else {
                for (int k = 0; k < 36; ++k) o->st.lastA[k] = A_rgb[k];
                for (int k = 0; k < 6; ++k) o->st.lastb[k] = b_rgb[k];
            }
            orc_ldlt_solve(6, o->st.lastA, o->st.lastb, result);
            o->st.iterations_run++;

            /* OdometryProvider.h:69-89 (computeUpdateSE3) */
            double upd[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, Rup[9];
            const double rvec[3] = {result[3], result[4], result[5]};
            orc_rodrigues(rvec, Rup);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) upd[r * 4 + c] = Rup[r * 3 + c];
                upd[r * 4 + 3] = result[r];
            }
            mul4d(upd, resultRt, resultRt);
            float Ro[9], to[3];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) Ro[r * 3 + c] = (float)resultRt[r * 4 + c];
                to[r] = (float)resultRt[r * 4 + 3];
            }
            /* RGBDOdometry.cpp:452-460: currentT = [Rprev|tprev] * rgbOdom.inverse();
             * an isometry inverse is (R^T, -R^T t). */
            float RoT[9], ti[3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) RoT[r * 3 + c] = Ro[c * 3 + r];
            for (int r = 0; r < 3; ++r)
                ti[r] = -RoT[r * 3 + 0] * to[0] + -RoT[r * 3 + 1] * to[1] + -RoT[r * 3 + 2] * to[2];
            mul3f(Rprev, RoT, Rcurr);
            for (int r = 0; r < 3; ++r) {
                float s = 0;
                for (int k = 0; k < 3; ++k) s += Rprev[r * 3 + k] * ti[k];
                tcurr[r] = s + tprev[r];
            }
        }
    }

done_levels:
    if (rgb) { /* :464-467 */
        const float dx = tcurr[0] - tprev[0], dy = tcurr[1] - tprev[1], dz = tcurr[2] - tprev[2];
        if (sqrtf(dx * dx + dy * dy + dz * dz) > 0.3) {
            memcpy(Rcurr, Rprev, sizeof(Rcurr));
            memcpy(tcurr, tprev, sizeof(tcurr));
        }
    }
    if (so3) /* :469-473 */
        for (int i = 0; i < NUM_PYRS; ++i) {
            uint8_t *t = o->last_next_image[i];
            o->last_next_image[i] = o->next_image[i];
            o->next_image[i] = t;
        }
    memcpy(trans, tcurr, sizeof(tcurr));
    memcpy(rot, Rcurr, sizeof(Rcurr));
}

void orc_odom_get_stats(const orc_odometry *o, orc_odom_stats *s) { *s = o->st; }

const float *orc_odom_buffer_f32(const orc_odometry *o, const char *name, int l) {
    if (!strcmp(name, "vmaps_curr")) return o->vmaps_curr[l];
    if (!strcmp(name, "nmaps_curr")) return o->nmaps_curr[l];
    if (!strcmp(name, "vmaps_g_prev")) return o->vmaps_g_prev[l];
    if (!strcmp(name, "nmaps_g_prev")) return o->nmaps_g_prev[l];
    if (!strcmp(name, "last_depth")) return o->last_depth[l];
    if (!strcmp(name, "next_depth")) return o->next_depth[l];
    if (!strcmp(name, "depth_pyr")) return o->depth_pyr[l];
    if (!strcmp(name, "cloud")) return o->cloud[l];
    return NULL;
}
const uint8_t *orc_odom_buffer_u8(const orc_odometry *o, const char *name, int l) {
    if (!strcmp(name, "last_image")) return o->last_image[l];
    if (!strcmp(name, "next_image")) return o->next_image[l];
    if (!strcmp(name, "last_next_image")) return o->last_next_image[l];
    return NULL;
}
const int16_t *orc_odom_buffer_i16(const orc_odometry *o, const char *name, int l) {
    if (!strcmp(name, "dIdx")) return o->dIdx[l];
    if (!strcmp(name, "dIdy")) return o->dIdy[l];
    return NULL;
}
