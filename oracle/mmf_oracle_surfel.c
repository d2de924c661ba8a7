/*
 * mmf_oracle_surfel.c -- CPU restatement of the reference's surfel path: index map, splat
 * prediction, data association + fusion, clean-up/compaction, first-frame initialisation,
 * bilateral depth filter and fill-in.  TEST INFRASTRUCTURE ONLY (see mmf_oracle.h).
 *
 * PARITY UNPINNED.  The reference runs this path as GLSL 3.30 through OpenGL/Pangolin; parts
 * of its behaviour are fixed-function GL state that is not in the tree.  Every such point is
 * written down as an ASSUMPTION below and is shared, bit for bit, by the HIP kernels:
 *   A1 point rasterisation: a size-1 GL point covers the pixel (floor(xw), floor(yw)); a point
 *      sprite of size s covers the pixels whose centres c satisfy xw-s/2 <= c < xw+s/2; points
 *      are clipped by their centre against -1 <= x,y,z <= 1; window coordinates are
 *      xw = (xn + 1) * cols/2, zw = 0.5 zn + 0.5, all in float32.
 *   A2 depth buffer: 24-bit unorm (Pangolin GlRenderBuffer default GL_DEPTH_COMPONENT24),
 *      d = (uint)(zw * 16777215 + 0.5) after clamping zw to [0,1]; GL_LESS with draw-order
 *      tie-break == minimum of (d << 32 | vertexId).
 *   A3 textures: NEAREST filtering, CLAMP_TO_EDGE, texel = floor(coord * size); RGBA8 colour
 *      reads return byte/255.
 *   A4 GLSL built-ins are taken as their C float equivalents (expf, acosf, sqrtf, roundf);
 *      normalize(v) = v * (1/sqrt(dot(v,v))); mat*vec sums left to right; no contraction.
 *   A5 gl_PointSize below 1 is clamped to 1.
 * Each function cites the shader / host lines it follows.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mmf_oracle.h"
#include "../include/mmf_math.h" /* shared bit-exact expf (see its header) */

/* -DORC_LIBM_EXP builds the checker with the C library's expf instead: the independent measure of how much the
 * shared definition matters (tests/test_oracle_surfel.py::test_libm_exp_variant) */
#ifdef ORC_LIBM_EXP
#define ORC_EXPF expf
#else
#define ORC_EXPF mmf_expf
#endif

typedef struct {
    float x, y, z;
} v3;
static inline v3 V3(float x, float y, float z) {
    v3 r = {x, y, z};
    return r;
}
static inline v3 v3add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 v3normalize(v3 a) { return v3scale(a, 1.0f / sqrtf(v3dot(a, a))); }
static inline float v3length(v3 a) { return sqrtf(v3dot(a, a)); }
/* row-major 4x4 * (p,1) and its upper-left 3x3 * n */
static inline v3 m4point(const float *m, v3 p) {
    return V3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
              m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
static inline v3 m4dir(const float *m, v3 n) {
    return V3(m[0] * n.x + m[1] * n.y + m[2] * n.z, m[4] * n.x + m[5] * n.y + m[6] * n.z,
              m[8] * n.x + m[9] * n.y + m[10] * n.z);
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int texel(float coord, int size) { return clampi((int)floorf(coord * (float)size), 0, size - 1); } /* A3 */

/* general 4x4 float inverse by cofactors (Eigen `pose.inverse()`, ModelProjection.cpp:108) */
void orc_inverse4f(const float m[16], float inv[16]) {
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

/* Shaders/color_encoding.glsl */
static inline float encode_color(float r, float g, float b) {
    int rgb = (int)roundf(r * 255.0f);
    rgb = (rgb << 8) + (int)roundf(g * 255.0f);
    rgb = (rgb << 8) + (int)roundf(b * 255.0f);
    return (float)rgb;
}
static inline v3 decode_color(float c) {
    const int ci = (int)c;
    return V3((float)(ci >> 16 & 0xFF) / 255.0f, (float)(ci >> 8 & 0xFF) / 255.0f, (float)(ci & 0xFF) / 255.0f);
}
/* Shaders/surfels.glsl:19-34 with cam = (cx, cy, 1/fx, 1/fy) */
static inline float get_radius(float depth, float norm_z, float inv_fx, float inv_fy) {
    const float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
    const float sqrt2 = 1.41421356237f;
    const float radius = (depth / meanFocal) * sqrt2;
    float radius_n = radius;
    radius_n = radius_n / fabsf(norm_z);
    radius_n = fminf(2.0f * radius, radius_n);
    return radius_n;
}
/* Shaders/surfels.glsl:36-46 */
static inline float confidence(float x, float y, float cx, float cy, float weighting) {
    const float maxRadDist = 400, twoSigmaSquared = 0.72f;
    const float px = x - cx, py = y - cy;
    const float radialDist = sqrtf(px * px + py * py) / maxRadDist;
    return ORC_EXPF((-(radialDist * radialDist) / twoSigmaSquared)) * weighting;
}
/* 24-bit depth key (A2) */
static inline uint32_t depth24(float zw) {
    if (!(zw >= 0.f)) zw = 0.f; /* also catches NaN */
    if (zw > 1.f) zw = 1.f;
    return (uint32_t)(zw * 16777215.0f + 0.5f);
}

/* texture coordinate of pixel centre i as the host builds it (Model.cpp:206-210,
 * FeedbackBuffer.cpp:44-47): float division plus a DOUBLE half-texel, rounded to float */
static inline float uv_coord(int i, int n) { return (float)((double)((float)i / (float)n) + 1.0 / (2 * (double)(float)n)); }

/* Shaders/geometry.glsl:22-40 -- vertex / central-difference normal from a depth texture */
static inline v3 get_vertex(const float *depth, int cols, int rows, float tx, float ty, float x, float y, float cx,
                            float cy, float ifx, float ify) {
    const float z = depth[texel(ty, rows) * cols + texel(tx, cols)];
    return V3((x - cx) * z * ifx, (y - cy) * z * ify, z);
}
static inline v3 get_normal(const float *depth, int cols, int rows, v3 p, float tx, float ty, float x, float y,
                            float cx, float cy, float ifx, float ify) {
    const v3 xf = get_vertex(depth, cols, rows, tx + (1.0f / cols), ty, x + 1, y, cx, cy, ifx, ify);
    const v3 xb = get_vertex(depth, cols, rows, tx - (1.0f / cols), ty, x - 1, y, cx, cy, ifx, ify);
    const v3 yf = get_vertex(depth, cols, rows, tx, ty + (1.0f / rows), x, y + 1, cx, cy, ifx, ify);
    const v3 yb = get_vertex(depth, cols, rows, tx, ty - (1.0f / rows), x, y - 1, cx, cy, ifx, ify);
    const v3 del_x = v3sub(v3scale(v3add(xb, p), 0.5f), v3scale(v3add(xf, p), 0.5f));
    const v3 del_y = v3sub(v3scale(v3add(yb, p), 0.5f), v3scale(v3add(yf, p), 0.5f));
    return v3normalize(v3cross(del_x, del_y));
}

/* ------------------------------------------------------------------------------------- */
/* MultiMotionFusion::filterDepth + Shaders/depth_bilateral_metric.frag:30-76             */
/* ------------------------------------------------------------------------------------- */
void orc_bilateral_filter(const float *depth, int cols, int rows, float maxD, float *out) {
    const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 555.556f;
    const int R = 6, D = R * 2 + 1;
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float value = depth[y * cols + x];
            if (value > maxD || value < 0.3f) {
                out[y * cols + x] = 0;
                continue;
            }
            const int tx = x - D / 2 + D < cols ? x - D / 2 + D : cols;
            const int ty = y - D / 2 + D < rows ? y - D / 2 + D : rows;
            float sum1 = 0, sum2 = 0;
            for (int cy = (y - D / 2 > 0 ? y - D / 2 : 0); cy < ty; ++cy)
                for (int cx = (x - D / 2 > 0 ? x - D / 2 : 0); cx < tx; ++cx) {
                    const float tmp = depth[cy * cols + cx];
                    const float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) +
                                         ((float)y - (float)cy) * ((float)y - (float)cy);
                    const float color2 = (value - tmp) * (value - tmp);
                    const float weight = ORC_EXPF(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
                    sum1 += tmp * weight;
                    sum2 += weight;
                }
            out[y * cols + x] = sum1 / sum2;
        }
}

/* ------------------------------------------------------------------------------------- */
/* First frame: FeedbackBuffer::compute x2 + Model::initialise                             */
/* (Shaders/vertex_feedback.vert/.geom, init_unstable.vert, Model.cpp:267-312).           */
/* Quirk kept: positions/colours come from the k-th valid RAW pixel, normals/radii from    */
/* the k-th valid FILTERED pixel, count = number of valid raw pixels.                      */
/* Pixels are visited column-major (FeedbackBuffer.cpp:41-49).                             */
/* ------------------------------------------------------------------------------------- */
static int feedback_pass(const uint8_t *rgb, const float *depth, int cols, int rows, float cx, float cy, float ifx,
                         float ify, int time, float maxDepth, orc_surfel *out) {
    int n = 0;
    for (int i = 0; i < cols; ++i)
        for (int j = 0; j < rows; ++j) {
            const float tx = uv_coord(i, cols), ty = uv_coord(j, rows);
            const float x = tx * cols, y = ty * rows;
            const v3 p = get_vertex(depth, cols, rows, tx, ty, x, y, cx, cy, ifx, ify);
            const v3 nl = get_normal(depth, cols, rows, p, tx, ty, x, y, cx, cy, ifx, ify);
            if (p.z <= 0 || p.z > maxDepth) continue; /* zVal == 0 => not emitted */
            const uint8_t *c = rgb + (size_t)(texel(ty, rows) * cols + texel(tx, cols)) * 3;
            orc_surfel s;
            s.pos[0] = p.x, s.pos[1] = p.y, s.pos[2] = p.z;
            s.pos[3] = confidence(x, y, cx, cy, 1.0f);
            s.col[0] = encode_color(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f);
            s.col[1] = 0;
            s.col[2] = 1.0f; /* sampled alpha of an RGB upload; overwritten by init_unstable */
            s.col[3] = (float)time;
            s.nrm[0] = nl.x, s.nrm[1] = nl.y, s.nrm[2] = nl.z;
            s.nrm[3] = get_radius(p.z, nl.z, ifx, ify);
            out[n++] = s;
        }
    return n;
}

int orc_surfel_initialise(const uint8_t *rgb, const float *depth_raw, const float *depth_filtered, int cols, int rows,
                          float cx, float cy, float fx, float fy, int time, float maxDepth, orc_surfel *out) {
    const float ifx = 1.0f / fx, ify = 1.0f / fy;
    orc_surfel *raw = (orc_surfel *)calloc((size_t)cols * rows, sizeof(orc_surfel));
    orc_surfel *fil = (orc_surfel *)calloc((size_t)cols * rows, sizeof(orc_surfel));
    const int nraw = feedback_pass(rgb, depth_raw, cols, rows, cx, cy, ifx, ify, time, maxDepth, raw);
    feedback_pass(rgb, depth_filtered, cols, rows, cx, cy, ifx, ify, time, maxDepth, fil);
    for (int k = 0; k < nraw; ++k) {
        out[k] = raw[k];
        out[k].col[1] = 0; /* init_unstable.vert */
        out[k].col[2] = 1;
        memcpy(out[k].nrm, fil[k].nrm, sizeof(fil[k].nrm));
    }
    free(raw);
    free(fil);
    return nraw;
}

/* ------------------------------------------------------------------------------------- */
/* ModelProjection::predictIndices + Shaders/index_map.vert/.frag                          */
/* ------------------------------------------------------------------------------------- */
void orc_predict_indices(const orc_surfel *s, int count, const float pose[16], float cx, float cy, float fx, float fy,
                         int cols, int rows, float maxDepth, int time, int timeDelta, uint32_t *index,
                         float *vertConf, float *colorTime, float *normRad) {
    float t_inv[16];
    orc_inverse4f(pose, t_inv);
    const size_t n = (size_t)cols * rows;
    uint64_t *key = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) key[i] = ~0ull;
    for (int id = 0; id < count; ++id) {
        const v3 h = m4point(t_inv, V3(s[id].pos[0], s[id].pos[1], s[id].pos[2]));
        if (h.z > maxDepth || h.z < 0 || (float)time - s[id].col[3] > (float)timeDelta) continue;
        const float xn = ((((fx * h.x) / h.z) + cx) - (cols * 0.5f)) / (cols * 0.5f);
        const float yn = ((((fy * h.y) / h.z) + cy) - (rows * 0.5f)) / (rows * 0.5f);
        const float zn = h.z / maxDepth;
        if (!(xn >= -1 && xn <= 1 && yn >= -1 && yn <= 1 && zn >= -1 && zn <= 1)) continue; /* A1 clip */
        const float xw = (xn + 1.0f) * (cols * 0.5f), yw = (yn + 1.0f) * (rows * 0.5f);
        const int px = (int)floorf(xw), py = (int)floorf(yw);
        if (px < 0 || py < 0 || px >= cols || py >= rows) continue;
        const uint64_t k = ((uint64_t)depth24(0.5f * zn + 0.5f) << 32) | (uint32_t)id;
        if (k < key[py * cols + px]) key[py * cols + px] = k;
    }
    for (size_t i = 0; i < n; ++i) {
        float *vc = vertConf + 4 * i, *ct = colorTime + 4 * i, *nr = normRad + 4 * i;
        if (key[i] == ~0ull) { /* glClear(0,0,0,0) */
            index[i] = 0;
            memset(vc, 0, 16), memset(ct, 0, 16), memset(nr, 0, 16);
            continue;
        }
        const uint32_t id = (uint32_t)key[i];
        const v3 h = m4point(t_inv, V3(s[id].pos[0], s[id].pos[1], s[id].pos[2]));
        const v3 nn = v3normalize(m4dir(t_inv, V3(s[id].nrm[0], s[id].nrm[1], s[id].nrm[2])));
        index[i] = id; /* vertexId 0 aliases "empty" -- kept (index_map.vert:49, data.vert:142) */
        vc[0] = h.x, vc[1] = h.y, vc[2] = h.z, vc[3] = s[id].pos[3];
        memcpy(ct, s[id].col, 16);
        nr[0] = nn.x, nr[1] = nn.y, nr[2] = nn.z, nr[3] = s[id].nrm[3];
    }
    free(key);
}

/* ------------------------------------------------------------------------------------- */
/* ModelProjection::combinedPredict + Shaders/splat.vert + combo_splat.frag                */
/* ------------------------------------------------------------------------------------- */
static inline v3 project_image(v3 p, float cx, float cy, float fx, float fy) {
    return V3(((fx * p.x) / p.z) + cx, ((fy * p.y) / p.z) + cy, p.z);
}

/* The splat.vert point sprite + one of the two fragment shaders that share it: combo_splat.frag
 * (image_rgba / vertexConf / normalRadius / time_out) or depth_splat.frag (depth_out). */
static void splat_render(const orc_surfel *s, int count, const float pose[16], float cx, float cy, float fx, float fy,
                         int cols, int rows, float maxDepth, float confThreshold, int time, int maxTime, int timeDelta,
                         uint8_t *image_rgba, float *vertexConf, float *normalRadius, uint16_t *time_out,
                         float *depth_out) {
    float t_inv[16];
    orc_inverse4f(pose, t_inv);
    const size_t n = (size_t)cols * rows;
    uint64_t *key = (uint64_t *)malloc(n * sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) key[i] = ~0ull;
    for (int pass = 0; pass < 2; ++pass) {
        /* pass 0: depth-tested winner per pixel; pass 1: attributes of the winners */
        for (int id = 0; id < count; ++id) {
            const v3 h = m4point(t_inv, V3(s[id].pos[0], s[id].pos[1], s[id].pos[2]));
            if (h.z > maxDepth || h.z < 0 || s[id].pos[3] < confThreshold ||
                (float)time - s[id].col[3] > (float)timeDelta || s[id].col[3] > (float)maxTime)
                continue;
            const float xn = ((((fx * h.x) / h.z) + cx) - (cols * 0.5f)) / (cols * 0.5f);
            const float yn = ((((fy * h.y) / h.z) + cy) - (rows * 0.5f)) / (rows * 0.5f);
            const float zn = h.z / maxDepth;
            if (!(xn >= -1 && xn <= 1 && yn >= -1 && yn <= 1 && zn >= -1 && zn <= 1)) continue;
            const v3 nrm = v3normalize(m4dir(t_inv, V3(s[id].nrm[0], s[id].nrm[1], s[id].nrm[2])));
            const float rad = s[id].nrm[3];
            const v3 x1 = v3scale(v3scale(v3normalize(V3((nrm.y - nrm.z), -nrm.x, nrm.x)), rad), 1.41421356f);
            const v3 y1 = v3cross(nrm, x1);
            const v3 p1 = project_image(v3add(h, x1), cx, cy, fx, fy), p2 = project_image(v3add(h, y1), cx, cy, fx, fy);
            const v3 p3 = project_image(v3sub(h, y1), cx, cy, fx, fy), p4 = project_image(v3sub(h, x1), cx, cy, fx, fy);
            const float xmin = fminf(p1.x, fminf(p2.x, fminf(p3.x, p4.x))), xmax = fmaxf(p1.x, fmaxf(p2.x, fmaxf(p3.x, p4.x)));
            const float ymin = fminf(p1.y, fminf(p2.y, fminf(p3.y, p4.y))), ymax = fmaxf(p1.y, fmaxf(p2.y, fmaxf(p3.y, p4.y)));
            float size = fmaxf(0.f, fmaxf(fabsf(xmax - xmin), fabsf(ymax - ymin)));
            if (!(size >= 1.0f)) size = 1.0f; /* A5 (also NaN) */
            const float xw = (xn + 1.0f) * (cols * 0.5f), yw = (yn + 1.0f) * (rows * 0.5f), hs = size * 0.5f;
            int x0 = (int)ceilf(xw - hs - 0.5f), x1i = (int)ceilf(xw + hs - 0.5f) - 1; /* xw-hs <= c < xw+hs */
            int y0 = (int)ceilf(yw - hs - 0.5f), y1i = (int)ceilf(yw + hs - 0.5f) - 1;
            x0 = x0 < 0 ? 0 : x0, y0 = y0 < 0 ? 0 : y0;
            x1i = x1i > cols - 1 ? cols - 1 : x1i, y1i = y1i > rows - 1 ? rows - 1 : y1i;
            for (int py = y0; py <= y1i; ++py)
                for (int px = x0; px <= x1i; ++px) {
                    const float fcx = px + 0.5f, fcy = py + 0.5f;
                    const v3 l = v3normalize(V3((fcx - cx) / fx, (fcy - cy) / fy, 1.0f));
                    const v3 corrected = v3scale(l, v3dot(h, nrm) / v3dot(l, nrm));
                    const v3 diff = v3sub(corrected, h);
                    if (v3dot(diff, diff) > rad * rad) continue; /* discard */
                    const uint64_t k = ((uint64_t)depth24((corrected.z / (2 * maxDepth)) + 0.5f) << 32) | (uint32_t)id;
                    const size_t pi = (size_t)py * cols + px;
                    if (pass == 0) {
                        if (k < key[pi]) key[pi] = k;
                    } else if (k == key[pi] && depth_out) { /* depth_splat.frag:39 */
                        depth_out[pi] = corrected.z;
                    } else if (k == key[pi]) {
                        const v3 col = decode_color(s[id].col[0]);
                        image_rgba[4 * pi + 0] = (uint8_t)(int)roundf(col.x * 255.0f);
                        image_rgba[4 * pi + 1] = (uint8_t)(int)roundf(col.y * 255.0f);
                        image_rgba[4 * pi + 2] = (uint8_t)(int)roundf(col.z * 255.0f);
                        image_rgba[4 * pi + 3] = 255;
                        const float z = corrected.z;
                        vertexConf[4 * pi + 0] = (fcx - cx) * z * (1.f / fx);
                        vertexConf[4 * pi + 1] = (fcy - cy) * z * (1.f / fy);
                        vertexConf[4 * pi + 2] = z;
                        vertexConf[4 * pi + 3] = s[id].pos[3];
                        normalRadius[4 * pi + 0] = nrm.x, normalRadius[4 * pi + 1] = nrm.y;
                        normalRadius[4 * pi + 2] = nrm.z, normalRadius[4 * pi + 3] = rad;
                        time_out[pi] = (uint16_t)(unsigned)s[id].col[2];
                    }
                }
        }
        if (pass == 0) { /* cleared targets */
            if (depth_out) {
                memset(depth_out, 0, 4 * n);
            } else {
                memset(image_rgba, 0, 4 * n);
                memset(vertexConf, 0, 16 * n);
                memset(normalRadius, 0, 16 * n);
                memset(time_out, 0, 2 * n);
            }
        }
    }
    free(key);
}

void orc_combined_predict(const orc_surfel *s, int count, const float pose[16], float cx, float cy, float fx, float fy,
                          int cols, int rows, float maxDepth, float confThreshold, int time, int maxTime,
                          int timeDelta, uint8_t *image_rgba, float *vertexConf, float *normalRadius,
                          uint16_t *time_out) {
    splat_render(s, count, pose, cx, cy, fx, fy, cols, rows, maxDepth, confThreshold, time, maxTime, timeDelta,
                 image_rgba, vertexConf, normalRadius, time_out, NULL);
}

/* ModelProjection::synthesizeDepth (ModelProjection.cpp:275-335) + splat.vert + depth_splat.frag:
 * the same sprites and depth test, the R32F target receives corrected_pos.z (0 where nothing lands) */
void orc_synthesize_depth(const orc_surfel *s, int count, const float pose[16], float cx, float cy, float fx, float fy,
                          int cols, int rows, float maxDepth, float confThreshold, int time, int maxTime,
                          int timeDelta, float *depth_out) {
    splat_render(s, count, pose, cx, cy, fx, fy, cols, rows, maxDepth, confThreshold, time, maxTime, timeDelta, NULL,
                 NULL, NULL, NULL, depth_out);
}

/* ------------------------------------------------------------------------------------- */
/* Model::fuse: data association (Shaders/data.vert) + update (Shaders/update.vert)        */
/* new_out receives the new unstable surfels (colour.w = -2) in draw order (column-major   */
/* pixel order, Model.cpp:204-210); returns their number.  Surfels are updated in place    */
/* (the reference ping-pongs two VBOs, Model.cpp:984-1047).                                */
/* ------------------------------------------------------------------------------------- */
int orc_fuse(orc_surfel *s, int count, const uint8_t *rgb, const float *depth_raw, const float *depth_filtered,
             const uint8_t *mask, const uint32_t *index, const float *vertConf, const float *normRad,
             const float pose[16], float cx, float cy, float fx, float fy, int cols, int rows, int time,
             float weighting, uint8_t maskID, float maxDepth, orc_surfel *new_out) {
    const float ifx = (float)(1.0 / fx), ify = (float)(1.0 / fy); /* Model.cpp:920-921 (double division) */
    const float scale = 1.0f;                                     /* ModelProjection::FACTOR */
    orc_surfel *upd = (orc_surfel *)calloc((size_t)count > 0 ? count : 1, sizeof(orc_surfel));
    uint8_t *has = (uint8_t *)calloc((size_t)count > 0 ? count : 1, 1);
    int nnew = 0;
    for (int i = 0; i < cols; ++i)
        for (int j = 0; j < rows; ++j) {
            const float tx = uv_coord(i, cols), ty = uv_coord(j, rows);
            const float x = tx * cols, y = ty * rows;
            const v3 vPosLocal = get_vertex(depth_raw, cols, rows, tx, ty, x, y, cx, cy, ifx, ify);
            const v3 vPos = m4point(pose, vPosLocal);
            const v3 vPos_f = get_vertex(depth_filtered, cols, rows, tx, ty, x, y, cx, cy, ifx, ify);
            const uint8_t *c = rgb + (size_t)(texel(ty, rows) * cols + texel(tx, cols)) * 3;
            const v3 nl = get_normal(depth_filtered, cols, rows, vPos_f, tx, ty, x, y, cx, cy, ifx, ify);
            const v3 ng = m4dir(pose, nl);
            orc_surfel m;
            m.pos[0] = vPos.x, m.pos[1] = vPos.y, m.pos[2] = vPos.z;
            m.pos[3] = confidence(x, y, cx, cy, weighting);
            m.col[0] = encode_color(c[0] / 255.0f, c[1] / 255.0f, c[2] / 255.0f);
            m.col[1] = 0, m.col[2] = (float)time, m.col[3] = 0;
            m.nrm[0] = ng.x, m.nrm[1] = ng.y, m.nrm[2] = ng.z;
            m.nrm[3] = get_radius(vPos_f.z, nl.z, ifx, ify);

            const int tm = ((int)(float)time) % 2;
            /* checkNeighbours (data.vert:59-78) on the RAW depth */
            const float zl = depth_raw[texel(ty, rows) * cols + texel(tx - (1.0f / cols), cols)];
            const float zu = depth_raw[texel(ty - (1.0f / rows), rows) * cols + texel(tx, cols)];
            const float zr = depth_raw[texel(ty, rows) * cols + texel(tx + (1.0f / cols), cols)];
            const float zd = depth_raw[texel(ty + (1.0f / rows), rows) * cols + texel(tx, cols)];
            const int neighbours = !(zl == 0) && !(zu == 0) && !(zr == 0) && !(zd == 0);
            if (!(((int)x) % 2 == tm && ((int)y) % 2 == tm && mask[texel(ty, rows) * cols + texel(tx, cols)] == maskID &&
                  neighbours && vPosLocal.z > 0 && vPosLocal.z <= maxDepth))
                continue;
            int operation = 0;
            uint32_t best = 0;
            const float indexXStep = (1.0f / (cols * scale)) * 0.5f, indexYStep = (1.0f / (rows * scale)) * 0.5f;
            float bestDist = 1000;
            const float windowMultiplier = 2;
            const float xl = (x - cx) * ifx, yl = (y - cy) * ify;
            const float lambda = sqrtf(xl * xl + yl * yl + 1);
            const v3 ray = V3(xl, yl, 1);
            for (float ii = tx - (scale * indexXStep * windowMultiplier); ii < tx + (scale * indexXStep * windowMultiplier);
                 ii += indexXStep)
                for (float jj = ty - (scale * indexYStep * windowMultiplier);
                     jj < ty + (scale * indexYStep * windowMultiplier); jj += indexYStep) {
                    const size_t t = (size_t)texel(jj, rows) * cols + texel(ii, cols);
                    const uint32_t current = index[t];
                    if (current > 0U) {
                        const float *vc = vertConf + 4 * t;
                        const float zdiff = (vc[2] - vPosLocal.z);
                        if (fabsf(zdiff * lambda) < 0.05f) {
                            const float dist = v3length(v3cross(ray, V3(vc[0], vc[1], vc[2])));
                            const float *nr = normRad + 4 * t;
                            const v3 nrv = V3(nr[0], nr[1], nr[2]);
                            /* angleBetween() < 0.5 rad (data.vert:80-83,154): acos is monotone on
                             * [-1,1], so |acos(c)| < 0.5 <=> c > cos(0.5); written without the
                             * transcendental so checker and kernel agree bit for bit (c outside
                             * [-1,1] or NaN fails both forms, up to the last ulp at c ~ 1) */
                            const float cosang = v3dot(nrv, nl) / (v3length(nrv) * v3length(nl));
                            if (dist < bestDist && (fabsf(nr[2]) < 0.75f || (cosang > 0.87758255f && cosang <= 1.0f))) {
                                operation = 1;
                                bestDist = dist;
                                best = current;
                            }
                        }
                    }
                }
            if (operation == 1) {
                m.col[3] = -1;
                /* first fragment in draw order wins the update texel (depth test at z = 0) */
                if (best < (uint32_t)count && !has[best]) {
                    has[best] = 1;
                    upd[best] = m;
                }
            } else {
                m.col[3] = -2;
                new_out[nnew++] = m;
            }
        }
    /* update.vert:38-111 */
    for (int k = 0; k < count; ++k) {
        if (!has[k]) continue;
        const orc_surfel *nw = &upd[k];
        orc_surfel *o = &s[k];
        const float c_k = o->pos[3], a = nw->pos[3];
        if (nw->nrm[3] < (1.0f + 0.5f) * o->nrm[3]) {
            for (int d = 0; d < 3; ++d) o->pos[d] = ((c_k * o->pos[d]) + (a * nw->pos[d])) / (c_k + a);
            const v3 oldCol = decode_color(o->col[0]), newCol = decode_color(nw->col[0]);
            const float ar = ((c_k * oldCol.x) + (a * newCol.x)) / (c_k + a);
            const float ag = ((c_k * oldCol.y) + (a * newCol.y)) / (c_k + a);
            const float ab = ((c_k * oldCol.z) + (a * newCol.z)) / (c_k + a);
            o->col[0] = encode_color(ar, ag, ab);
            o->col[3] = (float)time;
            float nr4[4];
            for (int d = 0; d < 4; ++d) nr4[d] = ((c_k * o->nrm[d]) + (a * nw->nrm[d])) / (c_k + a);
            const v3 nn = v3normalize(V3(nr4[0], nr4[1], nr4[2]));
            o->nrm[0] = nn.x, o->nrm[1] = nn.y, o->nrm[2] = nn.z, o->nrm[3] = nr4[3];
            o->pos[3] = c_k + a;
        } else {
            o->pos[3] = c_k + a;
            o->col[3] = (float)time;
        }
    }
    free(upd);
    free(has);
    return nnew;
}

/* ------------------------------------------------------------------------------------- */
/* Model::clean + Shaders/copy_unstable.vert:53-150 (.geom): filter + ordered compaction   */
/* of the existing surfels followed by the new unstable ones.  The deformation-graph part  */
/* (:152-335) only runs with nodes > 0, i.e. never (loop closure is disabled).             */
/* ------------------------------------------------------------------------------------- */
static int clean_one(orc_surfel *v, const float t_inv[16], float cx, float cy, float fx, float fy, int cols, int rows,
                     int time, int timeDelta, float confThreshold, float outlierCoeff, uint8_t maskID,
                     const uint32_t *index, const float *vertConf, const float *colorTime, const float *depth_in,
                     const uint8_t *mask) {
    int test = 1;
    const float scale = 1.0f;
    const v3 localPos = m4point(t_inv, V3(v->pos[0], v->pos[1], v->pos[2]));
    const float x = ((fx * localPos.x) / localPos.z) + cx, y = ((fy * localPos.y) / localPos.z) + cy;
    const v3 localNorm = v3normalize(m4dir(t_inv, V3(v->nrm[0], v->nrm[1], v->nrm[2])));
    const float x_n = x / cols, y_n = y / rows;
    const float stepX = 1.0f / cols, stepY = 1.0f / rows;
    const float indexXStep = stepX * 0.5f / scale, indexYStep = stepY * 0.5f / scale;
    const float windowMultiplier = 2;
    int count = 0, zCount = 0, violationCount = 0;
    float avgViolation = 0;
    if ((float)time - v->col[3] < (float)timeDelta && localPos.z > 0 && x > 0 && y > 0 && x < cols && y < rows) {
        for (float i = x_n - (scale * indexXStep * windowMultiplier); i < x_n + (scale * indexXStep * windowMultiplier);
             i += indexXStep)
            for (float j = y_n - (scale * indexYStep * windowMultiplier);
                 j < y_n + (scale * indexYStep * windowMultiplier); j += indexYStep) {
                const size_t t = (size_t)texel(j, rows) * cols + texel(i, cols);
                const uint32_t current = index[t];
                if (current > 0U) {
                    const float *vc = vertConf + 4 * t, *ct = colorTime + 4 * t;
                    const float dx = vc[0] - localPos.x, dy = vc[1] - localPos.y;
                    if (ct[2] < v->col[2] && vc[3] > confThreshold && vc[2] > localPos.z &&
                        vc[2] - localPos.z < 0.01f && sqrtf(dx * dx + dy * dy) < v->nrm[3] * 1.4f)
                        count++;
                    if (ct[3] == (float)time && vc[3] > confThreshold && vc[2] > localPos.z &&
                        vc[2] - localPos.z > 0.01f && fabsf(localNorm.z) > 0.85f)
                        zCount++;
                }
            }
        for (float i = x_n - stepX; i <= x_n + stepX; i += stepX)
            for (float j = y_n - stepY; j <= y_n + stepY; j += stepY) {
                const float d = depth_in[texel(j, rows) * cols + texel(i, cols)] - localPos.z;
                if (d > 0.03f) {
                    violationCount++;
                    avgViolation += d;
                }
            }
    }
    if (count > 8 || zCount > 4) test = 0;
    if (v->col[3] == -2) v->col[3] = (float)time;
    if ((v->col[3] == -1 || (((float)time - v->col[3]) > 20 && v->pos[3] < confThreshold))) test = 0;
    if (v->col[3] > 0 && (float)time - v->col[3] > (float)timeDelta) test = 1;
    if (violationCount > 0) {
        avgViolation /= violationCount;
        v->pos[3] *= 1.0f / (1 + outlierCoeff * avgViolation);
        const size_t t = (size_t)texel(y_n, rows) * cols + texel(x_n, cols);
        const float wDepth = depth_in[t];
        if (mask[t] != maskID && (wDepth > localPos.z - 0.05f && wDepth < localPos.z + 0.05f))
            v->pos[3] *= (0.5f + 0.5f * (1 - outlierCoeff / 10.0f));
    }
    return test;
}

int orc_clean(const orc_surfel *s, int count, const orc_surfel *new_unstable, int nnew, const float pose[16],
              float cx, float cy, float fx, float fy, int cols, int rows, int time, int timeDelta,
              float confThreshold, float outlierCoeff, uint8_t maskID, const uint32_t *index, const float *vertConf,
              const float *colorTime, const float *depth_filtered, const uint8_t *mask, orc_surfel *out) {
    float t_inv[16];
    orc_inverse4f(pose, t_inv);
    int n = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const orc_surfel *src = pass == 0 ? s : new_unstable;
        const int m = pass == 0 ? count : nnew;
        for (int k = 0; k < m; ++k) {
            orc_surfel v = src[k];
            if (clean_one(&v, t_inv, cx, cy, fx, fy, cols, rows, time, timeDelta, confThreshold, outlierCoeff, maskID,
                          index, vertConf, colorTime, depth_filtered, mask))
                out[n++] = v;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------- */
/* Model::performFillIn + Shaders/fill_vertex.frag, fill_normal.frag, fill_rgb.frag         */
/* ------------------------------------------------------------------------------------- */
void orc_fill_in(const float *vertex_pred, const float *normal_pred, const uint8_t *image_pred_rgba,
                 const float *depth_filtered, const uint8_t *rgb, int cols, int rows, float cx, float cy, float fx,
                 float fy, int passthrough_geom, int passthrough_rgb, float *vertex_out, float *normal_out,
                 uint8_t *image_out_rgba) {
    const float ifx = 1.0f / fx, ify = 1.0f / fy; /* FillIn.cpp:93-94 (float division) */
    for (int py = 0; py < rows; ++py)
        for (int px = 0; px < cols; ++px) {
            const size_t i = (size_t)py * cols + px;
            /* full-screen quad: texcoord at the pixel centre */
            const float tx = (px + 0.5f) / cols, ty = (py + 0.5f) / rows;
            const int ix = (int)(tx * cols), iy = (int)(ty * rows);
            if (vertex_pred[4 * i + 2] == 0 || passthrough_geom == 1) {
                const float z = depth_filtered[texel(ty, rows) * cols + texel(tx, cols)];
                vertex_out[4 * i + 0] = (ix - cx) * z * ifx;
                vertex_out[4 * i + 1] = (iy - cy) * z * ify;
                vertex_out[4 * i + 2] = z;
                vertex_out[4 * i + 3] = 1;
            } else {
                memcpy(vertex_out + 4 * i, vertex_pred + 4 * i, 16);
            }
            if (normal_pred[4 * i + 2] == 0 || passthrough_geom == 1) {
                const v3 p = get_vertex(depth_filtered, cols, rows, tx, ty, (float)ix, (float)iy, cx, cy, ifx, ify);
                /* int overload of getNormal: forward differences (geometry.glsl:43-59) */
                const v3 vx = get_vertex(depth_filtered, cols, rows, tx + (1.0f / cols), ty, (float)(ix + 1), (float)iy, cx,
                                         cy, ifx, ify);
                const v3 vy = get_vertex(depth_filtered, cols, rows, tx, ty + (1.0f / rows), (float)ix, (float)(iy + 1), cx,
                                         cy, ifx, ify);
                const v3 nn = v3normalize(v3cross(v3sub(vx, p), v3sub(vy, p)));
                normal_out[4 * i + 0] = nn.x, normal_out[4 * i + 1] = nn.y, normal_out[4 * i + 2] = nn.z;
                normal_out[4 * i + 3] = 1;
            } else {
                memcpy(normal_out + 4 * i, normal_pred + 4 * i, 16);
            }
            const uint8_t *e = image_pred_rgba + 4 * i;
            if (e[0] / 255.0f + e[1] / 255.0f + e[2] / 255.0f == 0 || passthrough_rgb == 1) {
                image_out_rgba[4 * i + 0] = rgb[3 * i + 0], image_out_rgba[4 * i + 1] = rgb[3 * i + 1];
                image_out_rgba[4 * i + 2] = rgb[3 * i + 2], image_out_rgba[4 * i + 3] = 255;
            } else {
                memcpy(image_out_rgba + 4 * i, e, 4);
            }
        }
}

/* MultiMotionFusion::requiresFillIn (MultiMotionFusion.cpp:877-895) + GPUResize::image:
 * nearest sample of the predicted colour at the centres of a (cols/20) x (rows/20) grid. */
int orc_requires_fill_in(const uint8_t *image_pred_rgba, int cols, int rows, float ratio) {
    const int dc = cols / 20, dr = rows / 20;
    int sum = 0;
    for (int j = 0; j < dr; ++j)
        for (int i = 0; i < dc; ++i) {
            const float tx = (i + 0.5f) / dc, ty = (j + 0.5f) / dr;
            const uint8_t *p = image_pred_rgba + 4 * ((size_t)texel(ty, rows) * cols + texel(tx, cols));
            sum += p[0] > 0 && p[1] > 0 && p[2] > 0;
        }
    return (float)sum / (float)(dr * dc) < ratio;
}
