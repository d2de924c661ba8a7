/*
 * mmf_oracle_superpoint.c -- CPU restatement of the SuperPoint keypoint network and its post-processing
 * (the north-star's "SuperPoint VGG encoder + descriptor head").
 *
 * TEST INFRASTRUCTURE ONLY (see mmf_oracle.h).  PARITY UNPINNED: the reference calls
 *     kp_predictor = std::make_shared<SuperPoint>(keypoint_predictor_path);       (Core/MultiMotionFusion.cpp:78)
 *     std::tie(coordinates[i], descriptors[i]) = kp_predictor->getFeatures(img);  (Core/MultiMotionFusion.cpp:233)
 * from the un-vendored dependency `super_point_inference` (branch `master`, unpinned, doc/install.sh:45-47),
 * which runs the MagicLeap "SuperPointNet.pt" TorchScript through libtorch.  Neither that library nor the
 * weights are in /root/reference, and the reference has no test or fixture for it.  What is restated here is
 * the PUBLISHED algorithm the dependency wraps (DeTone, Malisiewicz, Rabinovich: "SuperPoint: Self-Supervised
 * Interest Point Detection and Description", CVPRW 2018, and the authors' demo_superpoint.py):
 *
 *   network   conv1a(1,64) conv1b(64,64) pool  conv2a(64,64) conv2b(64,64) pool  conv3a(64,128) conv3b(128,128)
 *             pool  conv4a(128,128) conv4b(128,128)           -- all 3x3, stride 1, zero pad 1, ReLU; pool = 2x2 max
 *             detector   convPa(128,256) 3x3 ReLU, convPb(256,65) 1x1
 *             descriptor convDa(128,256) 3x3 ReLU, convDb(256,256) 1x1, then L2 normalisation over channels
 *   heatmap   dense = exp(semi) / (sum_c exp(semi_c) + 1e-5); drop channel 64 (dustbin);
 *             heat[8*hc + i][8*wc + j] = dense[8*i + j][hc][wc]
 *   keypoints heat >= conf_thresh; greedy non-maximum suppression (strongest first, Chebyshev radius nms_dist);
 *             points closer than `border` to the image edge dropped afterwards; strongest first
 *   descriptor bilinear sample of the normalised coarse descriptor map at the keypoint (grid_sample with the
 *             corner-aligned convention of the demo's PyTorch), L2 normalised again
 *   what the caller sees (PointTracker.cpp:40-41): coordinates normalised to [0,1) = (x / W, y / H).
 *
 * Arithmetic (ours to define -- cuDNN / MKL-DNN summation orders are implementation details): every
 * convolution output is ONE fmaf chain starting at 0, in this order
 *     for c0 in 0, 32, 64, ... (blocks of 32 input channels)
 *       for ky in 0..2, for kx in 0..2      (1x1: a single tap)
 *         for ci in c0 .. min(c0 + 32, Cin) - 1
 *           acc = fmaf(in[y + ky - 1][x + kx - 1][ci], w[co][ci][ky][kx], acc)     (0 outside the image)
 *     out = acc + bias[co];  ReLU: out < 0 ? 0 : out
 * which is the order the gfx950 f32 MFMA kernel (csrc/superpoint_kernels.hpp) accumulates in, so the two
 * agree bit for bit.  exp is the shared mmf_expf (include/mmf_math.h).  tests/test_oracle_superpoint.py pins
 * this file against torch.nn.functional (fp32, CPU) within 1e-4.
 *
 * Layouts: activations are channels-last [H][W][C] float32; weights are PyTorch's [Cout][Cin][kh][kw].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mmf_math.h"
#include "mmf_oracle.h"

#define SP_KBLOCK 32

/* weights repacked to chain order: wk[kidx][co] */
static float *repack_weights(const float *w, int cin, int cout, int taps, int *nk_out) {
    const int nk = cin * taps;
    float *wk = (float *)malloc(sizeof(float) * (size_t)nk * cout);
    int kidx = 0;
    for (int c0 = 0; c0 < cin; c0 += SP_KBLOCK)
        for (int tap = 0; tap < taps; ++tap)
            for (int ci = c0; ci < cin && ci < c0 + SP_KBLOCK; ++ci, ++kidx)
                for (int co = 0; co < cout; ++co) wk[(size_t)kidx * cout + co] = w[((size_t)co * cin + ci) * taps + tap];
    *nk_out = nk;
    return wk;
}

/* one output pixel: all Cout chains advance together (vectorises over co; each chain stays sequential) */
#if defined(__x86_64__)
__attribute__((target("avx2,fma")))
#endif
static void conv_pixel_fma(const float *in, int H, int W, int in_stride, int cin, int taps, int y, int x, const float *wk,
                           const float *bias, int cout, int relu, float *acc, float *out) {
    for (int co = 0; co < cout; ++co) acc[co] = 0.f;
    const float *wrow = wk;
    for (int c0 = 0; c0 < cin; c0 += SP_KBLOCK)
        for (int tap = 0; tap < taps; ++tap) {
            const int ky = taps == 9 ? tap / 3 - 1 : 0, kx = taps == 9 ? tap % 3 - 1 : 0;
            const int yy = y + ky, xx = x + kx;
            const int inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
            for (int ci = c0; ci < cin && ci < c0 + SP_KBLOCK; ++ci, wrow += cout) {
                const float v = inside ? in[((size_t)yy * W + xx) * in_stride + ci] : 0.f;
                for (int co = 0; co < cout; ++co) acc[co] = __builtin_fmaf(v, wrow[co], acc[co]);
            }
        }
    for (int co = 0; co < cout; ++co) {
        const float o = acc[co] + bias[co];
        out[co] = relu && o < 0.f ? 0.f : o;
    }
}

static void conv_pixel_libm(const float *in, int H, int W, int in_stride, int cin, int taps, int y, int x, const float *wk,
                            const float *bias, int cout, int relu, float *acc, float *out) {
    for (int co = 0; co < cout; ++co) acc[co] = 0.f;
    const float *wrow = wk;
    for (int c0 = 0; c0 < cin; c0 += SP_KBLOCK)
        for (int tap = 0; tap < taps; ++tap) {
            const int ky = taps == 9 ? tap / 3 - 1 : 0, kx = taps == 9 ? tap % 3 - 1 : 0;
            const int yy = y + ky, xx = x + kx;
            const int inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
            for (int ci = c0; ci < cin && ci < c0 + SP_KBLOCK; ++ci, wrow += cout) {
                const float v = inside ? in[((size_t)yy * W + xx) * in_stride + ci] : 0.f;
                for (int co = 0; co < cout; ++co) acc[co] = fmaf(v, wrow[co], acc[co]);
            }
        }
    for (int co = 0; co < cout; ++co) {
        const float o = acc[co] + bias[co];
        out[co] = relu && o < 0.f ? 0.f : o;
    }
}

/* taps = 9 (3x3, pad 1) or 1 (1x1).  `in` has `in_stride` floats per pixel of which the first cin are read. */
void orc_sp_conv(const float *in, int H, int W, int in_stride, int cin, const float *w, const float *bias, int cout, int taps,
                 int relu, float *out) {
    int nk;
    float *wk = repack_weights(w, cin, cout, taps, &nk);
    int hw_fma = 0;
#if defined(__x86_64__)
    hw_fma = __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
#endif
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (size_t)cout);
#pragma omp for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float *o = out + ((size_t)y * W + x) * cout;
                if (hw_fma)
                    conv_pixel_fma(in, H, W, in_stride, cin, taps, y, x, wk, bias, cout, relu, acc, o);
                else
                    conv_pixel_libm(in, H, W, in_stride, cin, taps, y, x, wk, bias, cout, relu, acc, o);
            }
        free(acc);
    }
    free(wk);
}

void orc_sp_maxpool2(const float *in, int H, int W, int C, float *out) {
    const int Ho = H / 2, Wo = W / 2;
    for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Wo; ++x)
            for (int c = 0; c < C; ++c) {
                const float a = in[((size_t)(2 * y) * W + 2 * x) * C + c], b = in[((size_t)(2 * y) * W + 2 * x + 1) * C + c];
                const float d = in[((size_t)(2 * y + 1) * W + 2 * x) * C + c], e = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * C + c];
                const float m0 = a > b ? a : b, m1 = d > e ? d : e;
                out[((size_t)y * Wo + x) * C + c] = m0 > m1 ? m0 : m1;
            }
}

/* the network input: [0,1] grey.  channels == 1: grey / 255;  channels == 3: (0.299 R + 0.587 G + 0.114 B) / 255 */
void orc_sp_input(const uint8_t *img, int H, int W, int channels, float *out) {
    for (size_t i = 0; i < (size_t)H * W; ++i) {
        if (channels == 1)
            out[i] = (float)img[i] / 255.0f;
        else {
            const uint8_t *p = img + i * channels;
            out[i] = ((0.299f * (float)p[0] + 0.587f * (float)p[1]) + 0.114f * (float)p[2]) / 255.0f;
        }
    }
}

/* L2 normalisation over the channels of every pixel: norm = sqrtf(fmaf chain of squares), x / norm */
void orc_sp_l2_normalize(float *desc, int npix, int C) {
    for (int p = 0; p < npix; ++p) {
        float *d = desc + (size_t)p * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = fmaf(d[c], d[c], s);
        const float n = sqrtf(s);
        for (int c = 0; c < C; ++c) d[c] = d[c] / n;
    }
}

/* weights[24]: {w, b} of conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa convPb convDa convDb.
 * semi: [H/8][W/8][65] raw detector logits; desc: [H/8][W/8][256] normalised coarse descriptors.
 * returns 0, or -1 when H or W is not a multiple of 8. */
int orc_sp_forward(const float *input, int H, int W, const float *const *weights, float *semi, float *desc) {
    if (H <= 0 || W <= 0 || H % 8 || W % 8) return -1;
    static const int cin[8] = {1, 64, 64, 64, 64, 128, 128, 128}, cout[8] = {64, 64, 64, 64, 128, 128, 128, 128};
    float *a = (float *)malloc(sizeof(float) * (size_t)H * W * 64), *b = (float *)malloc(sizeof(float) * (size_t)H * W * 64);
    memcpy(a, input, sizeof(float) * (size_t)H * W);
    int h = H, w = W;
    for (int l = 0; l < 8; ++l) {
        orc_sp_conv(a, h, w, cin[l], cin[l], weights[2 * l], weights[2 * l + 1], cout[l], 9, 1, b);
        if ((l & 1) && l < 7) {
            orc_sp_maxpool2(b, h, w, cout[l], a);
            h /= 2, w /= 2;
        } else {
            float *t = a;
            a = b, b = t;
        }
    }
    /* a = conv4b output [h][w][128] */
    float *head = (float *)malloc(sizeof(float) * (size_t)h * w * 256);
    orc_sp_conv(a, h, w, 128, 128, weights[16], weights[17], 256, 9, 1, head);
    orc_sp_conv(head, h, w, 256, 256, weights[18], weights[19], 65, 1, 0, semi);
    orc_sp_conv(a, h, w, 128, 128, weights[20], weights[21], 256, 9, 1, head);
    orc_sp_conv(head, h, w, 256, 256, weights[22], weights[23], 256, 1, 0, desc);
    orc_sp_l2_normalize(desc, h * w, 256);
    free(a), free(b), free(head);
    return 0;
}

/* demo_superpoint.py: softmax with the +1e-5 in the denominator, dustbin dropped, depth-to-space by 8 */
void orc_sp_heatmap(const float *semi, int Hc, int Wc, float *heat) {
    const int W = Wc * 8;
    for (int hc = 0; hc < Hc; ++hc)
        for (int wc = 0; wc < Wc; ++wc) {
            const float *s = semi + ((size_t)hc * Wc + wc) * 65;
            float e[65], sum = 0.f;
            for (int c = 0; c < 65; ++c) e[c] = mmf_expf(s[c]), sum = sum + e[c];
            sum = sum + 0.00001f;
            for (int c = 0; c < 64; ++c) heat[(size_t)(hc * 8 + c / 8) * W + wc * 8 + c % 8] = e[c] / sum;
        }
}

typedef struct {
    float conf;
    int idx;
} sp_cand;
static int cand_cmp(const void *pa, const void *pb) {
    const sp_cand *a = (const sp_cand *)pa, *b = (const sp_cand *)pb;
    if (a->conf != b->conf) return a->conf > b->conf ? -1 : 1; /* strongest first */
    return a->idx < b->idx ? -1 : (a->idx > b->idx);           /* ties: row-major order */
}

/* nms_fast of the demo: visit candidates strongest first; a visited candidate that is not yet suppressed is
 * kept and suppresses every candidate within Chebyshev distance nms_dist; afterwards points with
 * x < border, x >= W - border, y < border or y >= H - border are dropped.  Output strongest first.
 * returns the number of keypoints (at most max_out are written). */
int orc_sp_keypoints(const float *heat, int H, int W, float conf_thresh, int nms_dist, int border, int max_out, int *xy,
                     float *conf) {
    sp_cand *c = (sp_cand *)malloc(sizeof(sp_cand) * (size_t)H * W);
    unsigned char *state = (unsigned char *)calloc((size_t)H * W, 1); /* 0 none, 1 candidate, 2 suppressed */
    int nc = 0;
    for (int i = 0; i < H * W; ++i)
        if (heat[i] >= conf_thresh) c[nc].conf = heat[i], c[nc].idx = i, ++nc, state[i] = 1;
    qsort(c, (size_t)nc, sizeof(sp_cand), cand_cmp);
    int n = 0;
    for (int k = 0; k < nc; ++k) {
        const int i = c[k].idx;
        if (state[i] != 1) continue;
        const int y = i / W, x = i % W;
        for (int yy = y - nms_dist; yy <= y + nms_dist; ++yy)
            for (int xx = x - nms_dist; xx <= x + nms_dist; ++xx)
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && state[(size_t)yy * W + xx] == 1) state[(size_t)yy * W + xx] = 2;
        state[i] = 3; /* kept */
        if (x < border || x >= W - border || y < border || y >= H - border) continue;
        if (n < max_out) xy[2 * n] = x, xy[2 * n + 1] = y, conf[n] = c[k].conf;
        ++n;
    }
    free(c), free(state);
    return n;
}

/* grid_sample(coarse_desc, [x / (W/2) - 1, y / (H/2) - 1]) with corner-aligned unnormalisation
 * ((g + 1) / 2 * (size - 1)), bilinear, zero padding, taps summed nw, ne, sw, se; then L2 normalised */
void orc_sp_sample_descriptors(const float *desc, int Hc, int Wc, const int *xy, int n, int H, int W, float *out) {
    const int C = 256;
    for (int k = 0; k < n; ++k) {
        const float gx = (float)xy[2 * k] / ((float)W / 2.0f) - 1.0f, gy = (float)xy[2 * k + 1] / ((float)H / 2.0f) - 1.0f;
        const float ix = ((gx + 1.0f) / 2.0f) * (float)(Wc - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(Hc - 1);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
        const float wx1 = ix - x0f, wx0 = (x0f + 1.0f) - ix, wy1 = iy - y0f, wy0 = (y0f + 1.0f) - iy;
        const float wnw = wx0 * wy0, wne = wx1 * wy0, wsw = wx0 * wy1, wse = wx1 * wy1;
        float *o = out + (size_t)k * C;
        for (int c = 0; c < C; ++c) {
            const float nw = (x0 >= 0 && x0 < Wc && y0 >= 0 && y0 < Hc) ? desc[((size_t)y0 * Wc + x0) * C + c] : 0.f;
            const float ne = (x1 >= 0 && x1 < Wc && y0 >= 0 && y0 < Hc) ? desc[((size_t)y0 * Wc + x1) * C + c] : 0.f;
            const float sw = (x0 >= 0 && x0 < Wc && y1 >= 0 && y1 < Hc) ? desc[((size_t)y1 * Wc + x0) * C + c] : 0.f;
            const float se = (x1 >= 0 && x1 < Wc && y1 >= 0 && y1 < Hc) ? desc[((size_t)y1 * Wc + x1) * C + c] : 0.f;
            o[c] = ((nw * wnw + ne * wne) + sw * wsw) + se * wse;
        }
        orc_sp_l2_normalize(o, 1, C);
    }
}
