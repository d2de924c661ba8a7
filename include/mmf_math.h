/*
 * mmf_math.h -- the one transcendental the surfel path needs, written out in IEEE float32
 * operations (explicit fmaf where a fused multiply-add is meant; both builds run without implicit
 * contraction) so that every build (gcc for the oracle, hipcc for gfx950) produces the SAME bits.  libm / ocml expf differ from each other in the last
 * ulp, which would make the bilateral filter and the surfel confidence only "close" between
 * the checker and the kernels; with a shared definition they are comparable bit for bit.
 * Accuracy against a correctly rounded exp: <= 2 ulp on [-87, 88] (tests/test_oracle_kat.py).
 *
 * The reference calls GLSL exp() here (Shaders/surfels.glsl:45, depth_bilateral_metric.frag:66),
 * whose precision is implementation defined.
 */
#ifndef MMF_MATH_H_
#define MMF_MATH_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define MMF_MATH_FN __host__ __device__ static inline
#else
#define MMF_MATH_FN static inline
#endif

MMF_MATH_FN float mmf_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283905f) return INFINITY;
    if (x < -103.0f) return 0.0f;
    /* x = n ln2 + r, |r| <= ln2/2, ln2 split so that n*hi is exact */
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    /* e^r by a degree-6 polynomial (Cephes expf coefficients), Horner form with fused steps: the
     * bilateral filter evaluates this 169 times per pixel, and an unfused step is two instructions */
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, r * r, r) + 1.0f;
    /* scale by 2^n in two exact steps (n in [-149, 128]) */
    int ni = (int)n;
    float s1, s2;
    int h = ni / 2;
    uint32_t b1 = (uint32_t)(h + 127) << 23, b2 = (uint32_t)(ni - h + 127) << 23;
    memcpy(&s1, &b1, 4);
    memcpy(&s2, &b2, 4);
    return (p * s1) * s2;
}

#endif /* MMF_MATH_H_ */
