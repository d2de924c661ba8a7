/*
 * mmf_oracle_slic.c -- CPU restatement of the super-pixel resampling the segmentation feeds on
 * (SURVEY.md 8(f) item 3): Slic::downsample<float>, Slic::downsampleThresholded<float>, Slic::downsample()
 * and Slic::upsample<unsigned char> (Core/Segmentation/Slic.h:48-146, Slic.cpp:72-112), as called from
 * Segmentation.cpp:177-178,220-221,683.
 *
 * TEST INFRASTRUCTURE ONLY (see mmf_oracle.h).  PARITY UNPINNED: no reference test covers these functions.
 * The label image itself comes from gSLICr (un-vendored, doc/install.sh) and is an INPUT here.
 *
 * Restated literally, including two quirks of the reference:
 *   - mapToHigh(index) divides by spixelY where spixelX is meant (Slic.h:196);
 *   - the float versions divide IN PLACE while walking the super-pixels in ascending order, so an empty
 *     super-pixel that resamples from a lower index reads an already divided value (Slic.h:72-81,113-122).
 * Float sums run in pixel order, one `+=` per pixel, like the reference's loops.
 */
#include <stdlib.h>
#include <string.h>

#include "mmf_oracle.h"

/* Slic.cpp:76-79 */
void orc_slic_counts(const int *labels, int npix, int nspix, int *counts) {
    memset(counts, 0, sizeof(int) * (size_t)nspix);
    for (int i = 0; i < npix; ++i) counts[labels[i]]++;
}

/* Slic.h:191-209 */
static int resample_empty_index(const int *labels, int width, int height, int S, int spx, int spy, int index) {
    const int hx = index % spx, hy = index / spy; /* sic: spixelY */
    int cx = (int)(hx * S + S * 0.5), cy = (int)(hy * S + S * 0.5);
    if (cy >= height) cy = height - 1;
    if (cx >= width) cx = width - 1;
    return labels[cx + cy * width];
}

/* Slic.h:48-83 (thresholded == 0) and :87-126 (thresholded != 0); out has (height/S)*(width/S) floats */
void orc_slic_downsample(const int *labels, int width, int height, int S, const float *image, int channels, int channel,
                         int thresholded, float min_threshold, float *out) {
    const int spx = width / S, spy = height / S, n = spx * spy, npix = width * height;
    int *counts = (int *)malloc(sizeof(int) * (size_t)n), *dcounts = (int *)calloc((size_t)n, sizeof(int));
    orc_slic_counts(labels, npix, n, counts);
    for (int s = 0; s < n; ++s) out[s] = 0.f;
    for (int i = 0; i < npix; ++i) {
        const float v = image[(size_t)i * channels + channel];
        if (thresholded) {
            if (v > min_threshold) out[labels[i]] += v, dcounts[labels[i]]++;
        } else {
            out[labels[i]] += v;
        }
    }
    for (int s = 0; s < n; ++s) {
        int cnt = thresholded ? dcounts[s] : counts[s], r = s;
        if (cnt == 0) {
            r = resample_empty_index(labels, width, height, S, spx, spy, s);
            cnt = counts[r];
        }
        out[s] = out[r] / (float)cnt;
    }
    free(counts), free(dcounts);
}

/* Slic.cpp:82-112; rgb: interleaved u8 with `channels` >= 3; out [n][3] = means of input channels (2, 1, 0)
 * (the reference sums (.b, .g, .r) of gSLICr's Vector4u, whose .r is the first byte) */
void orc_slic_downsample_rgb(const int *labels, int width, int height, int S, const uint8_t *rgb, int channels, uint8_t *out) {
    const int spx = width / S, spy = height / S, n = spx * spy, npix = width * height;
    int *counts = (int *)malloc(sizeof(int) * (size_t)n), *sums = (int *)calloc((size_t)n * 3, sizeof(int));
    orc_slic_counts(labels, npix, n, counts);
    for (int i = 0; i < npix; ++i)
        for (int k = 0; k < 3; ++k) sums[labels[i] * 3 + k] += rgb[(size_t)i * channels + 2 - k];
    for (int s = 0; s < n; ++s) {
        int cnt = counts[s], r = s;
        if (cnt == 0) {
            r = resample_empty_index(labels, width, height, S, spx, spy, s);
            cnt = counts[r];
        }
        for (int k = 0; k < 3; ++k) {
            const int v = sums[r * 3 + k] / cnt; /* cv::Vec3b(int...) saturates */
            out[s * 3 + k] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
    free(counts), free(sums);
}

/* Slic.h:133-146 with Tin = Tout = unsigned char */
void orc_slic_upsample_u8(const int *labels, int npix, const uint8_t *map, uint8_t *out) {
    for (int i = 0; i < npix; ++i) out[i] = map[labels[i]];
}
