// superpoint_host.hpp -- host side of the SuperPoint C ABI (included by mmf_hip.hip; uses its error macros,
// mmf_ctx, grid1d and device_scan).  Replaces `SuperPoint(path)` / `SuperPoint::getFeatures(img)` of the
// un-vendored super_point_inference package as MultiMotionFusion uses it (Core/MultiMotionFusion.cpp:78,233).
#pragma once
#include <vector>

#include "superpoint_kernels.hpp"

namespace mmf {

// packed order = the order the lanes of sp_conv_mfma_kernel read: [z][step][g][t][kh][n][j] with
// step = block * taps + tap, input channel = 32*block + 8*g + 2*j + kh, output channel = (z*nt + t)*32 + n
static std::vector<float> sp_pack_weights(const float* w, int cin, int cout, int taps, int nt) {
    const int chunks = cin / kSpKBlock, steps = chunks * taps, zt = (cout + 32 * nt - 1) / (32 * nt);
    std::vector<float> out((size_t)zt * steps * 1024 * nt, 0.f);
    for (int z = 0; z < zt; ++z)
        for (int chunk = 0; chunk < chunks; ++chunk)
            for (int tap = 0; tap < taps; ++tap)
                for (int g = 0; g < 4; ++g)
                    for (int t = 0; t < nt; ++t)
                        for (int kh = 0; kh < 2; ++kh)
                            for (int n = 0; n < 32; ++n)
                                for (int j = 0; j < 4; ++j) {
                                    const int ci = 32 * chunk + 8 * g + 2 * j + kh, co = (z * nt + t) * 32 + n;
                                    if (co >= cout) continue;
                                    const size_t idx =
                                        ((((((size_t)z * steps + chunk * taps + tap) * 4 + g) * nt + t) * 2 + kh) * 32 + n) * 4 + j;
                                    out[idx] = w[((size_t)co * cin + ci) * taps + tap];
                                }
    return out;
}

struct SpLayer {
    int cin = 0, cout = 0, taps = 9, nt = 1;
    bool pool = false, relu = true;
    float* wpack = nullptr;
    float* bias = nullptr;
};

// widest output-channel tile that still gives the chip a few workgroups per CU
static int sp_choose_nt(int H, int W, int cout) {
    const int tiles = ((W + kSpTileW - 1) / kSpTileW) * ((H + kSpTileH - 1) / kSpTileH);
    const int ct = (cout + 31) / 32;
    for (int nt = 4; nt > 1; nt >>= 1)
        if (ct % nt == 0 && tiles * (ct / nt) >= 1024) return nt;
    return 1;
}

template <int NT>
static void sp_launch_nt(hipStream_t s, dim3 grid, int taps, bool pool, const SpConvArgs& a) {
    if (taps == 9 && pool)
        hipLaunchKernelGGL((sp_conv_mfma_kernel<NT, 9, true>), grid, dim3(256), 0, s, a);
    else if (taps == 9)
        hipLaunchKernelGGL((sp_conv_mfma_kernel<NT, 9, false>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((sp_conv_mfma_kernel<NT, 1, false>), grid, dim3(256), 0, s, a);
}

static void sp_launch_conv(hipStream_t s, const SpLayer& L, const float* in, int in_stride, float* out, int out_stride, int H,
                           int W) {
    SpConvArgs a;
    a.in = in, a.wpack = L.wpack, a.bias = L.bias, a.out = out;
    a.in_stride = in_stride, a.out_stride = out_stride;
    a.H = H, a.W = W, a.cin = L.cin, a.cout = L.cout, a.relu = L.relu ? 1 : 0;
    const dim3 grid((W + kSpTileW - 1) / kSpTileW, (H + kSpTileH - 1) / kSpTileH, (L.cout + 32 * L.nt - 1) / (32 * L.nt));
    if (L.nt == 4)
        sp_launch_nt<4>(s, grid, L.taps, L.pool, a);
    else if (L.nt == 2)
        sp_launch_nt<2>(s, grid, L.taps, L.pool, a);
    else
        sp_launch_nt<1>(s, grid, L.taps, L.pool, a);
}

static SpConvArgs sp_conv_args(const SpLayer& L, const float* in, int in_stride, float* out, int out_stride, int H, int W) {
    SpConvArgs a;
    a.in = in, a.wpack = L.wpack, a.bias = L.bias, a.out = out;
    a.in_stride = in_stride, a.out_stride = out_stride;
    a.H = H, a.W = W, a.cin = L.cin, a.cout = L.cout, a.relu = L.relu ? 1 : 0;
    return a;
}

// the two 1x1 heads (both NT = 1, no ReLU, same image) as one launch
static void sp_launch_head_pair(hipStream_t s, const SpConvArgs& a, const SpConvArgs& b) {
    SpConvPair q;
    q.a = a, q.b = b, q.za = (a.cout + 31) / 32;
    const dim3 grid((a.W + kSpTileW - 1) / kSpTileW, (a.H + kSpTileH - 1) / kSpTileH, q.za + (b.cout + 31) / 32);
    hipLaunchKernelGGL((sp_conv_mfma_pair_kernel<1, 1, false>), grid, dim3(256), 0, s, q);
}

}  // namespace mmf

struct mmf_superpoint {
    mmf_ctx* ctx = nullptr;
    int max_w = 0, max_h = 0, max_kp = 0;
    float *w1a = nullptr, *b1a = nullptr;  // conv1a: [9][64] + [64]
    // conv1b conv2a conv2b conv3a conv3b conv4a conv4b, {convPa | convDa} fused (512 outputs), convPb, convDb
    mmf::SpLayer L[10];
    float *act0 = nullptr, *act1 = nullptr, *head = nullptr, *semi = nullptr, *desc = nullptr, *heat = nullptr;
    uint8_t* state = nullptr;
    unsigned *flags = nullptr, *prefix = nullptr, *block_sums = nullptr, *counters = nullptr;
    unsigned* host_counters = nullptr;  // pinned
    int* kp_xy = nullptr;
    float *kp_conf = nullptr, *kp_desc = nullptr;
    int cur_w = 0, cur_h = 0;
    void* slab = nullptr;
};

static int sp_upload(float** dst, const float* src, size_t n, hipStream_t s) {
    MMF_HIP_TRY(hipMalloc(dst, n * sizeof(float)));
    MMF_HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyHostToDevice, s));
    MMF_HIP_TRY(hipStreamSynchronize(s));  // src may be a temporary
    return MMF_OK;
}

static int sp_make_layer(mmf::SpLayer& L, const float* w, const float* b, int cin, int cout, int taps, bool pool, bool relu, int H,
                         int W, hipStream_t s) {
    L.cin = cin, L.cout = cout, L.taps = taps, L.pool = pool, L.relu = relu;
    L.nt = mmf::sp_choose_nt(H, W, cout);
    const std::vector<float> packed = mmf::sp_pack_weights(w, cin, cout, taps, L.nt);
    if (int rc = sp_upload(&L.wpack, packed.data(), packed.size(), s)) return rc;
    return sp_upload(&L.bias, b, (size_t)cout, s);
}

extern "C" void mmf_superpoint_destroy(mmf_superpoint* sp) {
    if (!sp) return;
    (void)hipSetDevice(sp->ctx->device);
    (void)hipStreamSynchronize(sp->ctx->stream);
    (void)hipFree(sp->w1a), (void)hipFree(sp->b1a);
    for (auto& L : sp->L) (void)hipFree(L.wpack), (void)hipFree(L.bias);
    (void)hipFree(sp->slab);
    (void)hipHostFree(sp->host_counters);
    delete sp;
}

extern "C" int mmf_superpoint_create(mmf_ctx* c, const float* const* weights, int max_width, int max_height, int max_keypoints,
                                     mmf_superpoint** out) {
    MMF_REQUIRE(c && weights && out, "mmf_superpoint_create: null argument");
    for (int i = 0; i < 24; ++i) MMF_REQUIRE(weights[i] != nullptr, "mmf_superpoint_create: 24 weight arrays expected");
    MMF_REQUIRE(max_width >= 8 && max_height >= 8 && max_width % 8 == 0 && max_height % 8 == 0,
                "mmf_superpoint_create: image sides must be multiples of 8");
    MMF_REQUIRE(max_keypoints > 0, "mmf_superpoint_create: max_keypoints must be positive");
    MMF_HIP_TRY(hipSetDevice(c->device));
    mmf_superpoint* sp = new (std::nothrow) mmf_superpoint();
    MMF_REQUIRE(sp != nullptr, "mmf_superpoint_create: out of host memory");
    sp->ctx = c, sp->max_w = max_width, sp->max_h = max_height, sp->max_kp = max_keypoints;
    int rc = MMF_OK;
    auto guard = [&](int r) {
        if (r && !rc) rc = r;
        return r == MMF_OK;
    };
    // conv1a [64][1][3][3] -> [tap][co]
    {
        float w[9 * 64];
        for (int co = 0; co < 64; ++co)
            for (int tap = 0; tap < 9; ++tap) w[tap * 64 + co] = weights[0][co * 9 + tap];
        guard(sp_upload(&sp->w1a, w, 9 * 64, c->stream)) && guard(sp_upload(&sp->b1a, weights[1], 64, c->stream));
    }
    const int H = max_height, W = max_width;
    static const int cin[7] = {64, 64, 64, 64, 128, 128, 128}, cout[7] = {64, 64, 64, 128, 128, 128, 128};
    static const int shift[7] = {0, 1, 1, 2, 2, 3, 3};
    static const bool pool[7] = {true, false, true, false, true, false, false};
    for (int l = 0; l < 7 && !rc; ++l)
        guard(sp_make_layer(sp->L[l], weights[2 * (l + 1)], weights[2 * (l + 1) + 1], cin[l], cout[l], 9, pool[l], true,
                            H >> shift[l], W >> shift[l], c->stream));
    if (!rc) {  // detector and descriptor 3x3 heads read the same input: one launch with 512 outputs
        std::vector<float> w((size_t)512 * 128 * 9), b(512);
        std::memcpy(w.data(), weights[16], sizeof(float) * 256 * 128 * 9);
        std::memcpy(w.data() + (size_t)256 * 128 * 9, weights[20], sizeof(float) * 256 * 128 * 9);
        std::memcpy(b.data(), weights[17], sizeof(float) * 256);
        std::memcpy(b.data() + 256, weights[21], sizeof(float) * 256);
        guard(sp_make_layer(sp->L[7], w.data(), b.data(), 128, 512, 9, false, true, H >> 3, W >> 3, c->stream));
    }
    if (!rc) guard(sp_make_layer(sp->L[8], weights[18], weights[19], 256, 65, 1, false, false, H >> 3, W >> 3, c->stream));
    if (!rc) guard(sp_make_layer(sp->L[9], weights[22], weights[23], 256, 256, 1, false, false, H >> 3, W >> 3, c->stream));

    const size_t npix = (size_t)H * W, ncell = npix / 64;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t o_act0 = carve(npix * 64 * 4), o_act1 = carve(npix / 4 * 64 * 4);
    const size_t o_head = carve(ncell * 512 * 4), o_semi = carve(ncell * 65 * 4), o_desc = carve(ncell * 256 * 4);
    const size_t o_heat = carve(npix * 4), o_state = carve(npix);
    const size_t o_flags = carve(npix * 4), o_prefix = carve(npix * 4);
    const size_t o_bsum = carve(((npix + mmf::kScanTile - 1) / mmf::kScanTile + 1) * 4), o_cnt = carve(64);
    const size_t o_xy = carve(npix * 8), o_conf = carve(npix * 4), o_kdesc = carve((size_t)max_keypoints * 256 * 4);
    if (!rc) {
        hipError_t e = hipMalloc(&sp->slab, off);
        if (e != hipSuccess) rc = fail(MMF_ERR_HIP, std::string("mmf_superpoint_create: hipMalloc: ") + hipGetErrorString(e));
    }
    if (!rc) {
        hipError_t e = hipHostMalloc(&sp->host_counters, 64, hipHostMallocDefault);
        if (e != hipSuccess) rc = fail(MMF_ERR_HIP, std::string("mmf_superpoint_create: hipHostMalloc: ") + hipGetErrorString(e));
    }
    if (rc) {
        mmf_superpoint_destroy(sp);
        return rc;
    }
    char* base = static_cast<char*>(sp->slab);
    sp->act0 = (float*)(base + o_act0), sp->act1 = (float*)(base + o_act1);
    sp->head = (float*)(base + o_head), sp->semi = (float*)(base + o_semi), sp->desc = (float*)(base + o_desc);
    sp->heat = (float*)(base + o_heat), sp->state = (uint8_t*)(base + o_state);
    sp->flags = (unsigned*)(base + o_flags), sp->prefix = (unsigned*)(base + o_prefix);
    sp->block_sums = (unsigned*)(base + o_bsum), sp->counters = (unsigned*)(base + o_cnt);
    sp->kp_xy = (int*)(base + o_xy), sp->kp_conf = (float*)(base + o_conf), sp->kp_desc = (float*)(base + o_kdesc);
    *out = sp;
    return MMF_OK;
}

// the network: image -> semi [H/8][W/8][65], desc [H/8][W/8][256] (normalised), heat [H][W]; all on the device
extern "C" int mmf_superpoint_forward(mmf_superpoint* sp, const uint8_t* image, int width, int height, int channels) {
    MMF_REQUIRE(sp && image, "mmf_superpoint_forward: null argument");
    MMF_REQUIRE(channels == 1 || channels == 3 || channels == 4, "mmf_superpoint_forward: 1, 3 or 4 channels");
    MMF_REQUIRE(width >= 8 && height >= 8 && width % 8 == 0 && height % 8 == 0,
                "mmf_superpoint_forward: image sides must be multiples of 8");
    MMF_REQUIRE(width <= sp->max_w && height <= sp->max_h && (size_t)width * height <= (size_t)sp->max_w * sp->max_h,
                "mmf_superpoint_forward: image larger than the object was created for");
    mmf_ctx* c = sp->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int H = height, W = width, npix = H * W;
    using namespace mmf;
    hipLaunchKernelGGL(sp_conv1a_kernel, grid1d((size_t)npix * 4), dim3(256), 0, s, image, channels, H, W, sp->w1a, sp->b1a,
                       sp->act0);
    sp_launch_conv(s, sp->L[0], sp->act0, 64, sp->act1, 64, H, W);              // conv1b + pool
    sp_launch_conv(s, sp->L[1], sp->act1, 64, sp->act0, 64, H / 2, W / 2);      // conv2a
    sp_launch_conv(s, sp->L[2], sp->act0, 64, sp->act1, 64, H / 2, W / 2);      // conv2b + pool
    sp_launch_conv(s, sp->L[3], sp->act1, 64, sp->act0, 128, H / 4, W / 4);     // conv3a
    sp_launch_conv(s, sp->L[4], sp->act0, 128, sp->act1, 128, H / 4, W / 4);    // conv3b + pool
    sp_launch_conv(s, sp->L[5], sp->act1, 128, sp->act0, 128, H / 8, W / 8);    // conv4a
    sp_launch_conv(s, sp->L[6], sp->act0, 128, sp->act1, 128, H / 8, W / 8);    // conv4b
    sp_launch_conv(s, sp->L[7], sp->act1, 128, sp->head, 512, H / 8, W / 8);    // convPa | convDa
    if (sp->L[8].nt == 1 && sp->L[9].nt == 1) {  // convPb + convDb: one launch
        sp_launch_head_pair(s, sp_conv_args(sp->L[8], sp->head, 512, sp->semi, 65, H / 8, W / 8),
                            sp_conv_args(sp->L[9], sp->head + 256, 512, sp->desc, 256, H / 8, W / 8));
    } else {
        sp_launch_conv(s, sp->L[8], sp->head, 512, sp->semi, 65, H / 8, W / 8);         // convPb
        sp_launch_conv(s, sp->L[9], sp->head + 256, 512, sp->desc, 256, H / 8, W / 8);  // convDb
    }
    const int ncell = npix / 64;
    hipLaunchKernelGGL((sp_l2_normalize_kernel<256>), dim3((ncell + 3) / 4), dim3(256), 0, s, sp->desc, ncell);
    hipLaunchKernelGGL(sp_heatmap_kernel, dim3((ncell + 3) / 4), dim3(256), 0, s, sp->semi, H / 8, W / 8, sp->heat);
    MMF_HIP_TRY(hipGetLastError());
    sp->cur_w = W, sp->cur_h = H;
    return MMF_OK;
}

// copies one of the last forward pass's results to the host: 0 semi, 1 coarse descriptors, 2 heat map
extern "C" int mmf_superpoint_download(mmf_superpoint* sp, int which, float* host, size_t count) {
    MMF_REQUIRE(sp && host, "mmf_superpoint_download: null argument");
    MMF_REQUIRE(sp->cur_w > 0, "mmf_superpoint_download: no forward pass yet");
    const size_t ncell = (size_t)sp->cur_w * sp->cur_h / 64;
    const float* src = which == 0 ? sp->semi : which == 1 ? sp->desc : which == 2 ? sp->heat : nullptr;
    const size_t n = which == 0 ? ncell * 65 : which == 1 ? ncell * 256 : ncell * 64;
    MMF_REQUIRE(src != nullptr && count == n, "mmf_superpoint_download: bad selector or size");
    MMF_HIP_TRY(hipSetDevice(sp->ctx->device));
    MMF_HIP_TRY(hipMemcpyAsync(host, src, n * sizeof(float), hipMemcpyDeviceToHost, sp->ctx->stream));
    MMF_HIP_TRY(hipStreamSynchronize(sp->ctx->stream));
    return MMF_OK;
}

// SuperPoint::getFeatures (Core/MultiMotionFusion.cpp:233).  `image` is a DEVICE pointer (interleaved u8).
// xy [max_keypoints][2] pixel coordinates, conf [max_keypoints], desc [max_keypoints][256]: HOST arrays,
// strongest keypoint first; *count = number written.  The caller normalises xy by (width, height)
// (PointTracker.cpp:40-41).  Synchronous.
extern "C" int mmf_superpoint_get_features(mmf_superpoint* sp, const uint8_t* image, int width, int height, int channels,
                                           float conf_thresh, int nms_dist, int border, int* xy, float* conf, float* desc,
                                           int* count) {
    MMF_REQUIRE(sp && xy && conf && desc && count, "mmf_superpoint_get_features: null argument");
    MMF_REQUIRE(nms_dist >= 0 && border >= 0, "mmf_superpoint_get_features: negative radius");
    int rc = mmf_superpoint_forward(sp, image, width, height, channels);
    if (rc) return rc;
    mmf_ctx* c = sp->ctx;
    hipStream_t s = c->stream;
    const int H = height, W = width, npix = H * W;
    using namespace mmf;
    hipLaunchKernelGGL(sp_nms_init_kernel, grid1d(npix), dim3(256), 0, s, sp->heat, npix, conf_thresh, sp->state);
    // passes until no candidate is undecided; the strongest undecided candidate always decides, so the loop
    // ends after at most npix passes (a handful on real heat maps); checked every 4 passes
    for (int batch = 0;; ++batch) {
        MMF_REQUIRE(batch <= npix / 4 + 1, "mmf_superpoint_get_features: suppression did not converge");
        for (int k = 0; k < 3; ++k)
            hipLaunchKernelGGL(sp_nms_pass_kernel, grid1d(npix), dim3(256), 0, s, sp->heat, H, W, nms_dist, sp->state,
                               &sp->counters[1]);
        MMF_HIP_TRY(hipMemsetAsync(&sp->counters[0], 0, 4, s));
        hipLaunchKernelGGL(sp_nms_pass_kernel, grid1d(npix), dim3(256), 0, s, sp->heat, H, W, nms_dist, sp->state,
                           &sp->counters[0]);
        MMF_HIP_TRY(hipGetLastError());
        MMF_HIP_TRY(hipMemcpyAsync(sp->host_counters, sp->counters, 4, hipMemcpyDeviceToHost, s));
        MMF_HIP_TRY(hipStreamSynchronize(s));
        if (sp->host_counters[0] == 0) break;
    }
    hipLaunchKernelGGL(sp_keep_flag_kernel, grid1d(npix), dim3(256), 0, s, sp->state, H, W, border, sp->flags);
    rc = device_scan(c, sp->flags, (unsigned)npix, sp->prefix, sp->block_sums, &sp->counters[2]);
    if (rc) return rc;
    hipLaunchKernelGGL(sp_keep_scatter_kernel, grid1d(npix), dim3(256), 0, s, sp->flags, sp->prefix, sp->heat, npix, W, npix,
                       sp->kp_xy, sp->kp_conf);
    MMF_HIP_TRY(hipGetLastError());
    MMF_HIP_TRY(hipMemcpyAsync(sp->host_counters, sp->counters, 16, hipMemcpyDeviceToHost, s));
    MMF_HIP_TRY(hipStreamSynchronize(s));
    const int found = (int)sp->host_counters[2];
    *count = 0;
    if (found == 0) return MMF_OK;
    // strongest first (ties in row-major order, which is the order they were compacted in): a permutation
    // of at most a few thousand records, done on the host like the final Eigen packing of the reference
    std::vector<int> hxy((size_t)found * 2);
    std::vector<float> hconf((size_t)found);
    MMF_HIP_TRY(hipMemcpyAsync(hxy.data(), sp->kp_xy, hxy.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    MMF_HIP_TRY(hipMemcpyAsync(hconf.data(), sp->kp_conf, hconf.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    MMF_HIP_TRY(hipStreamSynchronize(s));
    std::vector<int> order((size_t)found);
    for (int k = 0; k < found; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return hconf[a] > hconf[b]; });
    const int n = found < sp->max_kp ? found : sp->max_kp;
    for (int k = 0; k < n; ++k) xy[2 * k] = hxy[2 * order[k]], xy[2 * k + 1] = hxy[2 * order[k] + 1], conf[k] = hconf[order[k]];
    MMF_HIP_TRY(hipMemcpyAsync(sp->kp_xy, xy, (size_t)n * 2 * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(sp_sample_kernel, dim3(n), dim3(256), 0, s, sp->desc, H / 8, W / 8, sp->kp_xy, H, W, sp->kp_desc);
    MMF_HIP_TRY(hipGetLastError());
    MMF_HIP_TRY(hipMemcpyAsync(desc, sp->kp_desc, (size_t)n * 256 * sizeof(float), hipMemcpyDeviceToHost, s));
    MMF_HIP_TRY(hipStreamSynchronize(s));
    *count = n;
    return MMF_OK;
}

// one convolution layer by itself (tests, tools): in [H][W][cin] and out on the device, w [cout][cin][k][k]
// and bias on the HOST; taps = 9 or 1; pool: 2x2 max after the ReLU; nt = 0 picks the tile width.
extern "C" int mmf_superpoint_conv(mmf_ctx* c, const float* in, int height, int width, int cin, const float* w, const float* bias,
                                   int cout, int taps, int relu, int pool, int nt, float* out) {
    MMF_REQUIRE(c && in && w && bias && out, "mmf_superpoint_conv: null argument");
    MMF_REQUIRE(taps == 9 || taps == 1, "mmf_superpoint_conv: 3x3 or 1x1");
    MMF_REQUIRE(cin > 0 && cin % 32 == 0 && cout > 0, "mmf_superpoint_conv: cin must be a multiple of 32");
    MMF_REQUIRE(width > 0 && height > 0 && (!pool || (taps == 9 && width % 2 == 0 && height % 2 == 0)),
                "mmf_superpoint_conv: bad size");
    MMF_REQUIRE(nt == 0 || nt == 1 || nt == 2 || nt == 4, "mmf_superpoint_conv: nt must be 0, 1, 2 or 4");
    MMF_HIP_TRY(hipSetDevice(c->device));
    mmf::SpLayer L;
    L.cin = cin, L.cout = cout, L.taps = taps, L.pool = pool != 0, L.relu = relu != 0;
    L.nt = nt ? nt : mmf::sp_choose_nt(height, width, cout);
    const std::vector<float> packed = mmf::sp_pack_weights(w, cin, cout, taps, L.nt);
    int rc = sp_upload(&L.wpack, packed.data(), packed.size(), c->stream);
    if (!rc) rc = sp_upload(&L.bias, bias, (size_t)cout, c->stream);
    if (!rc) {
        mmf::sp_launch_conv(c->stream, L, in, cin, out, cout, height, width);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(MMF_ERR_HIP, std::string("mmf_superpoint_conv: ") + hipGetErrorString(e));
    }
    (void)hipFree(L.wpack), (void)hipFree(L.bias);
    return rc;
}
