// mmf_hip.hip -- C ABI (include/mmf_hip.h) over the gfx950 tracking kernels.
// Built only for gfx950 with -ffp-contract=off (see multimotionfusion_amd/build.py).
#include "../../include/mmf_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "map_kernels.hpp"
#include "prep_batch.hpp"
#include "pose_algebra.hpp"
#include "track_kernels.hpp"
#include "tunables.hpp"
#include "gn_fused.hpp"

using namespace mmf;

// The enqueue side of a chain of dependent kernel launches: launches go out in call order on one stream, the first error
// is kept (`flush`).  (Rounds 2-3 and again round 5 could replay such a chain as a hipGraph -- one host call per chain, the
// arguments of the nodes that changed replaced.  With one model a graph launch reaches the GPU ~10 us later than the first
// kernel of a launch-by-launch chain; with eight models it takes the call's start from 330 to 80 us of the calling thread's
// time and the frame is no shorter: LABNOTES.md.)
struct Enqueuer {
    hipStream_t stream = nullptr;
    hipError_t err = hipSuccess;

    explicit Enqueuer(hipStream_t s) : stream(s) {}

    template <typename... KArgs, typename... Args>
    void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args&&... args) {
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, static_cast<KArgs>(args)...);
        const hipError_t e = hipGetLastError();
        if (err == hipSuccess) err = e;
    }
    hipError_t flush() const { return err; }  // everything launched so far is on the stream; the first error, if any
};

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define MMF_HIP_TRY(expr)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail(MMF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__) + " (" +  \
                                         __FILE__ + ":" + std::to_string(__LINE__) + ")");        \
    } while (0)

#define MMF_REQUIRE(cond, msg)                                  \
    do {                                                        \
        if (!(cond)) return fail(MMF_ERR_INVALID, (msg));       \
    } while (0)

// Waiting for a short dependent chain: hipStreamSynchronize / hipEventSynchronize park the thread and wake it ~20-30 us
// after the work is done (measured as idle gaps in the kernel trace); the frame has two such waits on its critical path.
// Poll first (the waits are a few hundred microseconds at most), park only when the work takes unusually long.
static hipError_t wait_stream(hipStream_t s) {
    for (int i = 0; i < 200000; ++i) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
    }
    return hipStreamSynchronize(s);
}
extern "C" int mmf_abi_version(void) { return MMF_ABI_VERSION; }
extern "C" const char* mmf_last_error(void) { return g_last_error.c_str(); }

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
constexpr int kMaxGrid = 2048;
constexpr int kMaxIcpGrid = 8192;  // the single-pass ICP producer needs one workgroup per BLOCK * PX pixels  // workgroups per reduction launch (grid-stride beyond)

struct mmf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    float* partials_f = nullptr;    // kMaxGrid records: in-launch (ticket) reductions
    float* partials_icp = nullptr;  // kMaxGrid records: icp_kernel -> its consumer
    int2* partials_res = nullptr;   // kMaxGrid {count, sigma} records: rgb_residual_kernel -> its consumer
    unsigned* ticket = nullptr;
    OdomState* scratch_state = nullptr;  // for the stand-alone *Step entry points
    OdomState* host_state = nullptr;     // pinned staging
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* match_ws = nullptr;  // descriptor matcher workspace: norms + arg-min keys, grown on demand
    size_t match_ws_rows = 0;
    void* slic_ws = nullptr;  // super-pixel resampling workspace (boxes, counts, sums), grown on demand
    size_t slic_ws_n = 0;
    char arch[64] = {0};
    int cu_count = 0;  // compute units of the device (the one-launch Gauss-Newton chain needs its grid resident at once)
};

static std::atomic<int> g_xcd_forced{-1};  // mmf_debug_set_xcd
extern "C" int mmf_ctx_create(int device, void* stream, int private_stream, mmf_ctx** out) {
    MMF_REQUIRE(out != nullptr, "mmf_ctx_create: out is null");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(MMF_ERR_NO_DEVICE, "mmf_ctx_create: no HIP device visible");
    MMF_REQUIRE(device >= 0 && device < count, "mmf_ctx_create: device index out of range");
    MMF_HIP_TRY(hipSetDevice(device));
    (void)tunables();  // the environment switches are read here, once
    mmf_ctx* c = new (std::nothrow) mmf_ctx();
    MMF_REQUIRE(c != nullptr, "mmf_ctx_create: out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    MMF_HIP_TRY(hipGetDeviceProperties(&prop, device));
    std::snprintf(c->arch, sizeof(c->arch), "%s", prop.gcnArchName);
    c->cu_count = prop.multiProcessorCount;
    if (std::strncmp(c->arch, "gfx950", 6) != 0) {
        std::string m = std::string("mmf_ctx_create: this library is built for gfx950 only, device is ") + c->arch;
        delete c;
        return fail(MMF_ERR_NO_DEVICE, m);
    }
    if (private_stream) {
        MMF_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    } else {
        c->stream = (hipStream_t)stream;
    }
    MMF_HIP_TRY(hipMalloc(&c->partials_f, sizeof(float) * kMaxGrid * kPartialStride));
    MMF_HIP_TRY(hipMalloc(&c->partials_icp, sizeof(float) * kMaxIcpGrid * kPartialStride));
    MMF_HIP_TRY(hipMalloc(&c->partials_res, sizeof(int2) * kMaxGrid));
    MMF_HIP_TRY(hipMalloc(&c->ticket, sizeof(unsigned) * kTicketWords));
    MMF_HIP_TRY(hipMemsetAsync(c->ticket, 0, sizeof(unsigned) * kTicketWords, c->stream));
    MMF_HIP_TRY(hipMalloc(&c->scratch_state, sizeof(OdomState)));
    MMF_HIP_TRY(hipMemsetAsync(c->scratch_state, 0, sizeof(OdomState), c->stream));
    MMF_HIP_TRY(hipHostMalloc(&c->host_state, sizeof(OdomState), hipHostMallocDefault));
    MMF_HIP_TRY(hipEventCreate(&c->ev0));
    MMF_HIP_TRY(hipEventCreate(&c->ev1));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    {  // (surfel_kernels.hpp: xcd_block)
        const int on = g_xcd_forced.load() < 0 ? (tunables().xcd_blocks ? 1 : 0) : g_xcd_forced.load();
        MMF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_blocks), &on, sizeof(on)));
    }
    *out = c;
    return MMF_OK;
}
// test / A-B hook: every XCD works on one contiguous eighth of a surfel pass's blocks (1, the default) or the blocks are dealt
// round-robin as the workgroups are (0); -1 = the default (MMF_XCD).  A device-wide word: set while no pass is running.
extern "C" unsigned mmf_debug_xcd_block(unsigned block, unsigned blocks) { return xcd_block_of(block, blocks); }  // (host: no device needed)
extern "C" int mmf_debug_set_xcd(int on) {
    g_xcd_forced.store(on < 0 ? -1 : (on ? 1 : 0));
    const int v = on < 0 ? (tunables().xcd_blocks ? 1 : 0) : (on ? 1 : 0);
    MMF_HIP_TRY(hipDeviceSynchronize());
    MMF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_blocks), &v, sizeof(v)));
    return MMF_OK;
}

extern "C" void mmf_ctx_destroy(mmf_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->partials_f);
    (void)hipFree(c->partials_icp);
    (void)hipFree(c->partials_res);
    (void)hipFree(c->ticket);
    (void)hipFree(c->scratch_state);
    (void)hipHostFree(c->host_state);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    (void)hipFree(c->match_ws);
    (void)hipFree(c->slic_ws);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int mmf_ctx_synchronize(mmf_ctx* c) {
    MMF_REQUIRE(c != nullptr, "mmf_ctx_synchronize: ctx is null");
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    return MMF_OK;
}

extern "C" void* mmf_ctx_stream(mmf_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int mmf_ctx_device_name(mmf_ctx* c, char* buf, size_t buflen) {
    MMF_REQUIRE(c && buf && buflen > 0, "mmf_ctx_device_name: bad arguments");
    std::snprintf(buf, buflen, "%s", c->arch);
    return MMF_OK;
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
static inline dim3 tile_grid(int cols, int rows) {
    return dim3((cols + kTileX - 1) / kTileX, (rows + kTileY - 1) / kTileY);
}
static inline dim3 tile_block() { return dim3(kTileX, kTileY); }

static inline int stride_elems(size_t step_bytes, int cols, size_t elem) {
    return step_bytes ? (int)(step_bytes / elem) : cols;
}

static inline int reduce_grid(int n, int per_block) {
    int g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return g;
}

static inline LevelIntr level_intr(float fx, float fy, float cx, float cy, int level) {
    const int div = 1 << level;  // types.cuh:94-98
    return LevelIntr{fx / div, fy / div, cx / div, cy / div};
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ICP reduction launch.  `variant` = GEN * 1000000 + PX * 10000 + BLOCK picks the kernel (0 = the
// tuned default): GEN 2 = icp_kernel2 gathering from a.prev_packed when it is set, GEN 1 = icp_kernel2
// with planar gathers, PX pixels per lane (1, 2 or 4; the vector forms need cols % PX == 0 and
// aligned rows), BLOCK threads per workgroup.  tools/sweep_icp.py on MI355X: every geometry of
// icp_kernel2 lands within 0.2 us at 640x480 (the launch is bound by its two memory round trips
// and the dispatch floor, no longer by instruction issue), so one default serves all levels.
static int icp_default_variant(int /*npix*/) {
    const int forced = tunables().icp_variant;  // tuning aid: MMF_ICP_VARIANT=<GEN*1000000 + PX*10000 + BLOCK>
    return forced > 0 ? forced : 2020256;
}

template <int W, int NV, int BLOCK, bool PACKED, int MODE>
static int launch_icp2_variant(mmf_ctx* c, OdomState* st, const IcpArgs& a, float* partials) {
    const int grid = (a.cols * a.rows + BLOCK * W * NV - 1) / (BLOCK * W * NV);  // <= kMaxIcpGrid, checked by the caller
    hipLaunchKernelGGL((icp_kernel2<W, NV, BLOCK, PACKED, MODE>), dim3(grid), dim3(BLOCK), 0, c->stream, st, a,
                       partials);
    return grid;
}

// largest float x with sqrtf(x) <= t, and smallest x with sqrtf(x) >= t: sqrtf is monotonic and
// correctly rounded, so comparing squares against these is EXACTLY the reference's comparison of
// the norms (reduce.cu:301-306) -- see icp_rows_v
static float sq_max_le(float t) {
    float x = t * t;
    while (std::sqrt(std::nextafter(x, INFINITY)) <= t) x = std::nextafter(x, INFINITY);
    while (std::sqrt(x) > t) x = std::nextafter(x, -INFINITY);
    return x;
}
static float sq_min_ge(float t) {
    float x = t * t;
    while (std::sqrt(std::nextafter(x, -INFINITY)) >= t) x = std::nextafter(x, -INFINITY);
    while (std::sqrt(x) < t) x = std::nextafter(x, INFINITY);
    return x;
}
static void icp_args_derive(IcpArgs& a) {
    a.dist_sq_max = sq_max_le(a.dist_thres);
    a.sine_sq_min = sq_min_ge(a.angle_thres);
    // i / cols == umulhi(i, magic) as long as i * cols < 2^32; larger images divide (0 selects that)
    a.cols_magic = (long long)a.cols * a.rows * a.cols < (1ll << 32) ? (unsigned)((1ull << 32) / (unsigned)a.cols) + 1u : 0u;
}

// pixels per lane the vector loads allow for these maps (4, 2 or 1)
static int icp_max_px(const IcpArgs& a, int px) {
    auto ok = [&](int k) {
        const uintptr_t m = (uintptr_t)k * 4 - 1;
        return (a.cols % k == 0) && (a.vmap_curr.stride % k == 0) && (a.nmap_curr.stride % k == 0) &&
               ((uintptr_t)a.vmap_curr.base & m) == 0 && ((uintptr_t)a.nmap_curr.base & m) == 0 &&
               (!a.err_map || (((uintptr_t)a.err_map & m) == 0 && a.err_stride % k == 0));
    };
    while (px > 1 && !ok(px)) px /= 2;
    return px;
}
// can ONE pass of kMaxIcpGrid workgroups cover the image?  (larger ones: the multi-pass instantiation)
static bool icp2_fits(const IcpArgs& a, int block, int px) {
    const long long n = (long long)a.cols * a.rows;
    return n * a.cols < (1ll << 32) && (long long)kMaxIcpGrid * block * px >= n;
}

// launches the ICP producer; *records_out = number of partial records it writes to c->partials_icp
template <int MODE>
static hipError_t launch_icp(mmf_ctx* c, OdomState* st, IcpArgs a, int variant = 0, int* records_out = nullptr,
                             float* partials = nullptr) {
    if (!partials) partials = c->partials_icp;
    if (variant == 0) variant = icp_default_variant(a.cols * a.rows);
    icp_args_derive(a);
    const int gen = variant / 1000000;
    const int px = icp_max_px(a, (variant / 10000) % 100);
    int block = variant % 10000;
    if (block != 64 && block != 128) block = 256;
    int grid = 0;
    if (icp2_fits(a, block, px)) {
        const bool packed = gen != 1 && a.prev_packed != nullptr;
#define MMF_ICP2(W, NV, B)                                                                 \
    grid = packed ? launch_icp2_variant<W, NV, B, true, MODE>(c, st, a, partials) : launch_icp2_variant<W, NV, B, false, MODE>(c, st, a, partials)
        switch (px * 10000 + block) {
            case 40256: MMF_ICP2(2, 2, 256); break;
            case 40128: MMF_ICP2(2, 2, 128); break;
            case 20128: MMF_ICP2(2, 1, 128); break;
            case 20064: MMF_ICP2(2, 1, 64); break;
            case 10128: MMF_ICP2(1, 1, 128); break;
            case 10064: MMF_ICP2(1, 1, 64); break;
            case 10256: MMF_ICP2(1, 1, 256); break;
            default: MMF_ICP2(2, 1, 256);
        }
#undef MMF_ICP2
    } else {  // very large images: kMaxIcpGrid workgroups walk the image, one pixel per lane and pass
        grid = kMaxIcpGrid;
        const bool packed = gen != 1 && a.prev_packed != nullptr;
        if (packed)
            hipLaunchKernelGGL((icp_kernel2<1, 1, 256, true, MODE, true>), dim3(grid), dim3(256), 0, c->stream, st, a, partials);
        else
            hipLaunchKernelGGL((icp_kernel2<1, 1, 256, false, MODE, true>), dim3(grid), dim3(256), 0, c->stream, st, a, partials);
    }
    if (records_out) *records_out = grid;
    return hipGetLastError();
}

static void unpack_se3_host(const float* tot, float* A, float* b) {  // reduce.cu:458-472
    int shift = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 7; ++j) {
            const float v = tot[shift++];
            if (j == 6)
                b[i] = v;
            else
                A[j * 6 + i] = A[i * 6 + j] = v;
        }
}

// ---------------------------------------------------------------------------------------------
// stand-alone device entry points (cudafuncs.cuh)
// ---------------------------------------------------------------------------------------------
extern "C" int mmf_icp_step(mmf_ctx* c, const float Rcurr[9], const float tcurr[3], const float* vmap_curr,
                            size_t vmap_curr_step, const float* nmap_curr, size_t nmap_curr_step,
                            const float Rprev_inv[9], const float tprev[3], const mmf_camera* intr,
                            const float* vmap_g_prev, size_t vmap_g_prev_step, const float* nmap_g_prev,
                            size_t nmap_g_prev_step, float dist_thres, float angle_thres, int cols, int rows,
                            float* A_host, float* b_host, float* residual_host, float* err_map_dev,
                            size_t err_map_step) {
    MMF_REQUIRE(c && Rcurr && tcurr && Rprev_inv && tprev && intr, "mmf_icp_step: null argument");
    MMF_REQUIRE(vmap_curr && nmap_curr && vmap_g_prev && nmap_g_prev, "mmf_icp_step: null map");
    MMF_REQUIRE(A_host && b_host && residual_host, "mmf_icp_step: null output");
    MMF_REQUIRE(cols > 0 && rows > 0 && (long long)cols * rows * 3 < (1ll << 31), "mmf_icp_step: bad size");
    MMF_HIP_TRY(hipSetDevice(c->device));
    OdomState* h = c->host_state;
    std::memcpy(h->Rcurr, Rcurr, sizeof(float) * 9);
    std::memcpy(h->tcurr, tcurr, sizeof(float) * 3);
    std::memcpy(h->Rprev_inv, Rprev_inv, sizeof(float) * 9);
    std::memcpy(h->tprev, tprev, sizeof(float) * 3);
    // Rprev .. tcurr are contiguous at the head of the struct
    MMF_HIP_TRY(hipMemcpyAsync(c->scratch_state, h, offsetof(OdomState, resultRt), hipMemcpyHostToDevice, c->stream));
    IcpArgs a;
    a.vmap_curr = MapView{vmap_curr, stride_elems(vmap_curr_step, cols, 4)};
    a.nmap_curr = MapView{nmap_curr, stride_elems(nmap_curr_step, cols, 4)};
    a.vmap_g_prev = MapView{vmap_g_prev, stride_elems(vmap_g_prev_step, cols, 4)};
    a.nmap_g_prev = MapView{nmap_g_prev, stride_elems(nmap_g_prev_step, cols, 4)};
    a.intr = LevelIntr{intr->fx, intr->fy, intr->cx, intr->cy};
    a.dist_thres = dist_thres;
    a.angle_thres = angle_thres;
    a.cols = cols;
    a.rows = rows;
    a.prev_packed = nullptr;
    a.err_map = err_map_dev;
    a.err_stride = stride_elems(err_map_step, cols, 4);
    int records = 0;
    MMF_HIP_TRY(launch_icp<FINISH_RAW>(c, c->scratch_state, a, 0, &records));
    hipLaunchKernelGGL((icp_finish_kernel<FINISH_RAW>), dim3(1), dim3(256), 0, c->stream, c->scratch_state,
                       c->partials_icp, (unsigned)records, a.intr);
    MMF_HIP_TRY(hipGetLastError());
    float tot[32];
    MMF_HIP_TRY(hipMemcpyAsync(tot, c->scratch_state->out_f, sizeof(float) * 32, hipMemcpyDeviceToHost, c->stream));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    unpack_se3_host(tot, A_host, b_host);
    residual_host[0] = tot[27];
    residual_host[1] = tot[28];
    return MMF_OK;
}

static RgbResidualArgs make_residual_args(float min_scale, const int16_t* dIdx, size_t dIdx_step, const int16_t* dIdy,
                                          size_t dIdy_step, const float* last_depth, size_t ld_step,
                                          const float* next_depth, size_t nd_step, const uint8_t* last_image,
                                          size_t li_step, const uint8_t* next_image, size_t ni_step,
                                          mmf_dataterm* corres, float max_depth_delta, int cols, int rows,
                                          float* err_map, size_t err_step) {
    RgbResidualArgs a;
    a.min_scale = min_scale;
    a.max_depth_delta = max_depth_delta;
    a.dIdx = dIdx;
    a.dIdy = dIdy;
    a.d_stride = stride_elems(dIdx_step, cols, 2);
    (void)dIdy_step;
    a.last_depth = last_depth;
    a.next_depth = next_depth;
    a.ld_stride = stride_elems(ld_step, cols, 4);
    a.nd_stride = stride_elems(nd_step, cols, 4);
    a.last_image = last_image;
    a.next_image = next_image;
    a.li_stride = stride_elems(li_step, cols, 1);
    a.ni_stride = stride_elems(ni_step, cols, 1);
    a.corres = corres;
    a.cols = cols;
    a.rows = rows;
    a.cols_magic = (unsigned)((1ull << 32) / (unsigned)cols) + 1u;
    a.err_map = err_map;
    a.err_stride = stride_elems(err_step, cols, 4);
    a.intr = LevelIntr{0, 0, 0, 0};
    a.extent = nullptr, a.extent_gen = 0u, a.extent_level = 0;
    return a;
}

// 4 pixels per lane needs whole 4-pixel groups per row and 16-byte aligned rows of every image
static bool residual_vec4_ok(const RgbResidualArgs& a) {
    auto al = [](const void* p, uintptr_t n) { return ((uintptr_t)p & (n - 1)) == 0; };
    return a.cols % 4 == 0 && a.ni_stride % 4 == 0 && a.d_stride % 4 == 0 && a.nd_stride % 4 == 0 &&
           al(a.next_image, 4) && al(a.dIdx, 8) && al(a.dIdy, 8) && al(a.next_depth, 16) &&
           (!a.err_map || (al(a.err_map, 16) && a.err_stride % 4 == 0)) && al(a.corres, 16) &&
           (long long)a.cols * a.rows * a.cols < (1ll << 32);  // index / cols by multiply-high
}

extern "C" int mmf_compute_rgb_residual(mmf_ctx* c, float min_scale, const int16_t* dIdx, size_t dIdx_step,
                                        const int16_t* dIdy, size_t dIdy_step, const float* last_depth,
                                        size_t last_depth_step, const float* next_depth, size_t next_depth_step,
                                        const uint8_t* last_image, size_t last_image_step, const uint8_t* next_image,
                                        size_t next_image_step, mmf_dataterm* corres_dev, float max_depth_delta,
                                        const float kt[3], const float krkinv[9], int cols, int rows,
                                        int* sigma_sum_host, int* count_host, float* err_map_dev,
                                        size_t err_map_step) {
    MMF_REQUIRE(c && dIdx && dIdy && last_depth && next_depth && last_image && next_image && corres_dev && kt &&
                    krkinv && sigma_sum_host && count_host,
                "mmf_compute_rgb_residual: null argument");
    MMF_REQUIRE(cols > 0 && rows > 0 && cols < 32768 && rows < 32768, "mmf_compute_rgb_residual: bad size");
    MMF_REQUIRE(dIdx_step == dIdy_step, "mmf_compute_rgb_residual: dIdx/dIdy steps differ");
    MMF_REQUIRE(aligned16(corres_dev), "mmf_compute_rgb_residual: corres must be 16-byte aligned");
    MMF_HIP_TRY(hipSetDevice(c->device));
    OdomState* h = c->host_state;
    std::memcpy(h->krkinv, krkinv, sizeof(float) * 9);
    std::memcpy(h->kt, kt, sizeof(float) * 3);
    MMF_HIP_TRY(hipMemcpyAsync(c->scratch_state->krkinv, h->krkinv, sizeof(float) * 12, hipMemcpyHostToDevice, c->stream));
    RgbResidualArgs a = make_residual_args(min_scale, dIdx, dIdx_step, dIdy, dIdy_step, last_depth, last_depth_step,
                                           next_depth, next_depth_step, last_image, last_image_step, next_image,
                                           next_image_step, corres_dev, max_depth_delta, cols, rows, err_map_dev,
                                           err_map_step);
    const bool vec4 = residual_vec4_ok(a);
    const int grid = reduce_grid(cols * rows, vec4 ? kBlock * 4 : kBlock);
    if (vec4)
        hipLaunchKernelGGL((rgb_residual_kernel<FINISH_RAW, 4>), dim3(grid), dim3(kBlock), 0, c->stream,
                           c->scratch_state, a, c->partials_res);
    else
        hipLaunchKernelGGL((rgb_residual_kernel<FINISH_RAW, 1>), dim3(grid), dim3(kBlock), 0, c->stream,
                           c->scratch_state, a, c->partials_res);
    MMF_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(residual_finish_kernel, dim3(1), dim3(256), 0, c->stream, c->scratch_state, c->partials_res,
                       (unsigned)grid);
    MMF_HIP_TRY(hipGetLastError());
    int tot[2];
    MMF_HIP_TRY(hipMemcpyAsync(tot, c->scratch_state->out_i, sizeof(tot), hipMemcpyDeviceToHost, c->stream));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    *count_host = tot[0];
    *sigma_sum_host = tot[1];
    return MMF_OK;
}

extern "C" int mmf_rgb_step(mmf_ctx* c, const mmf_dataterm* corres_dev, float sigma, const float* cloud_dev, float fx,
                            float fy, const int16_t* dIdx, size_t dIdx_step, const int16_t* dIdy, size_t dIdy_step,
                            float sobel_scale, int cols, int rows, float* A_host, float* b_host) {
    MMF_REQUIRE(c && corres_dev && cloud_dev && dIdx && dIdy && A_host && b_host, "mmf_rgb_step: null argument");
    MMF_REQUIRE(cols > 0 && rows > 0, "mmf_rgb_step: bad size");
    MMF_REQUIRE(dIdx_step == dIdy_step, "mmf_rgb_step: dIdx/dIdy steps differ");
    MMF_REQUIRE(aligned16(corres_dev), "mmf_rgb_step: corres must be 16-byte aligned");
    MMF_HIP_TRY(hipSetDevice(c->device));
    c->host_state->sigmaVal = sigma;
    MMF_HIP_TRY(hipMemcpyAsync(&c->scratch_state->sigmaVal, &c->host_state->sigmaVal, sizeof(float),
                               hipMemcpyHostToDevice, c->stream));
    RgbStepArgs a;
    a.next_level = 0;
    a.final_step = 0;
    a.extent = nullptr, a.extent_gen = 0u, a.extent_level = 0;
    a.cols_magic = 0;
    a.residual_partials = nullptr;
    a.residual_records = 0;
    a.icp_partials = nullptr;
    a.icp_records = 0;
    a.corres = corres_dev;
    a.cloud = cloud_dev, a.cloud4 = nullptr;
    a.fx = fx;
    a.fy = fy;
    a.dIdx = dIdx;
    a.dIdy = dIdy;
    a.d_stride = stride_elems(dIdx_step, cols, 2);
    a.sobel_scale = sobel_scale;
    a.cols = cols;
    a.rows = rows;
    a.intr = LevelIntr{0, 0, 0, 0};
    if ((cols * rows) % 4 == 0) {
        const int grid = reduce_grid(cols * rows, kBlock * 4);
        hipLaunchKernelGGL((rgb_step_kernel<FINISH_RAW, 4>), dim3(grid), dim3(kBlock), 0, c->stream, c->scratch_state,
                           a, c->partials_f, c->ticket, BatchDelta{}, ChainGeom{});
    } else {
        const int grid = reduce_grid(cols * rows, kBlock);
        hipLaunchKernelGGL((rgb_step_kernel<FINISH_RAW, 1>), dim3(grid), dim3(kBlock), 0, c->stream, c->scratch_state,
                           a, c->partials_f, c->ticket, BatchDelta{}, ChainGeom{});
    }
    MMF_HIP_TRY(hipGetLastError());
    float tot[32];
    MMF_HIP_TRY(hipMemcpyAsync(tot, c->scratch_state->out_f, sizeof(float) * 32, hipMemcpyDeviceToHost, c->stream));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    unpack_se3_host(tot, A_host, b_host);
    return MMF_OK;
}

extern "C" int mmf_so3_step(mmf_ctx* c, const uint8_t* last_image, size_t last_image_step, const uint8_t* next_image,
                            size_t next_image_step, const float image_basis[9], const float kinv[9],
                            const float krlr[9], int cols, int rows, float* A_host, float* b_host,
                            float* residual_host) {
    MMF_REQUIRE(c && last_image && next_image && image_basis && kinv && krlr && A_host && b_host && residual_host,
                "mmf_so3_step: null argument");
    MMF_REQUIRE(cols > 2 && rows > 2, "mmf_so3_step: bad size");
    MMF_HIP_TRY(hipSetDevice(c->device));
    OdomState* h = c->host_state;
    std::memcpy(h->imageBasis, image_basis, sizeof(float) * 9);
    std::memcpy(h->kinv, kinv, sizeof(float) * 9);
    std::memcpy(h->krlr, krlr, sizeof(float) * 9);
    MMF_HIP_TRY(hipMemcpyAsync(c->scratch_state->imageBasis, h->imageBasis, sizeof(float) * 27, hipMemcpyHostToDevice,
                               c->stream));
    So3Args a;
    a.last_image = last_image;
    a.next_image = next_image;
    a.l_stride = stride_elems(last_image_step, cols, 1);
    a.n_stride = stride_elems(next_image_step, cols, 1);
    a.cols = cols;
    a.rows = rows;
    a.cols_magic = (long long)cols * rows * cols < (1ll << 32) ? (unsigned)((1ull << 32) / (unsigned)cols) + 1u : 0u;
    a.intr = LevelIntr{0, 0, 0, 0};
    const int grid = reduce_grid(cols * rows, kBlock);
    hipLaunchKernelGGL((so3_kernel<FINISH_RAW>), dim3(grid), dim3(kBlock), 0, c->stream, c->scratch_state, a,
                       c->partials_f, c->ticket);
    MMF_HIP_TRY(hipGetLastError());
    float tot[32];
    MMF_HIP_TRY(hipMemcpyAsync(tot, c->scratch_state->out_f, sizeof(float) * 32, hipMemcpyDeviceToHost, c->stream));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    int shift = 0;  // reduce.cu:1135-1149
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 4; ++j) {
            const float v = tot[shift++];
            if (j == 3)
                b_host[i] = v;
            else
                A_host[j * 3 + i] = A_host[i * 3 + j] = v;
        }
    residual_host[0] = tot[9];
    residual_host[1] = tot[10];
    return MMF_OK;
}

// ---- map / pyramid entry points ---------------------------------------------------------------
static int launch_create_vmap(mmf_ctx* c, LevelIntr in, const float* depth, int d_stride, int cols, int rows,
                              float* vmap, int v_stride, float cutoff) {
    hipLaunchKernelGGL(create_vmap_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream, depth, d_stride, cols,
                       rows, vmap, v_stride, 1.f / in.fx, 1.f / in.fy, in.cx, in.cy, cutoff);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_create_vmap(mmf_ctx* c, const mmf_camera* intr, const float* depth, size_t depth_step, int cols,
                               int rows, float* vmap, size_t vmap_step, float depth_cutoff) {
    MMF_REQUIRE(c && intr && depth && vmap && cols > 0 && rows > 0, "mmf_create_vmap: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_create_vmap(c, LevelIntr{intr->fx, intr->fy, intr->cx, intr->cy}, depth,
                              stride_elems(depth_step, cols, 4), cols, rows, vmap, stride_elems(vmap_step, cols, 4),
                              depth_cutoff);
}

static int launch_create_nmap(mmf_ctx* c, const float* vmap, int v_stride, int cols, int rows, float* nmap,
                              int n_stride) {
    hipLaunchKernelGGL(create_nmap_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream, rows, cols, vmap,
                       v_stride, nmap, n_stride);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_create_nmap(mmf_ctx* c, const float* vmap, size_t vmap_step, int cols, int rows, float* nmap,
                               size_t nmap_step) {
    MMF_REQUIRE(c && vmap && nmap && cols > 0 && rows > 0, "mmf_create_nmap: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_create_nmap(c, vmap, stride_elems(vmap_step, cols, 4), cols, rows, nmap,
                              stride_elems(nmap_step, cols, 4));
}

static int launch_transform(mmf_ctx* c, const float* vs, const float* ns, int s_stride, int cols, int rows,
                            const float R[9], const float t[3], float* vd, float* nd, int d_stride) {
    m33 Rm;
    std::memcpy(Rm.m, R, sizeof(float) * 9);
    hipLaunchKernelGGL(transform_maps_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream, rows, cols, vs, ns,
                       s_stride, Rm, f3{t[0], t[1], t[2]}, vd, nd, d_stride);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_transform_maps(mmf_ctx* c, const float* vmap_src, const float* nmap_src, size_t src_step, int cols,
                                  int rows, const float R[9], const float t[3], float* vmap_dst, float* nmap_dst,
                                  size_t dst_step) {
    MMF_REQUIRE(c && vmap_src && nmap_src && R && t && vmap_dst && nmap_dst && cols > 0 && rows > 0,
                "mmf_transform_maps: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_transform(c, vmap_src, nmap_src, stride_elems(src_step, cols, 4), cols, rows, R, t, vmap_dst,
                            nmap_dst, stride_elems(dst_step, cols, 4));
}

static int launch_copy_maps(mmf_ctx* c, const float* v_rgba, const float* n_rgba, int cols, int rows, float* vd,
                            float* nd, int d_stride) {
    hipLaunchKernelGGL(copy_maps_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream, rows, cols,
                       reinterpret_cast<const float4*>(v_rgba), reinterpret_cast<const float4*>(n_rgba), vd, nd,
                       d_stride);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_copy_maps(mmf_ctx* c, const float* vmap_rgba, const float* nmap_rgba, int cols, int rows,
                             float* vmap_dst, float* nmap_dst, size_t dst_step) {
    MMF_REQUIRE(c && vmap_rgba && nmap_rgba && vmap_dst && nmap_dst && cols > 0 && rows > 0,
                "mmf_copy_maps: bad argument");
    MMF_REQUIRE(aligned16(vmap_rgba) && aligned16(nmap_rgba), "mmf_copy_maps: RGBA images must be 16-byte aligned");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_copy_maps(c, vmap_rgba, nmap_rgba, cols, rows, vmap_dst, nmap_dst, stride_elems(dst_step, cols, 4));
}

template <bool NORMALIZE>
static int launch_resize(mmf_ctx* c, const float* in, int i_stride, int in_cols, int in_rows, float* out,
                         int o_stride) {
    const int dcols = in_cols / 2, drows = in_rows / 2;
    hipLaunchKernelGGL((resize_map_kernel<NORMALIZE>), tile_grid(dcols, drows), tile_block(), 0, c->stream, drows,
                       dcols, in_rows, in, i_stride, out, o_stride);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_resize_vmap(mmf_ctx* c, const float* in, size_t in_step, int in_cols, int in_rows, float* out,
                               size_t out_step) {
    MMF_REQUIRE(c && in && out && in_cols > 1 && in_rows > 1, "mmf_resize_vmap: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_resize<false>(c, in, stride_elems(in_step, in_cols, 4), in_cols, in_rows, out,
                                stride_elems(out_step, in_cols / 2, 4));
}

extern "C" int mmf_resize_nmap(mmf_ctx* c, const float* in, size_t in_step, int in_cols, int in_rows, float* out,
                               size_t out_step) {
    MMF_REQUIRE(c && in && out && in_cols > 1 && in_rows > 1, "mmf_resize_nmap: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_resize<true>(c, in, stride_elems(in_step, in_cols, 4), in_cols, in_rows, out,
                               stride_elems(out_step, in_cols / 2, 4));
}

static int launch_intensity(mmf_ctx* c, const uint8_t* img, int i_stride, int channels, int cols, int rows,
                            uint8_t* dst, int d_stride) {
    hipLaunchKernelGGL(image_to_intensity_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream, img, i_stride,
                       channels, cols, rows, dst, d_stride);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_image_bgr_to_intensity(mmf_ctx* c, const uint8_t* img, size_t img_step, int channels, int cols,
                                          int rows, uint8_t* dst, size_t dst_step) {
    MMF_REQUIRE(c && img && dst && cols > 0 && rows > 0, "mmf_image_bgr_to_intensity: bad argument");
    MMF_REQUIRE(channels == 3 || channels == 4, "mmf_image_bgr_to_intensity: channels must be 3 or 4");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_intensity(c, img, img_step ? (int)img_step : cols * channels, channels, cols, rows, dst,
                            stride_elems(dst_step, cols, 1));
}

static int launch_vertices_to_depth(mmf_ctx* c, const float* vmap_rgba, int cols, int rows, float cutoff, float* dst,
                                    int d_stride) {
    hipLaunchKernelGGL(vertices_to_depth_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream,
                       reinterpret_cast<const float4*>(vmap_rgba), cols, rows, dst, d_stride, cutoff);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_vertices_to_depth(mmf_ctx* c, const float* vmap_rgba, int cols, int rows, float cutoff, float* dst,
                                     size_t dst_step) {
    MMF_REQUIRE(c && vmap_rgba && dst && cols > 0 && rows > 0, "mmf_vertices_to_depth: bad argument");
    MMF_REQUIRE(aligned16(vmap_rgba), "mmf_vertices_to_depth: RGBA image must be 16-byte aligned");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_vertices_to_depth(c, vmap_rgba, cols, rows, cutoff, dst, stride_elems(dst_step, cols, 4));
}

static int launch_project(mmf_ctx* c, const float* depth, int d_stride, int cols, int rows, LevelIntr in,
                          float* cloud, float* cloud4 = nullptr) {
    hipLaunchKernelGGL(project_points_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream, depth, d_stride, cols,
                       rows, cloud, 1.0f / in.fx, 1.0f / in.fy, in.cx, in.cy, reinterpret_cast<float4*>(cloud4));
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_project_to_point_cloud(mmf_ctx* c, const float* depth, size_t depth_step, int cols, int rows,
                                          const mmf_camera* intr, int level, float* cloud) {
    MMF_REQUIRE(c && depth && intr && cloud && cols > 0 && rows > 0 && level >= 0 && level < 16,
                "mmf_project_to_point_cloud: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_project(c, depth, stride_elems(depth_step, cols, 4), cols, rows,
                          level_intr(intr->fx, intr->fy, intr->cx, intr->cy, level), cloud);
}

static int launch_pyrdown_f(mmf_ctx* c, const float* src, int s_stride, int scols, int srows, float* dst,
                            int d_stride) {
    const int dcols = scols / 2, drows = srows / 2;
    hipLaunchKernelGGL(pyrdown_gauss_f_kernel, tile_grid(dcols, drows), tile_block(), 0, c->stream, src, s_stride,
                       scols, srows, dst, d_stride, dcols, drows);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_pyr_down_gauss_f(mmf_ctx* c, const float* src, size_t src_step, int src_cols, int src_rows,
                                    float* dst, size_t dst_step) {
    MMF_REQUIRE(c && src && dst && src_cols > 1 && src_rows > 1, "mmf_pyr_down_gauss_f: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_pyrdown_f(c, src, stride_elems(src_step, src_cols, 4), src_cols, src_rows, dst,
                            stride_elems(dst_step, src_cols / 2, 4));
}

static int launch_pyrdown_u8(mmf_ctx* c, const uint8_t* src, int s_stride, int scols, int srows, uint8_t* dst,
                             int d_stride) {
    const int dcols = scols / 2, drows = srows / 2;
    hipLaunchKernelGGL(pyrdown_uchar_gauss_kernel, tile_grid(dcols, drows), tile_block(), 0, c->stream, src, s_stride,
                       scols, srows, dst, d_stride, dcols, drows);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_pyr_down_uchar_gauss(mmf_ctx* c, const uint8_t* src, size_t src_step, int src_cols, int src_rows,
                                        uint8_t* dst, size_t dst_step) {
    MMF_REQUIRE(c && src && dst && src_cols > 1 && src_rows > 1, "mmf_pyr_down_uchar_gauss: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_pyrdown_u8(c, src, stride_elems(src_step, src_cols, 1), src_cols, src_rows, dst,
                             stride_elems(dst_step, src_cols / 2, 1));
}

static int launch_derivative(mmf_ctx* c, const uint8_t* src, int s_stride, int cols, int rows, int16_t* dx,
                             int dx_stride, int16_t* dy, int dy_stride) {
    hipLaunchKernelGGL(derivative_kernel, tile_grid(cols, rows), tile_block(), 0, c->stream, src, s_stride, cols, rows,
                       dx, dx_stride, dy, dy_stride);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_compute_derivative_images(mmf_ctx* c, const uint8_t* src, size_t src_step, int cols, int rows,
                                             int16_t* dx, size_t dx_step, int16_t* dy, size_t dy_step) {
    MMF_REQUIRE(c && src && dx && dy && cols > 0 && rows > 0, "mmf_compute_derivative_images: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    return launch_derivative(c, src, stride_elems(src_step, cols, 1), cols, rows, dx, stride_elems(dx_step, cols, 2),
                             dy, stride_elems(dy_step, cols, 2));
}

// ---------------------------------------------------------------------------------------------
// RGBDOdometry object
// ---------------------------------------------------------------------------------------------
constexpr int kMaxTimedLaunches = 48;  // 19 iterations x (producer + step)

struct mmf_odom {
    mmf_ctx* ctx = nullptr;
    int width = 0, height = 0;
    float cx = 0, cy = 0, fx = 0, fy = 0;
    float dist_thres = 0, angle_thres = 0;
    float sobel_scale = 0.125f;         // RGBDOdometry.cpp:31-32
    float max_depth_delta_rgb = 0.07f;  // :33
    float max_depth_rgb = 6.0f;         // :34
    float min_grad[MMF_NUM_PYRS] = {5, 3, 1};  // :103-105

    // one slab of HBM; every buffer below points into it
    void* slab = nullptr;
    size_t slab_bytes = 0;
    float *vmaps_tmp = nullptr, *nmaps_tmp = nullptr;
    float *vmaps_g_prev[MMF_NUM_PYRS], *nmaps_g_prev[MMF_NUM_PYRS];
    float *vmaps_curr[MMF_NUM_PYRS], *nmaps_curr[MMF_NUM_PYRS];
    float *last_depth[MMF_NUM_PYRS], *next_depth[MMF_NUM_PYRS], *depth_pyr[MMF_NUM_PYRS];
    uint8_t *last_image[MMF_NUM_PYRS], *next_image[MMF_NUM_PYRS], *last_next_image[MMF_NUM_PYRS];
    int16_t *dIdx[MMF_NUM_PYRS], *dIdy[MMF_NUM_PYRS];
    float* cloud[MMF_NUM_PYRS];
    float* cloud4[MMF_NUM_PYRS];  // the same points as {X, Y, Z, 1/Z} (16-byte records: what gn_iter_kernel gathers)
    mmf_dataterm* corres[MMF_NUM_PYRS];
    float* prev_packed[MMF_NUM_PYRS];  // model vertex + normal, pixel interleaved 24-byte records (the ICP gather side)
    // reduction scratch of the Gauss-Newton loop and the two error images, inside the slab: every odometry object has
    // its own (concurrent chains on different streams, or one batched chain addressing model m at slab(m) - slab(0))
    float* gn_partials_f = nullptr;
    float* gn_partials_icp = nullptr;
    int2* gn_partials_res = nullptr;
    unsigned* gn_ticket = nullptr;
    float *icp_err = nullptr, *rgb_err = nullptr;  // Model::icpError / rgbError (R32F), written on the last level-0 iteration
    // extent.hpp: three boxes of four words (extent_of_level), noted by the model-side preparation's depth jobs when the owner asks for it (extent_gen != 0:
    // the number of the frame they were noted for), read by the two-launch chain's passes
    unsigned long long* extent = nullptr;
    unsigned extent_gen = 0;
    // prep_batch.hpp (PrepJob::rect_*): the box an object model's model-side preparation last found its prediction non-zero
    // in (device: two slots of four ints, the preparation's number & 1), and whether it describes the buffers (any
    // preparation of the whole frame leaves it unknown)
    int* prep_box = nullptr;
    unsigned prep_gen = 0;
    bool prep_box_known = false;
    unsigned sensor_gen = 0;     // number of the last sensor-side depth preparation (its smallest depth: extent words 18 / 19)
    float sensor_cutoff = 0.f;   // ... and the depth cut-off its vertex maps were made with
    // an OBJECT model (set by the orchestrator): the two-launch chain walks its images with a quarter of the workgroups
    // (track_kernels.hpp: ChainGeom), batched or alone
    bool sparse = false;
    // ... and the one-launch chain by its extents with a fraction of the workgroups (gn_fused.hpp: gn_iter_mixed_kernel), as
    // many as the box of its own depth needed in its LAST chain plus a quarter (OdomState::gn_need; 0: no chain yet)
    int gn_need[MMF_NUM_PYRS] = {0, 0, 0};
    int last_gn_fault = 0;          // OdomState::gn_fault of the last result picked up (odom_finish_tracking)
    bool walked_by_extent = false;  // the last chain enqueued walked this model by its extents (gn_iter_mixed_kernel)
    bool two_launch_once = false;  // the next chain this odometry leads is the two-launch chain (a frame tracked again)
    OdomState* state = nullptr;  // device
    OdomState* host_result = nullptr;  // pinned, device visible: odom_publish_kernel writes it, the host polls publish_seq
    OdomState* host_result_dev = nullptr;
    unsigned publish_seq = 0;          // sequence number of the last chain enqueued
    bool defer_publish = false;        // set by the orchestrator: the hand-over to the host + the fusion weight ride on the
    FrameRider rider;                  // frame's next resolve launch (frame_rider.hpp) instead of ending the chain
    bool have_tmp = false;  // vmaps_tmp filled by an initICP* call (ordering contract)
    // The reference COPIES its inputs at each init* call (RGBDOdometry.cpp:125,130; Model.cpp:359-388).
    // An owner that guarantees the images stay untouched until getIncrementalTransformation returns
    // (the native orchestrator: they live in its model / frame slabs) sets alias_inputs, and the three
    // device-to-device copies per frame (2 x 4.9 MB + 1.2 MB at 640x480) become pointer assignments.
    bool alias_inputs = false;
    const float *vtmp = nullptr, *ntmp = nullptr;  // what populateRGBDData / copyMaps read: own copy or alias
    const float* depth_l0 = nullptr;               // level 0 of the depth pyramid: own copy or alias
    // set by odom_prepare_batched: gradients and point clouds are already built, and next_depth is
    // last_depth (both come from the same prediction, RGBDOdometry.cpp:179 -- see odom_populate_rgbd)
    bool prep_batched = false;
    bool so3_prefetched = false;  // this frame's SO3 pre-alignment already ran (odom_prefetch_so3)
    const OdomState* so3_stage = nullptr;  // ... in this state (nullptr: in this odometry's own)
    // The gradient images are double buffered: the batched image-side preparation writes grad_w_*, the chain reads dIdx /
    // dIdy, and odom_adopt_gradients swaps the two when a prepared frame becomes the frame that is tracked -- so that the
    // NEXT frame's image side can be prepared while this frame's chain still reads its gradients.
    int16_t *grad_w_dx[MMF_NUM_PYRS], *grad_w_dy[MMF_NUM_PYRS];
    bool grad_pending = false;  // grad_w_* hold a prepared frame's gradients
    // The one-launch-per-iteration chain has a barrier inside every launch: its workgroups must all become resident.  Two
    // such launches from different streams can each hold part of the GPU and wait for the rest forever, so an owner that
    // keeps several chains in flight at once (the orchestrator without batching) clears this and gets the two-launch chain.
    bool exclusive_chain = true;
    // the beginning of the next tracking already ran, with these arguments, on the last launch of the preparation enqueued ahead
    // of it (track_kernels.hpp: prep_batch_begin_kernel); begin_spec_ok: the caller vouches that nothing touched the state since
    bool begin_spec_valid = false, begin_spec_ok = false;
    BeginArgs begin_spec;
    bool retry_so3_prefetched = false;  // what the last odom_enqueue_tracking consumed (a chain that gives up is enqueued again)
    const OdomState* retry_so3_stage = nullptr;
    bool pending_icp = false, pending_so3 = false;  // mode of the tracking call that is in flight (enqueue -> finish)
    hipStream_t track_stream = nullptr;             // the stream that call's chain runs on (the batch leader's)
    mmf_odom* result_of = nullptr;                  // the odometry (batch leader) whose chain publishes this one's result
    // measurement mode (mmf_odom_enable_timing): every producer / rgb_step launch of a tracking call carries its own
    // start / stop events (hipExtLaunchKernelGGL: the dispatch's own begin / end timestamps), the whole chain two more
    int timing = 0;  // 1: chain events only; 2: also every kernel of the chain
    hipEvent_t ev_kernel[2 * kMaxTimedLaunches] = {};
    int timed_kind[kMaxTimedLaunches] = {};  // level * 2 + (0 producer | 1 rgb_step)
    int n_timed = 0;
    hipEvent_t ev_chain[2] = {};
    mmf_odom_timing timing_acc;
    mmf_odom_stats stats;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int mmf_odom_create(mmf_ctx* c, int width, int height, float cx, float cy, float fx, float fy,
                               float dist_thresh, float angle_thresh, mmf_odom** out) {
    MMF_REQUIRE(c && out, "mmf_odom_create: null argument");
    MMF_REQUIRE(width >= 32 && height >= 32 && width % 4 == 0 && height % 4 == 0 && width < 32768 && height < 32768,
                "mmf_odom_create: width/height must be multiples of 4, >= 32");
    MMF_HIP_TRY(hipSetDevice(c->device));
    mmf_odom* o = new (std::nothrow) mmf_odom();
    MMF_REQUIRE(o != nullptr, "mmf_odom_create: out of host memory");
    o->ctx = c;
    o->width = width;
    o->height = height;
    o->cx = cx;
    o->cy = cy;
    o->fx = fx;
    o->fy = fy;
    o->dist_thres = dist_thresh;
    o->angle_thres = angle_thresh;
    std::memset(&o->stats, 0, sizeof(o->stats));
    std::memset(&o->timing_acc, 0, sizeof(o->timing_acc));
    o->stats.lastICPCount = o->stats.lastRGBCount = o->stats.lastSO3Count = (float)(width * height);  // :24-28

    // carve every pyramid buffer out of one allocation, each 256-byte aligned
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    const size_t n0 = (size_t)width * height;
    size_t o_vt = carve(4 * n0 * 4), o_nt = carve(4 * n0 * 4);
    size_t o_vgp[3], o_ngp[3], o_vc[3], o_nc[3], o_ld[3], o_nd[3], o_dp[3], o_li[3], o_ni[3], o_lni[3], o_dx[3],
        o_dy[3], o_dx2[3], o_dy2[3], o_cl[3], o_c4[3], o_co[3], o_pp[3];
    for (int i = 0; i < MMF_NUM_PYRS; ++i) {
        const size_t n = (size_t)(width >> i) * (height >> i);
        o_vgp[i] = carve(3 * n * 4);
        o_ngp[i] = carve(3 * n * 4);
        o_vc[i] = carve(3 * n * 4);
        o_nc[i] = carve(3 * n * 4);
        o_ld[i] = carve(n * 4);
        o_nd[i] = carve(n * 4);
        o_dp[i] = carve(n * 4);
        o_li[i] = carve(n);
        o_ni[i] = carve(n);
        o_lni[i] = carve(n);
        o_dx[i] = carve(n * 2);
        o_dy[i] = carve(n * 2);
        o_dx2[i] = carve(n * 2);
        o_dy2[i] = carve(n * 2);
        o_cl[i] = carve(3 * n * 4);
        o_c4[i] = carve(4 * n * 4);
        o_co[i] = carve(n * sizeof(mmf_dataterm));
        o_pp[i] = carve(n * 6 * sizeof(float));
    }
    size_t o_state = carve(sizeof(OdomState));
    const size_t o_pf = carve(sizeof(float) * kMaxGrid * kPartialStride), o_pi = carve(sizeof(float) * kMaxIcpGrid * kPartialStride),
                 o_pr = carve(sizeof(int2) * kMaxGrid), o_tk = carve(sizeof(unsigned) * kTicketWords), o_ei = carve(n0 * 4),
                 o_er = carve(n0 * 4), o_ex = carve(kExtentWords * sizeof(unsigned long long) + 64);  // (+ prep_box)
    o->slab_bytes = off;
    hipError_t e = hipMalloc(&o->slab, o->slab_bytes);
    if (e != hipSuccess) {
        delete o;
        return fail(MMF_ERR_HIP, std::string("mmf_odom_create: hipMalloc: ") + hipGetErrorString(e));
    }
    MMF_HIP_TRY(hipMemsetAsync(o->slab, 0, o->slab_bytes, c->stream));
    char* base = static_cast<char*>(o->slab);
    o->vmaps_tmp = (float*)(base + o_vt);
    o->nmaps_tmp = (float*)(base + o_nt);
    for (int i = 0; i < MMF_NUM_PYRS; ++i) {
        o->vmaps_g_prev[i] = (float*)(base + o_vgp[i]);
        o->nmaps_g_prev[i] = (float*)(base + o_ngp[i]);
        o->vmaps_curr[i] = (float*)(base + o_vc[i]);
        o->nmaps_curr[i] = (float*)(base + o_nc[i]);
        o->last_depth[i] = (float*)(base + o_ld[i]);
        o->next_depth[i] = (float*)(base + o_nd[i]);
        o->depth_pyr[i] = (float*)(base + o_dp[i]);
        o->last_image[i] = (uint8_t*)(base + o_li[i]);
        o->next_image[i] = (uint8_t*)(base + o_ni[i]);
        o->last_next_image[i] = (uint8_t*)(base + o_lni[i]);
        o->dIdx[i] = (int16_t*)(base + o_dx[i]);
        o->dIdy[i] = (int16_t*)(base + o_dy[i]);
        o->grad_w_dx[i] = (int16_t*)(base + o_dx2[i]);
        o->grad_w_dy[i] = (int16_t*)(base + o_dy2[i]);
        o->cloud[i] = (float*)(base + o_cl[i]);
        o->cloud4[i] = (float*)(base + o_c4[i]);
        o->corres[i] = (mmf_dataterm*)(base + o_co[i]);
        o->prev_packed[i] = (float*)(base + o_pp[i]);
    }
    o->state = (OdomState*)(base + o_state);
    o->gn_partials_f = (float*)(base + o_pf), o->gn_partials_icp = (float*)(base + o_pi);
    o->gn_partials_res = (int2*)(base + o_pr), o->gn_ticket = (unsigned*)(base + o_tk);
    o->icp_err = (float*)(base + o_ei), o->rgb_err = (float*)(base + o_er);
    o->extent = (unsigned long long*)(base + o_ex);
    o->prep_box = (int*)(base + o_ex + kExtentWords * sizeof(unsigned long long));
    MMF_HIP_TRY(hipHostMalloc(&o->host_result, sizeof(OdomState), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(o->host_result, 0, sizeof(OdomState));
    MMF_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&o->host_result_dev), o->host_result, 0));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    *out = o;
    return MMF_OK;
}

// The odometry of a model ANOTHER rank tracks (fusion_orchestrator.hpp: a shard rank's bookkeeping copies): the host
// struct only -- statistics as they arrive with the pose exchange -- no slab, no events, nothing a kernel could be given.
static int odom_create_bookkeeping(mmf_ctx* c, int width, int height, float cx, float cy, float fx, float fy, mmf_odom** out) {
    mmf_odom* o = new (std::nothrow) mmf_odom();
    MMF_REQUIRE(o != nullptr, "mmf_odom_create: out of host memory");
    o->ctx = c;
    o->width = width, o->height = height;
    o->cx = cx, o->cy = cy, o->fx = fx, o->fy = fy;
    std::memset(&o->stats, 0, sizeof(o->stats));
    std::memset(&o->timing_acc, 0, sizeof(o->timing_acc));
    o->stats.lastICPCount = o->stats.lastRGBCount = o->stats.lastSO3Count = (float)(width * height);
    *out = o;
    return MMF_OK;
}

extern "C" void mmf_odom_destroy(mmf_odom* o) {
    if (!o) return;
    (void)hipSetDevice(o->ctx->device);
    (void)hipStreamSynchronize(o->ctx->stream);
    (void)hipFree(o->slab);
    (void)hipHostFree(o->host_result);
    for (hipEvent_t e : o->ev_kernel)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : o->ev_chain)
        if (e) (void)hipEventDestroy(e);
    delete o;
}

extern "C" int mmf_odom_build_depth_pyramid(mmf_odom* o, const float* depth_l0, size_t step) {
    MMF_REQUIRE(o && depth_l0, "mmf_odom_build_depth_pyramid: null argument");
    mmf_ctx* c = o->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const size_t s = step ? step : (size_t)o->width * 4;
    if (o->alias_inputs && s == (size_t)o->width * 4) {
        o->depth_l0 = depth_l0;
    } else {
        MMF_HIP_TRY(hipMemcpy2DAsync(o->depth_pyr[0], (size_t)o->width * 4, depth_l0, s, (size_t)o->width * 4, o->height,
                                     hipMemcpyDeviceToDevice, c->stream));
        o->depth_l0 = o->depth_pyr[0];
    }
    for (int i = 1; i < MMF_NUM_PYRS; ++i) {  // Model.cpp:378-382
        int rc = launch_pyrdown_f(c, i == 1 ? o->depth_l0 : o->depth_pyr[i - 1], o->width >> (i - 1), o->width >> (i - 1),
                                  o->height >> (i - 1), o->depth_pyr[i], o->width >> i);
        if (rc) return rc;
    }
    return MMF_OK;
}

extern "C" int mmf_odom_init_icp(mmf_odom* o, const float* const depth_pyr[MMF_NUM_PYRS],
                                 const size_t steps[MMF_NUM_PYRS], float depth_cutoff) {
    MMF_REQUIRE(o != nullptr, "mmf_odom_init_icp: null odometry");
    mmf_ctx* c = o->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < MMF_NUM_PYRS; ++i) {  // RGBDOdometry.cpp:112-115
        const int cols = o->width >> i, rows = o->height >> i;
        const float* d = depth_pyr ? depth_pyr[i] : (i == 0 && o->depth_l0 ? o->depth_l0 : o->depth_pyr[i]);
        MMF_REQUIRE(d != nullptr, "mmf_odom_init_icp: null pyramid level");
        const int ds = (depth_pyr && steps && steps[i]) ? (int)(steps[i] / 4) : cols;
        int rc = launch_create_vmap(c, level_intr(o->fx, o->fy, o->cx, o->cy, i), d, ds, cols, rows, o->vmaps_curr[i],
                                    cols, depth_cutoff);
        if (rc) return rc;
        rc = launch_create_nmap(c, o->vmaps_curr[i], cols, cols, rows, o->nmaps_curr[i], cols);
        if (rc) return rc;
    }
    return MMF_OK;
}

static int odom_take_prediction(mmf_odom* o, const float* vert_rgba, const float* norm_rgba, float** vdst,
                                float** ndst) {
    mmf_ctx* c = o->ctx;
    o->prep_batched = false;
    const size_t bytes = (size_t)4 * o->width * o->height * sizeof(float);
    // the reference copies both textures into vmaps_tmp / nmaps_tmp (RGBDOdometry.cpp:125,130);
    // vmaps_tmp is read again by initRGB*/populateRGBDData (:179)
    if (o->alias_inputs) {
        o->vtmp = vert_rgba, o->ntmp = norm_rgba;
    } else {
        MMF_HIP_TRY(hipMemcpyAsync(o->vmaps_tmp, vert_rgba, bytes, hipMemcpyDeviceToDevice, c->stream));
        MMF_HIP_TRY(hipMemcpyAsync(o->nmaps_tmp, norm_rgba, bytes, hipMemcpyDeviceToDevice, c->stream));
        o->vtmp = o->vmaps_tmp, o->ntmp = o->nmaps_tmp;
    }
    o->have_tmp = true;
    int rc = launch_copy_maps(c, o->vtmp, o->ntmp, o->width, o->height, vdst[0], ndst[0], o->width);
    if (rc) return rc;
    for (int i = 1; i < MMF_NUM_PYRS; ++i) {
        rc = launch_resize<false>(c, vdst[i - 1], o->width >> (i - 1), o->width >> (i - 1), o->height >> (i - 1),
                                  vdst[i], o->width >> i);
        if (rc) return rc;
        rc = launch_resize<true>(c, ndst[i - 1], o->width >> (i - 1), o->width >> (i - 1), o->height >> (i - 1),
                                 ndst[i], o->width >> i);
        if (rc) return rc;
    }
    return MMF_OK;
}

extern "C" int mmf_odom_init_icp_from_prediction(mmf_odom* o, const float* vert_rgba, const float* norm_rgba,
                                                 float depth_cutoff) {
    (void)depth_cutoff;  // unused by the reference as well (RGBDOdometry.cpp:120-141)
    MMF_REQUIRE(o && vert_rgba && norm_rgba, "mmf_odom_init_icp_from_prediction: null argument");
    MMF_HIP_TRY(hipSetDevice(o->ctx->device));
    return odom_take_prediction(o, vert_rgba, norm_rgba, o->vmaps_curr, o->nmaps_curr);
}

extern "C" int mmf_odom_init_icp_model(mmf_odom* o, const float* vert_rgba, const float* norm_rgba,
                                       float depth_cutoff, const float pose[16]) {
    (void)depth_cutoff;  // unused by the reference as well (RGBDOdometry.cpp:143-175)
    MMF_REQUIRE(o && vert_rgba && norm_rgba && pose, "mmf_odom_init_icp_model: null argument");
    MMF_HIP_TRY(hipSetDevice(o->ctx->device));
    int rc = odom_take_prediction(o, vert_rgba, norm_rgba, o->vmaps_g_prev, o->nmaps_g_prev);
    if (rc) return rc;
    const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    const float t[3] = {pose[3], pose[7], pose[11]};
    for (int i = 0; i < MMF_NUM_PYRS; ++i) {
        const int cols = o->width >> i, rows = o->height >> i;
        rc = launch_transform(o->ctx, o->vmaps_g_prev[i], o->nmaps_g_prev[i], cols, cols, rows, R, t,
                              o->vmaps_g_prev[i], o->nmaps_g_prev[i], cols);
        if (rc) return rc;
        const int n = cols * rows;
        hipLaunchKernelGGL(pack_prev_kernel, dim3((n + 255) / 256), dim3(256), 0, o->ctx->stream, o->vmaps_g_prev[i],
                           o->nmaps_g_prev[i], n, o->prev_packed[i]);
        MMF_HIP_TRY(hipGetLastError());
    }
    return MMF_OK;
}

// RGBDOdometry.cpp:177-194 (populateRGBDData); mask pyramids are never read and are omitted
static int odom_populate_rgbd(mmf_odom* o, const uint8_t* rgb, size_t step, int channels, float** depths,
                              uint8_t** images) {
    mmf_ctx* c = o->ctx;
    o->prep_batched = false;
    if (!o->have_tmp)
        return fail(MMF_ERR_STATE, "initRGB*/initRGBModel needs a preceding initICPModel / initICP(prediction): "
                                   "it reads vmaps_tmp (RGBDOdometry.cpp:197,202)");
    int rc = launch_vertices_to_depth(c, o->vtmp, o->width, o->height, o->max_depth_rgb, depths[0], o->width);
    if (rc) return rc;
    for (int i = 0; i + 1 < MMF_NUM_PYRS; ++i) {
        rc = launch_pyrdown_f(c, depths[i], o->width >> i, o->width >> i, o->height >> i, depths[i + 1],
                              o->width >> (i + 1));
        if (rc) return rc;
    }
    rc = launch_intensity(c, rgb, step ? (int)step : o->width * channels, channels, o->width, o->height, images[0],
                          o->width);
    if (rc) return rc;
    for (int i = 0; i + 1 < MMF_NUM_PYRS; ++i) {
        rc = launch_pyrdown_u8(c, images[i], o->width >> i, o->width >> i, o->height >> i, images[i + 1],
                               o->width >> (i + 1));
        if (rc) return rc;
    }
    return MMF_OK;
}

extern "C" int mmf_odom_init_rgb(mmf_odom* o, const uint8_t* rgb, size_t step, int channels) {
    MMF_REQUIRE(o && rgb && (channels == 3 || channels == 4), "mmf_odom_init_rgb: bad argument");
    MMF_HIP_TRY(hipSetDevice(o->ctx->device));
    return odom_populate_rgbd(o, rgb, step, channels, o->next_depth, o->next_image);
}

extern "C" int mmf_odom_init_rgb_model(mmf_odom* o, const uint8_t* rgb, size_t step, int channels) {
    MMF_REQUIRE(o && rgb && (channels == 3 || channels == 4), "mmf_odom_init_rgb_model: bad argument");
    MMF_HIP_TRY(hipSetDevice(o->ctx->device));
    return odom_populate_rgbd(o, rgb, step, channels, o->last_depth, o->last_image);
}

extern "C" int mmf_odom_init_first_rgb(mmf_odom* o, const uint8_t* rgb, size_t step, int channels) {
    MMF_REQUIRE(o && rgb && (channels == 3 || channels == 4), "mmf_odom_init_first_rgb: bad argument");
    mmf_ctx* c = o->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    int rc = launch_intensity(c, rgb, step ? (int)step : o->width * channels, channels, o->width, o->height,
                              o->last_next_image[0], o->width);
    if (rc) return rc;
    for (int i = 0; i + 1 < MMF_NUM_PYRS; ++i) {  // RGBDOdometry.cpp:212-214
        rc = launch_pyrdown_u8(c, o->last_next_image[i], o->width >> i, o->width >> i, o->height >> i,
                               o->last_next_image[i + 1], o->width >> (i + 1));
        if (rc) return rc;
    }
    return MMF_OK;
}

// ---- the whole per-frame preparation in four launches (prep_batch.hpp) -------------------------
static size_t prep_big_job() {  // pixels from which a job's workgroups take four tiles each; MMF_PREP_BIG=0: never
    const long n = tunables().prep_big;
    return n < 0 ? (size_t)200000 : (n > 0 ? (size_t)n : ~(size_t)0);
}
struct PrepBuilder {  // the jobs of one stage (possibly of several models); launched kMaxPrepJobs at a time
    std::vector<PrepJob> jobs;
    bool critical = false;  // PrepBatch::critical
    PrepJob& add(int op, int cols, int rows) {
        if (jobs.capacity() < 96) jobs.reserve(96);  // references handed out stay valid while a stage is being filled
        jobs.emplace_back();
        PrepJob& j = jobs.back();
        std::memset(&j, 0, sizeof(j));
        j.op = op;
        j.cols = cols, j.rows = rows;
        j.gx = (cols + kTileX - 1) / kTileX;
        j.reps = (size_t)cols * rows >= prep_big_job() ? 4 : 1;  // (PrepJob::reps)
        return j;
    }
    // rider: the beginning of the tracking these jobs prepare, on one more workgroup of the stage's LAST launch (BeginRider)
    template <int CAP>
    static int fill(PrepBatchT<CAP>& b, const std::vector<PrepJob>& jobs, size_t first, bool critical) {
        b.njobs = 0;
        b.critical = critical ? 1 : 0;
        int blocks = 0;
        for (size_t k = first; k < jobs.size() && b.njobs < CAP; ++k) {
            PrepJob& j = b.job[b.njobs++];
            j = jobs[k];
            j.first_block = blocks;
            blocks += j.rect_now ? j.rect_groups : j.gx * ((j.rows + kTileY * j.reps - 1) / (kTileY * j.reps));
        }
        return blocks;
    }
    int launch(Enqueuer& q, const BeginRider* rider = nullptr) {
        if (jobs.size() > (size_t)kMaxPrepJobs && !rider) {  // several models' jobs: the wide table (prep_batch.hpp)
            for (size_t first = 0; first < jobs.size(); first += kMaxPrepJobsWide) {
                PrepBatchWide b;
                const int blocks = fill(b, jobs, first, critical);
                q.launch(prep_batch_wide_kernel, dim3(blocks), tile_block(), b);
            }
            jobs.clear();
            return MMF_OK;
        }
        for (size_t first = 0; first < jobs.size(); first += kMaxPrepJobs) {
            PrepBatch b;
            const int blocks = fill(b, jobs, first, critical);
            if (rider && first + kMaxPrepJobs >= jobs.size()) {
                BeginRider r = *rider;
                r.prep_blocks = blocks;
                q.launch(prep_batch_begin_kernel, dim3(blocks + 1), tile_block(), b, r);
            } else {
                q.launch(prep_batch_kernel, dim3(blocks), tile_block(), b);
            }
        }
        jobs.clear();
        return MMF_OK;
    }
};
struct PrepStages {  // the four dependent launches of a frame's preparation
    PrepBuilder stage[4];
    void set_critical(bool on) {
        for (PrepBuilder& pb : stage) pb.critical = on;
    }
    int launch(Enqueuer& q, const BeginRider* rider = nullptr) {  // recorded; the caller flushes.  rider: on the last launch
        int last = -1;
        for (int k = 0; k < 4; ++k)
            if (!stage[k].jobs.empty()) last = k;
        for (int k = 0; k < 4; ++k)
            if (int rc = stage[k].launch(q, k == last ? rider : nullptr)) return rc;
        return MMF_OK;
    }
    int launch(hipStream_t stream, const BeginRider* rider = nullptr) {
        Enqueuer q(stream);
        if (int rc = launch(q, rider)) return rc;
        MMF_HIP_TRY(q.flush());
        return MMF_OK;
    }
    bool empty() const {
        for (const PrepBuilder& pb : stage)
            if (!pb.jobs.empty()) return false;
        return true;
    }
};

// Everything initICPModel + initRGBModel + generateCUDATextures/initICP + initRGB + the gradient and
// point-cloud passes of getIncrementalTransformation compute (RGBDOdometry.cpp:108-235, 332-334;
// Model.cpp:359-407), for inputs that stay untouched until tracking returns (the native orchestrator).
// `sel` (device, may be null): when *sel != 0 the prediction is read from the alt_* images instead.
// `side`: the jobs split into those that depend only on the NEW sensor frame (filtered depth, RGB: vertex / normal
// maps, depth and intensity pyramids, gradients) and those that depend on the model's prediction and pose.
// The orchestrator can run the first group for frame t+1 on a second stream while frame t is still being fused
// (mmf_fusion_prefetch_frame); PREP_ALL is both groups in the same four launches.
enum PrepSide { PREP_INPUT_IMAGE = 1, PREP_INPUT_DEPTH = 2, PREP_MODEL_SIDE = 4, PREP_ALL = 7 };

static std::atomic<int> g_prep_rect{-1};  // -1: MMF_PREP_RECT decides (default on); 0 / 1: mmf_debug_set_prep_rect
extern "C" int mmf_debug_set_prep_rect(int on) {
    g_prep_rect.store(on < 0 ? -1 : (on ? 1 : 0));
    return MMF_OK;
}
static void odom_prepare_collect(PrepStages& stages, mmf_odom* o, const float* depth_filtered, float depth_cutoff, const uint8_t* rgb,
                                 int rgb_channels, const float* pred_vertex, const float* pred_normal,
                                 const uint8_t* pred_image, int pred_channels, const float pose[16],
                                 const int* sel = nullptr, const float* alt_vertex = nullptr,
                                 const float* alt_normal = nullptr, const uint8_t* alt_image = nullptr,
                                 int side = PREP_ALL, int sel_total = 0, float sel_ratio = 0.f, unsigned ext_gen = 0,
                                 const int* pred_box = nullptr) {
    // pred_box (device, level-0 pixels {x0, y0, x1, y1}; an OBJECT model's model side only): where the prediction's images are
    // non-zero -- the jobs then cover the hull of that box and of the one the previous preparation saw (PrepJob::rect_*)
    const bool in_img = (side & PREP_INPUT_IMAGE) != 0, in_depth = (side & PREP_INPUT_DEPTH) != 0;
    const bool model_side = (side & PREP_MODEL_SIDE) != 0;
    size_t jobs_before[4];
    for (int k = 0; k < 4; ++k) jobs_before[k] = stages.stage[k].jobs.size();
    const int W = o->width, H = o->height;
    const size_t n0 = (size_t)W * H;
    // camera-frame model pyramids (before the transform into the global frame) live in the two 4*N-float
    // staging images, which this path does not need as copies: 3*N*(1 + 1/4 + 1/16) floats each
    float* uv[3] = {o->vmaps_tmp, o->vmaps_tmp + 3 * n0, o->vmaps_tmp + 3 * n0 + 3 * (n0 / 4)};
    float* un[3] = {o->nmaps_tmp, o->nmaps_tmp + 3 * n0, o->nmaps_tmp + 3 * n0 + 3 * (n0 / 4)};
    // The coarsest level's model-side products (transformed + packed maps, point cloud) are computed in the SAME stage as
    // its pyramid step, from the level above (PREP_RESIZE_TP, PREP_PYR_PROJECT): the model side is three dependent launches
    // instead of four.  MMF_PREP_MERGE=0: the four-stage form (A/B aid).
    const bool merge_last = tunables().prep_merge != 0;
    // ... and level 0 and the first pyramid step are computed straight from the prediction's images (PREP_TEX_*): two
    // dependent launches.  MMF_PREP_MERGE=1: only the last stage merged (A/B aid).
    const bool merge_first = tunables().prep_merge >= 2;
    // The model maps in the global frame go out as the packed records (and {X, Y, Z, 1/Z} point records) the chains gather
    // from; the planar copies and the AoS cloud of the first-generation kernels (25 -> 14 MB written per frame at level 0)
    // only for the first-generation ICP kernel (MMF_ICP_VARIANT 1xxxxxx) or on request: MMF_PREP_PLANAR=1.
    const bool planar = tunables().prep_planar || icp_default_variant(0) / 1000000 == 1;
    // The sensor frame's normal map of a level is computed in the same job as its vertex map, from the depth image
    // (PREP_VMAP_NMAP): the depth side is three dependent launches instead of four.  MMF_PREP_VN=0: apart (A/B aid).
    const bool merge_vn = tunables().prep_vn;
    // ext_gen != 0: the jobs that write the model's depth and vertex pyramids note the extent of what is valid (extent.hpp)
    const bool note_extent = model_side && merge_first && merge_last && ext_gen != 0;
    auto noted = [&](PrepJob& j, int lvl) {  // (the depth jobs: extent.hpp, extent_of_level)
        if (note_extent) j.ext = o->extent + 4 * lvl, j.ext_gen = ext_gen;
    };
    if (model_side) o->extent_gen = note_extent ? ext_gen : 0u;
    auto intr_f = [&](PrepJob& j, int lvl, bool cutoff_too, float cutoff) {
        const LevelIntr in = level_intr(o->fx, o->fy, o->cx, o->cy, lvl);
        j.f[0] = 1.f / in.fx, j.f[1] = 1.f / in.fy, j.f[2] = in.cx, j.f[3] = in.cy;
        if (cutoff_too) j.f[4] = cutoff;
    };
    auto pyr = [&](PrepBuilder& pb, int op, const void* src, void* dst, int lvl) {  // level lvl-1 -> lvl
        PrepJob& j = pb.add(op, W >> lvl, H >> lvl);
        j.src0 = src, j.dst0 = dst;
        j.scols = W >> (lvl - 1), j.srows = H >> (lvl - 1);
    };
    auto level_jobs = [&](PrepBuilder& pb, int lvl) {  // jobs whose inputs are the level-lvl images
        const int cols = W >> lvl, rows = H >> lvl;
        if (model_side && merge_first && lvl == 0) {  // from the prediction's images
            const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
            PrepJob& t = pb.add(PREP_TEX_TP, cols, rows);
            t.src0 = pred_vertex, t.src1 = pred_normal, t.sel = sel, t.alt0 = alt_vertex, t.alt1 = alt_normal;
            t.dst0 = planar ? o->vmaps_g_prev[0] : nullptr, t.dst1 = planar ? o->nmaps_g_prev[0] : nullptr, t.dst2 = o->prev_packed[0];
            for (int k = 0; k < 9; ++k) t.f[k] = R[k];
            t.f[9] = pose[3], t.f[10] = pose[7], t.f[11] = pose[11];
            if (note_extent) t.aabb = o->extent, t.ext_gen = ext_gen;  // (extent.hpp: the pixel box and depth range of the valid vertices)
            PrepJob& p = pb.add(PREP_TEX_PROJECT, cols, rows);
            p.src0 = pred_vertex, p.sel = sel, p.alt0 = alt_vertex;
            p.dst0 = planar ? o->cloud[0] : nullptr, p.dst1 = o->cloud4[0], p.dst2 = o->last_depth[0];
            intr_f(p, 0, false, 0.f);
            p.f[4] = o->max_depth_rgb;
            noted(p, 0);
            PrepJob& il = pb.add(PREP_INTENSITY, W, H);
            il.src0 = pred_image, il.dst0 = o->last_image[0], il.scols = W * pred_channels, il.channels = pred_channels;
            il.sel = sel, il.alt0 = alt_image;
        } else if (model_side && !(merge_last && lvl == MMF_NUM_PYRS - 1)) {
            PrepJob& t = pb.add(PREP_TRANSFORM_PACK, cols, rows);
            t.src0 = uv[lvl], t.src1 = un[lvl];
            t.dst0 = planar ? o->vmaps_g_prev[lvl] : nullptr, t.dst1 = planar ? o->nmaps_g_prev[lvl] : nullptr, t.dst2 = o->prev_packed[lvl];
            const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
            for (int k = 0; k < 9; ++k) t.f[k] = R[k];
            t.f[9] = pose[3], t.f[10] = pose[7], t.f[11] = pose[11];
            PrepJob& p = pb.add(PREP_PROJECT, cols, rows);
            p.src0 = o->last_depth[lvl], p.dst0 = planar ? o->cloud[lvl] : nullptr, p.dst1 = o->cloud4[lvl];
            intr_f(p, lvl, false, 0.f);
        }
        if (in_img) {
            PrepJob& d = pb.add(PREP_DERIV, cols, rows);
            d.src0 = o->next_image[lvl], d.dst0 = o->grad_w_dx[lvl], d.dst1 = o->grad_w_dy[lvl];  // (odom_adopt_gradients)
        }
        if (in_depth && !merge_vn) {
            PrepJob& nm = pb.add(PREP_NMAP, cols, rows);
            nm.src0 = o->vmaps_curr[lvl], nm.dst0 = o->nmaps_curr[lvl];
        }
    };
    auto vmap_job = [&](PrepBuilder& pb, int lvl, const float* depth) {
        PrepJob& j = pb.add(merge_vn ? PREP_VMAP_NMAP : PREP_VMAP, W >> lvl, H >> lvl);
        j.src0 = depth, j.dst0 = o->vmaps_curr[lvl], j.dst1 = o->nmaps_curr[lvl];
        intr_f(j, lvl, true, depth_cutoff);
        if (lvl == 0) {  // (extent.hpp: the sensor frame's smallest valid depth, for the object models' error-image launches)
            j.zmin = o->extent, j.zmin_gen = ++o->sensor_gen;
            o->sensor_cutoff = depth_cutoff;
        }
    };
    auto down_jobs = [&](PrepBuilder& pb, int lvl, const float* depth_src) {  // level lvl-1 -> lvl of every pyramid
        if (in_img) pyr(pb, PREP_PYRDOWN_U8, o->next_image[lvl - 1], o->next_image[lvl], lvl);
        if (model_side && merge_first && lvl == 1) {  // from the prediction's images
            const int cols = W >> 1, rows = H >> 1;
            PrepJob& d = pb.add(PREP_TEX_PYR_F, cols, rows);
            d.src0 = pred_vertex, d.sel = sel, d.alt0 = alt_vertex, d.scols = W, d.srows = H;
            d.dst0 = o->last_depth[1], d.f[0] = o->max_depth_rgb;
            noted(d, 1);
            PrepJob& u = pb.add(PREP_TEX_PYR_U8, cols, rows);
            u.src0 = pred_image, u.sel = sel, u.alt0 = alt_image, u.scols = W, u.srows = H, u.channels = pred_channels;
            u.dst0 = o->last_image[1];
            PrepJob& r = pb.add(PREP_TEX_RESIZE, cols, rows);
            r.src0 = pred_vertex, r.src1 = pred_normal, r.sel = sel, r.alt0 = alt_vertex, r.alt1 = alt_normal, r.scols = W, r.srows = H;
            r.dst0 = uv[1], r.dst1 = un[1];
        } else if (model_side && merge_last && lvl == MMF_NUM_PYRS - 1) {
            pyr(pb, PREP_PYRDOWN_U8, o->last_image[lvl - 1], o->last_image[lvl], lvl);
            const int cols = W >> lvl, rows = H >> lvl;
            PrepJob& t = pb.add(PREP_RESIZE_TP, cols, rows);
            t.src0 = uv[lvl - 1], t.src1 = un[lvl - 1], t.scols = W >> (lvl - 1), t.srows = H >> (lvl - 1);
            t.dst0 = planar ? o->vmaps_g_prev[lvl] : nullptr, t.dst1 = planar ? o->nmaps_g_prev[lvl] : nullptr, t.dst2 = o->prev_packed[lvl];
            const float R[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
            for (int k = 0; k < 9; ++k) t.f[k] = R[k];
            t.f[9] = pose[3], t.f[10] = pose[7], t.f[11] = pose[11];
            PrepJob& p = pb.add(PREP_PYR_PROJECT, cols, rows);
            p.src0 = o->last_depth[lvl - 1], p.scols = W >> (lvl - 1), p.srows = H >> (lvl - 1);
            p.dst0 = planar ? o->cloud[lvl] : nullptr, p.dst1 = o->cloud4[lvl], p.dst2 = o->last_depth[lvl];
            intr_f(p, lvl, false, 0.f);
            noted(p, lvl);
        } else if (model_side) {
            pyr(pb, PREP_PYRDOWN_F, o->last_depth[lvl - 1], o->last_depth[lvl], lvl);
            pyr(pb, PREP_PYRDOWN_U8, o->last_image[lvl - 1], o->last_image[lvl], lvl);
            pyr(pb, PREP_RESIZE_V, uv[lvl - 1], uv[lvl], lvl);
            pyr(pb, PREP_RESIZE_N, un[lvl - 1], un[lvl], lvl);
        }
        (void)depth_src;
    };

    if (model_side) {
        o->depth_l0 = depth_filtered;
        o->vtmp = pred_vertex, o->ntmp = pred_normal;
        o->have_tmp = true;
    }

    {   // stage 1: inputs -> level 0 (and level 1 of the depth pyramid)
        PrepBuilder& pb = stages.stage[0];
        if (in_depth) {
            pyr(pb, PREP_PYRDOWN_F, depth_filtered, o->depth_pyr[1], 1);
            vmap_job(pb, 0, depth_filtered);
        }
        if (in_img) {
            PrepJob& in = pb.add(PREP_INTENSITY, W, H);
            in.src0 = rgb, in.dst0 = o->next_image[0], in.scols = W * rgb_channels, in.channels = rgb_channels;
        }
        if (model_side && !merge_first) {
            PrepJob& v = pb.add(PREP_V2D, W, H);
            v.src0 = pred_vertex, v.dst0 = o->last_depth[0], v.f[0] = o->max_depth_rgb;
            v.sel = sel, v.alt0 = alt_vertex;
            PrepJob& il = pb.add(PREP_INTENSITY, W, H);
            il.src0 = pred_image, il.dst0 = o->last_image[0], il.scols = W * pred_channels, il.channels = pred_channels;
            il.sel = sel, il.alt0 = alt_image;
            PrepJob& cm = pb.add(PREP_COPY_MAPS, W, H);
            cm.src0 = pred_vertex, cm.src1 = pred_normal, cm.dst0 = uv[0], cm.dst1 = un[0];
            cm.sel = sel, cm.alt0 = alt_vertex, cm.alt1 = alt_normal;
        }
    }
    {   // stage 2: level 0 -> level 1 (and level 2 of the depth pyramid)
        PrepBuilder& pb = stages.stage[1];
        if (in_depth) {
            pyr(pb, PREP_PYRDOWN_F, o->depth_pyr[1], o->depth_pyr[2], 2);
            vmap_job(pb, 1, o->depth_pyr[1]);
        }
        // The model side's level-0 products come straight from the prediction's images (merge_first) and nothing of the
        // preparation reads them: they go into the LAST of its two launches, beside the small jobs of levels 1 and 2, and the
        // first launch is the three quarter-size pyramid jobs alone (MMF_PREP_L0_LATE=0: in the first, as before).
        const bool l0_late = tunables().prep_l0_late;
        level_jobs(model_side && merge_first && merge_last && l0_late && !in_img && !in_depth ? stages.stage[2] : pb, 0);
        down_jobs(pb, 1, nullptr);
    }
    {   // stage 3: level 1 -> level 2
        PrepBuilder& pb = stages.stage[2];
        if (in_depth) vmap_job(pb, 2, o->depth_pyr[2]);
        level_jobs(pb, 1);
        down_jobs(pb, 2, nullptr);
    }
    {   // stage 4: level 2
        PrepBuilder& pb = stages.stage[3];
        level_jobs(pb, 2);
    }
    if (model_side) {
        const bool rect_on = g_prep_rect.load() < 0 ? tunables().prep_rect : g_prep_rect.load() != 0;
        const bool rect = pred_box != nullptr && side == PREP_MODEL_SIDE && sel == nullptr && rect_on && o->prep_box != nullptr;
        if (rect) {
            const unsigned g = ++o->prep_gen;
            bool first = true;
            for (int k = 0; k < 4; ++k)
                for (size_t q = jobs_before[k]; q < stages.stage[k].jobs.size(); ++q) {
                    PrepJob& j = stages.stage[k].jobs[q];
                    int lvl = 0;
                    while ((W >> lvl) > j.cols && lvl < MMF_NUM_PYRS - 1) ++lvl;
                    j.rect_now = pred_box, j.rect_prev = o->prep_box_known ? o->prep_box + 4 * ((g + 1u) & 1u) : nullptr;
                    j.rect_level = lvl;
                    j.rect_groups = lvl == 0 ? 48 : (lvl == 1 ? 32 : 16);
                    j.rect_store = first ? o->prep_box + 4 * (g & 1u) : nullptr;
                    first = false;
                }
            o->prep_box_known = true;
        } else {
            o->prep_box_known = false;
        }
        o->prep_batched = true;
    }
    if (in_img) o->grad_pending = true;
    if (sel && sel_total)  // *sel is a count (PrepJob::sel_total)
        for (PrepBuilder& pb : stages.stage)
            for (PrepJob& j : pb.jobs)
                if (j.sel == sel) j.sel_total = sel_total, j.sel_ratio = sel_ratio;
}
// the gradients the batched preparation wrote last become the ones the chain reads
static void odom_adopt_gradients(mmf_odom* o) {
    if (!o->grad_pending) return;
    o->grad_pending = false;
    for (int i = 0; i < MMF_NUM_PYRS; ++i) std::swap(o->dIdx[i], o->grad_w_dx[i]), std::swap(o->dIdy[i], o->grad_w_dy[i]);
}

static int odom_prepare_batched(mmf_odom* o, const float* depth_filtered, float depth_cutoff, const uint8_t* rgb,
                                int rgb_channels, const float* pred_vertex, const float* pred_normal,
                                const uint8_t* pred_image, int pred_channels, const float pose[16],
                                const int* sel = nullptr, const float* alt_vertex = nullptr,
                                const float* alt_normal = nullptr, const uint8_t* alt_image = nullptr,
                                int side = PREP_ALL, hipStream_t stream = nullptr, Enqueuer* q = nullptr) {
    PrepStages stages;
    odom_prepare_collect(stages, o, depth_filtered, depth_cutoff, rgb, rgb_channels, pred_vertex, pred_normal, pred_image,
                         pred_channels, pose, sel, alt_vertex, alt_normal, alt_image, side);
    if (q) return stages.launch(*q);
    return stages.launch(stream ? stream : o->ctx->stream);
}

static IcpArgs odom_icp_args(mmf_odom* o, int level, float* err_map) {
    const int cols = o->width >> level, rows = o->height >> level;
    IcpArgs a;
    a.vmap_curr = MapView{o->vmaps_curr[level], cols};
    a.nmap_curr = MapView{o->nmaps_curr[level], cols};
    a.vmap_g_prev = MapView{o->vmaps_g_prev[level], cols};
    a.nmap_g_prev = MapView{o->nmaps_g_prev[level], cols};
    a.intr = level_intr(o->fx, o->fy, o->cx, o->cy, level);
    a.dist_thres = o->dist_thres;
    a.angle_thres = o->angle_thres;
    a.cols = cols;
    a.rows = rows;
    a.prev_packed = o->prev_packed[level];
    a.err_map = err_map;
    a.err_stride = cols;
    icp_args_derive(a);
    return a;
}

// the ten launches of the SO3 pre-alignment (RGBDOdometry.cpp:239-310): last frame's image against this frame's
// at level 2 -- no model, no pose
static int odom_enqueue_so3(mmf_odom* o, Enqueuer& q, float* partials = nullptr, unsigned* ticket = nullptr, OdomState* state = nullptr) {
    if (!partials) partials = o->gn_partials_f, ticket = o->gn_ticket;
    if (!state) state = o->state;
    const int lvl = 2, cols = o->width >> lvl, rows = o->height >> lvl;
    So3Args a;
    a.last_image = o->last_next_image[lvl];
    a.next_image = o->next_image[lvl];
    a.l_stride = a.n_stride = cols;
    a.cols = cols;
    a.rows = rows;
    a.intr = level_intr(o->fx, o->fy, o->cx, o->cy, 2);
    a.cols_magic = (unsigned)((1ull << 32) / (unsigned)cols) + 1u;
    const int grid = reduce_grid(cols * rows, kBlock);
    for (int i = 0; i < 10; ++i) q.launch((so3_kernel<FINISH_GN>), dim3(grid), dim3(kBlock), state, a, partials, ticket);
    return MMF_OK;
}

// the SO3 pre-alignment of the NEXT frame ahead of its tracking, on `stream` (after that frame's intensity
// pyramid): its begin part, then the ten launches; getIncrementalTransformation then skips both
// `stage`: the state the pre-alignment runs in (its images are o's); the caller tells the consumer (mmf_odom::so3_stage)
static int odom_prefetch_so3(mmf_odom* o, Enqueuer& q, float* partials, unsigned* ticket, OdomState* stage) {
    q.launch(so3_begin_kernel, dim3(1), dim3(64), stage, level_intr(o->fx, o->fy, o->cx, o->cy, 2));
    return odom_enqueue_so3(o, q, partials, ticket, stage);
}

// RGBDOdometry::getIncrementalTransformation (RGBDOdometry.cpp:217-477), device resident:
// every kernel below is enqueued back to back on the context's stream; the data-dependent
// `break`s of the reference (:285-292, :376-378) become flags in the device state that make the
// remaining launches of that loop return immediately.
// first half: everything up to and including the copy of the result towards the host is enqueued on the
// context's stream, nothing waits.  The orchestrator enqueues the chains of all its models (one stream each)
// before it waits for the first result.
// The models one chain of launches tracks: o[0] leads (its stream, its launch arguments), the others ride at their
// slab offsets (BatchDelta).  All share the sensor-side images (fusion_orchestrator.hpp) and the configuration.
struct TrackBatch {
    int n = 1;
    mmf_odom* o[kMaxBatch];
    BatchDelta bd;
    BeginPoses poses;
};

static bool odom_batchable(mmf_odom* o, int rgb_only, float icp_weight, int pyramid, int fast_odom);
// Launch geometry of gn_iter_kernel at a level of cols x rows pixels: a workgroup = the solver wave + the pixel waves.  The
// launch is a chain of latencies -- records, solve, two memory round trips, the count barrier -- and every workgroup waits at
// that barrier for the slowest one, so (measured on MI355X, tools/ab_libs.sh): at most one workgroup per CU (two on a CU
// reach the barrier 1.6 us after the others); exactly four pixel waves, one per SIMD, where that fits (a fifth doubles up on
// one SIMD and is the long pole of every arithmetic phase: what 640x480 suffered with four pixels per lane, 300 lanes in 256
// workgroups); as few pixels per lane as those two allow (a lane's serial arithmetic is on the critical path).  Five pixels
// per lane exist for 640x480: 240 workgroups x 256 lanes x 5.
// MMF_GN_PX="p0,p1,p2" forces the pixels per lane of a level, MMF_GN_GROUPS the workgroup limit (tuning aids).
struct GnGeometry {
    int px, lanes, threads, groups;
};
// mixed: a launch that also carries object models walked by their extents (gn_iter_mixed_kernel).  Its register count allows
// three waves per SIMD, and the GPU places a second 5-wave workgroup on a CU only with four (measured: tools/gn_mixed_probe.py
// -- the object models' workgroups started when the camera model's had finished); workgroups of FOUR waves (the solver wave
// + three pixel waves, 192 pixel lanes) take one wave slot per SIMD and three of them share a CU.
static bool gn_geometry(int level, int cols, int rows, GnGeometry* out, bool mixed = false) {
    if (mixed && tunables().gn_mixed_lanes > 0 && tunables().gn_mixed_lanes < kBlock) {
        GnGeometry base;
        if (!gn_geometry(level, cols, rows, &base, false) || base.lanes != kBlock) return false;
        const int lanes = std::max(32, tunables().gn_mixed_lanes);
        const int groups = (cols * rows / base.px + lanes - 1) / lanes;
        if (groups > kGnMaxGroups) return false;
        *out = GnGeometry{base.px, lanes, (lanes + 63) / 64 * 64 + 64, groups};
        return true;
    }
    const int* forced = tunables().gn_px;
    const int max_groups = tunables().gn_groups;
    const int n = cols * rows;
    const int want = (level >= 0 && level < 3) ? forced[level] : 0;
    // a lane's pixels lie in one row; the window words are 4-byte aligned columns
    auto allowed = [&](int px) { return cols % px == 0 && cols % 4 == 0 && (!(want == 1 || want == 2 || want == 4 || want == 5) || px == want); };
    static const int kPx[4] = {1, 2, 4, 5};
    for (int k = 0; k < 4; ++k) {  // four pixel waves per workgroup (three, 192 lanes: no different, tools/ab_env.sh)
        const int px = kPx[k];
        const int groups = (n / px + kBlock - 1) / kBlock;
        if (!allowed(px) || groups > max_groups || groups > kGnMaxGroups) continue;
        *out = GnGeometry{px, kBlock, kBlock + 64, groups};
        return true;
    }
    for (int k = 3; k >= 0; --k) {  // larger workgroups, as few of them as the limit asks for
        const int px = kPx[k];
        if (!allowed(px)) continue;
        const int total = n / px;
        const int lanes = std::max(kBlock, (total + max_groups - 1) / max_groups);
        const int groups = (total + lanes - 1) / lanes;
        if (lanes > 64 * (kGnMaxWaves - 1) || groups > kGnMaxGroups) continue;
        *out = GnGeometry{px, lanes, (lanes + 63) / 64 * 64 + 64, groups};
        return true;
    }
    return false;
}
struct GnChainPlan {  // how the one-launch chain walks the models of a batch (gn_fused.hpp: GnBatchGeom)
    unsigned sparse_mask = 0, ext_gen = 0;
    int groups[MMF_NUM_PYRS][kMaxBatch];  // workgroups of model m at level l
};
static bool odom_fused_chain_ok(mmf_odom* o, int rgb_only, float icp_weight, int pyramid, int fast_odom, const TrackBatch* batch = nullptr,
                                bool sparse_on = false, GnChainPlan* plan = nullptr);
static unsigned fused_max_models();
// workgroups of an OBJECT model per launch of the one-launch chain at pyramid level l (a property of the model, batched or
// alone: its float sums keep their order): its photometric box must fit them in ONE pass (32 x 1280 pixels at 640x480 level 0,
// a box of 200 x 200), its ICP rectangle takes as many passes as it needs
static std::atomic<int> g_sparse_groups{0};  // test hook (mmf_debug_set_sparse_groups): that many at every level instead
// need: the lanes the model's box took in its last chain (0: unknown -- 32 / 24 / 16 workgroups, a box of 200 x 200 at 640x480);
// lanes: pixel lanes of a workgroup.  Returns dense_groups when the model is better walked like the camera model.
static int gn_sparse_groups(int level, int dense_groups, int need, int lanes) {
    static const int k[MMF_NUM_PYRS] = {32, 24, 16};
    const int forced = g_sparse_groups.load();
    if (forced > 0) return std::min(dense_groups, forced);
    int want = k[level < MMF_NUM_PYRS ? level : MMF_NUM_PYRS - 1];
    if (need > 0) want = std::max(8, (need + need / 4 + lanes - 1) / lanes + 1);
    return want * 4 >= dense_groups * 3 ? dense_groups : want;
}
extern "C" int mmf_debug_set_sparse_groups(int n) {
    g_sparse_groups.store(n > 0 ? n : 0);
    return MMF_OK;
}
static std::atomic<bool> g_gn_latched_off{false};
static std::atomic<int> g_track_cull{-1};  // -1: tunables().track_cull; 0 / 1: mmf_debug_set_track_cull
extern "C" int mmf_debug_set_track_cull(int mode) {
    g_track_cull.store(mode < 0 ? -1 : (mode ? 1 : 0));
    return MMF_OK;
}
static std::atomic<int> g_gn_force_fault{0};
static std::atomic<int> g_sparse_check{0};
// test hooks of the object models' walk in the one-launch chain (gn_fused.hpp: gn_sparse_icp_box): checking mode on / off, and the
// number of correspondences the last chain of an odometry accepted outside the rectangle it would have walked (must be 0)
extern "C" int mmf_debug_set_sparse_check(int on) {
    g_sparse_check.store(on ? 1 : 0);
    return MMF_OK;
}
extern "C" int mmf_debug_odom_sparse_outside(mmf_odom* o, unsigned* outside, int* walked_by_extent, int rect[9]) {
    if (!o || !o->host_result) return fail(MMF_ERR_INVALID, "mmf_debug_odom_sparse_outside: null argument");
    if (outside) *outside = o->host_result->gn_dbg_outside;
    if (walked_by_extent) *walked_by_extent = o->walked_by_extent ? 1 : 0;
    if (rect) {
        std::memcpy(rect, o->host_result->gn_dbg_rect, sizeof(int) * 6);
        std::memcpy(rect + 6, o->gn_need, sizeof(int) * 3);
    }
    return MMF_OK;
}
static std::atomic<int> g_gn_recoveries{0};
// odom_finish_tracking: the one-launch chain gave up; track again (odom_retrack_prepare).  A private sentinel, outside the
// public mmf_status range (include/mmf_hip.h: 0 and small negatives): it never leaves the library (gn_retry_twice).
constexpr int kGnRetry = 0x6e726574;
static int gn_retry_twice() { return fail(MMF_ERR_STATE, "odometry: tracking gave up twice (the two-launch chain reported a fault)"); }

static bool odom_sparse_on() { return (g_track_cull.load() < 0 ? tunables().track_cull : g_track_cull.load()) != 0; }
static void odom_begin_rider_used();
// what odom_begin_kernel is told (RGBDOdometry.cpp:221-228, 237, 252-255, 316-328); every byte defined (the blocks are compared)
static BeginArgs odom_begin_args(const mmf_odom* o, const float trans[3], const float rot[9], int rgb_only, float icp_weight, int pyramid,
                                 int fast_odom, int so3, bool so3_prefetched, const OdomState* so3_stage, bool fused_chain) {
    const bool icp = !rgb_only && icp_weight > 0;  // :221-222
    const bool rgb = rgb_only || icp_weight < 100;
    const int iterations[MMF_NUM_PYRS] = {fast_odom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0};  // :312-314
    int first_iter_level = MMF_NUM_PYRS - 1;  // the coarsest level that runs iterations
    while (first_iter_level > 0 && !iterations[first_iter_level]) --first_iter_level;
    BeginArgs b;
    std::memset(&b, 0, sizeof(b));
    std::memcpy(b.trans, trans, sizeof(b.trans));
    std::memcpy(b.rot, rot, sizeof(b.rot));
    b.rgb_only = rgb_only ? 1 : 0;
    b.icp = icp ? 1 : 0;
    b.rgb = rgb ? 1 : 0;
    b.so3 = so3 ? 1 : 0;
    b.icp_weight = icp_weight;
    b.so3_intr = level_intr(o->fx, o->fy, o->cx, o->cy, 2);
    b.so3_prefetched = (so3 && so3_prefetched) ? 1 : 0;
    b.so3_stage = b.so3_prefetched ? so3_stage : nullptr;
    // nothing runs between the beginning and the first level's begin unless the SO3 loop does: one launch
    b.fold_level_begin = (!so3 || so3_prefetched) ? 1 : 0;
    // the one-launch chain has no per-level begin: its first launch reads what this one prepares
    b.first_intr = level_intr(o->fx, o->fy, o->cx, o->cy, fused_chain ? first_iter_level : MMF_NUM_PYRS - 1);
    return b;
}
static int odom_enqueue_tracking(mmf_odom* o, const float trans[3], const float rot[9], int rgb_only, float icp_weight,
                                 int pyramid, int fast_odom, int so3, float* icp_err_dev, float* rgb_err_dev,
                                 const TrackBatch* batch = nullptr) {
    mmf_ctx* c = o->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
#ifdef MMF_STAMPS  // (diagnostic builds: MMF_DBG_NO_ERR=1 leaves the error images out, so that a chain's last launch is an ordinary one)
    if (std::getenv("MMF_DBG_NO_ERR")) icp_err_dev = rgb_err_dev = nullptr;
#endif
    const bool icp = !rgb_only && icp_weight > 0;  // :221-222
    const bool rgb = rgb_only || icp_weight < 100;
    const unsigned ny = batch ? (unsigned)batch->n : 1u;
    BatchDelta bd;
    BeginPoses poses;
    std::memset(&bd, 0, sizeof(bd));
    std::memset(&poses, 0, sizeof(poses));
    if (batch) bd = batch->bd, poses = batch->poses;
    MMF_REQUIRE(ny == 1 || odom_batchable(o, rgb_only, icp_weight, pyramid, fast_odom), "odom_enqueue_tracking: not batchable");

    if (rgb && !o->prep_batched)
        for (int i = 0; i < MMF_NUM_PYRS; ++i) {  // :230-235
            int rc = launch_derivative(c, o->next_image[i], o->width >> i, o->width >> i, o->height >> i, o->dIdx[i],
                                       o->width >> i, o->dIdy[i], o->width >> i);
            if (rc) return rc;
        }

    const int iterations[MMF_NUM_PYRS] = {fast_odom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0};  // :312-314
    // both terms on and every level fits: ONE launch per iteration (gn_fused.hpp) instead of producer + step
    // (more than three models: the batched two-launch chain is as fast (four) or faster -- 8 models 1.40 ms against 1.60 --
    // because a model's workgroups hold their CUs at the count barrier while the next models' wait for a place)
    // Object models (extent.hpp): in the two-launch chain (track_kernels.hpp: ChainGeom) their passes skip what lies outside
    // the model's own depth when the preparation noted its extents for this frame, and they walk their images with a
    // quarter of the workgroups; in the one-launch chain (gn_fused.hpp: gn_iter_mixed_kernel) they walk their extents with a
    // fraction of the workgroups, which is what lets ALL models of a frame be resident in one launch.
    // MMF_TRACK_CULL=0 / mmf_debug_set_track_cull(0): every model like the first.
    const bool sparse_on = odom_sparse_on();
    bool two_launch_once = false;  // (odom_retrack_prepare: this frame is being tracked again)
    for (unsigned m = 0; m < ny; ++m) {
        mmf_odom* om = batch ? batch->o[m] : o;
        two_launch_once = two_launch_once || om->two_launch_once;
        om->two_launch_once = false;
    }
    GnChainPlan plan;
    for (unsigned m = 0; m < ny; ++m) (batch ? batch->o[m] : o)->walked_by_extent = false;
    const bool fused_chain = !two_launch_once && odom_fused_chain_ok(o, rgb_only, icp_weight, pyramid, fast_odom, batch, sparse_on, &plan) &&
                             (ny == 1 || (ny <= fused_max_models() && odom_batchable(o, rgb_only, icp_weight, pyramid, fast_odom)));
    const bool lead_sparse = sparse_on && o->sparse;
    bool foll_sparse = sparse_on && batch && ny > 1;
    unsigned foll_gen = foll_sparse ? batch->o[1]->extent_gen : 0u;
    for (unsigned m = 1; m < ny && foll_sparse; ++m) {
        foll_sparse = batch->o[m]->sparse;
        if (batch->o[m]->extent_gen != foll_gen) foll_gen = 0u;
    }
    if (!foll_sparse) foll_gen = 0u;
    // (the kernels drop the first model's extent in a batch; a sparse model tracked alone keeps its own)
    const unsigned cull_gen = ny > 1 ? foll_gen : (lead_sparse ? o->extent_gen : 0u);
    auto quarter = [](int full) { return std::max(std::min(full, 32), (full + 3) / 4); };
    int first_iter_level = MMF_NUM_PYRS - 1;  // the coarsest level that runs iterations
    while (first_iter_level > 0 && !iterations[first_iter_level]) --first_iter_level;

    const BeginArgs b = odom_begin_args(o, trans, rot, rgb_only, icp_weight, pyramid, fast_odom, so3, o->so3_prefetched, o->so3_stage, fused_chain);
    // nothing runs between the beginning and the first level's begin unless the SO3 loop does: one launch
    const bool fold_first_level = b.fold_level_begin != 0;
    o->n_timed = 0;
    if (o->timing) MMF_HIP_TRY(hipEventRecord(o->ev_chain[0], c->stream));
    // from here to the last step: kernels only, enqueued one by one in call order (Enqueuer keeps the first error)
    Enqueuer q(c->stream);
    // (the beginning may have run already, on the last launch of the preparation enqueued ahead of this frame: odom_begin_rider)
    const bool begun = o->begin_spec_valid && o->begin_spec_ok && !batch && std::memcmp(&b, &o->begin_spec, sizeof(b)) == 0;
    o->begin_spec_valid = o->begin_spec_ok = false;
    if (!begun)
        q.launch(odom_begin_kernel, dim3(ny), dim3(64), o->state, b, bd, poses);
    else
        odom_begin_rider_used();

    o->retry_so3_prefetched = o->so3_prefetched, o->retry_so3_stage = o->so3_stage;  // (odom_retrack_prepare)
    const bool so3_ran_here = so3 && !o->so3_prefetched;  // in the leader's state: shared with the others at the first level begin
    if (so3 && !o->so3_prefetched) {  // :239-310
        int rc = odom_enqueue_so3(o, q);
        if (rc) return rc;
    }
    o->so3_prefetched = false;
    o->so3_stage = nullptr;

    GnIterArgs final_args;
    bool final_pending = false;
    std::memset(&final_args, 0, sizeof(final_args));
    if (fused_chain) {
        for (unsigned m = 0; m < ny; ++m) (batch ? batch->o[m] : o)->walked_by_extent = ((plan.sparse_mask >> m) & 1u) != 0;
        int it = 0;
        bool first = true;
        GnIterArgs a;
        std::memset(&a, 0, sizeof(a));
        a.poll_sleep = tunables().gn_sleep;
        a.max_polls = kGnMaxPolls;
        a.check_sparse = g_sparse_check.load();
        bool force_fault = false;
        for (int n = g_gn_force_fault.load(); n > 0 && !force_fault;) force_fault = g_gn_force_fault.compare_exchange_weak(n, n - 1);
        for (int i = MMF_NUM_PYRS - 1; i >= 0; --i) {
            if (!iterations[i]) continue;
            const int cols = o->width >> i, rows = o->height >> i;
            const LevelIntr in = level_intr(o->fx, o->fy, o->cx, o->cy, i);
            if (!o->prep_batched) {  // :332-334
                MMF_HIP_TRY(q.flush());
                int rc = launch_project(c, o->last_depth[i], cols, cols, rows, in, o->cloud[i], o->cloud4[i]);
                if (rc) return rc;
            }
            if (first && !fold_first_level)  // the SO3 loop ran in between: seed resultRt from its result now
                q.launch(gn_level_begin_kernel, dim3(ny), dim3(64), o->state, 1, in, bd, so3_ran_here ? 1 : 0);
            first = false;
            const float min_scale = (float)(std::pow((double)o->min_grad[i], 2.0) / std::pow((double)o->sobel_scale, 2.0));
            GnGeometry geo;
            MMF_REQUIRE(gn_geometry(i, cols, rows, &geo, plan.sparse_mask != 0), "odom_enqueue_tracking: no launch geometry for this level");
            const int px = geo.px, groups = geo.groups;
            a.lanes = geo.lanes;
            // object models walked by their extents: one one-dimensional grid, a geometry per model (GnBatchGeom)
            GnBatchGeom gg;
            std::memset(&gg, 0, sizeof(gg));
            const bool mixed = plan.sparse_mask != 0;
            if (mixed) {
                for (unsigned m = 0; m < (unsigned)kMaxBatch; ++m) gg.start[m + 1] = gg.start[m] + (m < ny ? plan.groups[i][m] : 0);
                gg.sparse_mask = plan.sparse_mask, gg.ext_gen = plan.ext_gen, gg.level = i, gg.extent = o->extent;
                gg.sensor = o->extent, gg.sensor_gen = o->sensor_gen, gg.sensor_cutoff = o->sensor_cutoff;
                gg.rotate = tunables().gn_obj_first ? gg.start[1] : 0;  // (the first model of a batch is the camera model)
            }
            for (int j = 0; j < iterations[i]; ++j) {
                const bool last_l0 = (i == 0 && j == iterations[i] - 1);
                a.ra = make_residual_args(min_scale, o->dIdx[i], 0, o->dIdy[i], 0, o->last_depth[i], 0,
                                          o->prep_batched ? o->last_depth[i] : o->next_depth[i], 0, o->last_image[i], 0,
                                          o->next_image[i], 0, o->corres[i], o->max_depth_delta_rgb, cols, rows,
                                          last_l0 ? rgb_err_dev : nullptr, 0);
                a.ra.intr = in;
                a.ia = odom_icp_args(o, i, last_l0 ? icp_err_dev : nullptr);
                a.cloud4 = reinterpret_cast<const float4*>(o->cloud4[i]);
                a.fx = in.fx, a.fy = in.fy, a.sobel_scale = o->sobel_scale;
                a.intr = in;
                a.ifx = 1.0 / (double)in.fx, a.ify = 1.0 / (double)in.fy;
                a.it = it;
                a.max_polls = (it == 2 && force_fault) ? 0 : kGnMaxPolls;
                const bool err = last_l0 && (icp_err_dev || rgb_err_dev);
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (o->timing >= 2 && o->n_timed < kMaxTimedLaunches) {
                    e0 = o->ev_kernel[2 * o->n_timed], e1 = o->ev_kernel[2 * o->n_timed + 1];
                    o->timed_kind[o->n_timed++] = i * 2;
                }
#define MMF_GN_LAUNCH(PXV, ERRV)                                                                                              \
    do {                                                                                                                      \
        if (mixed && e0)                                                                                                      \
            hipExtLaunchKernelGGL((gn_iter_mixed_kernel<PXV, ERRV>), dim3(gg.start[kMaxBatch]), dim3(geo.threads), 0, c->stream, \
                                  e0, e1, 0, o->state, a, bd, gg);                                                            \
        else if (mixed)                                                                                                       \
            q.launch((gn_iter_mixed_kernel<PXV, ERRV>), dim3(gg.start[kMaxBatch]), dim3(geo.threads), o->state, a, bd, gg);     \
        else if (e0)                                                                                                          \
            hipExtLaunchKernelGGL((gn_iter_kernel<PXV, ERRV>), dim3(groups, ny), dim3(geo.threads), 0, c->stream, e0, e1, 0,      \
                                  o->state, a, bd);                                                                           \
        else                                                                                                                  \
            q.launch((gn_iter_kernel<PXV, ERRV>), dim3(groups, ny), dim3(geo.threads), o->state, a, bd);                       \
    } while (0)
                switch (px * 2 + (err ? 1 : 0)) {
                    case 11: MMF_GN_LAUNCH(5, true); break;
                    case 10: MMF_GN_LAUNCH(5, false); break;
                    case 9: MMF_GN_LAUNCH(4, true); break;
                    case 8: MMF_GN_LAUNCH(4, false); break;
                    case 5: MMF_GN_LAUNCH(2, true); break;
                    case 4: MMF_GN_LAUNCH(2, false); break;
                    case 3: MMF_GN_LAUNCH(1, true); break;
                    default: MMF_GN_LAUNCH(1, false); break;
                }
#undef MMF_GN_LAUNCH
                MMF_HIP_TRY(hipGetLastError());
                ++it;
            }
        }
        // the last solve + RGBDOdometry.cpp:464-467: one workgroup per model
        a.it = it;
        a.intr = level_intr(o->fx, o->fy, o->cx, o->cy, 0);
        a.ifx = 1.0 / (double)a.intr.fx, a.ify = 1.0 / (double)a.intr.fy;
        final_args = a;
        // the chain's last solve shares a launch with the hand-over of the result to the host (below), unless the chain is
        // being timed as such (its closing event lies between the two)
        if (o->timing)
            q.launch(gn_final_kernel, dim3(ny), dim3(kBlock), o->state, a, bd);
        else
            final_pending = true;
    }
    bool first_level = true;
    bool end_folded = fused_chain;  // odom_end ran in the finishing lane of the frame's last rgb_step (or in gn_final_kernel)
    bool begin_folded = fold_first_level;  // this level's gn_level_begin already ran (in odom_begin_kernel, or in the
                                           // last rgb_step of the level before)
    for (int i = MMF_NUM_PYRS - 1; i >= 0 && !fused_chain; --i) {
        const int cols = o->width >> i, rows = o->height >> i;
        const LevelIntr in = level_intr(o->fx, o->fy, o->cx, o->cy, i);
        if (rgb && !o->prep_batched) {  // :332-334
            MMF_HIP_TRY(q.flush());
            int rc = launch_project(c, o->last_depth[i], cols, cols, rows, in, o->cloud[i], o->cloud4[i]);
            if (rc) return rc;
        }
        if (!begin_folded)
            q.launch(gn_level_begin_kernel, dim3(ny), dim3(64), o->state, first_level ? 1 : 0, in, bd,
                     (first_level && so3_ran_here) ? 1 : 0);
        first_level = false;
        begin_folded = false;

        for (int j = 0; j < iterations[i]; ++j) {
            const bool last_l0 = (i == 0 && j == iterations[i] - 1);
            int res_records = 0, icp_records = 0;
            RgbResidualArgs ra;
            bool res_vec4 = false;
            if (rgb) {  // :363-371
                const float min_scale = (float)(std::pow((double)o->min_grad[i], 2.0) / std::pow((double)o->sobel_scale, 2.0));
                ra = make_residual_args(min_scale, o->dIdx[i], 0, o->dIdy[i], 0, o->last_depth[i], 0,
                                        o->prep_batched ? o->last_depth[i] : o->next_depth[i], 0,
                                        o->last_image[i], 0, o->next_image[i], 0, o->corres[i], o->max_depth_delta_rgb,
                                        cols, rows, last_l0 ? rgb_err_dev : nullptr, 0);
                ra.intr = in;
                ra.extent = cull_gen ? o->extent : nullptr, ra.extent_gen = cull_gen, ra.extent_level = i;
                res_vec4 = residual_vec4_ok(ra);
                res_records = reduce_grid(cols * rows, res_vec4 ? kBlock * 4 : kBlock);
            }
            ChainGeom geom;
            std::memset(&geom, 0, sizeof(geom));
            IcpArgs ia;
            int ipx = 1;
            if (icp) {
                ia = odom_icp_args(o, i, last_l0 ? icp_err_dev : nullptr);
                ipx = std::min(2, icp_max_px(ia, 2));
            }
            const bool fuse_producers = rgb && icp && res_vec4 && icp2_fits(ia, kBlock, ipx) &&
                                        icp_default_variant(cols * rows) / 1000000 != 1;
            if (fuse_producers) {  // ICP reduction + correspondence pass side by side in one launch
                icp_records = (cols * rows + kBlock * ipx - 1) / (kBlock * ipx);
                if (lead_sparse)  // (every model of this launch, then)
                    res_records = quarter(res_records);
                else if (foll_sparse)
                    geom.res_f = (unsigned)quarter(res_records);
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (o->timing >= 2 && o->n_timed < kMaxTimedLaunches) {
                    e0 = o->ev_kernel[2 * o->n_timed], e1 = o->ev_kernel[2 * o->n_timed + 1];
                    o->timed_kind[o->n_timed++] = i * 2;
                }
                if (!e0 && ipx == 2)
                    q.launch((track_producer_kernel<2, true>), dim3(icp_records + res_records, ny), dim3(kBlock), o->state, ia,
                             (unsigned)icp_records, ra, o->gn_partials_icp, o->gn_partials_res, bd, geom);
                else if (!e0)
                    q.launch((track_producer_kernel<1, true>), dim3(icp_records + res_records, ny), dim3(kBlock), o->state, ia,
                             (unsigned)icp_records, ra, o->gn_partials_icp, o->gn_partials_res, bd, geom);
                else if (ipx == 2)
                    hipExtLaunchKernelGGL((track_producer_kernel<2, true>), dim3(icp_records + res_records, ny), dim3(kBlock), 0,
                                          c->stream, e0, e1, 0, o->state, ia, (unsigned)icp_records, ra, o->gn_partials_icp,
                                          o->gn_partials_res, bd, geom);
                else
                    hipExtLaunchKernelGGL((track_producer_kernel<1, true>), dim3(icp_records + res_records, ny), dim3(kBlock), 0,
                                          c->stream, e0, e1, 0, o->state, ia, (unsigned)icp_records, ra, o->gn_partials_icp,
                                          o->gn_partials_res, bd, geom);
                MMF_HIP_TRY(hipGetLastError());
            } else {
                MMF_REQUIRE(ny == 1, "odom_enqueue_tracking: this level cannot be batched");
                MMF_HIP_TRY(q.flush());
                if (rgb) {
                    if (res_vec4)
                        hipLaunchKernelGGL((rgb_residual_kernel<FINISH_GN, 4>), dim3(res_records), dim3(kBlock), 0,
                                           c->stream, o->state, ra, o->gn_partials_res);
                    else
                        hipLaunchKernelGGL((rgb_residual_kernel<FINISH_GN, 1>), dim3(res_records), dim3(kBlock), 0,
                                           c->stream, o->state, ra, o->gn_partials_res);
                    MMF_HIP_TRY(hipGetLastError());
                }
                if (icp) {  // :403-410
                    MMF_HIP_TRY(launch_icp<FINISH_GN>(c, o->state, ia, 0, &icp_records, o->gn_partials_icp));
                    if (!rgb) {  // ICP-only tracking: one workgroup sums the records, solves, updates the pose
                        hipLaunchKernelGGL((icp_finish_kernel<FINISH_GN>), dim3(1), dim3(256), 0, c->stream, o->state,
                                           o->gn_partials_icp, (unsigned)icp_records, in);
                        MMF_HIP_TRY(hipGetLastError());
                    }
                }
            }
            if (rgb) {  // :418-423, then :425-460 in the finishing workgroup
                RgbStepArgs a;
                a.residual_partials = o->gn_partials_res;
                a.residual_records = (unsigned)res_records;
                a.icp_partials = o->gn_partials_icp;
                a.icp_records = (unsigned)icp_records;
                a.corres = o->corres[i];
                a.cloud = nullptr, a.cloud4 = reinterpret_cast<const float4*>(o->cloud4[i]);
                a.fx = in.fx;
                a.fy = in.fy;
                a.dIdx = o->dIdx[i];
                a.dIdy = o->dIdy[i];
                a.d_stride = cols;
                a.sobel_scale = o->sobel_scale;
                a.cols = cols;
                a.rows = rows;
                a.cols_magic = ra.cols_magic;
                a.intr = in;
                a.extent = ra.extent, a.extent_gen = ra.extent_gen, a.extent_level = ra.extent_level;
                // the last step of a level also does the next level's gn_level_begin (one launch less per
                // level).  Not with rgbOnly: its divergence `break` skips the finishing lane.
                a.next_level = 0;
                a.final_step = (last_l0 && !rgb_only) ? 1 : 0;  // odom_end in the finishing lane (which rgbOnly's `break` skips)
                end_folded = a.final_step != 0;
                if (j == iterations[i] - 1 && i > 0 && iterations[i - 1] > 0 && !rgb_only) {
                    a.next_level = 1;
                    a.intr = level_intr(o->fx, o->fy, o->cx, o->cy, i - 1);  // only read by the finishing lane's rgb_prepare
                    begin_folded = true;
                }
                int grid = reduce_grid(cols * rows, kBlock * 4);  // width, height are multiples of 4
                if (fuse_producers && lead_sparse)
                    grid = quarter(grid);
                else if (fuse_producers && foll_sparse)
                    geom.step_f = (unsigned)quarter(grid);
                hipEvent_t e0 = nullptr, e1 = nullptr;
                if (o->timing >= 2 && o->n_timed < kMaxTimedLaunches) {
                    e0 = o->ev_kernel[2 * o->n_timed], e1 = o->ev_kernel[2 * o->n_timed + 1];
                    o->timed_kind[o->n_timed++] = i * 2 + 1;
                }
                if (!e0 && res_vec4)  // the 4-pixel correspondence pass wrote compact records
                    q.launch((rgb_step_kernel<FINISH_GN, 4, true>), dim3(grid, ny), dim3(kBlock), o->state, a, o->gn_partials_f,
                             o->gn_ticket, bd, geom);
                else if (!e0)
                    q.launch((rgb_step_kernel<FINISH_GN, 4, false>), dim3(grid, ny), dim3(kBlock), o->state, a, o->gn_partials_f,
                             o->gn_ticket, bd, geom);
                else if (res_vec4)
                    hipExtLaunchKernelGGL((rgb_step_kernel<FINISH_GN, 4, true>), dim3(grid, ny), dim3(kBlock), 0, c->stream, e0, e1,
                                          0, o->state, a, o->gn_partials_f, o->gn_ticket, bd, geom);
                else
                    hipExtLaunchKernelGGL((rgb_step_kernel<FINISH_GN, 4, false>), dim3(grid, ny), dim3(kBlock), 0, c->stream, e0,
                                          e1, 0, o->state, a, o->gn_partials_f, o->gn_ticket, bd, geom);
                MMF_HIP_TRY(hipGetLastError());
            }
        }
    }

    if (!end_folded) q.launch(odom_end_kernel, dim3(ny), dim3(64), o->state, bd);
    MMF_HIP_TRY(q.flush());
    if (o->timing) MMF_HIP_TRY(hipEventRecord(o->ev_chain[1], c->stream));
    PublishTargets to;
    std::memset(&to, 0, sizeof(to));
    const unsigned seq = ++o->publish_seq;
    for (unsigned m = 0; m < ny; ++m) {  // every model's result towards the host, on the chain's stream
        mmf_odom* om = batch ? batch->o[m] : o;
        to.host[m] = om->host_result_dev;
        om->publish_seq = seq;
        // a follower of a batch takes the LEADER's counter: whatever its own earlier chains left in the pinned flag, it is not
        // this chain's number before this chain has written it
        om->host_result->publish_seq = ~seq;
        om->pending_icp = icp, om->pending_so3 = so3 != 0;
        om->track_stream = c->stream;
        om->result_of = o;
        if (om != o) om->so3_prefetched = false;
        // the image ring (:469-473).  Host bookkeeping only -- the launches above hold their pointers by value -- and
        // done here rather than when the result is picked up, so that the next frame's sensor-side preparation can be
        // enqueued while this chain runs
        if (so3)
            for (int i = 0; i < MMF_NUM_PYRS; ++i) std::swap(om->last_next_image[i], om->next_image[i]);
    }
    o->rider = FrameRider();
    if (final_pending && o->defer_publish && ny == 1) {
        hipLaunchKernelGGL(gn_final_kernel, dim3(ny), dim3(kBlock), 0, c->stream, o->state, final_args, bd);
        o->rider.st = o->state, o->rider.host = o->host_result_dev, o->rider.seq = seq;
    } else if (final_pending)
        hipLaunchKernelGGL(gn_final_publish_kernel, dim3(ny), dim3(kBlock), 0, c->stream, o->state, final_args, bd, to, seq);
    else
        hipLaunchKernelGGL(odom_publish_kernel, dim3(ny), dim3(128), 0, c->stream, o->state, to, seq, bd);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

// can several models ride one chain (every level on the fused producer path)?  Same tests as the loop above.
static bool odom_batchable(mmf_odom* o, int rgb_only, float icp_weight, int pyramid, int fast_odom) {
    const bool icp = !rgb_only && icp_weight > 0, rgb = rgb_only || icp_weight < 100;
    if (!icp || !rgb || rgb_only || !o->prep_batched || icp_default_variant(0) / 1000000 == 1) return false;
    const int iterations[MMF_NUM_PYRS] = {fast_odom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0};
    for (int i = 0; i < MMF_NUM_PYRS; ++i) {
        if (!iterations[i]) continue;
        const int cols = o->width >> i, rows = o->height >> i;
        RgbResidualArgs ra = make_residual_args(1.f, o->dIdx[i], 0, o->dIdy[i], 0, o->last_depth[i], 0, o->last_depth[i], 0,
                                                o->last_image[i], 0, o->next_image[i], 0, o->corres[i], o->max_depth_delta_rgb,
                                                cols, rows, nullptr, 0);
        IcpArgs ia = odom_icp_args(o, i, nullptr);
        if (!residual_vec4_ok(ra) || !icp2_fits(ia, kBlock, std::min(2, icp_max_px(ia, 2)))) return false;
    }
    return true;
}

// can the chain run as one launch per iteration (gn_iter_kernel)?  Both terms on, four pixels per lane at every level,
// every level's grid small enough to be resident at once (the count barrier inside the launch).  MMF_GN_FUSED=0 keeps the
// two-launch chain (A/B runs, and the test that compares the two).
static std::atomic<int> g_gn_fused{-1};  // -1: MMF_GN_FUSED decides (default on); 0 / 1: mmf_debug_set_gn_fused
extern "C" int mmf_debug_set_gn_fused(int on) {
    g_gn_fused.store(on < 0 ? -1 : (on ? 1 : 0));
    if (on) g_gn_latched_off.store(false);  // (a test that asks for the one-launch chain again gets it)
    return MMF_OK;
}
// test hook: the next n one-launch chains of this process give up at their third launch (the count barrier of that launch
// polls zero times), as if a workgroup had never arrived: exercises the recovery below without another process on the GPU
extern "C" int mmf_debug_force_gn_fault(int n) {
    g_gn_force_fault.store(n < 0 ? 0 : n);
    return MMF_OK;
}
// how often a one-launch chain gave up and its frame was tracked again on the two-launch chain, and whether the one-launch
// chain is still in use (it is latched off by the first such event: whatever kept its workgroups from being resident
// together -- another process on this GPU -- is likely still there)
extern "C" int mmf_gn_chain_status(int* recoveries, int* one_launch_chain_in_use) {
    if (recoveries) *recoveries = g_gn_recoveries.load();
    if (one_launch_chain_in_use) *one_launch_chain_in_use = g_gn_latched_off.load() ? 0 : 1;
    return MMF_OK;
}
static unsigned fused_max_models() {  // MMF_GN_FUSED_MAX: up to how many models one one-launch chain carries
    return (unsigned)tunables().gn_fused_max;
}
// How many workgroups of gn_iter_kernel<px, .> with `threads` threads the device holds at once: the occupancy the runtime
// reports for the kernel (registers, LDS) x the compute units.  The launch spins on its own workgroups (count barrier), so a
// grid beyond this could only time out; such sizes, partitioned or smaller devices take the two-launch chain.
static long long gn_resident_groups(mmf_ctx* c, int px, int threads, bool mixed = false) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> per_cu;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(px + (mixed ? 100 : 0), threads);
    auto it = per_cu.find(key);
    if (it == per_cu.end()) {
        int nb = 0, nb2 = 0;
        hipError_t e = hipErrorInvalidValue, e2 = hipSuccess;
        switch (px + (mixed ? 100 : 0)) {  // (both variants of a launch shape: the smaller occupancy bounds the chain)
#define MMF_OCC(PXV, KERNEL)                                                                                  \
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, KERNEL<PXV, true>, threads, 0);                  \
    e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, KERNEL<PXV, false>, threads, 0);               \
    break
            case 5: MMF_OCC(5, gn_iter_kernel);
            case 4: MMF_OCC(4, gn_iter_kernel);
            case 2: MMF_OCC(2, gn_iter_kernel);
            case 1: MMF_OCC(1, gn_iter_kernel);
            case 105: MMF_OCC(5, gn_iter_mixed_kernel);
            case 104: MMF_OCC(4, gn_iter_mixed_kernel);
            case 102: MMF_OCC(2, gn_iter_mixed_kernel);
            case 101: MMF_OCC(1, gn_iter_mixed_kernel);
#undef MMF_OCC
            default: break;
        }
        if (e2 != hipSuccess) e = e2;
        nb = std::min(nb, nb2);
        if (e != hipSuccess) nb = 0, (void)hipGetLastError();
        it = per_cu.emplace(key, nb).first;
    }
    return (long long)it->second * c->cu_count;
}
static bool odom_fused_chain_ok(mmf_odom* o, int rgb_only, float icp_weight, int pyramid, int fast_odom, const TrackBatch* batch,
                                bool sparse_on, GnChainPlan* plan) {
    const int models = batch ? batch->n : 1;
    // the models walked by their extents: object models whose preparation noted extents for THIS frame (one number for all)
    GnChainPlan pl;
    for (int m = 0; m < models && sparse_on; ++m) {
        const mmf_odom* om = batch ? batch->o[m] : o;
        if (!om->sparse || om->extent_gen == 0) continue;
        if (pl.ext_gen == 0) pl.ext_gen = om->extent_gen;
        if (om->extent_gen == pl.ext_gen) pl.sparse_mask |= 1u << m;
    }
    const bool env_enabled = tunables().gn_fused;
    const int forced = g_gn_fused.load();
    const bool enabled = (forced < 0 ? env_enabled : forced != 0) && o->exclusive_chain && !g_gn_latched_off.load();
    const bool icp = !rgb_only && icp_weight > 0, rgb = rgb_only || icp_weight < 100;
    if (!enabled || !icp || !rgb || rgb_only) return false;
    const int iterations[MMF_NUM_PYRS] = {fast_odom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0};
    for (int i = 0; i < MMF_NUM_PYRS; ++i) {
        if (!iterations[i]) continue;
        const int cols = o->width >> i, rows = o->height >> i;
        RgbResidualArgs ra = make_residual_args(1.f, o->dIdx[i], 0, o->dIdy[i], 0, o->last_depth[i], 0,
                                                o->prep_batched ? o->last_depth[i] : o->next_depth[i], 0, o->last_image[i], 0,
                                                o->next_image[i], 0, o->corres[i], o->max_depth_delta_rgb, cols, rows, o->rgb_err, 0);
        IcpArgs ia = odom_icp_args(o, i, o->icp_err);
        if (!residual_vec4_ok(ra) || icp_max_px(ia, 4) != 4 || !ia.prev_packed) return false;
        GnGeometry geo;
        if (!gn_geometry(i, cols, rows, &geo, pl.sparse_mask != 0)) return false;
        // the count barrier inside the launch needs every workgroup of it resident at once
        if (geo.lanes > kGnSparseLanes) pl.sparse_mask = 0;  // (the sparse walk keeps a workgroup's correspondences in LDS: sized for 256 lanes)
        long long total = 0;
        for (int m = 0; m < models; ++m) {
            const mmf_odom* om = batch ? batch->o[m] : o;
            pl.groups[i][m] = ((pl.sparse_mask >> m) & 1u) ? gn_sparse_groups(i, geo.groups, om->gn_need[i], geo.lanes) : geo.groups;
            total += pl.groups[i][m];
        }
        if (total > gn_resident_groups(o->ctx, geo.px, geo.threads, pl.sparse_mask != 0)) return false;
    }
    if (plan) *plan = pl;
    return true;
}

// The beginning of the NEXT tracking of one model, to ride the last launch of the preparation that is being enqueued ahead of
// it (track_kernels.hpp: prep_batch_begin_kernel): `pose` = where that tracking will start (the model's pose now), so3_stage =
// the staged pre-alignment of that frame (null: none, the tracking runs it itself).  odom_enqueue_tracking skips its own
// odom_begin_kernel when it is about to pass these very arguments and the caller says the state was left alone (begin_spec_ok).
static std::atomic<int> g_begin_rider{-1};      // -1: MMF_BEGIN_RIDER decides (default on); 0 / 1: mmf_debug_set_begin_rider
static std::atomic<int> g_begin_riders_used{0};  // chains that found their beginning done (mmf_debug_begin_rider_count)
extern "C" int mmf_debug_set_begin_rider(int on) {
    g_begin_rider.store(on < 0 ? -1 : (on ? 1 : 0));
    return MMF_OK;
}
extern "C" int mmf_debug_begin_rider_count(void) { return g_begin_riders_used.load(); }
static void odom_begin_rider_used() { g_begin_riders_used.fetch_add(1); }
static bool odom_begin_rider(mmf_odom* o, const float pose[16], int rgb_only, float icp_weight, int pyramid, int fast_odom, int so3,
                             const OdomState* so3_stage, BeginRider* out) {
    o->begin_spec_valid = o->begin_spec_ok = false;
    const int forced = g_begin_rider.load();
    if (!(forced < 0 ? tunables().begin_rider : forced != 0) || o->two_launch_once) return false;
    const bool fused = odom_fused_chain_ok(o, rgb_only, icp_weight, pyramid, fast_odom, nullptr, odom_sparse_on(), nullptr);
    const float trans[3] = {pose[3], pose[7], pose[11]};
    const float rot[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
    std::memset(out, 0, sizeof(*out));
    out->st = o->state;
    out->a = odom_begin_args(o, trans, rot, rgb_only, icp_weight, pyramid, fast_odom, so3, so3_stage != nullptr, so3_stage, fused);
    o->begin_spec = out->a;
    o->begin_spec_valid = true;
    return true;
}

// second half: wait for the stream, hand the result out
static int odom_finish_tracking(mmf_odom* o, float trans[3], float rot[9]) {
    mmf_ctx* c = o->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    // the chain it rode on publishes into this odometry's pinned state and then stores the sequence number
    if (o->result_of) {
        const volatile unsigned* flag = &o->host_result->publish_seq;
        bool seen = false;
        const auto t_poll = std::chrono::steady_clock::now();
        while (!seen) {  // a few seconds of polling at most, then the stream says why
            for (int i = 0; i < 4096 && !seen; ++i) seen = *flag == o->publish_seq;
            if (seen || std::chrono::steady_clock::now() - t_poll > std::chrono::seconds(5)) break;
        }
        if (!seen) {
            MMF_HIP_TRY(hipStreamSynchronize(o->track_stream));
            MMF_REQUIRE(*flag == o->publish_seq, "odometry: the tracking result did not reach the host");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        MMF_HIP_TRY(wait_stream(c->stream));
    }
    o->track_stream = nullptr, o->result_of = nullptr;
    const bool icp = o->pending_icp, so3 = o->pending_so3;
    if (o->timing) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, o->ev_chain[0], o->ev_chain[1]) == hipSuccess) {
            o->timing_acc.chain_us_sum += ms * 1e3;
            o->timing_acc.chains += 1;
        }
        for (int k = 0; k < o->n_timed; ++k)
            if (hipEventElapsedTime(&ms, o->ev_kernel[2 * k], o->ev_kernel[2 * k + 1]) == hipSuccess) {
                const int lvl = o->timed_kind[k] >> 1, step = o->timed_kind[k] & 1;
                double* sum = step ? o->timing_acc.rgb_step_us_sum : o->timing_acc.producer_us_sum;
                double* mn = step ? o->timing_acc.rgb_step_us_min : o->timing_acc.producer_us_min;
                int* n = step ? o->timing_acc.rgb_step_launches : o->timing_acc.producer_launches;
                sum[lvl] += ms * 1e3;
                if (n[lvl] == 0 || ms * 1e3 < mn[lvl]) mn[lvl] = ms * 1e3;
                n[lvl] += 1;
            }
        o->n_timed = 0;
    }

    const OdomState* r = o->host_result;
    if (o->walked_by_extent) std::memcpy(o->gn_need, r->gn_need, sizeof(o->gn_need));  // (valid also when the chain gave up)
    if (r->gn_fault) {  // the one-launch chain gave up: nothing of its result is valid
        o->last_gn_fault = r->gn_fault;
        return kGnRetry;
    }
    o->last_gn_fault = 0;
    std::memcpy(trans, r->trans_out, sizeof(float) * 3);
    std::memcpy(rot, r->rot_out, sizeof(float) * 9);
    // members the reference leaves untouched in a given mode keep their previous values
    if (icp) {
        o->stats.lastICPError = r->st.lastICPError;
        o->stats.lastICPCount = r->st.lastICPCount;
    }
    o->stats.lastRGBError = r->st.lastRGBError;
    o->stats.lastRGBCount = r->st.lastRGBCount;
    if (so3) {
        o->stats.lastSO3Error = r->st.lastSO3Error;
        o->stats.lastSO3Count = r->st.lastSO3Count;
    }
    if (r->st.iterations_run > 0) {
        std::memcpy(o->stats.lastA, r->st.lastA, sizeof(double) * 36);
        std::memcpy(o->stats.lastb, r->st.lastb, sizeof(double) * 6);
    }
    o->stats.iterations_run = r->st.iterations_run;
    o->stats.so3_iterations_run = r->st.so3_iterations_run;
    return MMF_OK;
}

// A one-launch chain gave up (odom_finish_tracking returned kGnRetry for a model it tracked): the process stops using that
// chain, and every odometry the chain drove is put back to where odom_enqueue_tracking found it -- the image ring (:469-473)
// and the prefetched SO3 pre-alignment it consumed -- so that the same call can be enqueued again and take the two-launch
// chain from the pose the frame started with.  RGBDOdometry.cpp:464-467: the call returns a pose or reverts, it never aborts.
static void odom_retrack_prepare(mmf_odom* const* odoms, int n, int so3) {
    // An object model whose extent did not fit its workgroups (kGnFaultExtent) says nothing about the GPU: this frame is tracked
    // again on the two-launch chain, the model's next chain is sized by what the box needed (odom_finish_tracking has taken
    // OdomState::gn_need over), and the one-launch chain stays in use.
    bool extent_only = true;
    for (int k = 0; k < n; ++k) {
        const int fault = odoms[k]->host_result ? odoms[k]->host_result->gn_fault : 0;
        if (fault != 0 && fault != kGnFaultExtent) extent_only = false;
    }
    if (!extent_only) g_gn_latched_off.store(true);
    g_gn_recoveries.fetch_add(1);
    for (int k = 0; k < n; ++k) {
        mmf_odom* o = odoms[k];
        o->two_launch_once = true;
        if (so3)
            for (int i = 0; i < MMF_NUM_PYRS; ++i) std::swap(o->last_next_image[i], o->next_image[i]);
        o->so3_prefetched = odoms[0]->retry_so3_prefetched, o->so3_stage = odoms[0]->retry_so3_stage;
        o->track_stream = nullptr, o->result_of = nullptr;
    }
}

extern "C" int mmf_odom_get_incremental_transformation(mmf_odom* o, float trans[3], float rot[9], int rgb_only,
                                                       float icp_weight, int pyramid, int fast_odom, int so3,
                                                       float* icp_err_dev, float* rgb_err_dev) {
    MMF_REQUIRE(o && trans && rot, "mmf_odom_get_incremental_transformation: null argument");
    const float trans0[3] = {trans[0], trans[1], trans[2]};
    float rot0[9];
    std::memcpy(rot0, rot, sizeof(rot0));
    int rc = odom_enqueue_tracking(o, trans, rot, rgb_only, icp_weight, pyramid, fast_odom, so3, icp_err_dev, rgb_err_dev);
    if (rc) return rc;
    rc = odom_finish_tracking(o, trans, rot);
    if (rc != kGnRetry) return rc;
    odom_retrack_prepare(&o, 1, so3);
    rc = odom_enqueue_tracking(o, trans0, rot0, rgb_only, icp_weight, pyramid, fast_odom, so3, icp_err_dev, rgb_err_dev);
    if (rc) return rc;
    rc = odom_finish_tracking(o, trans, rot);
    return rc == kGnRetry ? gn_retry_twice() : rc;
}

// measurement mode: per-launch durations of the Gauss-Newton kernels from the dispatches' own timestamps
extern "C" int mmf_odom_enable_timing(mmf_odom* o, int on) {
    MMF_REQUIRE(o != nullptr, "mmf_odom_enable_timing: null odometry");
    MMF_HIP_TRY(hipSetDevice(o->ctx->device));
    if (on && !o->ev_chain[0]) {
        for (hipEvent_t& e : o->ev_kernel) MMF_HIP_TRY(hipEventCreate(&e));
        for (hipEvent_t& e : o->ev_chain) MMF_HIP_TRY(hipEventCreate(&e));
    }
    o->timing = on < 0 ? 0 : (on > 2 ? 2 : on);
    std::memset(&o->timing_acc, 0, sizeof(o->timing_acc));
    return MMF_OK;
}
extern "C" int mmf_odom_get_timing(mmf_odom* o, mmf_odom_timing* out) {
    MMF_REQUIRE(o && out, "mmf_odom_get_timing: null argument");
    *out = o->timing_acc;
    return MMF_OK;
}

extern "C" int mmf_odom_get_stats(mmf_odom* o, mmf_odom_stats* out) {
    MMF_REQUIRE(o && out, "mmf_odom_get_stats: null argument");
    *out = o->stats;
    return MMF_OK;
}

// RGBDOdometry.cpp:479 -- lastA.lu().inverse(); Gauss-Jordan with partial pivoting on the host
extern "C" int mmf_odom_get_covariance(mmf_odom* o, double cov[36]) {
    MMF_REQUIRE(o && cov, "mmf_odom_get_covariance: null argument");
    double a[6][12];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            a[i][j] = o->stats.lastA[i * 6 + j];
            a[i][6 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 6; ++col) {
        int piv = col;
        for (int r = col + 1; r < 6; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (piv != col)
            for (int k = 0; k < 12; ++k) std::swap(a[piv][k], a[col][k]);
        const double d = a[col][col];
        for (int k = 0; k < 12; ++k) a[col][k] /= d;
        for (int r = 0; r < 6; ++r)
            if (r != col) {
                const double f = a[r][col];
                for (int k = 0; k < 12; ++k) a[r][k] -= f * a[col][k];
            }
    }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) cov[i * 6 + j] = a[i][6 + j];
    return MMF_OK;
}

extern "C" int mmf_odom_buffer(mmf_odom* o, const char* name, int level, void** dev_ptr, size_t* bytes) {
    MMF_REQUIRE(o && name && dev_ptr && bytes && level >= 0 && level < MMF_NUM_PYRS, "mmf_odom_buffer: bad argument");
    const size_t n = (size_t)(o->width >> level) * (o->height >> level);
    const std::string s(name);
    void* p = nullptr;
    size_t b = 0;
    if (s == "vmaps_curr") p = o->vmaps_curr[level], b = 3 * n * 4;
    else if (s == "nmaps_curr") p = o->nmaps_curr[level], b = 3 * n * 4;
    else if (s == "vmaps_g_prev") p = o->vmaps_g_prev[level], b = 3 * n * 4;
    else if (s == "nmaps_g_prev") p = o->nmaps_g_prev[level], b = 3 * n * 4;
    else if (s == "last_depth") p = o->last_depth[level], b = n * 4;
    else if (s == "next_depth") p = o->next_depth[level], b = n * 4;
    else if (s == "depth_pyr") p = o->depth_pyr[level], b = n * 4;
    else if (s == "cloud") p = o->cloud[level], b = 3 * n * 4;
    else if (s == "cloud4") p = o->cloud4[level], b = 4 * n * 4;         // {X, Y, Z, 1/Z} per pixel
    else if (s == "prev_packed") p = o->prev_packed[level], b = 6 * n * 4;  // {vertex, normal} per pixel
    else if (s == "last_image") p = o->last_image[level], b = n;
    else if (s == "next_image") p = o->next_image[level], b = n;
    else if (s == "last_next_image") p = o->last_next_image[level], b = n;
    else if (s == "dIdx") p = o->dIdx[level], b = n * 2;
    else if (s == "dIdy") p = o->dIdy[level], b = n * 2;
    else if (s == "corres") p = o->corres[level], b = n * sizeof(mmf_dataterm);
    else if (s == "icp_error") p = o->icp_err, b = (size_t)o->width * o->height * 4;  // Model::icpError / rgbError (R32F, full size;
    else if (s == "rgb_error") p = o->rgb_err, b = (size_t)o->width * o->height * 4;  // `level` is ignored)
    else return fail(MMF_ERR_INVALID, "mmf_odom_buffer: unknown buffer name '" + s + "'");
    *dev_ptr = p;
    *bytes = b;
    return MMF_OK;
}

extern "C" int mmf_odom_download(mmf_odom* o, const char* name, int level, void* host_dst, size_t host_bytes) {
    MMF_REQUIRE(host_dst != nullptr, "mmf_odom_download: null destination");
    void* p = nullptr;
    size_t b = 0;
    int rc = mmf_odom_buffer(o, name, level, &p, &b);
    if (rc) return rc;
    MMF_REQUIRE(b == host_bytes, "mmf_odom_download: size mismatch");
    MMF_HIP_TRY(hipSetDevice(o->ctx->device));
    MMF_HIP_TRY(hipMemcpyAsync(host_dst, p, b, hipMemcpyDeviceToHost, o->ctx->stream));
    MMF_HIP_TRY(hipStreamSynchronize(o->ctx->stream));
    return MMF_OK;
}

extern "C" int mmf_odom_time_icp_kernel(mmf_odom* o, int level, int reps, int variant, float* mean_us_out) {
    MMF_REQUIRE(o && mean_us_out && level >= 0 && level < MMF_NUM_PYRS && reps > 0, "mmf_odom_time_icp_kernel: bad argument");
    mmf_ctx* c = o->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    IcpArgs a = odom_icp_args(o, level, nullptr);
    // the state's pose fields are whatever the last getIncrementalTransformation left (or zero);
    // FINISH_RAW only writes out_f
    for (int w = 0; w < 3; ++w) MMF_HIP_TRY(launch_icp<FINISH_RAW>(c, o->state, a, variant));
    MMF_HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; ++r) MMF_HIP_TRY(launch_icp<FINISH_RAW>(c, o->state, a, variant));
    MMF_HIP_TRY(hipEventRecord(c->ev1, c->stream));
    MMF_HIP_TRY(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    MMF_HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *mean_us_out = ms * 1000.0f / reps;
    return MMF_OK;
}

// =============================================================================================
// Surfel model (Core/Model/Model.{h,cpp} + Core/Model/ModelProjection.{h,cpp}), pure HIP
// =============================================================================================
#include "surfel_kernels.hpp"
#include "pass_rect.hpp"

constexpr int kCountWord = 8, kCountSeqWord = 9;  // in mmf_model::host_totals

struct mmf_model {
    mmf_ctx* ctx = nullptr;
    int width = 0, height = 0;
    float cx = 0, cy = 0, fx = 0, fy = 0;
    unsigned char id = 0;
    float conf_threshold = 10.f;
    float max_depth = FLT_MAX;  // Model::maxDepth (Model.h:129, set per object from the segmentation: MultiMotionFusion.cpp:486,586)
    int capacity = 0;
    unsigned long long tex_gen = 0;    // bumped by every pass that rewrites the prediction / fill-in images
    // generation of the thumbnail counters (thumbnail_count_px): bumped only by the resolve passes that count into them, so that
    // a stand-alone performFillIn (which rewrites images but counts nothing) cannot flip the slot a reader looks at
    unsigned long long thumb_gen = 0;
    const float* pose_dev = nullptr;    // likewise the pose itself and computeFusionWeight(1) (OdomState::pose_out,
    const float* weight_dev = nullptr;  // fusion_weight) for a fuse pass; `weighting` is then the multiplier
    const int* abort_dev = nullptr;    // ... and the word that tells such a pass that the result it would read is void (MMF_SPECULATION_GUARD)
    const float* t_inv_dev = nullptr;  // set by the orchestrator around projections it enqueues before the tracked pose has
                                       // reached the host: the device copy of inverse(pose) (OdomState::pose_inv)
    FrameRider rider;  // set by the orchestrator for the predict() right behind a chain: carried by its resolve launch
    float pose[16];
    unsigned count = 0;  // host copy of the number of surfels in set[cur]
    // After clean() the new count is on its way to host_totals (asynchronous copy on the stream); until a host
    // decision needs it, launches are sized by count_bound and read the exact value from totals[0] on the device.
    bool count_pending = false;
    unsigned count_bound = 0;

    void* slab = nullptr;
    size_t slab_bytes = 0;
    SurfelSoA set[2];  // ping-pong surfel stores (Model.cpp:174-188: two VBOs)
    int cur = 0;
    SurfelSoA meas;    // per-pixel measurement of the data pass (the "update map" + newUnstableBuffer)
    SurfelSoA cand2;   // second per-pixel candidate set (first frame: filtered-depth pass)
    unsigned* flags_a = nullptr;   // count + npix
    unsigned* flags_b = nullptr;   // npix
    unsigned* prefix_a = nullptr;  // count + npix
    unsigned* prefix_b = nullptr;  // npix
    unsigned* block_sums = nullptr;
    unsigned* totals = nullptr;    // [0] scan total a, [1] scan total b, [2] thumbnail count
    unsigned* winner = nullptr;    // capacity
    float2* conf_time = nullptr;   // capacity + npix
    unsigned long long* keys = nullptr;  // npix
    float4* rays = nullptr;              // npix: the pixels' normalised viewing rays (splat_ray_kernel), transposed
    // sparse index map (ModelProjection.cpp:28-41)
    unsigned* index = nullptr;
    float4 *vertConf = nullptr, *colorTime = nullptr, *normRad = nullptr;
    // splat prediction (ModelProjection.cpp:47-55)
    uchar4* image = nullptr;
    float4 *vertexConf = nullptr, *normalRadius = nullptr;
    unsigned short* time_tex = nullptr;
    float* synth_depth = nullptr;  // ModelProjection::synthesizeDepth target (depthTexture, R32F)
    void* export_rm = nullptr;     // row-major copies of the (transposed) index-map images for mmf_model_texture
    // fill-in (Shaders/FillIn.cpp)
    float4 *fill_vertex = nullptr, *fill_normal = nullptr;
    uchar4* fill_image = nullptr;
    unsigned* host_totals = nullptr;      // pinned, device visible: words 0..3 = copies of totals[], kCountWord / kCountSeqWord =
    unsigned* host_totals_dev = nullptr;  // the count the last clean pass published and its sequence number
    unsigned count_seq = 0;
    hipStream_t count_stream = nullptr;  // the stream the last clean pass ran on when it is not the context's (a batched pass)
    // pass_rect.hpp: where this model's key image was written and where its images are non-zero (device), the numbers of its
    // projection / index-resolve / prediction-resolve launches, and whether the non-zero boxes describe the images (a
    // full-frame pass leaves them unknown: the next restricted resolve then covers the whole image once)
    PassBoxes* boxes = nullptr;
    unsigned kgen = 0, igen = 0, sgen = 0;
    bool idx_nz_known = false, spl_nz_known = false;
};

static Cam make_cam(const mmf_model* m, bool double_reciprocal) {
    Cam c;
    c.cx = m->cx, c.cy = m->cy, c.fx = m->fx, c.fy = m->fy;
    if (double_reciprocal) {  // Model.cpp:920-921: 1.0 / fx in double, then to float
        c.ifx = (float)(1.0 / m->fx), c.ify = (float)(1.0 / m->fy);
    } else {  // FeedbackBuffer.cpp:86-87, FillIn.cpp:93-94: 1.0f / fx
        c.ifx = 1.0f / m->fx, c.ify = 1.0f / m->fy;
    }
    return c;
}

static void inverse4f_host(const float* m, float* inv) { inverse4f(m, inv); }  // Eigen `pose.inverse()` (ModelProjection.cpp:108)

static inline dim3 grid1d(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// exclusive scan of n flags -> prefix, grand total -> *total_dev
static int device_scan(mmf_ctx* c, const unsigned* flags, unsigned n, unsigned* prefix, unsigned* block_sums,
                       unsigned* total_dev) {
    const unsigned nblocks = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblocks), dim3(kScanBlock), 0, c->stream, flags, n, block_sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(kScanBlock), 0, c->stream, block_sums, nblocks, total_dev);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nblocks), dim3(kScanBlock), 0, c->stream, flags, n, block_sums, prefix);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_model_create(mmf_ctx* c, int width, int height, float cx, float cy, float fx, float fy,
                                unsigned char id, float conf_threshold, int max_surfels, mmf_model** out) {
    MMF_REQUIRE(c && out, "mmf_model_create: null argument");
    MMF_REQUIRE(width >= 32 && height >= 32 && width % 4 == 0 && height % 4 == 0, "mmf_model_create: bad size");
    MMF_HIP_TRY(hipSetDevice(c->device));
    mmf_model* m = new (std::nothrow) mmf_model();
    MMF_REQUIRE(m != nullptr, "mmf_model_create: out of host memory");
    m->ctx = c;
    m->width = width, m->height = height;
    m->cx = cx, m->cy = cy, m->fx = fx, m->fy = fy;
    m->id = id;
    m->conf_threshold = conf_threshold;
    m->capacity = max_surfels > 0 ? max_surfels : 1024 * 1024;  // Model::MAX_VERTICES (Model.cpp:119-126)
    for (int i = 0; i < 16; ++i) m->pose[i] = (i % 5 == 0) ? 1.f : 0.f;
    const size_t npix = (size_t)width * height, cap = (size_t)m->capacity;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 256);
        return at;
    };
    size_t o_set[2][3], o_meas[3], o_cand[3];
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < 3; ++k) o_set[s][k] = carve(cap * 16);
    for (int k = 0; k < 3; ++k) o_meas[k] = carve(npix * 16);
    for (int k = 0; k < 3; ++k) o_cand[k] = carve(npix * 16);
    const size_t o_fa = carve((cap + npix) * 4), o_fb = carve(npix * 4), o_pa = carve((cap + npix) * 4),
                 o_pb = carve(npix * 4), o_bs = carve(((cap + npix) / 256 + 2) * 4), o_tot = carve(64),
                 o_win = carve(cap * 4), o_ct = carve((cap + npix) * 8), o_keys = carve(npix * 8), o_rays = carve(npix * 16),
                 o_idx = carve(npix * 4), o_vc = carve(npix * 16), o_ctm = carve(npix * 16), o_nr = carve(npix * 16),
                 o_img = carve(npix * 4), o_vxc = carve(npix * 16), o_nrr = carve(npix * 16), o_tt = carve(npix * 2), o_sd = carve(npix * 4), o_ex = carve(npix * 52),
                 o_fv = carve(npix * 16), o_fn = carve(npix * 16), o_fi = carve(npix * 4);
    m->slab_bytes = off;
    hipError_t e = hipMalloc(&m->slab, m->slab_bytes);
    if (e != hipSuccess) {
        delete m;
        return fail(MMF_ERR_HIP, std::string("mmf_model_create: hipMalloc: ") + hipGetErrorString(e));
    }
    MMF_HIP_TRY(hipMemsetAsync(m->slab, 0, m->slab_bytes, c->stream));
    MMF_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->boxes), sizeof(PassBoxes)));
    MMF_HIP_TRY(hipMemsetAsync(m->boxes, 0, sizeof(PassBoxes), c->stream));
    char* b = static_cast<char*>(m->slab);
    for (int s = 0; s < 2; ++s)
        m->set[s] = SurfelSoA{(float4*)(b + o_set[s][0]), (float4*)(b + o_set[s][1]), (float4*)(b + o_set[s][2])};
    m->meas = SurfelSoA{(float4*)(b + o_meas[0]), (float4*)(b + o_meas[1]), (float4*)(b + o_meas[2])};
    m->cand2 = SurfelSoA{(float4*)(b + o_cand[0]), (float4*)(b + o_cand[1]), (float4*)(b + o_cand[2])};
    m->flags_a = (unsigned*)(b + o_fa), m->flags_b = (unsigned*)(b + o_fb);
    m->prefix_a = (unsigned*)(b + o_pa), m->prefix_b = (unsigned*)(b + o_pb);
    m->block_sums = (unsigned*)(b + o_bs), m->totals = (unsigned*)(b + o_tot);
    m->winner = (unsigned*)(b + o_win), m->conf_time = (float2*)(b + o_ct);
    m->keys = (unsigned long long*)(b + o_keys);
    m->rays = (float4*)(b + o_rays);
    m->index = (unsigned*)(b + o_idx);
    m->vertConf = (float4*)(b + o_vc), m->colorTime = (float4*)(b + o_ctm), m->normRad = (float4*)(b + o_nr);
    m->image = (uchar4*)(b + o_img), m->vertexConf = (float4*)(b + o_vxc), m->normalRadius = (float4*)(b + o_nrr);
    m->time_tex = (unsigned short*)(b + o_tt);
    m->synth_depth = (float*)(b + o_sd);
    m->export_rm = (void*)(b + o_ex);
    m->fill_vertex = (float4*)(b + o_fv), m->fill_normal = (float4*)(b + o_fn), m->fill_image = (uchar4*)(b + o_fi);
    hipLaunchKernelGGL(fill_u32_kernel, grid1d(cap), dim3(256), 0, c->stream, m->winner, cap, kNoWinner);
    // the key image starts empty and every resolve kernel hands it back empty
    hipLaunchKernelGGL(fill_u64_kernel, grid1d(npix), dim3(256), 0, c->stream, m->keys, npix, kEmptyKey);
    hipLaunchKernelGGL(splat_ray_kernel, grid1d(npix), dim3(256), 0, c->stream, make_cam(m, false), width, height, m->rays);
    MMF_HIP_TRY(hipGetLastError());
    MMF_HIP_TRY(hipHostMalloc(&m->host_totals, 64, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(m->host_totals, 0, 64);
    MMF_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&m->host_totals_dev), m->host_totals, 0));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    *out = m;
    return MMF_OK;
}

// The model of a rigid body another rank owns: id, thresholds and pose on the host, no surfel store, no images
// (mmf_fusion_set_shard; the round-2 advisor's finding: every rank allocated every model's stores)
static int model_create_bookkeeping(mmf_ctx* c, int width, int height, float cx, float cy, float fx, float fy, unsigned char id,
                                    float conf_threshold, mmf_model** out) {
    mmf_model* m = new (std::nothrow) mmf_model();
    MMF_REQUIRE(m != nullptr, "mmf_model_create: out of host memory");
    m->ctx = c;
    m->width = width, m->height = height;
    m->cx = cx, m->cy = cy, m->fx = fx, m->fy = fy;
    m->id = id;
    m->conf_threshold = conf_threshold;
    m->capacity = 0;
    for (int i = 0; i < 16; ++i) m->pose[i] = (i % 5 == 0) ? 1.f : 0.f;
    *out = m;
    return MMF_OK;
}

extern "C" void mmf_model_destroy(mmf_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->ctx->device);
    (void)hipStreamSynchronize(m->ctx->stream);
    (void)hipFree(m->slab);
    (void)hipFree(m->boxes);
    (void)hipHostFree(m->host_totals);
    delete m;
}

extern "C" int mmf_model_set_pose(mmf_model* m, const float pose[16]) {
    MMF_REQUIRE(m && pose, "mmf_model_set_pose: null argument");
    std::memcpy(m->pose, pose, sizeof(m->pose));
    return MMF_OK;
}
extern "C" int mmf_model_get_pose(mmf_model* m, float pose[16]) {
    MMF_REQUIRE(m && pose, "mmf_model_get_pose: null argument");
    std::memcpy(pose, m->pose, sizeof(m->pose));
    return MMF_OK;
}
// Model::setMaxDepth / setConfidenceThreshold (Model.h:224-230)
extern "C" int mmf_model_set_max_depth(mmf_model* m, float max_depth) {
    MMF_REQUIRE(m != nullptr, "mmf_model_set_max_depth: null model");
    m->max_depth = max_depth;
    return MMF_OK;
}
extern "C" int mmf_model_set_confidence_threshold(mmf_model* m, float conf_threshold) {
    MMF_REQUIRE(m != nullptr, "mmf_model_set_confidence_threshold: null model");
    m->conf_threshold = conf_threshold;
    return MMF_OK;
}
extern "C" float mmf_model_confidence_threshold(mmf_model* m) { return m ? m->conf_threshold : 0.f; }
extern "C" int mmf_model_id(mmf_model* m) { return m ? (int)m->id : -1; }
// makes m->count exact again (after the stream has passed the clean pass that produced it)
static int model_resolve_count(mmf_model* m) {
    if (!m->count_pending) return MMF_OK;
    // only the copy that follows the clean pass is awaited, not whatever has been enqueued since (a stream
    // synchronisation here idled the GPU for ~45 us per frame: the splat of the frame was already in the queue)
    // (it arrives without a runtime copy: the clean pass stores it into pinned memory, then the sequence number)
    const volatile unsigned* flag = &m->host_totals[kCountSeqWord];
    bool seen = false;
    const auto t_poll = std::chrono::steady_clock::now();
    while (!seen) {
        for (int i = 0; i < 4096 && !seen; ++i) seen = *flag == m->count_seq;
        if (seen || std::chrono::steady_clock::now() - t_poll > std::chrono::seconds(5)) break;
    }
    if (!seen) {
        MMF_HIP_TRY(hipStreamSynchronize(m->count_stream ? m->count_stream : m->ctx->stream));
        MMF_REQUIRE(*flag == m->count_seq, "model: the surfel count did not reach the host");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const unsigned total = m->host_totals[kCountWord];
    m->count = total < (unsigned)m->capacity ? total : (unsigned)m->capacity;
    m->count_pending = false;
    return MMF_OK;
}

extern "C" int mmf_model_count(mmf_model* m, unsigned* count) {
    MMF_REQUIRE(m && count, "mmf_model_count: null argument");
    const int rc_count = model_resolve_count(m);
    if (rc_count) return rc_count;
    *count = m->count;
    return MMF_OK;
}

// MultiMotionFusion::filterDepth (MultiMotionFusion.cpp:897-904)
static int filter_depth_on(mmf_ctx* c, Enqueuer& q, const float* depth, int cols, int rows, float max_depth, float* out) {
    MMF_REQUIRE(c && depth && out && cols > 0 && rows > 0, "mmf_filter_depth: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    if (cols % 2 == 0 && ((uintptr_t)depth & 7u) == 0 && ((uintptr_t)out & 7u) == 0)  // two pixels per lane
        q.launch(bilateral_filter2_kernel, tile_grid(cols / 2, rows), tile_block(), depth, cols, rows, max_depth, out);
    else
        q.launch(bilateral_filter_kernel, tile_grid(cols, rows), tile_block(), depth, cols, rows, max_depth, out);
    return MMF_OK;
}
static int filter_depth_on(mmf_ctx* c, hipStream_t stream, const float* depth, int cols, int rows, float max_depth,
                           float* out) {
    Enqueuer q(stream);
    if (int rc = filter_depth_on(c, q, depth, cols, rows, max_depth, out)) return rc;
    MMF_HIP_TRY(q.flush());
    return MMF_OK;
}

extern "C" int mmf_filter_depth(mmf_ctx* c, const float* depth, int cols, int rows, float max_depth, float* out) {
    MMF_REQUIRE(c != nullptr, "mmf_filter_depth: bad argument");
    return filter_depth_on(c, c->stream, depth, cols, rows, max_depth, out);
}

static int model_read_totals(mmf_model* m) {
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipMemcpyAsync(m->host_totals, m->totals, 16, hipMemcpyDeviceToHost, c->stream));
    MMF_HIP_TRY(hipStreamSynchronize(c->stream));
    return MMF_OK;
}

// Model::initialise (Model.cpp:267-312) with the two FeedbackBuffer::compute passes it consumes
extern "C" int mmf_model_initialise(mmf_model* m, const uint8_t* rgb, const float* depth_raw,
                                    const float* depth_filtered, int time, float max_depth) {
    MMF_REQUIRE(m && rgb && depth_raw && depth_filtered, "mmf_model_initialise: null argument");
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const int npix = m->width * m->height;
    const Cam cam = make_cam(m, false);
    hipLaunchKernelGGL(feedback_kernel, grid1d(npix), dim3(256), 0, c->stream, rgb, depth_raw, m->width, m->height, cam,
                       time, max_depth, m->meas, m->flags_a);
    hipLaunchKernelGGL(feedback_kernel, grid1d(npix), dim3(256), 0, c->stream, rgb, depth_filtered, m->width, m->height,
                       cam, time, max_depth, m->cand2, m->flags_b);
    MMF_HIP_TRY(hipGetLastError());
    int rc = device_scan(c, m->flags_a, npix, m->prefix_a, m->block_sums, &m->totals[0]);
    if (rc) return rc;
    rc = device_scan(c, m->flags_b, npix, m->prefix_b, m->block_sums, &m->totals[1]);
    if (rc) return rc;
    hipLaunchKernelGGL(init_scatter_kernel, grid1d(npix), dim3(256), 0, c->stream, npix, m->meas, m->flags_a,
                       m->prefix_a, m->cand2, m->flags_b, m->prefix_b, m->set[m->cur], (unsigned)m->capacity);
    MMF_HIP_TRY(hipGetLastError());
    rc = model_read_totals(m);
    if (rc) return rc;
    // "both raw and filtered have the same amount of vertices" (Model.cpp:292); capped by the buffer
    m->count = m->host_totals[0] < (unsigned)m->capacity ? m->host_totals[0] : (unsigned)m->capacity;
    m->count_pending = false;
    return MMF_OK;
}

static IndexArgs model_index_args(mmf_model* m, int time, float depth_cutoff, int time_delta) {
    IndexArgs a;
    inverse4f_host(m->pose, a.t_inv.m);
    a.t_inv_dev = m->t_inv_dev;
    a.abort_dev = m->abort_dev;
    a.c = make_cam(m, false);
    a.cols = m->width, a.rows = m->height;
    a.maxDepth = depth_cutoff;
    a.time = time, a.timeDelta = time_delta;
    return a;
}

// ModelProjection::predictIndices (ModelProjection.cpp:94-143)
// projected: the surfels are in the key image already (model_fuse with then_index); only the resolve is left
static int model_predict_indices(mmf_model* m, int time, float depth_cutoff, int time_delta, bool projected) {
    MMF_REQUIRE(m != nullptr, "mmf_model_predict_indices: null model");
    if (int rc0 = model_resolve_count(m)) return rc0;
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const size_t npix = (size_t)m->width * m->height;
    const IndexArgs a = model_index_args(m, time, depth_cutoff, time_delta);
    m->idx_nz_known = false;  // (pass_rect.hpp: a full-frame resolve; the non-zero box is not kept)
    // frame_rider.hpp: when this is the frame's first projection, its first launch carries the tracking result's hand-over
    // to the host and its second the fusion weight (each a few microseconds on one extra workgroup, shorter than its carrier)
    FrameRider publish = m->rider, weight = m->rider;
    publish.what = 1u, weight.what = 2u;
    m->rider = FrameRider();
    MMF_REQUIRE(!(projected && publish.st), "mmf_model_predict_indices: a tracking result to hand over, but no projection launch to carry it");
    if (!projected && (m->count || publish.st))
        hipLaunchKernelGGL(index_map_kernel, dim3((unsigned)((m->count + 255) / 256) + (publish.st ? 1u : 0u)), dim3(256), 0, c->stream,
                           m->set[m->cur], (int)m->count, a, m->keys, publish);
    hipLaunchKernelGGL(index_resolve_kernel, dim3((unsigned)((npix + 255) / 256) + (weight.st ? 1u : 0u)), dim3(256), 0, c->stream,
                       m->set[m->cur], a, m->keys, m->index, m->vertConf, m->colorTime, m->normRad, weight);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}
extern "C" int mmf_model_predict_indices(mmf_model* m, int time, float depth_cutoff, int time_delta) {
    return model_predict_indices(m, time, depth_cutoff, time_delta, false);
}

// the two thumbnail counters of thumbnail_count_px (surfel_kernels.hpp): the count of the latest prediction is [thumb_gen & 1]
static unsigned* model_thumb_counts(mmf_model* m) { return &m->totals[4]; }
// splat_kernel's launch: a fixed number of workgroups that deal the surfels out among their waves (surfel_kernels.hpp)
// 512 workgroups = two waves per SIMD for a small store (an object model); a store of half a surfel per pixel and more
// needs the latency of its LDS searches and ray look-ups hidden: 2048 workgroups (combinedPredict 55 -> 52 us on the
// headline loop's 246 k surfels, 124 -> 107 us on 740 k; 1024 / 1280 / 4096: within 1 us of 2048)
static dim3 splat_grid(size_t bound, bool deep = false) {
    const unsigned wgs = tunables().splat_wgs > 0 ? (unsigned)tunables().splat_wgs : (deep ? 2048u : 512u);
    const size_t one_per_thread = (bound + 255) / 256;
    return dim3((unsigned)std::max<size_t>(1, std::min<size_t>(one_per_thread, wgs)));
}
#ifdef MMF_SPLAT_COUNT
extern "C" int mmf_debug_splat_counts(unsigned long long out[4], int reset) {
    MMF_HIP_TRY(hipDeviceSynchronize());
    MMF_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_splat_dbg), 32));
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        MMF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_splat_dbg), z, 32));
    }
    return MMF_OK;
}
#endif
extern "C" int mmf_debug_depth_keys(mmf_ctx* c, const float* z_dev, int n, float max_depth, unsigned* fast_dev, unsigned* divided_dev) {
    MMF_REQUIRE(c && z_dev && fast_dev && divided_dev && n > 0, "mmf_debug_depth_keys: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(depth_key_probe_kernel, grid1d((size_t)n), dim3(256), 0, c->stream, z_dev, n, max_depth, fast_dev, divided_dev);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}
static std::atomic<int> g_splat_bound{-2};  // the early depth test of a deep store; -2: what the environment says (MMF_SPLAT_BOUND), else by the surfel count; -1 / 0 / 1: mmf_debug_set_splat_bound
extern "C" int mmf_debug_set_splat_bound(int mode) {
    g_splat_bound.store(mode < 0 ? -1 : (mode ? 1 : 0));
    return MMF_OK;
}
// ModelProjection::combinedPredict(ACTIVE) (ModelProjection.cpp:187-269); Model.h:210-214
// fill_rgb / fill_depth != nullptr: Model::performFillIn in the same pass as the resolve (the orchestrator's predict)
static int model_combined_predict(mmf_model* m, float depth_cutoff, int time, int max_time, int time_delta,
                                  const uint8_t* fill_rgb, const float* fill_depth, int frame_to_frame_rgb, int lost) {
    MMF_REQUIRE(m != nullptr, "mmf_model_combined_predict: null model");
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    ++m->tex_gen;
    ++m->thumb_gen;
    m->spl_nz_known = false;  // (pass_rect.hpp)
    SplatArgs a;
    inverse4f_host(m->pose, a.t_inv.m);
    a.t_inv_dev = m->t_inv_dev;
    a.abort_dev = m->abort_dev;
    a.c = make_cam(m, false);
    a.cols = m->width, a.rows = m->height;
    a.maxDepth = depth_cutoff;
    a.confThreshold = m->conf_threshold;
    a.time = time, a.maxTime = max_time, a.timeDelta = time_delta;
    // right after clean() the exact count may still be in flight to the host: size the launch by the bound
    // and let the kernel read the count on the device instead of waiting for it
    const unsigned launch_count = m->count_pending ? m->count_bound : m->count;
    // a DEEP store (two surfels per pixel and more: occluded layers) looks at the key image before it evaluates a fragment
    // (surfel_kernels.hpp, splat_kernel<true>): a fraction of the fragments and of their atomics.  Same images either way.
    const int bound_mode = g_splat_bound.load() == -2 ? tunables().splat_bound : g_splat_bound.load();
    const size_t npix_s = (size_t)m->width * m->height;
    const bool deep_store = (size_t)launch_count >= 2 * npix_s;
    const bool deep = bound_mode < 0 ? deep_store : bound_mode != 0;
    a.early_z = deep ? 1 : 0;
    a.rays = m->rays;
    // An OBJECT model (no fill-in, no deep store): the rasterising pass also notes the box of its sprites and the resolve keeps it
    // as the box the prediction is non-zero in (PassBoxes::spl_nz) -- the model-side preparation that follows walks that box
    // instead of the frame.  The grid is sized by what the store can plausibly hold rather than by the bound (see
    // models_combined_predict_rect).
    const bool keep_box = m->boxes != nullptr && !(fill_rgb && fill_depth) && !deep && m->id != 0;
    if (keep_box) {
        ++m->kgen, ++m->sgen;
        const size_t plausible = std::min<size_t>(launch_count, (size_t)m->count * 2 + 8192);
        if (launch_count)
            hipLaunchKernelGGL(splat_box_kernel, splat_grid(plausible, plausible >= npix_s / 2), dim3(256), 0, c->stream, m->set[m->cur],
                               (int)launch_count, a, m->keys, m->count_pending ? m->totals : nullptr, m->boxes, m->kgen);
        hipLaunchKernelGGL(splat_resolve_keep_box_kernel, dim3(splat_tile_grid(m->width, m->height)), dim3(256), 0, c->stream, m->set[m->cur], a, m->keys,
                           m->image, m->vertexConf, m->normalRadius, m->time_tex, model_thumb_counts(m), (int)(m->thumb_gen & 1), m->boxes, m->kgen,
                           m->sgen);
        m->spl_nz_known = true;
        MMF_HIP_TRY(hipGetLastError());
        return MMF_OK;
    }
    if (launch_count)
        hipLaunchKernelGGL(deep ? splat_kernel<true> : splat_kernel<false>, splat_grid(launch_count, (size_t)launch_count >= npix_s / 2), dim3(256), 0, c->stream,
                           m->set[m->cur], (int)launch_count, a, m->keys, m->count_pending ? m->totals : nullptr);
    if (fill_rgb && fill_depth) {  // (a pending frame rider stays for the predictIndices that follows: frame_rider.hpp)
        hipLaunchKernelGGL(splat_resolve_fill_kernel, dim3(splat_tile_grid(m->width, m->height)), dim3(256), 0,
                           c->stream, m->set[m->cur], a, m->keys, m->image, m->vertexConf, m->normalRadius, m->time_tex, fill_depth,
                           fill_rgb, lost ? 1 : 0, (lost || frame_to_frame_rgb) ? 1 : 0, m->fill_vertex, m->fill_normal,
                           m->fill_image, model_thumb_counts(m), (int)(m->thumb_gen & 1));
    } else
        hipLaunchKernelGGL(splat_resolve_kernel, dim3(splat_tile_grid(m->width, m->height)), dim3(256), 0, c->stream, m->set[m->cur], a, m->keys, m->image,
                           m->vertexConf, m->normalRadius, m->time_tex, model_thumb_counts(m), (int)(m->thumb_gen & 1));
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_model_combined_predict(mmf_model* m, float depth_cutoff, int time, int max_time, int time_delta) {
    return model_combined_predict(m, depth_cutoff, time, max_time, time_delta, nullptr, nullptr, 0, 0);
}

// ModelProjection::synthesizeDepth (ModelProjection.cpp:275-335): same sprites and depth test as
// combinedPredict, only corrected_pos.z is kept
extern "C" int mmf_model_synthesize_depth(mmf_model* m, float depth_cutoff, float conf_threshold, int time, int max_time,
                                          int time_delta) {
    MMF_REQUIRE(m != nullptr, "mmf_model_synthesize_depth: null model");
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    SplatArgs a;
    inverse4f_host(m->pose, a.t_inv.m);
    a.t_inv_dev = m->t_inv_dev;
    a.abort_dev = m->abort_dev;
    a.c = make_cam(m, false);
    a.cols = m->width, a.rows = m->height;
    a.maxDepth = depth_cutoff;
    a.confThreshold = conf_threshold;
    a.time = time, a.maxTime = max_time, a.timeDelta = time_delta;
    a.early_z = 0, a.rays = m->rays;
    if (int rc0 = model_resolve_count(m)) return rc0;
    if (m->count)
        hipLaunchKernelGGL(splat_kernel<false>, splat_grid(m->count), dim3(256), 0, c->stream, m->set[m->cur], (int)m->count, a,
                           m->keys, nullptr);
    hipLaunchKernelGGL(splat_depth_resolve_kernel, dim3(splat_tile_grid(m->width, m->height)), dim3(256), 0, c->stream, m->set[m->cur], a, m->keys,
                       m->synth_depth);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

// Model::fuse (Model.cpp:893-1048): data association + update; `weighting` = computeFusionWeight()
// then_index: the update pass also projects every surfel into the key image with these arguments -- the first half of
// the predictIndices that follows a fuse (fuse_update_index_kernel); the caller continues with model_predict_indices(projected)
static int model_fuse(mmf_model* m, int time, const uint8_t* rgb, const uint8_t* mask, const float* depth_raw,
                      const float* depth_filtered, float depth_cutoff, float weighting, const IndexArgs* then_index) {
    MMF_REQUIRE(m && rgb && mask && depth_raw && depth_filtered, "mmf_model_fuse: null argument");
    if (int rc0 = model_resolve_count(m)) return rc0;
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    FuseArgs a;
    std::memcpy(a.pose.m, m->pose, sizeof(m->pose));
    a.c = make_cam(m, true);
    a.cols = m->width, a.rows = m->height;
    a.time = time;
    a.weighting = weighting;
    a.pose_dev = m->pose_dev, a.weight_dev = m->weight_dev, a.weight_mult = weighting;
    a.abort_dev = m->abort_dev;
    a.maskID = m->id;
    a.maxDepth = depth_cutoff < m->max_depth ? depth_cutoff : m->max_depth;  // std::min(depthCutoff, maxDepth) (Model.cpp:928)
    a.count = (int)m->count;
    hipLaunchKernelGGL(fuse_data_kernel, grid1d((size_t)((m->width + 1) / 2) * ((m->height + 1) / 2)), dim3(256), 0, c->stream, rgb, depth_raw, depth_filtered, mask,
                       m->index, m->vertConf, m->normRad, a, m->meas, m->flags_b, m->winner);
    if (m->count && then_index)
        hipLaunchKernelGGL(fuse_update_index_kernel, grid1d(m->count), dim3(256), 0, c->stream, m->set[m->cur], (int)m->count,
                           m->meas, time, m->winner, *then_index, m->keys);
    else if (m->count)
        hipLaunchKernelGGL(fuse_update_kernel, grid1d(m->count), dim3(256), 0, c->stream, m->set[m->cur], (int)m->count,
                           m->meas, time, m->winner);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}
extern "C" int mmf_model_fuse(mmf_model* m, int time, const uint8_t* rgb, const uint8_t* mask, const float* depth_raw,
                              const float* depth_filtered, float depth_cutoff, float weighting) {
    return model_fuse(m, time, rgb, mask, depth_raw, depth_filtered, depth_cutoff, weighting, nullptr);
}

// Model::clean (Model.cpp:1050-1182); must follow mmf_model_fuse + mmf_model_predict_indices of the same frame
extern "C" int mmf_model_clean(mmf_model* m, int time, int time_delta, float depth_cutoff, const float* depth_filtered,
                               const uint8_t* mask, float outlier_coeff) {
    (void)depth_cutoff;  // only consumed by the deformation part, which never runs (nodes == 0)
    MMF_REQUIRE(m && depth_filtered && mask, "mmf_model_clean: null argument");
    if (int rc0 = model_resolve_count(m)) return rc0;
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const int npix = m->width * m->height;
    CleanArgs a;
    inverse4f_host(m->pose, a.t_inv.m);
    a.t_inv_dev = m->t_inv_dev;
    a.abort_dev = m->abort_dev;
    a.c = make_cam(m, false);
    a.cols = m->width, a.rows = m->height;
    a.time = time, a.timeDelta = time_delta;
    a.confThreshold = m->conf_threshold;
    a.outlierCoeff = outlier_coeff;
    a.maskID = m->id;
    a.count = (int)m->count;
    a.npix = npix;
    const unsigned n = m->count + npix;
    hipLaunchKernelGGL(clean_flag_kernel, grid1d(n), dim3(256), 0, c->stream, m->set[m->cur], m->meas, m->flags_b, a,
                       m->index, m->vertConf, m->colorTime, depth_filtered, mask, m->flags_a, m->conf_time, m->block_sums);
    hipLaunchKernelGGL(clean_scatter_kernel, grid1d(n), dim3(256), 0, c->stream, m->set[m->cur], m->meas, (int)m->count,
                       npix, m->flags_a, m->block_sums, m->conf_time, m->set[1 - m->cur], m->capacity, &m->totals[0],
                       m->host_totals_dev + kCountWord, ++m->count_seq, m->abort_dev);
    MMF_HIP_TRY(hipGetLastError());
    // glGetQueryObjectuiv(countQuery) in the reference (Model.cpp:1166) stalls for the count; here the scatter's last
    // workgroup stores it into the host's pinned words, where the next call that needs it on the host picks it up
    m->count_bound = n < (unsigned)m->capacity ? n : (unsigned)m->capacity;
    m->count_pending = true;
    m->count_stream = nullptr;
    m->cur = 1 - m->cur;
    return MMF_OK;
}

// ---- the same passes for SEVERAL models, one launch per pass (surfel_kernels.hpp: *_batched_kernel), all on `st` ----
// predictIndices -> fuse -> predictIndices -> clean (MultiMotionFusion.cpp:791-816) of n <= kMaxPassBatch models without
// frame riders, device-side poses or fill-in (object models): what mmf_model_predict_indices, model_fuse(then_index),
// model_predict_indices(projected) and mmf_model_clean enqueue for each of them, model by model, as seven launches in all.
static int models_fuse_clean_batched(mmf_model* const* ms, int n, hipStream_t st, int time, int time_delta, float depth_cutoff,
                                     const uint8_t* rgb, const uint8_t* mask, const float* depth_raw, const float* depth_filtered,
                                     float outlier_coeff, const float* weighting) {
    MMF_REQUIRE(ms && n >= 1 && n <= kMaxPassBatch && rgb && mask && depth_raw && depth_filtered && weighting,
                "models_fuse_clean_batched: bad argument");
    MMF_HIP_TRY(hipSetDevice(ms[0]->ctx->device));
    PassBatch<index_map_item> b_map;
    PassBatch<index_resolve_item> b_res;
    PassBatch<fuse_data_item> b_fuse;
    PassBatch<fuse_update_index_item> b_upd;
    PassBatch<clean_flag_item> b_flag;
    PassBatch<clean_scatter_item> b_scat;
    unsigned g_map = 0, g_res = 0, g_fuse = 0, g_upd = 0, g_clean = 0;
    for (int k = 0; k < n; ++k) {
        mmf_model* m = ms[k];
        MMF_REQUIRE(m && m->rider.st == nullptr && !m->t_inv_dev && !m->pose_dev && !m->abort_dev,
                    "models_fuse_clean_batched: a model with a pending hand-over or a device-side pose");
        if (int rc0 = model_resolve_count(m)) return rc0;
        m->idx_nz_known = false;
        const unsigned npix = (unsigned)(m->width * m->height);
        const IndexArgs ia = model_index_args(m, time, depth_cutoff, time_delta);
        index_map_item& im = b_map.m[k];
        im.s = m->set[m->cur], im.count = (int)m->count, im.a_in = ia, im.keys = m->keys, im.rider = FrameRider();
        im.grid = (m->count + 255u) / 256u;
        index_resolve_item& ir = b_res.m[k];
        ir.s = m->set[m->cur], ir.a_in = ia, ir.keys = m->keys, ir.index = m->index, ir.vertConf = m->vertConf, ir.colorTime = m->colorTime,
        ir.normRad = m->normRad, ir.rider = FrameRider();
        ir.grid = (npix + 255u) / 256u;
        FuseArgs fa;
        std::memcpy(fa.pose.m, m->pose, sizeof(m->pose));
        fa.c = make_cam(m, true);
        fa.cols = m->width, fa.rows = m->height;
        fa.time = time;
        fa.weighting = weighting[k];
        fa.pose_dev = nullptr, fa.weight_dev = nullptr, fa.weight_mult = weighting[k];
        fa.abort_dev = nullptr;
        fa.maskID = m->id;
        fa.maxDepth = depth_cutoff < m->max_depth ? depth_cutoff : m->max_depth;  // std::min(depthCutoff, maxDepth) (Model.cpp:928)
        fa.count = (int)m->count;
        fuse_data_item& fd = b_fuse.m[k];
        fd.rgb = rgb, fd.depth_raw = depth_raw, fd.depth_fil = depth_filtered, fd.mask = mask, fd.index = m->index, fd.vertConf = m->vertConf,
        fd.normRad = m->normRad, fd.a_in = fa, fd.meas = m->meas, fd.new_flags = m->flags_b, fd.winner = m->winner;
        fd.grid = ((unsigned)((m->width + 1) / 2) * (unsigned)((m->height + 1) / 2) + 255u) / 256u;
        fuse_update_index_item& fu = b_upd.m[k];
        fu.s = m->set[m->cur], fu.count = (int)m->count, fu.meas = m->meas, fu.time = time, fu.winner = m->winner, fu.a_in = ia, fu.keys = m->keys;
        fu.grid = (m->count + 255u) / 256u;
        CleanArgs ca;
        inverse4f_host(m->pose, ca.t_inv.m);
        ca.t_inv_dev = nullptr, ca.abort_dev = nullptr;
        ca.c = make_cam(m, false);
        ca.cols = m->width, ca.rows = m->height;
        ca.time = time, ca.timeDelta = time_delta;
        ca.confThreshold = m->conf_threshold;
        ca.outlierCoeff = outlier_coeff;
        ca.maskID = m->id;
        ca.count = (int)m->count;
        ca.npix = (int)npix;
        const unsigned nc = m->count + npix;
        clean_flag_item& cf = b_flag.m[k];
        cf.s = m->set[m->cur], cf.meas = m->meas, cf.new_flags = m->flags_b, cf.a_in = ca, cf.index = m->index, cf.vertConf = m->vertConf,
        cf.colorTime = m->colorTime, cf.depth_in = depth_filtered, cf.mask = mask, cf.keep = m->flags_a, cf.conf_time = m->conf_time,
        cf.block_sums = m->block_sums;
        cf.grid = (nc + 255u) / 256u;
        clean_scatter_item& cs = b_scat.m[k];
        cs.s = m->set[m->cur], cs.meas = m->meas, cs.count = (int)m->count, cs.npix = (int)npix, cs.keep = m->flags_a, cs.block_sums = m->block_sums,
        cs.conf_time = m->conf_time, cs.dst = m->set[1 - m->cur], cs.capacity = m->capacity, cs.total_out = &m->totals[0],
        cs.total_host = m->host_totals_dev + kCountWord, cs.seq = ++m->count_seq, cs.abort_dev = nullptr;
        cs.grid = cf.grid;
        g_map = std::max(g_map, im.grid), g_res = std::max(g_res, ir.grid), g_fuse = std::max(g_fuse, fd.grid);
        g_upd = std::max(g_upd, fu.grid), g_clean = std::max(g_clean, cf.grid);
        // the host's bookkeeping of mmf_model_clean
        m->count_bound = nc < (unsigned)m->capacity ? nc : (unsigned)m->capacity;
        m->count_pending = true;
        m->count_stream = st;
        m->cur = 1 - m->cur;
    }
    const dim3 blk(256);
    if (g_map) hipLaunchKernelGGL(index_map_batched_kernel, dim3(g_map, n), blk, 0, st, b_map);
    hipLaunchKernelGGL(index_resolve_batched_kernel, dim3(g_res, n), blk, 0, st, b_res);
    hipLaunchKernelGGL(fuse_data_batched_kernel, dim3(g_fuse, n), blk, 0, st, b_fuse);
    if (g_upd) hipLaunchKernelGGL(fuse_update_index_batched_kernel, dim3(g_upd, n), blk, 0, st, b_upd);
    hipLaunchKernelGGL(index_resolve_batched_kernel, dim3(g_res, n), blk, 0, st, b_res);
    hipLaunchKernelGGL(clean_flag_batched_kernel, dim3(g_clean, n), blk, 0, st, b_flag);
    hipLaunchKernelGGL(clean_scatter_batched_kernel, dim3(g_clean, n), blk, 0, st, b_scat);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

// Model::combinedPredict(ACTIVE) (ModelProjection.cpp:187-269) of n models without fill-in, two launches in all
static int models_combined_predict_batched(mmf_model* const* ms, int n, hipStream_t st, float depth_cutoff, int time, int max_time,
                                           int time_delta) {
    MMF_REQUIRE(ms && n >= 1 && n <= kMaxPassBatch, "models_combined_predict_batched: bad argument");
    MMF_HIP_TRY(hipSetDevice(ms[0]->ctx->device));
    PassBatch<splat_item> b_splat;
    PassBatch<splat_resolve_item> b_res;
    unsigned g_splat = 0, g_res = 0;
    for (int k = 0; k < n; ++k) {
        mmf_model* m = ms[k];
        MMF_REQUIRE(m && m->rider.st == nullptr && !m->t_inv_dev && !m->abort_dev, "models_combined_predict_batched: a model with a device-side pose");
        ++m->tex_gen;
        ++m->thumb_gen;
        m->spl_nz_known = false;
        SplatArgs a;
        inverse4f_host(m->pose, a.t_inv.m);
        a.t_inv_dev = nullptr, a.abort_dev = nullptr;
        a.c = make_cam(m, false);
        a.cols = m->width, a.rows = m->height;
        a.maxDepth = depth_cutoff;
        a.confThreshold = m->conf_threshold;
        a.time = time, a.maxTime = max_time, a.timeDelta = time_delta;
        a.early_z = 0;
        a.rays = m->rays;
        const unsigned launch_count = m->count_pending ? m->count_bound : m->count;
        const size_t npix_s = (size_t)m->width * m->height;
        splat_item& sp = b_splat.m[k];
        sp.s = m->set[m->cur], sp.count = (int)launch_count, sp.a_in = a, sp.keys = m->keys, sp.count_dev = m->count_pending ? m->totals : nullptr;
        sp.grid = launch_count ? splat_grid(launch_count, (size_t)launch_count >= npix_s / 2).x : 0u;
        splat_resolve_item& sr = b_res.m[k];
        sr.s = m->set[m->cur], sr.a_in = a, sr.keys = m->keys, sr.image = m->image, sr.vertexConf = m->vertexConf, sr.normalRadius = m->normalRadius,
        sr.time_out = m->time_tex, sr.thumb = model_thumb_counts(m), sr.gen = (int)(m->thumb_gen & 1);
        sr.grid = splat_tile_grid(m->width, m->height);
        g_splat = std::max(g_splat, sp.grid), g_res = std::max(g_res, sr.grid);
    }
    if (g_splat) hipLaunchKernelGGL(splat_batched_kernel, dim3(g_splat, n), dim3(256), 0, st, b_splat);
    hipLaunchKernelGGL(splat_resolve_batched_kernel, dim3(g_res, n), dim3(256), 0, st, b_res);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}
// ---- the same passes RESTRICTED to where the models are (pass_rect.hpp): nine launches for all of them ----
// mask_boxes: the id image's boxes of this frame (mask_boxes_kernel, generation mask_gen), device
static int models_fuse_clean_rect(mmf_model* const* ms, int n, hipStream_t st, int time, int time_delta, float depth_cutoff,
                                  const uint8_t* rgb, const uint8_t* mask, const float* depth_raw, const float* depth_filtered,
                                  float outlier_coeff, const float* weighting, const unsigned long long* mask_boxes, unsigned mask_gen) {
    MMF_REQUIRE(ms && n >= 1 && n <= kMaxPassBatch && rgb && mask && depth_raw && depth_filtered && weighting && mask_boxes,
                "models_fuse_clean_rect: bad argument");
    MMF_HIP_TRY(hipSetDevice(ms[0]->ctx->device));
    PassBatch<index_map_box_item> b_map;
    PassBatch<index_resolve_rect_item> b_res1, b_res2;
    PassBatch<fuse_data_rect_item> b_fuse;
    PassBatch<fuse_update_index_box_item> b_upd;
    PassBatch<clean_rect_item> b_clean;
    unsigned g_map = 0, g_upd = 0;
    for (int k = 0; k < n; ++k) {
        mmf_model* m = ms[k];
        MMF_REQUIRE(m && m->boxes && m->rider.st == nullptr && !m->t_inv_dev && !m->pose_dev && !m->abort_dev,
                    "models_fuse_clean_rect: a model with a pending hand-over or a device-side pose");
        if (int rc0 = model_resolve_count(m)) return rc0;
        const unsigned npix = (unsigned)(m->width * m->height);
        const IndexArgs ia = model_index_args(m, time, depth_cutoff, time_delta);
        // predictIndices: projection (generation kgen + 1) and its resolve
        index_map_box_item& im = b_map.m[k];
        im.s = m->set[m->cur], im.count = (int)m->count, im.a = ia, im.keys = m->keys, im.boxes = m->boxes, im.kgen = ++m->kgen;
        im.grid = (m->count + 255u) / 256u;
        index_resolve_rect_item& r1 = b_res1.m[k];
        r1.s = m->set[m->cur], r1.a = ia, r1.keys = m->keys, r1.index = m->index, r1.vertConf = m->vertConf, r1.colorTime = m->colorTime,
        r1.normRad = m->normRad, r1.boxes = m->boxes, r1.kgen = m->kgen, r1.igen = ++m->igen, r1.prev_whole = m->idx_nz_known ? 0 : 1;
        m->idx_nz_known = true;
        // fuse: data association over the id's box, then the update pass with the next projection (generation kgen + 1)
        FuseArgs fa;
        std::memcpy(fa.pose.m, m->pose, sizeof(m->pose));
        fa.c = make_cam(m, true);
        fa.cols = m->width, fa.rows = m->height;
        fa.time = time;
        fa.weighting = weighting[k];
        fa.pose_dev = nullptr, fa.weight_dev = nullptr, fa.weight_mult = weighting[k];
        fa.abort_dev = nullptr;
        fa.maskID = m->id;
        fa.maxDepth = depth_cutoff < m->max_depth ? depth_cutoff : m->max_depth;  // std::min(depthCutoff, maxDepth) (Model.cpp:928)
        fa.count = (int)m->count;
        fuse_data_rect_item& fd = b_fuse.m[k];
        fd.rgb = rgb, fd.depth_raw = depth_raw, fd.depth_fil = depth_filtered, fd.mask = mask, fd.index = m->index, fd.vertConf = m->vertConf,
        fd.normRad = m->normRad, fd.a = fa, fd.meas = m->meas, fd.new_flags = m->flags_b, fd.winner = m->winner;
        fd.mask_box = mask_boxes + 4 * (size_t)m->id, fd.mask_gen = mask_gen;
        fuse_update_index_box_item& fu = b_upd.m[k];
        fu.s = m->set[m->cur], fu.count = (int)m->count, fu.meas = m->meas, fu.time = time, fu.winner = m->winner, fu.a = ia, fu.keys = m->keys,
        fu.boxes = m->boxes, fu.kgen = ++m->kgen;
        fu.grid = (m->count + 255u) / 256u;
        index_resolve_rect_item& r2 = b_res2.m[k];
        r2 = r1;
        r2.kgen = m->kgen, r2.igen = ++m->igen, r2.prev_whole = 0;
        // clean
        CleanArgs ca;
        inverse4f_host(m->pose, ca.t_inv.m);
        ca.t_inv_dev = nullptr, ca.abort_dev = nullptr;
        ca.c = make_cam(m, false);
        ca.cols = m->width, ca.rows = m->height;
        ca.time = time, ca.timeDelta = time_delta;
        ca.confThreshold = m->conf_threshold;
        ca.outlierCoeff = outlier_coeff;
        ca.maskID = m->id;
        ca.count = (int)m->count;
        ca.npix = (int)npix;
        clean_rect_item& cl = b_clean.m[k];
        cl.s = m->set[m->cur], cl.meas = m->meas, cl.new_flags = m->flags_b, cl.a = ca, cl.index = m->index, cl.vertConf = m->vertConf,
        cl.colorTime = m->colorTime, cl.depth_in = depth_filtered, cl.mask = mask, cl.keep = m->flags_a, cl.conf_time = m->conf_time,
        cl.block_sums = m->block_sums, cl.dst = m->set[1 - m->cur], cl.capacity = m->capacity, cl.total_out = &m->totals[0],
        cl.total_host = m->host_totals_dev + kCountWord, cl.seq = ++m->count_seq, cl.mask_box = fd.mask_box, cl.mask_gen = mask_gen;
        g_map = std::max(g_map, im.grid), g_upd = std::max(g_upd, fu.grid);
        // the host's bookkeeping of mmf_model_clean (the count's bound: every pixel could still be a candidate)
        const unsigned nc = m->count + npix;
        m->count_bound = nc < (unsigned)m->capacity ? nc : (unsigned)m->capacity;
        m->count_pending = true;
        m->count_stream = st;
        m->cur = 1 - m->cur;
    }
    const dim3 blk(256), rect(kRectGroups, n);
    if (g_map) hipLaunchKernelGGL(index_map_box_batched_kernel, dim3(g_map, n), blk, 0, st, b_map);
    hipLaunchKernelGGL(index_resolve_rect_batched_kernel, rect, blk, 0, st, b_res1);
    hipLaunchKernelGGL(fuse_data_rect_batched_kernel, rect, blk, 0, st, b_fuse);
    if (g_upd) hipLaunchKernelGGL(fuse_update_index_box_batched_kernel, dim3(g_upd, n), blk, 0, st, b_upd);
    hipLaunchKernelGGL(index_resolve_rect_batched_kernel, rect, blk, 0, st, b_res2);
    hipLaunchKernelGGL(clean_flag_rect_batched_kernel, rect, blk, 0, st, b_clean);
    hipLaunchKernelGGL(clean_scatter_rect_batched_kernel, rect, blk, 0, st, b_clean);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

static int models_combined_predict_rect(mmf_model* const* ms, int n, hipStream_t st, float depth_cutoff, int time, int max_time,
                                        int time_delta) {
    MMF_REQUIRE(ms && n >= 1 && n <= kMaxPassBatch, "models_combined_predict_rect: bad argument");
    MMF_HIP_TRY(hipSetDevice(ms[0]->ctx->device));
    PassBatch<splat_box_item> b_splat;
    PassBatch<splat_resolve_rect_item> b_res;
    unsigned g_splat = 0;
    for (int k = 0; k < n; ++k) {
        mmf_model* m = ms[k];
        MMF_REQUIRE(m && m->boxes && m->rider.st == nullptr && !m->t_inv_dev && !m->abort_dev, "models_combined_predict_rect: a model with a device-side pose");
        ++m->tex_gen;
        ++m->thumb_gen;
        SplatArgs a;
        inverse4f_host(m->pose, a.t_inv.m);
        a.t_inv_dev = nullptr, a.abort_dev = nullptr;
        a.c = make_cam(m, false);
        a.cols = m->width, a.rows = m->height;
        a.maxDepth = depth_cutoff;
        a.confThreshold = m->conf_threshold;
        a.time = time, a.maxTime = max_time, a.timeDelta = time_delta;
        a.early_z = 0;
        a.rays = m->rays;
        const unsigned launch_count = m->count_pending ? m->count_bound : m->count;
        const size_t npix_s = (size_t)m->width * m->height;
        splat_box_item& sp = b_splat.m[k];
        sp.s = m->set[m->cur], sp.count = (int)launch_count, sp.a = a, sp.keys = m->keys, sp.count_dev = m->count_pending ? m->totals : nullptr;
        sp.boxes = m->boxes, sp.kgen = ++m->kgen;
        // The bound right after clean() is count + every pixel; the waves deal the surfels out whatever the grid is, and every
        // wave with a surfel notes its sprites' box: size the grid by what the store can plausibly hold (the count before the
        // clean pass, doubled, + 8192) -- 4 000 one-surfel waves noting into four words took 180 us
        const size_t plausible = std::min<size_t>(launch_count, (size_t)m->count * 2 + 8192);
        sp.grid = launch_count ? splat_grid(plausible, plausible >= npix_s / 2).x : 0u;
        splat_resolve_rect_item& sr = b_res.m[k];
        sr.s = m->set[m->cur], sr.a = a, sr.keys = m->keys, sr.image = m->image, sr.vertexConf = m->vertexConf, sr.normalRadius = m->normalRadius,
        sr.time_out = m->time_tex, sr.thumb = model_thumb_counts(m), sr.gen = (int)(m->thumb_gen & 1);
        sr.boxes = m->boxes, sr.kgen = m->kgen, sr.sgen = ++m->sgen, sr.prev_whole = m->spl_nz_known ? 0 : 1;
        m->spl_nz_known = true;
        g_splat = std::max(g_splat, sp.grid);
    }
    if (g_splat) hipLaunchKernelGGL(splat_box_batched_kernel, dim3(g_splat, n), dim3(256), 0, st, b_splat);
    hipLaunchKernelGGL(splat_resolve_rect_batched_kernel, dim3(kRectGroups, n), dim3(256), 0, st, b_res);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}
// can a model's combinedPredict go into a batch (models_combined_predict_batched)?  Not a deep store (its rasterising pass is
// another kernel), not one with pending device-side inputs
static bool model_predict_batchable(const mmf_model* m) {
    const unsigned launch_count = m->count_pending ? m->count_bound : m->count;
    const int bound_mode = g_splat_bound.load() == -2 ? tunables().splat_bound : g_splat_bound.load();
    const bool deep = bound_mode < 0 ? (size_t)launch_count >= 2 * (size_t)m->width * m->height : bound_mode != 0;
    return !deep && m->rider.st == nullptr && !m->t_inv_dev && !m->abort_dev && !m->pose_dev;
}

// Model::performFillIn (Model.cpp:1607-1616)
extern "C" int mmf_model_perform_fill_in(mmf_model* m, const uint8_t* rgb, const float* depth_filtered,
                                         int frame_to_frame_rgb, int lost) {
    MMF_REQUIRE(m && rgb && depth_filtered, "mmf_model_perform_fill_in: null argument");
    ++m->tex_gen;
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const int npix = m->width * m->height;
    hipLaunchKernelGGL(fill_in_kernel, grid1d(npix), dim3(256), 0, c->stream, m->vertexConf, m->normalRadius, m->image,
                       depth_filtered, rgb, m->width, m->height, make_cam(m, false), lost ? 1 : 0,
                       (lost || frame_to_frame_rgb) ? 1 : 0, m->fill_vertex, m->fill_normal, m->fill_image);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

// MultiMotionFusion::requiresFillIn (MultiMotionFusion.cpp:877-895)
extern "C" int mmf_model_requires_fill_in(mmf_model* m, float ratio, int* result) {
    MMF_REQUIRE(m && result, "mmf_model_requires_fill_in: null argument");
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const int dc = m->width / 20, dr = m->height / 20;
    MMF_HIP_TRY(hipMemsetAsync(&m->totals[2], 0, 4, c->stream));
    hipLaunchKernelGGL(thumbnail_count_kernel, grid1d(dc * dr), dim3(256), 0, c->stream, m->image, m->width, m->height,
                       &m->totals[2]);
    MMF_HIP_TRY(hipGetLastError());
    int rc = model_read_totals(m);
    if (rc) return rc;
    *result = ((float)m->host_totals[2] / (float)(dr * dc) < ratio) ? 1 : 0;
    return MMF_OK;
}

// Model::getModel() (Model.h:297, Model.cpp:357: the vertex buffer the last fuse / clean left = vbos[target]) as a view of the
// surfel store in HBM: three float4 arrays and the number of surfels in them.  The pointers stay valid until
// the next fuse / clean / initialise of this model (the two stores ping-pong).
extern "C" int mmf_model_surfel_arrays(mmf_model* m, const float** pos_conf, const float** colour_time, const float** normal_radius,
                                       unsigned* count) {
    MMF_REQUIRE(m && pos_conf && colour_time && normal_radius && count, "mmf_model_surfel_arrays: null argument");
    if (int rc0 = model_resolve_count(m)) return rc0;
    *pos_conf = reinterpret_cast<const float*>(m->set[m->cur].pos);
    *colour_time = reinterpret_cast<const float*>(m->set[m->cur].col);
    *normal_radius = reinterpret_cast<const float*>(m->set[m->cur].nrm);
    *count = m->count;
    return MMF_OK;
}

// Model::downloadMap (Model.cpp:1353-1384): 48-byte AoS surfels {pos+conf, colour/time, normal+radius}
extern "C" int mmf_model_download_map(mmf_model* m, float* host_aos, unsigned max_surfels, unsigned* count_out) {
    MMF_REQUIRE(m && host_aos && count_out, "mmf_model_download_map: null argument");
    if (int rc0 = model_resolve_count(m)) return rc0;
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const unsigned n = m->count < max_surfels ? m->count : max_surfels;
    *count_out = n;
    if (!n) return MMF_OK;
    // stage through the (free) other surfel set as an AoS image
    float4* aos = m->set[1 - m->cur].pos;  // 3 * 16 B * capacity contiguous? no: use its three arrays in turn
    (void)aos;
    float4* stage = nullptr;
    MMF_HIP_TRY(hipMalloc(&stage, (size_t)n * 48));
    hipLaunchKernelGGL(soa_to_aos_kernel, grid1d(n), dim3(256), 0, c->stream, m->set[m->cur], (int)n, stage);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(host_aos, stage, (size_t)n * 48, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(MMF_ERR_HIP, std::string("mmf_model_download_map: ") + hipGetErrorString(e));
    return MMF_OK;
}

// test / restore hook: replace the store's content with host AoS surfels
extern "C" int mmf_model_upload_map(mmf_model* m, const float* host_aos, unsigned count) {
    MMF_REQUIRE(m && (host_aos || count == 0) && count <= (unsigned)m->capacity, "mmf_model_upload_map: bad argument");
    mmf_ctx* c = m->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    if (int rc0 = model_resolve_count(m)) return rc0;  // lets an in-flight count land before it is replaced
    m->count = count;
    if (!count) return MMF_OK;
    float4* stage = nullptr;
    MMF_HIP_TRY(hipMalloc(&stage, (size_t)count * 48));
    hipError_t e = hipMemcpyAsync(stage, host_aos, (size_t)count * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(aos_to_soa_kernel, grid1d(count), dim3(256), 0, c->stream, stage, (int)count, m->set[m->cur]);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(stage);
    if (e != hipSuccess) return fail(MMF_ERR_HIP, std::string("mmf_model_upload_map: ") + hipGetErrorString(e));
    return MMF_OK;
}

// device images of the projections (the reference's GPUTexture getters, ModelProjection.h:52-77,
// Model.h:232-244).  names: index vertConf colorTime normRad | image vertexConf normalRadius time |
// depth (synthesizeDepth) | fillVertex fillNormal fillImage
extern "C" int mmf_model_texture(mmf_model* m, const char* name, void** dev_ptr, size_t* bytes) {
    MMF_REQUIRE(m && name && dev_ptr && bytes, "mmf_model_texture: null argument");
    const size_t npix = (size_t)m->width * m->height;
    const std::string s(name);
    void* p = nullptr;
    size_t b = 0;
    // the index map lives transposed in HBM (index_map_kernel); the getters hand out a row-major copy
    // made on the model's stream
    auto exported = [&](auto* src, size_t off) {
        using T = typename std::remove_pointer<decltype(src)>::type;
        T* dst = reinterpret_cast<T*>(reinterpret_cast<char*>(m->export_rm) + off);
        hipLaunchKernelGGL((untranspose_kernel<T>), grid1d(npix), dim3(256), 0, m->ctx->stream, src, m->width, m->height, dst);
        return (void*)dst;
    };
    if (s == "index") p = exported(m->index, 0), b = npix * 4;
    else if (s == "vertConf") p = exported(m->vertConf, npix * 4), b = npix * 16;
    else if (s == "colorTime") p = exported(m->colorTime, npix * 20), b = npix * 16;
    else if (s == "normRad") p = exported(m->normRad, npix * 36), b = npix * 16;
    else if (s == "image") p = m->image, b = npix * 4;
    else if (s == "vertexConf") p = m->vertexConf, b = npix * 16;
    else if (s == "normalRadius") p = m->normalRadius, b = npix * 16;
    else if (s == "time") p = m->time_tex, b = npix * 2;
    else if (s == "depth") p = m->synth_depth, b = npix * 4;
    else if (s == "fillVertex") p = m->fill_vertex, b = npix * 16;
    else if (s == "fillNormal") p = m->fill_normal, b = npix * 16;
    else if (s == "fillImage") p = m->fill_image, b = npix * 4;
    else return fail(MMF_ERR_INVALID, "mmf_model_texture: unknown name '" + s + "'");
    *dev_ptr = p;
    *bytes = b;
    return MMF_OK;
}

// =============================================================================================
// Keypoint descriptor matching (Core/Utils/PointTracker.cpp:100-114) on the f32 matrix cores
// =============================================================================================
#include "match_kernels.hpp"

extern "C" int mmf_match_descriptors(mmf_ctx* c, const float* query, int nq, const float* train, int nt, int dim,
                                     float max_distance, int* train_idx, float* distance) {
    MMF_REQUIRE(c && (query || nq == 0) && (train || nt == 0) && (train_idx || nq == 0) && (distance || nq == 0),
                "mmf_match_descriptors: null argument");
    MMF_REQUIRE(nq >= 0 && nt >= 0 && dim > 0 && dim % 8 == 0, "mmf_match_descriptors: dim must be a positive multiple of 8");
    MMF_REQUIRE(((uintptr_t)query & 15u) == 0 && ((uintptr_t)train & 15u) == 0, "mmf_match_descriptors: 16-byte aligned rows");
    MMF_HIP_TRY(hipSetDevice(c->device));
    if (nq == 0) return MMF_OK;
    // workspace: [qn | tn] floats, then [row keys | col keys]
    const size_t rows = (size_t)nq + (size_t)nt;
    if (rows > c->match_ws_rows) {
        MMF_HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->match_ws);
        c->match_ws = nullptr;
        c->match_ws_rows = 0;
        const size_t cap = rows + rows / 2 + 256;
        MMF_HIP_TRY(hipMalloc(&c->match_ws, cap * (sizeof(float) + sizeof(unsigned long long))));
        c->match_ws_rows = cap;
    }
    unsigned long long* keys = static_cast<unsigned long long*>(c->match_ws);
    float* norms = reinterpret_cast<float*>(keys + c->match_ws_rows);
    unsigned long long *row_best = keys, *col_best = keys + nq;
    float *qn = norms, *tn = norms + nq;
    if (nt == 0)  // nothing to match against: every query stays unmatched
        hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, keys, rows, kNoMatchKey);
    if (nt > 0) {  // the norms kernel also resets the arg-min keys
        hipLaunchKernelGGL(row_norms_kernel, dim3((nq + 31) / 32 + (nt + 31) / 32), dim3(64), 0, c->stream, query, nq, train, nt,
                           dim, qn, tn, row_best, col_best, kNoMatchKey);
        if (dim % kMatchSlab == 0)  // 64 x 64 workgroup tiles with LDS-shared operands
            hipLaunchKernelGGL(match_tile64_kernel, dim3((nt + 63) / 64, (nq + 63) / 64), dim3(256), 0, c->stream, query, train,
                               qn, tn, nq, nt, dim, row_best, col_best);
        else
            hipLaunchKernelGGL(match_tile_kernel, dim3((nt + 31) / 32, (nq + 31) / 32), dim3(64), 0, c->stream, query, train, qn,
                               tn, nq, nt, dim, row_best, col_best);
    }
    hipLaunchKernelGGL(match_cross_check_kernel, dim3((nq + 255) / 256), dim3(256), 0, c->stream, row_best, col_best, nq,
                       max_distance, train_idx, distance);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

// =============================================================================================
// RigidRANSAC (Core/Utils/RigidRANSAC.{h,cpp}): host code, see rigid_ransac.hpp
// =============================================================================================
#include "rigid_ransac.hpp"

struct mmf_ransac {
    mmf::RigidRANSAC impl;
    mmf_ransac(int it, float thr, float frac) : impl(it, thr, frac) {}
};

static void isometry_to_4x4(const mmf::Isometry3f& T, float out[16]) {
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out[r * 4 + c] = T.R[r * 3 + c];
        out[r * 4 + 3] = T.t[r];
    }
    out[12] = out[13] = out[14] = 0.f;
    out[15] = 1.f;
}

extern "C" int mmf_rigid_fit(const float* p0, const float* p1, int n, const unsigned char* mask, float T[16]) {
    MMF_REQUIRE(p0 && p1 && T && n > 0, "mmf_rigid_fit: bad argument");
    isometry_to_4x4(mmf::rigid_fit(p0, p1, n, mask), T);
    return MMF_OK;
}

extern "C" int mmf_rigid_apply(const float T[16], const float* p0, const float* p1, int n, float* distance) {
    MMF_REQUIRE(T && p0 && p1 && distance && n >= 0, "mmf_rigid_apply: bad argument");
    mmf::Isometry3f I;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) I.R[r * 3 + c] = T[r * 4 + c];
        I.t[r] = T[r * 4 + 3];
    }
    mmf::rigid_apply(I, p0, p1, n, distance);
    return MMF_OK;
}

extern "C" int mmf_ransac_create(int iterations, float inlier_threshold, float inlier_fraction, mmf_ransac** out) {
    MMF_REQUIRE(out && iterations >= 0, "mmf_ransac_create: bad argument");
    *out = new (std::nothrow) mmf_ransac(iterations, inlier_threshold, inlier_fraction);
    MMF_REQUIRE(*out != nullptr, "mmf_ransac_create: out of host memory");
    return MMF_OK;
}

extern "C" void mmf_ransac_destroy(mmf_ransac* r) { delete r; }

extern "C" int mmf_ransac_estimate(mmf_ransac* r, const float* p0, const float* p1, int n, const unsigned char* mask,
                                   float T[16], float* error, unsigned char* inlier, int* has_inlier) {
    MMF_REQUIRE(r && p0 && p1 && T && error, "mmf_ransac_estimate: null argument");
    MMF_REQUIRE(n >= 3, "mmf_ransac_estimate: needs at least 3 correspondences (RigidRANSAC.cpp:133)");
    const mmf::RigidRANSAC::Result res = r->impl.estimate(p0, p1, n, mask);
    isometry_to_4x4(res.transformation, T);
    *error = res.error;
    if (has_inlier) *has_inlier = res.inlier.empty() ? 0 : 1;
    if (inlier)
        for (int i = 0; i < n; ++i) inlier[i] = res.inlier.empty() ? 0 : res.inlier[i];
    return MMF_OK;
}

// ---------------------------------------------------------------------------------------------
// SuperPoint keypoint network (Core/MultiMotionFusion.cpp:78,233)
// ---------------------------------------------------------------------------------------------
#include "superpoint_host.hpp"

// ---------------------------------------------------------------------------------------------
// Super-pixel resampling for the segmentation (Core/Segmentation/Slic.h:48-146)
// ---------------------------------------------------------------------------------------------
#include "slic_kernels.hpp"

struct SlicWs {
    mmf::SlicBox* box;
    int *counts, *dcounts, *rgb_sums;
    float* sums;
};

static int slic_workspace(mmf_ctx* c, int n, SlicWs* ws) {
    if ((size_t)n > c->slic_ws_n) {
        MMF_HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->slic_ws);
        c->slic_ws = nullptr, c->slic_ws_n = 0;
        const size_t cap = (size_t)n + n / 2 + 64;
        MMF_HIP_TRY(hipMalloc(&c->slic_ws, cap * (sizeof(mmf::SlicBox) + 6 * sizeof(int))));
        c->slic_ws_n = cap;
    }
    const size_t cap = c->slic_ws_n;
    ws->box = static_cast<mmf::SlicBox*>(c->slic_ws);
    ws->counts = reinterpret_cast<int*>(ws->box + cap);
    ws->dcounts = ws->counts + cap;
    ws->rgb_sums = ws->dcounts + cap;
    ws->sums = reinterpret_cast<float*>(ws->rgb_sums + 3 * cap);
    return MMF_OK;
}

static int slic_check(const int* labels, int width, int height, int spixel_size, const char* who) {
    if (!labels || width <= 0 || height <= 0) return fail(MMF_ERR_INVALID, std::string(who) + ": bad label image");
    // Slic::Slic asserts spixelSize in (10, 256) (Slic.cpp:23)
    if (!(spixel_size > 10 && spixel_size < 256) || width / spixel_size < 1 || height / spixel_size < 1)
        return fail(MMF_ERR_INVALID, std::string(who) + ": super-pixel size must be in (10, 256) and fit the image");
    return MMF_OK;
}

extern "C" int mmf_slic_downsample(mmf_ctx* c, const int* labels, int width, int height, int spixel_size, const float* image,
                                   int channels, int channel, int thresholded, float min_threshold, float* out,
                                   int* counts_out) {
    MMF_REQUIRE(c && image && out, "mmf_slic_downsample: null argument");
    MMF_REQUIRE(channels >= 1 && channel >= 0 && channel < channels, "mmf_slic_downsample: bad channel");
    if (int rc = slic_check(labels, width, height, spixel_size, "mmf_slic_downsample")) return rc;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const int spx = width / spixel_size, spy = height / spixel_size, n = spx * spy, npix = width * height;
    SlicWs ws;
    if (int rc = slic_workspace(c, n, &ws)) return rc;
    using namespace mmf;
    hipLaunchKernelGGL(slic_reset_kernel, grid1d(n), dim3(256), 0, c->stream, n, ws.box, ws.counts, ws.rgb_sums);
    hipLaunchKernelGGL(slic_census_kernel, grid1d(npix), dim3(256), 0, c->stream, labels, width, height, n, ws.box, ws.counts,
                       (const uint8_t*)nullptr, 0, ws.rgb_sums);
    hipLaunchKernelGGL(slic_sum_kernel, dim3((n + 3) / 4), dim3(256), 0, c->stream, labels, width, n, image, channels, channel,
                       thresholded ? 1 : 0, min_threshold, ws.box, ws.sums, ws.dcounts);
    hipLaunchKernelGGL(slic_finish_kernel, grid1d(n), dim3(256), 0, c->stream, labels, width, height, spixel_size, spx, spy,
                       ws.counts, thresholded ? ws.dcounts : ws.counts, ws.sums, out);
    MMF_HIP_TRY(hipGetLastError());
    if (counts_out) MMF_HIP_TRY(hipMemcpyAsync(counts_out, ws.counts, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    return MMF_OK;
}

extern "C" int mmf_slic_downsample_rgb(mmf_ctx* c, const int* labels, int width, int height, int spixel_size,
                                       const uint8_t* rgb, int channels, uint8_t* out) {
    MMF_REQUIRE(c && rgb && out, "mmf_slic_downsample_rgb: null argument");
    MMF_REQUIRE(channels == 3 || channels == 4, "mmf_slic_downsample_rgb: 3 or 4 channels");
    if (int rc = slic_check(labels, width, height, spixel_size, "mmf_slic_downsample_rgb")) return rc;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const int spx = width / spixel_size, spy = height / spixel_size, n = spx * spy, npix = width * height;
    SlicWs ws;
    if (int rc = slic_workspace(c, n, &ws)) return rc;
    using namespace mmf;
    hipLaunchKernelGGL(slic_reset_kernel, grid1d(n), dim3(256), 0, c->stream, n, ws.box, ws.counts, ws.rgb_sums);
    hipLaunchKernelGGL(slic_census_kernel, grid1d(npix), dim3(256), 0, c->stream, labels, width, height, n, ws.box, ws.counts,
                       rgb, channels, ws.rgb_sums);
    hipLaunchKernelGGL(slic_finish_rgb_kernel, grid1d(n), dim3(256), 0, c->stream, labels, width, height, spixel_size, spx, spy,
                       ws.counts, ws.rgb_sums, out);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

extern "C" int mmf_slic_upsample_u8(mmf_ctx* c, const int* labels, int width, int height, const uint8_t* map, int nspix,
                                    uint8_t* out) {
    MMF_REQUIRE(c && labels && map && out && width > 0 && height > 0 && nspix > 0, "mmf_slic_upsample_u8: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    const int npix = width * height;
    hipLaunchKernelGGL(mmf::slic_upsample_u8_kernel, grid1d(npix), dim3(256), 0, c->stream, labels, npix, nspix, map, out);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

// the two exponentials of the surfel path as the gfx950 build evaluates them -- mmf_expf (include/mmf_math.h: surfel
// confidence, SuperPoint softmax, scalar bilateral filter) and the packed expf_nonpositive2 of the two-pixel bilateral
// filter -- on caller-supplied arguments, so that a test can hold BOTH against a float64 exp instead of against the
// oracle that shares their source
__global__ void debug_expf_kernel(const float* __restrict__ x, int n, float* __restrict__ out_mmf, float* __restrict__ out_pk) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const float a = x[i], b = i + 1 < n ? x[i + 1] : 0.f;
    out_mmf[i] = mmf_expf(a);
    if (i + 1 < n) out_mmf[i + 1] = mmf_expf(b);
    const v2fs arg[1] = {v2fs{a, b}};
    v2fs e[1];
    expf_nonpositive2<1>(arg, e);
    out_pk[i] = e[0].x;
    if (i + 1 < n) out_pk[i + 1] = e[0].y;
}
extern "C" int mmf_debug_expf(mmf_ctx* c, const float* x_dev, int n, float* out_mmf_dev, float* out_packed_dev) {
    MMF_REQUIRE(c && x_dev && out_mmf_dev && out_packed_dev && n > 0, "mmf_debug_expf: bad argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(debug_expf_kernel, grid1d(((size_t)n + 1) / 2), dim3(256), 0, c->stream, x_dev, n, out_mmf_dev, out_packed_dev);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}

#ifdef MMF_STAMPS
// diagnostic builds only (tools/rgb_step_probe.py): phase-stamp buffer of the instrumented kernels
extern "C" int mmf_debug_set_stamps(void* dev_buf) {
    unsigned long long* p = static_cast<unsigned long long*>(dev_buf);
    MMF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_mmf_dbg), &p, sizeof(p)));
    MMF_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_mmf_dbg_solve), &p, sizeof(p)));
    return MMF_OK;
}
#endif

// =============================================================================================
// Orchestrator: MultiMotionFusion::processFrame / predict (Core/MultiMotionFusion.cpp:207-854, 863-875)
// =============================================================================================
#include "fusion_orchestrator.hpp"

// =============================================================================================
// Per-rigid-body shard across GPUs: frame broadcast + pose all-gather over RCCL
// =============================================================================================
#include "shard_rccl.hpp"
