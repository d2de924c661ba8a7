// Diagnostic: where does an icp_kernel launch spend its time?  (GPU box)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DMMF_STAMPS tools/latency_probe.hip -o /tmp/probe
// Phase stamps are 100 MHz wall_clock64 ticks of thread 0 of every workgroup; the build is not
// the shipped one (stamps + forced waits), read its SHARES not its length.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../include/mmf_hip.h"
#include "../multimotionfusion_amd/csrc/track_kernels.hpp"
using namespace mmf;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 12345) p[0] = 1; }

template <int PX, int BLOCK>
int run(int cols, int rows, OdomState* st, const float* vmap, const float* nmap, float* partials, unsigned long long* dbg) {
    IcpArgs a;
    a.vmap_curr = a.vmap_g_prev = MapView{vmap, cols};
    a.nmap_curr = a.nmap_g_prev = MapView{nmap, cols};
    a.intr = LevelIntr{528.f * cols / 640, 528.f * cols / 640, cols / 2.f, rows / 2.f};
    a.dist_thres = 0.1f; a.angle_thres = 0.342f; a.cols = cols; a.rows = rows; a.err_map = nullptr; a.err_stride = cols;
    const int n = cols * rows;
    const int grid = (n + BLOCK * PX - 1) / (BLOCK * PX);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) icp_kernel<PX, BLOCK, 0><<<grid, BLOCK>>>(st, a, partials);
    CK(hipDeviceSynchronize());
    const int reps = 300;
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) icp_kernel<PX, BLOCK, 0><<<grid, BLOCK>>>(st, a, partials); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * 8);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, tend = 0; double ph[6] = {0};
    for (int b = 0; b < grid; ++b) { t0 = std::min(t0, h[b * 8]); tend = std::max(tend, h[b * 8 + 5]); for (int k = 1; k < 6; ++k) ph[k] += (double)(h[b * 8 + k] - h[b * 8 + k - 1]); }
    unsigned long long last_start = 0; for (int b = 0; b < grid; ++b) last_start = std::max(last_start, h[b * 8]);
    printf("%4dx%-4d PX=%d BLOCK=%4d grid=%4d: %6.2f us/launch | in-kernel first-start..last-end %5.2f us, start spread %5.2f us | mean phase us: load %.2f proj %.2f gather %.2f rows %.2f reduce+store %.2f\n",
           cols, rows, PX, BLOCK, grid, ms * 1000 / reps, (tend - t0) * 0.01, (last_start - t0) * 0.01, ph[1] / grid * 0.01, ph[2] / grid * 0.01, ph[3] / grid * 0.01, ph[4] / grid * 0.01, ph[5] / grid * 0.01);
    return 0;
}

int main() {
    const int W = 640, H = 480;
    std::vector<float> vm(3 * W * H), nm(3 * W * H);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const float z = 2.f + 0.001f * x; vm[y * W + x] = z * (x - 320) / 528.f; vm[(y + H) * W + x] = z * (y - 240) / 528.f; vm[(y + 2 * H) * W + x] = z;
        nm[y * W + x] = 0; nm[(y + H) * W + x] = 0; nm[(y + 2 * H) * W + x] = 1; }
    float *dv, *dn, *partials; OdomState* st; unsigned long long* dbg;
    CK(hipMalloc(&dv, vm.size() * 4)); CK(hipMalloc(&dn, nm.size() * 4)); CK(hipMalloc(&partials, 4096 * 128)); CK(hipMalloc(&st, sizeof(OdomState))); CK(hipMalloc(&dbg, 4096 * 64));
    CK(hipMemcpy(dv, vm.data(), vm.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, nm.data(), nm.size() * 4, hipMemcpyHostToDevice));
    OdomState hs; memset(&hs, 0, sizeof(hs));
    for (int k = 0; k < 9; ++k) hs.Rcurr[k] = hs.Rprev_inv[k] = hs.Rprev[k] = (k % 4 == 0);
    CK(hipMemcpy(st, &hs, sizeof(hs), hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_mmf_dbg), &dbg, sizeof(dbg)));
    // launch floor: empty kernels back to back
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int g : {1, 75, 300, 1200}) {
        for (int i = 0; i < 5; ++i) empty_kernel<<<g, 256>>>(nullptr);
        CK(hipEventRecord(e0)); for (int i = 0; i < 300; ++i) empty_kernel<<<g, 256>>>(nullptr); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel grid %4d: %.2f us/launch\n", g, ms * 1000 / 300);
    }
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int c = W >> lvl, r = H >> lvl;  // the level-0 arrays are simply re-read with smaller extents
        if (run<4, 256>(c, r, st, dv, dn, partials, dbg)) return 1;
        if (run<2, 256>(c, r, st, dv, dn, partials, dbg)) return 1;
        if (run<1, 256>(c, r, st, dv, dn, partials, dbg)) return 1;
        if (run<1, 512>(c, r, st, dv, dn, partials, dbg)) return 1;
    }
    return 0;
}
