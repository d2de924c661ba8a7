// Diagnostic (GPU box): time the conv1b-shaped MFMA kernel with parts of its data path removed, to see what
// bounds it.  Build variants with -DSP_PROBE_NO_B / -DSP_PROBE_NO_A / -DSP_PROBE_NO_STAGE.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 [-D...] -o /tmp/probe tools/sp_conv_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../multimotionfusion_amd/csrc/superpoint_kernels.hpp"

#ifndef PROBE_NT
#define PROBE_NT 2
#endif
#ifndef PROBE_POOL
#define PROBE_POOL true
#endif

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 640, H = argc > 2 ? atoi(argv[2]) : 480;
    const int cin = argc > 3 ? atoi(argv[3]) : 64, cout = argc > 4 ? atoi(argv[4]) : 64;
    const int reps = 20;
    float *in, *wp, *bias, *out;
    const size_t n_in = (size_t)H * W * cin, n_w = (size_t)cin * 9 * ((cout + 31) / 32 * 32), n_out = (size_t)H * W * cout;
    hipMalloc(&in, n_in * 4), hipMalloc(&wp, n_w * 4), hipMalloc(&bias, cout * 4), hipMalloc(&out, n_out * 4);
    std::vector<float> h(n_in);
    for (size_t i = 0; i < n_in; ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.f - 0.5f;
    hipMemcpy(in, h.data(), n_in * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < n_w; ++i) h[i % n_in] = (float)((i * 40503u) >> 6 & 1023) / 8192.f - 0.06f;
    hipMemcpy(wp, h.data(), n_w * 4, hipMemcpyHostToDevice);
    hipMemset(bias, 0, cout * 4);
    mmf::SpConvArgs a;
    a.in = in, a.wpack = wp, a.bias = bias, a.out = out, a.in_stride = cin, a.out_stride = cout;
    a.H = H, a.W = W, a.cin = cin, a.cout = cout, a.relu = 1;
    const dim3 grid((W + 15) / 16, (H + 7) / 8, (cout + 32 * PROBE_NT - 1) / (32 * PROBE_NT));
#ifdef SP_PROBE_STAMPS
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    unsigned long long* stamps;
    hipMalloc(&stamps, nwg * 64);
    hipMemset(stamps, 0, nwg * 64);
    hipMemcpyToSymbol(HIP_SYMBOL(mmf::g_sp_stamps), &stamps, sizeof(stamps));
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mmf::sp_conv_mfma_kernel<PROBE_NT, 9, PROBE_POOL>), grid, dim3(256), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mmf::sp_conv_mfma_kernel<PROBE_NT, 9, PROBE_POOL>), grid, dim3(256), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, fl = 2.0 * H * W * cin * cout * 9;
    printf("%dx%d %d->%d NT=%d: %.1f us  %.1f TFLOP/s (%.1f%% of 157.3)\n", W, H, cin, cout, PROBE_NT, us, fl / us / 1e6,
           fl / us / 1e6 / 157.3 * 100);
#ifdef SP_PROBE_STAMPS
    {  // last launch's stamps: 100 MHz real-time ticks (10 ns)
        std::vector<unsigned long long> hs(nwg * 8);
        hipMemcpy(hs.data(), stamps, nwg * 64, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        double pro = 0, loop = 0, epi = 0;
        for (size_t w = 0; w < nwg; ++w) {
            const unsigned long long* q = &hs[w * 8];
            t0 = q[0] < t0 ? q[0] : t0, t1 = q[3] > t1 ? q[3] : t1;
            pro += (double)(q[1] - q[0]), loop += (double)(q[2] - q[1]), epi += (double)(q[3] - q[2]);
        }
        printf("stamps: kernel span %.1f us; per workgroup mean prologue %.2f us, K loop %.2f us, epilogue %.2f us\n",
               (t1 - t0) * 0.01, pro / nwg * 0.01, loop / nwg * 0.01, epi / nwg * 0.01);
        // concurrency profile: workgroups inside their K loop, sampled every 10 us
        for (unsigned long long t = t0; t < t1; t += 1000) {
            int in_loop = 0, alive = 0;
            for (size_t w = 0; w < nwg; ++w) {
                const unsigned long long* q = &hs[w * 8];
                alive += q[0] <= t && t < q[3];
                in_loop += q[1] <= t && t < q[2];
            }
            printf("  t=%5.0f us alive %4d in-loop %4d\n", (t - t0) * 0.01, alive, in_loop);
        }
    }
#endif
    return 0;
}
