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
    return 0;
}
