// standalone check of grid_reduce.hpp (build: hipcc --offload-arch=gfx950 -O3 tools/test_grid_reduce.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../multimotionfusion_amd/csrc/grid_reduce.hpp"
using namespace mmf;
template <int NV, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(float* __restrict__ partials, unsigned* __restrict__ tickets, float* __restrict__ out, int n) {
    __shared__ GridReduceLds<float, BLOCK> lds;
    float v[NV];
    for (int j = 0; j < NV; ++j) v[j] = 0.f;
    for (int i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        for (int j = 0; j < NV; ++j) v[j] += (float)(j + 1);
    if (!grid_reduce<NV, BLOCK>(v, partials, tickets, lds)) return;
    if (threadIdx.x == 0) for (int j = 0; j < NV; ++j) out[j] = lds.total[j];
}
#ifndef NVT
#define NVT 29
#endif
int main() {
    float *partials, *out; unsigned* tickets;
    hipMalloc(&partials, 2048 * 32 * 4); hipMalloc(&tickets, kTicketWords * 4); hipMalloc(&out, 128);
    hipMemset(tickets, 0, kTicketWords * 4);
    for (int grid : {1, 3, 12, 16, 17, 75, 300, 1200, 2048}) {
        int n = grid * 256 - 7;
        hipMemset(out, 0, 128);
        for (int rep = 0; rep < 3; ++rep) k<NVT, 256><<<grid, 256>>>(partials, tickets, out, n);
        float h[32]; hipMemcpy(h, out, 128, hipMemcpyDeviceToHost); bool allok = true; for (int j = 0; j < NVT; ++j) allok &= (h[j] == (float)(j + 1) * n); h[28] = allok ? 29.0f * n : -1.f;
        std::vector<unsigned> t(kTicketWords); hipMemcpy(t.data(), tickets, kTicketWords * 4, hipMemcpyDeviceToHost);
        unsigned tsum = 0; for (auto x : t) tsum += x;
        printf("grid %4d n %7d: out[0]=%.0f (want %d) out[28]=%.0f (want %.0f) tickets_left=%u err=%s\n", grid, n, h[0], n, h[28],
               29.0 * n, tsum, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
