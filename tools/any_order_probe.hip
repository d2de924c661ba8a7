// Does hipExtAnyOrderLaunch let a kernel start before the previous kernel of the same stream has finished on this GPU?
// (hip_ext.h says the flag is "not supported on AMD GFX9xx boards" for the module launch API.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/any_order_probe tools/any_order_probe.hip && /tmp/any_order_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin_kernel(unsigned long long* stamps, int slot, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamps[2 * slot] = t0;
    while (wall_clock64() - t0 < ticks) {
    }
    if (threadIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}

int main() {
    unsigned long long* stamps;
    hipHostMalloc((void**)&stamps, 64 * sizeof(unsigned long long), hipHostMallocDefault);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const unsigned long long ticks_50us = (unsigned long long)rate_khz * 50 / 1000;
    for (int flags = 0; flags <= 1; ++flags) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, stamps, 0, ticks_50us);
            hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags ? hipExtAnyOrderLaunch : 0, stamps, 1,
                                  ticks_50us / 10);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, stamps, 2, ticks_50us / 10);
            hipStreamSynchronize(s);
            const double us = 1000.0 / rate_khz;
            printf("flags %d: A %.1f..%.1f  B starts %.1f ends %.1f  C starts %.1f (us after A's start)\n", flags, 0.0,
                   (stamps[1] - stamps[0]) * us, ((double)stamps[2] - (double)stamps[0]) * us, ((double)stamps[3] - (double)stamps[0]) * us,
                   ((double)stamps[4] - (double)stamps[0]) * us);
        }
    }
    return 0;
}
