// Diagnostic (GPU box): does work enqueued on stream A AFTER other streams were told to wait for an event of A start
// late?  A: a chain of kernels, event E behind it.  B, C: wait for E, then chains of their own -- enqueued before the
// host waits for A.  Then the host polls an event of A and launches one more kernel on A.  Every kernel logs
// wall_clock64() (100 MHz) at its start and end.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_stall_probe.hip -o /tmp/ssp && /tmp/ssp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void work(unsigned long long* log, int slot, int spin, float* p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) log[2 * slot] = wall_clock64();
    float x = threadIdx.x;
    for (int k = 0; k < spin; ++k) x = x * 1.0001f + 1.0f;
    if (x == 12345.f) p[0] = x;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) log[2 * slot + 1] = wall_clock64();
}

int main() {
    unsigned long long* log; CK(hipHostMalloc(&log, 8 * 512));
    float* d; CK(hipMalloc(&d, 4));
    hipStream_t A, B, C;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    hipEvent_t E, R;
    CK(hipEventCreateWithFlags(&E, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&R, hipEventDisableTiming));
    const int spin = 4000;  // ~10 us
    for (int mode = 0; mode < 3; ++mode) {  // 0: side work enqueued after the host wait; 1: before it; 2: before, no kernels on A after R
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 512; ++i) log[i] = 0;
            int slot = 0;
            auto side = [&]() {
                CK(hipStreamWaitEvent(B, E, 0));
                CK(hipStreamWaitEvent(C, E, 0));
                for (int i = 0; i < 5; ++i) work<<<256, 256, 0, B>>>(log, 100 + i, spin * 2, d);
                for (int i = 0; i < 12; ++i) work<<<64, 256, 0, C>>>(log, 120 + i, spin / 2, d);
                return 0;
            };
            for (int i = 0; i < 30; ++i) work<<<300, 256, 0, A>>>(log, slot++, spin, d);
            CK(hipEventRecord(R, A));
            if (mode != 2)
                for (int i = 0; i < 4; ++i) work<<<300, 256, 0, A>>>(log, slot++, spin, d);  // "early projections"
            CK(hipEventRecord(E, A));
            if (mode >= 1 && side()) return 1;
            while (hipEventQuery(R) == hipErrorNotReady) {}
            for (int i = 0; i < 6; ++i) work<<<300, 256, 0, A>>>(log, slot++, spin, d);  // the fusion passes
            if (mode == 0 && side()) return 1;
            CK(hipDeviceSynchronize());
            const unsigned long long t0 = log[2 * 29 + 1];  // end of A's 30th kernel
            auto us = [&](unsigned long long t) { return t ? ((double)t - (double)t0) / 100.0 : -1.0; };
            const int firstAfter = mode == 2 ? 30 : 34;
            printf("mode %d: A's kernel after the host wait starts at %+7.1f us (previous A kernel ended %+7.1f); B runs %+7.1f .. %+7.1f, C runs %+7.1f .. %+7.1f; A ends %+7.1f\n",
                   mode, us(log[2 * firstAfter]), us(log[2 * (firstAfter - 1) + 1]), us(log[200]), us(log[2 * 104 + 1]), us(log[240]),
                   us(log[2 * 131 + 1]), us(log[2 * (slot - 1) + 1]));
        }
    }
    return 0;
}
