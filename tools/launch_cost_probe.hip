// Diagnostic (GPU box): HOST cost of one kernel launch on a stream, three ways: hipLaunchKernelGGL (what the library uses),
// hipModuleLaunchKernel on a cached hipFunction_t with a kernelParams array, and the same with one packed argument
// buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER).  A frame is ~60 launches: each microsecond per launch is 60 us of host time.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_cost_probe.hip -o /tmp/lcp && /tmp/lcp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Big { float v[64]; float* p; int n; };  // ~270 bytes by value, like the Gauss-Newton kernels' arguments

__global__ void work(Big b, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = b.v[i & 63];
    for (int k = 0; k < spin; ++k) x = x * 1.0001f + 1.0f;
    if (i < b.n) b.p[i] = x;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int n = 76800, per = 40, reps = 300;
    float* d; CK(hipMalloc(&d, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Big b{}; b.p = d; b.n = n;
    int spin = 1500;  // ~6 us kernels: the GPU stays behind the host, the queue never drains
    hipFunction_t fn; CK(hipGetFuncBySymbol(&fn, (const void*)work));
    for (int mode = 0; mode < 3; ++mode) {
        double host = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipStreamSynchronize(s));
            const double t0 = now_us();
            for (int i = 0; i < per; ++i) {
                b.v[0] = (float)i;
                if (mode == 0) {
                    hipLaunchKernelGGL(work, dim3(300), dim3(256), 0, s, b, spin);
                } else if (mode == 1) {
                    void* params[2] = {&b, &spin};
                    CK(hipModuleLaunchKernel(fn, 300, 1, 1, 256, 1, 1, 0, s, params, nullptr));
                } else {
                    struct { Big b; int spin; } packed{b, spin};
                    size_t size = sizeof(packed);
                    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &packed, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
                    CK(hipModuleLaunchKernel(fn, 300, 1, 1, 256, 1, 1, 0, s, nullptr, extra));
                }
            }
            host += now_us() - t0;
        }
        CK(hipStreamSynchronize(s));
        printf("%s: %.2f us of host time per launch\n", mode == 0 ? "hipLaunchKernelGGL          " : mode == 1 ? "hipModuleLaunchKernel params" : "hipModuleLaunchKernel packed", host / reps / per);
    }
    return 0;
}
