// Diagnostic (GPU box): per-kernel cost of a chain of dependent tiny kernels, launched (a) one by one on a
// stream, (b) as one captured hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_probe.hip -o /tmp/graph_probe && /tmp/graph_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
    const int n = 19200, reps = 200;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int g : {1, 75, 600}) {
        for (int i = 0; i < 10; ++i) tiny<<<g, 256, 0, s>>>(d, n);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) tiny<<<g, 256, 0, s>>>(d, n);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const float stream_us = ms * 1000 / reps;

        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < reps; ++i) tiny<<<g, 256, 0, s>>>(d, n);
        CK(hipStreamEndCapture(s, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(exec, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %4d: stream %.2f us/kernel, graph %.2f us/kernel\n", g, stream_us, ms * 1000 / reps);
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
