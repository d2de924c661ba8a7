// Diagnostic (GPU box): HOST cost of enqueueing a Gauss-Newton-like chain (38 dependent kernels, ~300-byte kernargs)
// (a) launch by launch, (b) as an explicitly built hipGraph replayed with one hipGraphExecKernelNodeSetParams,
// (c) the same with 19 nodes updated per replay; and the GPU time of the chain each way.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_host_probe.hip -o /tmp/ghp && /tmp/ghp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Big { float v[64]; float* p; int n; };  // ~270 bytes by value, like IcpArgs + RgbResidualArgs

__global__ void work(Big b, int spin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x = b.v[i & 63];
    for (int k = 0; k < spin; ++k) x = x * 1.0001f + 1.0f;
    if (i < b.n) b.p[i] = x;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int n = 76800, nodes = 38, frames = 200;
    float* d; CK(hipMalloc(&d, n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Big b{}; b.p = d; b.n = n;
    for (int spin : {200, 2000}) {  // ~3 us and ~8 us kernels
        // (a) stream launches
        for (int i = 0; i < nodes; ++i) work<<<300, 256, 0, s>>>(b, spin);
        CK(hipStreamSynchronize(s));
        double host = 0; float gpu = 0;
        for (int f = 0; f < frames; ++f) {
            CK(hipEventRecord(e0, s));
            const double t0 = now_us();
            for (int i = 0; i < nodes; ++i) { b.v[0] = (float)f; work<<<300, 256, 0, s>>>(b, spin); }
            host += now_us() - t0;
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); gpu += ms * 1000;
        }
        printf("spin %4d stream: host %.1f us per chain (%.2f per launch), GPU %.1f us per chain\n", spin, host / frames, host / frames / nodes, gpu / frames);

        // (b, c) explicit graph
        hipGraph_t graph; CK(hipGraphCreate(&graph, 0));
        std::vector<hipGraphNode_t> node(nodes);
        std::vector<Big> args(nodes, b);
        std::vector<int> spins(nodes, spin);
        std::vector<void*> ptrs(2 * nodes);
        for (int i = 0; i < nodes; ++i) {
            ptrs[2 * i] = &args[i], ptrs[2 * i + 1] = &spins[i];
            hipKernelNodeParams p{};
            p.func = (void*)work; p.gridDim = dim3(300); p.blockDim = dim3(256); p.sharedMemBytes = 0; p.kernelParams = &ptrs[2 * i]; p.extra = nullptr;
            CK(hipGraphAddKernelNode(&node[i], graph, i ? &node[i - 1] : nullptr, i ? 1 : 0, &p));
        }
        hipGraphExec_t exec; CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int upd : {1, 19}) {
            for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, s));
            CK(hipStreamSynchronize(s));
            host = 0, gpu = 0;
            double host_set = 0;
            for (int f = 0; f < frames; ++f) {
                CK(hipEventRecord(e0, s));
                const double t0 = now_us();
                for (int u = 0; u < upd; ++u) {
                    const int i = u * 2 % nodes;
                    args[i].v[0] = (float)f;
                    hipKernelNodeParams p{};
                    p.func = (void*)work; p.gridDim = dim3(300); p.blockDim = dim3(256); p.kernelParams = &ptrs[2 * i];
                    CK(hipGraphExecKernelNodeSetParams(exec, node[i], &p));
                }
                const double t1 = now_us();
                CK(hipGraphLaunch(exec, s));
                host += now_us() - t0, host_set += t1 - t0;
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); gpu += ms * 1000;
            }
            printf("spin %4d graph, %2d node updates: host %.1f us per chain (%.1f in SetParams), GPU %.1f us per chain\n", spin, upd, host / frames,
                   host_set / frames, gpu / frames);
        }
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    return 0;
}
