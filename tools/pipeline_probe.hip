// GPU box: what a chain of dependent launches costs per link when the dependency is (a) the stream's order and (b) a word in
// device memory that launch k + 1 -- enqueued on a second stream, resident beside launch k -- polls until launch k has set it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pipeline_probe tools/pipeline_probe.hip && /tmp/pipeline_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kShards = 16, kStride = 32;  // counters of one link: 16 words, 128 bytes apart
__device__ __forceinline__ void arrive(unsigned* flags, int k) {
    __hip_atomic_fetch_add(&flags[(k * kShards + (blockIdx.x % kShards)) * kStride], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0 of the workgroup: lanes 0..15 read a shard each
__device__ __forceinline__ bool all_arrived(unsigned* flags, int k, int groups) {
    const int lane = threadIdx.x & 63;
    unsigned v = lane < kShards ? __hip_atomic_load(&flags[(k * kShards + lane) * kStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    for (int d = 8; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return __shfl(v, 0) >= (unsigned)groups;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// `work` rounds of a dependent load chain per workgroup stand for the iteration's own work (~0.6 us each)
__global__ __launch_bounds__(320) void link_kernel(unsigned* flags, int k, int groups, int mode, const unsigned* chase, int work, unsigned* sink, unsigned long long* stuck) {
    unsigned v = threadIdx.x;
    if (mode == 1 && k > 0) {  // wait for launch k - 1: every workgroup's wave 0 polls (bounded)
        if (threadIdx.x < 64) {
            int polls = 0;
            while (!all_arrived(flags, k - 1, groups)) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > 200000) { if (threadIdx.x == 0) atomicAdd(stuck, 1ull); break; }
            }
        }
        __syncthreads();
    }
    for (int w = 0; w < work; ++w) v = chase[(v * 97u + blockIdx.x) & 0xFFFFu];
    if (v == 0xFFFFFFFFu) sink[0] = v;
    __syncthreads();
    if (threadIdx.x == 0) arrive(flags, k);
}

// (c) ONE launch that runs all the links: between links every workgroup arrives at a counter and its wave 0 polls it
__global__ __launch_bounds__(320) void persistent_kernel(unsigned* flags, int links, int groups, const unsigned* chase, int work, unsigned* sink, unsigned long long* stuck, int sleep) {
    unsigned v = threadIdx.x;
    for (int k = 0; k < links; ++k) {
        for (int w = 0; w < work; ++w) v = chase[(v * 97u + blockIdx.x) & 0xFFFFu];
        if (v == 0xFFFFFFFFu) sink[0] = v;
        __syncthreads();
        if (threadIdx.x < 64) {
            if (threadIdx.x == 0) arrive(flags, k);
            int polls = 0;
            while (!all_arrived(flags, k, groups)) {
                if (sleep) __builtin_amdgcn_s_sleep(1);
                if (++polls > 200000) { if (threadIdx.x == 0) atomicAdd(stuck, 1ull); break; }
            }
        }
        __syncthreads();
    }
}

int main() {
    const int N = 40, groups = 240;
    unsigned *flags, *chase, *sink;
    unsigned long long* stuck;
    CK(hipMalloc(&flags, N * kShards * kStride * 4)); CK(hipMalloc(&chase, 65536 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&stuck, 8));
    std::vector<unsigned> h(65536);
    for (unsigned i = 0; i < 65536; ++i) h[i] = (i * 2654435761u) >> 8;
    CK(hipMemcpy(chase, h.data(), 65536 * 4, hipMemcpyHostToDevice)); CK(hipMemset(stuck, 0, 8));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, j;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    for (int work : {0, 4, 8}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f, sum = 0;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipMemsetAsync(flags, 0, N * kShards * kStride * 4, s[0]));
                CK(hipEventRecord(j, s[0])); CK(hipStreamWaitEvent(s[1], j, 0));
                CK(hipStreamSynchronize(s[0]));
                CK(hipEventRecord(e0, s[0]));
                for (int k = 0; k < N; ++k)
                    hipLaunchKernelGGL(link_kernel, dim3(groups), dim3(320), 0, s[mode ? (k & 1) : 0], flags, k, groups, mode, chase, work, sink, stuck);
                if (mode) { CK(hipEventRecord(j, s[1])); CK(hipStreamWaitEvent(s[0], j, 0)); }
                CK(hipEventRecord(e1, s[0]));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
            }
            unsigned long long st; CK(hipMemcpy(&st, stuck, 8, hipMemcpyDeviceToHost));
            if (mode == 1) {
                for (int g2 : {240, 150, 75}) for (int sl = 0; sl < 2; ++sl) {
                    float b2 = 1e9f;
                    for (int rep = 0; rep < 8; ++rep) {
                        CK(hipMemsetAsync(flags, 0, N * kShards * kStride * 4, s[0]));
                        CK(hipEventRecord(e0, s[0]));
                        hipLaunchKernelGGL(persistent_kernel, dim3(g2), dim3(320), 0, s[0], flags, N, g2, chase, work, sink, stuck, sl);
                        CK(hipEventRecord(e1, s[0]));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep >= 2) b2 = ms < b2 ? ms : b2;
                    }
                    printf("work %d  one launch, %3d groups, counter barrier per link%s: %.2f us per link (whole launch / %d)\n", work, g2, sl ? " (sleep)" : "        ", b2 / N * 1e3, N);
                }
            }
            printf("work %d  %s: %.2f us per link (best %.2f), polls given up %llu\n", work, mode ? "two streams + polled word" : "one stream                ", sum / 10 / N * 1e3, best / N * 1e3, st);
        }
    }
    return 0;
}
