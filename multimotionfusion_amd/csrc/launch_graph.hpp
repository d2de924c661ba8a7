// launch_graph.hpp -- the enqueue side of a chain of dependent kernel launches.
//
// Rounds 2-3 could record such a chain and replay it as a hipGraph (explicit kernel nodes, per-frame node-parameter updates):
// 38 launches cost the host 140 us one by one and 8.7 us as a graph -- but a frame here waits for the GPU, and a graph launch
// reaches it ~10 us later than the first kernel of a launch-by-launch chain (frame 0.550-0.575 against 0.517-0.525 ms).  The
// replay was never the default and is gone (LABNOTES.md has the numbers); what remains is the call-site interface: launches go
// out in call order, the first error is kept.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

namespace mmf {

struct Enqueuer {
    hipStream_t stream = nullptr;
    hipError_t err = hipSuccess;

    explicit Enqueuer(hipStream_t s) : stream(s) {}

    template <typename... KArgs, typename... Args>
    void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args&&... args) {
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, static_cast<KArgs>(args)...);
        const hipError_t e = hipGetLastError();
        if (err == hipSuccess) err = e;
    }
    // everything launched so far is on the stream; the first error, if any
    hipError_t flush() const { return err; }
};

}  // namespace mmf
