// launch_graph.hpp -- a chain of dependent kernel launches, enqueued with ONE host call (opt-in: MMF_GRAPHS=1).
//
// MEASURED AND NOT ADOPTED AS THE DEFAULT: it cuts the host's enqueue time as described below, but a frame here waits
// for the GPU, not for the host, and a graph launch reaches the GPU later than the first kernel of a launch-by-launch
// chain (numbers at graphs_enabled() in mmf_hip.hip).  Kept for hosts that cannot keep up with ~90 launches per frame.
//
// The reference synchronises with the host >= 67 times per model per frame (RGBDOdometry.cpp:217-477); this
// implementation keeps the Gauss-Newton state on the device and enqueues ~40 launches per model back to back.  What
// is left of the host's share is the enqueue itself: 3-4 us per hipLaunchKernelGGL, 110-140 us per chain -- as much as
// the GPU needs for a third of the chain, and the part of a frame that varies with the load of the host
// (tools/graph_host_probe.hip: 38 launches with ~300-byte arguments, 140 us launch by launch, 8.7 us as a graph with one
// node's arguments replaced, 0.75 us per further node whose arguments changed; the GPU time of the chain is the same or
// slightly lower).
//
// A call site records its launches into a LaunchPlan (kernel, grid, block, a copy of every argument).  GraphCache
// keeps a few instantiated hipGraphs per site -- linear chains of kernel nodes -- and replays the one whose kernels
// match, after replacing the arguments of the nodes that differ from what that graph last ran with (the whole
// argument block is compared: nothing has to be declared "constant" by hand, so a stale pointer cannot slip through).
// Sites whose buffers alternate with the frame parity end up with two graphs, each a perfect match every other frame.
// Anything that goes wrong in the graph API falls back to launching the plan kernel by kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>
#include <tuple>
#include <vector>

namespace mmf {

struct PlannedLaunch {
    void* func = nullptr;
    dim3 grid, block;
    size_t blob_off = 0, blob_len = 0;  // this launch's arguments inside LaunchPlan::blob
    size_t arg_first = 0, arg_count = 0;  // its entries in LaunchPlan::arg_off
};

struct LaunchPlan {
    std::vector<PlannedLaunch> k;
    std::vector<unsigned char> blob;
    std::vector<size_t> arg_off;

    void clear() { k.clear(), blob.clear(), arg_off.clear(); }
    bool empty() const { return k.empty(); }

    template <typename T>
    void push_arg(const T& v) {
        const size_t al = alignof(T) > 8 ? alignof(T) : 8;  // every argument at least 8-byte aligned inside the blob
        const size_t off = (blob.size() + al - 1) / al * al;
        blob.resize(off + sizeof(T), 0);
        std::memcpy(blob.data() + off, &v, sizeof(T));
        arg_off.push_back(off);
    }

    template <typename... KArgs, typename... Args>
    void add(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args&&... args) {
        static_assert(sizeof...(KArgs) == sizeof...(Args), "launch: wrong number of kernel arguments");
        PlannedLaunch p;
        p.func = reinterpret_cast<void*>(kernel);
        p.grid = grid, p.block = block;
        p.blob_off = (blob.size() + 15) / 16 * 16;
        blob.resize(p.blob_off, 0);
        p.arg_first = arg_off.size();
        const std::tuple<KArgs...> converted(static_cast<KArgs>(args)...);  // the conversions a direct launch would do
        std::apply([&](const KArgs&... a) { (push_arg(a), ...); }, converted);
        p.arg_count = arg_off.size() - p.arg_first;
        p.blob_len = blob.size() - p.blob_off;
        k.push_back(p);
    }

    // kernelParams of launch i (pointers into `from`, which holds a copy of this plan's blob layout)
    void params(size_t i, unsigned char* from, std::vector<void*>& out) const {
        out.clear();
        for (size_t a = 0; a < k[i].arg_count; ++a) out.push_back(from + arg_off[k[i].arg_first + a]);
    }
};

inline bool same_dim(dim3 a, dim3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

class GraphCache {
  public:
    static constexpr int kEntries = 4;
    ~GraphCache() { release(); }

    void release() {
        for (Entry& e : e_) drop(e);
    }

    // enqueue `plan` on `stream`; returns a hipError_t (the kernel-by-kernel fallback's, if it came to that)
    hipError_t run(LaunchPlan& plan, hipStream_t stream) {
        if (plan.empty()) return hipSuccess;
        ++clock_;
        Entry* best = nullptr;
        size_t best_diff = ~size_t(0);
        for (Entry& e : e_) {
            if (!e.exec || !same_structure(e.plan, plan)) continue;
            const size_t d = differing(e.plan, plan);
            if (d < best_diff) best_diff = d, best = &e;
        }
        // a match that needs most of its nodes rewritten is a different configuration: give it a graph of its own
        // while there is room (the frame-parity case), else rewrite
        if (best && best_diff * 4 > plan.k.size() && has_free()) best = nullptr;
        if (!best) {
            best = victim();
            drop(*best);
            if (!build(*best, plan)) {
                drop(*best);
                return launch_direct(plan, stream);
            }
        } else if (!update(*best, plan)) {
            drop(*best);
            return launch_direct(plan, stream);
        }
        best->used = clock_;
        const hipError_t err = hipGraphLaunch(best->exec, stream);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            drop(*best);
            return launch_direct(plan, stream);
        }
        return hipSuccess;
    }

    static hipError_t launch_direct(LaunchPlan& plan, hipStream_t stream) {
        std::vector<void*> ptrs;
        for (size_t i = 0; i < plan.k.size(); ++i) {
            plan.params(i, plan.blob.data(), ptrs);
            const hipError_t err = hipLaunchKernel(plan.k[i].func, plan.k[i].grid, plan.k[i].block, ptrs.data(), 0, stream);
            if (err != hipSuccess) return err;
        }
        return hipSuccess;
    }

  private:
    struct Entry {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        std::vector<hipGraphNode_t> node;
        LaunchPlan plan;  // what the graph's nodes currently hold
        unsigned long long used = 0;
    };
    Entry e_[kEntries];
    unsigned long long clock_ = 0;

    static void drop(Entry& e) {
        if (e.exec) (void)hipGraphExecDestroy(e.exec);
        if (e.graph) (void)hipGraphDestroy(e.graph);
        e.exec = nullptr, e.graph = nullptr;
        e.node.clear(), e.plan.clear();
    }
    bool has_free() const {
        for (const Entry& e : e_)
            if (!e.exec) return true;
        return false;
    }
    Entry* victim() {
        Entry* v = &e_[0];
        for (Entry& e : e_) {
            if (!e.exec) return &e;
            if (e.used < v->used) v = &e;
        }
        return v;
    }
    static bool same_structure(const LaunchPlan& a, const LaunchPlan& b) {
        if (a.k.size() != b.k.size() || a.arg_off.size() != b.arg_off.size()) return false;
        for (size_t i = 0; i < a.k.size(); ++i)
            if (a.k[i].func != b.k[i].func || a.k[i].blob_len != b.k[i].blob_len || a.k[i].blob_off != b.k[i].blob_off) return false;
        return true;
    }
    static bool node_differs(const LaunchPlan& a, const LaunchPlan& b, size_t i) {
        return !same_dim(a.k[i].grid, b.k[i].grid) || !same_dim(a.k[i].block, b.k[i].block) ||
               std::memcmp(a.blob.data() + a.k[i].blob_off, b.blob.data() + b.k[i].blob_off, a.k[i].blob_len) != 0;
    }
    static size_t differing(const LaunchPlan& a, const LaunchPlan& b) {
        size_t d = 0;
        for (size_t i = 0; i < a.k.size(); ++i) d += node_differs(a, b, i) ? 1 : 0;
        return d;
    }
    static hipKernelNodeParams node_params(const LaunchPlan& plan, size_t i, std::vector<void*>& ptrs) {
        hipKernelNodeParams p;
        std::memset(&p, 0, sizeof(p));
        p.func = plan.k[i].func;
        p.gridDim = plan.k[i].grid, p.blockDim = plan.k[i].block;
        p.sharedMemBytes = 0;
        p.kernelParams = ptrs.data();
        p.extra = nullptr;
        return p;
    }
    static bool build(Entry& e, const LaunchPlan& plan) {
        e.plan = plan;
        if (hipGraphCreate(&e.graph, 0) != hipSuccess) return (void)hipGetLastError(), false;
        e.node.resize(plan.k.size());
        std::vector<void*> ptrs;
        for (size_t i = 0; i < plan.k.size(); ++i) {
            e.plan.params(i, e.plan.blob.data(), ptrs);
            const hipKernelNodeParams p = node_params(e.plan, i, ptrs);
            if (hipGraphAddKernelNode(&e.node[i], e.graph, i ? &e.node[i - 1] : nullptr, i ? 1 : 0, &p) != hipSuccess)
                return (void)hipGetLastError(), false;
        }
        if (hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0) != hipSuccess) return (void)hipGetLastError(), false;
        return true;
    }
    static bool update(Entry& e, const LaunchPlan& plan) {
        std::vector<void*> ptrs;
        for (size_t i = 0; i < plan.k.size(); ++i) {
            if (!node_differs(e.plan, plan, i)) continue;
            std::memcpy(e.plan.blob.data() + e.plan.k[i].blob_off, plan.blob.data() + plan.k[i].blob_off, plan.k[i].blob_len);
            e.plan.k[i].grid = plan.k[i].grid, e.plan.k[i].block = plan.k[i].block;
            e.plan.params(i, e.plan.blob.data(), ptrs);
            const hipKernelNodeParams p = node_params(e.plan, i, ptrs);
            if (hipGraphExecKernelNodeSetParams(e.exec, e.node[i], &p) != hipSuccess) return (void)hipGetLastError(), false;
        }
        return true;
    }
};

// The enqueue side of a call site: launches are recorded and go out together at flush() -- through the site's graph
// cache, or one by one when there is none (`cache == nullptr`: the measurement modes, which wrap single launches in
// events).  Order on the stream is the order of the calls either way.
struct Enqueuer {
    hipStream_t stream = nullptr;
    GraphCache* cache = nullptr;
    LaunchPlan plan;
    hipError_t err = hipSuccess;

    Enqueuer(hipStream_t s, GraphCache* c) : stream(s), cache(c) {}

    template <typename... KArgs, typename... Args>
    void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, Args&&... args) {
        if (cache) {
            plan.add(kernel, grid, block, std::forward<Args>(args)...);
        } else {
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, static_cast<KArgs>(args)...);
            const hipError_t e = hipGetLastError();
            if (err == hipSuccess) err = e;
        }
    }
    // everything recorded so far is on the stream when this returns
    hipError_t flush() {
        if (cache && !plan.empty()) {
            const hipError_t e = cache->run(plan, stream);
            if (err == hipSuccess) err = e;
            plan.clear();
        }
        return err;
    }
};

}  // namespace mmf
