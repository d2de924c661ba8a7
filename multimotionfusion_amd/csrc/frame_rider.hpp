// frame_rider.hpp -- tracker work that nothing on the model's stream waits for, carried by one extra workgroup of the
// frame's first projection pass.
//
// The chain's last launch used to end with (a) the copy of the odometry state into the host's pinned struct + the
// sequence number the host spins on and (b) Model::computeFusionWeight of the new pose for the early fuse pass
// (Model.cpp:876-891): one wave each, 3.5 us and 3.8 us behind the 5.6 us of the solve (rocprofv3: 13.0 us for the
// launch, 9.2 without the weight), and the frame's first projection waited for both.  That projection -- the first
// predictIndices (MultiMotionFusion.cpp:792): index_map_kernel, then index_resolve_kernel -- reads neither: each of its two
// launches gets one more workgroup, dispatched first: the hand-over on the first, the weight on the second.
// (The resolve of a prediction used to be a carrier too; the weight's local arrays gave that whole kernel a scratch frame
// and 82 registers instead of 37.)
// The first reader of the weight is fuse_data_kernel, two launches later; the host has the pose ~2 us EARLIER than from
// the end of the 13 us launch.
#pragma once
#include "odom_state.hpp"
#include "pose_algebra.hpp"

namespace mmf {

struct FrameRider {
    OdomState* st = nullptr;    // nullptr: nothing rides
    OdomState* host = nullptr;  // pinned, device visible: the copy the host polls
    unsigned seq = 0;
    unsigned what = 3;  // bit 0: the hand-over to the host, bit 1: the fusion weight (the projection's two launches carry one each)
};

// Model::computeFusionWeight (Model.cpp:876-891) of the tracked pose against lastPose = the pose the chain started from
// (Model.cpp:412) with weightMultiplier 1 (the multiplier is the last factor: the fuse pass applies it).  One lane.
__device__ __forceinline__ void odom_fusion_weight(OdomState* st) {
    float last[16];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) last[r * 4 + c] = st->Rprev[r * 3 + c];
        last[r * 4 + 3] = st->tprev[r];
    }
    last[12] = last[13] = last[14] = 0.f, last[15] = 1.f;
    float inv[16];
    for (int k = 0; k < 16; ++k) inv[k] = st->pose_inv[k];
    st->fusion_weight = mmf::host::compute_fusion_weight(inv, last, 1.0f);
}

// One whole wave (lane = 0 .. 63): the state words up to publish_seq into the host's struct, a system-scope fence executed
// by the wave as a whole (every lane's stores are out before lane 0 publishes), then the sequence number.
__device__ __forceinline__ void odom_publish_wave(const OdomState* st, OdomState* host, unsigned seq, unsigned lane) {
    const unsigned* src = reinterpret_cast<const unsigned*>(st);
    unsigned* dst = reinterpret_cast<unsigned*>(host);
    constexpr unsigned kWords = offsetof(OdomState, publish_seq) / 4;
    constexpr unsigned kRounds = (kWords + 63) / 64;
    // all of a lane's words are loaded before any is stored: as a load-store loop this was kRounds dependent round trips
    unsigned v[kRounds];
#pragma unroll
    for (unsigned r = 0; r < kRounds; ++r)
        v[r] = __hip_atomic_load(src + min(lane + 64 * r, kWords - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (unsigned r = 0; r < kRounds; ++r)
        if (lane + 64 * r < kWords) dst[lane + 64 * r] = v[r];
    __threadfence_system();
    if (lane == 0) __hip_atomic_store(&host->publish_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the riding workgroup (>= 128 threads): wave 0 publishes, lane 0 of wave 1 evaluates the weight
// PARTS: what this carrier's code can do at all (bit 0 hand-over, bit 1 weight) -- the weight's SVD is 60 registers that a
// carrier which never computes it need not reserve for every one of its waves
template <unsigned PARTS = 3u>
__device__ __forceinline__ void frame_rider_run(const FrameRider& r) {
    if (threadIdx.x < 64) {
        if ((PARTS & 1u) && (r.what & 1u)) odom_publish_wave(r.st, r.host, r.seq, threadIdx.x);
    } else if (threadIdx.x == 64) {
        if ((PARTS & 2u) && (r.what & 2u)) odom_fusion_weight(r.st);
    }
}

}  // namespace mmf
