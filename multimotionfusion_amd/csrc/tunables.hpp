// tunables.hpp -- every environment switch of the library, read ONCE (the first mmf_ctx_create) into one struct.
// None changes a result: they select between bit-identical ways of scheduling the same work (A/B aids, measured in
// LABNOTES.md) or size a launch.  Defaults are the shipped configuration.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mmf {

struct Tunables {
    // ---- Gauss-Newton chain (gn_fused.hpp, mmf_hip.hip: gn_geometry, odom_fused_chain_ok) ----
    bool gn_fused = true;     // MMF_GN_FUSED=0: always the two-launch chain (producer + step)
    int gn_fused_max = 8;     // MMF_GN_FUSED_MAX: models one one-launch chain carries at most (the residency check decides below that)
    int gn_px[3] = {0, 0, 0}; // MMF_GN_PX="p0,p1,p2": pixels per lane of a level (0 = by geometry)
    int gn_groups = 256;      // MMF_GN_GROUPS: workgroups per model per launch at most
    int gn_mixed_lanes = 192; // MMF_GN_MIXED_LANES: pixel lanes of a workgroup in a launch that carries object models (256: as the others)
    int gn_sleep = 1;         // MMF_GN_SLEEP: s_sleep(1) repetitions between two polls of the count barrier
    bool gn_obj_first = true; // MMF_GN_OBJ_FIRST=0: a shared launch dispatches the camera model's workgroups first
    int icp_variant = -1;     // MMF_ICP_VARIANT: shape of the stand-alone ICP kernel (-1 = by size)
    // ---- preparation jobs (prep_batch.hpp) ----
    int prep_merge = 2;       // MMF_PREP_MERGE=0|1|2: four / three / two model-side preparation stages
    bool prep_vn = true;      // MMF_PREP_VN=0: a level's vertex and normal maps as two jobs
    bool prep_l0_late = true; // MMF_PREP_L0_LATE=0: the model side's level-0 jobs in its first launch
    long prep_big = -1;       // MMF_PREP_BIG=<pixels>|0: from how many pixels a job's workgroups take four tiles each
    bool prep_planar = false; // MMF_PREP_PLANAR=1: also write the planar model maps and the AoS point cloud
    bool begin_rider = true;  // MMF_BEGIN_RIDER=0: odom_begin_kernel always as a launch of its own (else: on the last launch of a preparation enqueued ahead of the frame)
    bool prep_rect = true;    // MMF_PREP_RECT=0: an object model's model-side preparation covers the whole frame (else: the box its prediction is non-zero in)
    // ---- orchestrator (fusion_orchestrator.hpp) ----
    int early_image = -1;     // MMF_EARLY_IMAGE=start|chain|off: where the next frame's image side is enqueued (2 / 1 / 0; -1: by the number of models)
    bool fuse_index = true;   // MMF_FUSE_INDEX=0: fuse's update pass and the index map's projection as two launches
    bool host_up_events = false;  // MMF_HOST_UP_EVENTS=1: host frames: always order a ring slot's upload by events
    bool host_trace = false;  // MMF_HOST_TRACE=1: the calling thread's timeline (stderr, every 100 calls)
    // ---- surfel passes ----
    int track_cull = 1;       // MMF_TRACK_CULL=0: object models are tracked like the camera model (whole image, full grids)
    int spec_prep_all = 2;    // MMF_SPEC_PREP_ALL=<n>: from n models per GPU on the next frame's model-side preparation of all models is enqueued at the end of the frame (0: never)
    int pass_batch = -1;      // MMF_PASS_BATCH=0|1|2: the object models' projection / fuse / clean / predict passes model by model on the models' own
                              // streams / as one launch per pass covering the whole frame / as one launch per pass restricted to where the models
                              // are (pass_rect.hpp).  All bit-identical.  -1 (default): 2 from four object models on a GPU, else 0 -- the restricted
                              // launches do 9 % less GPU work with a fifth of the launches, which pays once the calling thread's ~11 launches per
                              // model are what the GPU waits for (LABNOTES r5)
    bool xcd_blocks = true;   // MMF_XCD=0: the blocks of a surfel pass are dealt to the XCDs round-robin, as the workgroups are (else: a contiguous eighth per XCD)
    int splat_wgs = 0;        // MMF_SPLAT_WGS: workgroups of a splat launch (0 = by the store's size)
    int splat_bound = -1;     // MMF_SPLAT_BOUND=0|1: the bounded depth test never / always (-1 = by the store's size)
};

inline const Tunables& tunables() {
    static const Tunables t = []() {
        Tunables v;
        auto flag = [](const char* name, bool dflt) {
            const char* e = std::getenv(name);
            return e ? (e[0] != '0' && e[0] != '\0') : dflt;
        };
        auto num = [](const char* name, long dflt) {
            const char* e = std::getenv(name);
            return e ? std::atol(e) : dflt;
        };
        v.gn_fused = flag("MMF_GN_FUSED", true);
        v.gn_fused_max = (int)num("MMF_GN_FUSED_MAX", 8);
        if (v.gn_fused_max < 1) v.gn_fused_max = 1;
        if (const char* e = std::getenv("MMF_GN_PX")) std::sscanf(e, "%d,%d,%d", &v.gn_px[0], &v.gn_px[1], &v.gn_px[2]);
        v.gn_groups = (int)num("MMF_GN_GROUPS", 256);
        v.gn_mixed_lanes = (int)num("MMF_GN_MIXED_LANES", 192);
        v.gn_sleep = (int)num("MMF_GN_SLEEP", 1);
        v.gn_obj_first = flag("MMF_GN_OBJ_FIRST", true);
        v.icp_variant = (int)num("MMF_ICP_VARIANT", -1);
        v.prep_merge = (int)num("MMF_PREP_MERGE", 2);
        v.prep_vn = flag("MMF_PREP_VN", true);
        v.prep_l0_late = flag("MMF_PREP_L0_LATE", true);
        v.prep_big = num("MMF_PREP_BIG", -1);
        v.prep_planar = flag("MMF_PREP_PLANAR", false);
        v.prep_rect = flag("MMF_PREP_RECT", true);
        v.begin_rider = flag("MMF_BEGIN_RIDER", true);
        if (const char* e = std::getenv("MMF_EARLY_IMAGE")) v.early_image = std::strcmp(e, "off") == 0 ? 0 : (std::strcmp(e, "chain") == 0 ? 1 : 2);
        v.fuse_index = flag("MMF_FUSE_INDEX", true);
        v.host_up_events = flag("MMF_HOST_UP_EVENTS", false);
        v.host_trace = flag("MMF_HOST_TRACE", false);
        v.track_cull = (int)num("MMF_TRACK_CULL", 1);
        v.spec_prep_all = (int)num("MMF_SPEC_PREP_ALL", 2);
        v.pass_batch = (int)num("MMF_PASS_BATCH", -1);
        v.xcd_blocks = flag("MMF_XCD", true);
        v.splat_wgs = (int)num("MMF_SPLAT_WGS", 0);
        if (std::getenv("MMF_SPLAT_BOUND")) v.splat_bound = num("MMF_SPLAT_BOUND", 0) ? 1 : 0;
        return v;
    }();
    return t;
}

}  // namespace mmf
