// fusion_orchestrator.hpp -- MultiMotionFusion::processFrame / predict (Core/MultiMotionFusion.cpp:207-854,
// 863-875) for a list of rigid-body models: the global (camera) model plus the object models the segmentation
// spawns.  Textually included at the end of mmf_hip.hip (it uses that file's static helpers).
//
// What stays with the caller: the segmentation itself (gSLICr + dense CRF, or a ground-truth id image) -- it is
// handed in per frame (mmf_segmentation) or pulled through a callback at the point where the reference calls
// performSegmentation (:412); relocalisation, ferns, deformation (closeLoops / reloc off).
//
// MI355X mapping: every model owns a LANE (a child context: its own stream + reduction scratch), so the
// latency-bound Gauss-Newton chains of different models (19 x two small launches each) overlap on the device;
// the chains of all models are enqueued before the first result is awaited.  What depends on the sensor frame
// only (bilateral filter, depth / intensity pyramids, vertex / normal maps, gradients) is computed ONCE on the
// fusion's own stream and aliased by every model's odometry -- the reference shares just the depth pyramid
// (Model::GPUSetup::depth_tmp, Model.h:103) and recomputes the rest per model.
#pragma once

#include <chrono>
#include <vector>

struct PoseLogItem {  // Model::PoseLogItem (Model.h:326-329)
    long long ts;
    float p[7];  // x y z qx qy qz qw
};

struct FusionModel {  // one ModelPointer of the reference's `models` list
    mmf_ctx* lane = nullptr;  // stream + reduction scratch (models[0]: the fusion's own context)
    bool own_lane = false;
    mmf_model* model = nullptr;
    mmf_odom* odom = nullptr;  // Model::frameToModel
    float last_pose[16];       // Model::lastPose
    int fill_in = 0;           // Model::allowsFillIn()
    long long unseen = 0;      // Model::unseenCount
    float* icp_error = nullptr;  // Model::icpError / rgbError (R32F, enableErrorRecording)
    float* rgb_error = nullptr;
    hipEvent_t ev_done = nullptr;
    bool tracking = false;  // a tracking call is in flight on the lane
    bool early_done = false;  // this frame's predict() + first predictIndices went out before the pose reached the host
    bool early_fused = false;  // ... and so did its fuse / predictIndices / clean
    // The model side of the tracker's preparation (prediction -> model pyramids in the global frame, point clouds,
    // intensity pyramid: Model::initICP's initICPModel / initRGBModel, Model.cpp:396-401) needs nothing of the next
    // sensor frame.  When this process runs ONE model it is enqueued at the END of a frame, behind the final predict(),
    // where the model's stream would otherwise idle until the side streams have finished the next frame's sensor side;
    // the next processFrame uses it if nothing it was computed from has changed since (pose, images, mode), else
    // prepares as before.
    bool spec_valid = false, spec_hit = false;
    float spec_pose[16];
    unsigned long long spec_tex_gen = 0;
    int spec_f2f = 0;
    std::vector<PoseLogItem> pose_log;
};

struct mmf_fusion {
    mmf_ctx* ctx = nullptr;
    mmf_fusion_config cfg;
    int width = 0, height = 0;
    float cx = 0, cy = 0, fx = 0, fy = 0;
    std::vector<FusionModel*> models;        // active; models[0] = globalModel
    std::vector<FusionModel*> preallocated;  // preallocatedModels (MultiMotionFusion.h:349)
    std::vector<FusionModel*> inactive;      // inactiveModels
    std::vector<int> scheduled_deactivation;
    int next_id = 0;                  // nextID (getNextModelID)
    // per-rigid-body shard: this process owns the models whose id has id % shard_world == shard_rank
    // (fusion_owner_of) and runs their track / predict / fuse / clean; the others only exist as bookkeeping here (ids,
    // thresholds, poses handed in through mmf_fusion_set_model_pose) -- their owners run on other GPUs
    int shard_rank = 0, shard_world = 1;
    double t_tracking_s = 0, t_frame_s = 0;  // host wall clock of the last processFrame: tracking phase, whole call
    double trace_us[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // MMF_HOST_TRACE
    long trace_calls = 0;
    float* depth_filtered = nullptr;  // = filtered[cur]
    uint8_t* mask = nullptr;          // textures[MASK]: all zeros unless enableMultipleModels
    bool mask_is_zero = false;
    int tick = 1;                     // MultiMotionFusion.cpp:36
    unsigned extent_seq = 0;          // number of the frame whose model-side preparation notes extents (extent.hpp): only ever grows
    unsigned long long* mask_boxes = nullptr;  // pass_rect.hpp: the boxes of the ids of the frame's id image (device, 256 x 4 words)
    unsigned mask_gen = 0;
    int tracking_ok = 1;
    const uint8_t* frame_rgb = nullptr;  // this frame's inputs (device), kept for predict()
    const float* frame_depth = nullptr;
    mmf_segmentation_fn seg_fn = nullptr;
    void* seg_user = nullptr;
    hipEvent_t ev_frame_ready = nullptr;  // fusion stream: the frame's shared inputs are complete
    // next-frame prefetch (mmf_fusion_prefetch_frame): the filter and the input-side preparation of frame t+1 run
    // on `side` while frame t is fused on the context's stream.  Two filtered-depth buffers: frame t's fuse /
    // clean / fill-in read one while frame t+1's filter writes the other.
    float* filtered[2] = {nullptr, nullptr};
    int cur = 0;
    // The next frame's SO3 pre-alignment runs in one of two states of its own (by the parity of the frame it is for): no
    // chain reads or writes them, so it can run while the current frame's chain is still at work; the chain's first launch
    // copies the result into every tracked model's state (BeginArgs::so3_stage).
    OdomState* so3_stage[2] = {nullptr, nullptr};
    int so3_stage_ready = -1;               // which of the two holds the pre-alignment of the upcoming frame; -1: none
    const uint8_t* image_pre_rgb = nullptr;  // the image side of this frame (intensity pyramid, gradients, SO3) is enqueued already
    hipStream_t side = nullptr;   // depth chain: filter, depth pyramid, vertex / normal maps
    hipStream_t side2 = nullptr;  // image chain: intensity pyramid, gradients, SO3 pre-alignment
    float* side_partials = nullptr;   // reduction scratch of the SO3 launches on side2 (never the context's: the
    unsigned* side_ticket = nullptr;  // main stream may be inside a reduction of its own at the same time)
    hipEvent_t ev_prefetch2_done = nullptr;
    hipEvent_t ev_inputs_free = nullptr;    // fusion stream: enqueued work no longer reads the odometry's input-side
                                            // buffers nor filtered[1 - cur]
    hipEvent_t ev_prefetch_done = nullptr;  // side stream: the prefetch has been enqueued up to here
    bool inputs_free_recorded = false;
    bool pre_valid = false;
    const uint8_t* pre_rgb = nullptr;
    const float* pre_depth = nullptr;
    // host FrameData hand-over (mmf_fusion_process_frame_host[_next]): pinned staging + device copies in a ring of three
    // (the frame being processed, the next one being uploaded, one whose readers may still be running), uploads on a
    // stream of their own
    static constexpr int kUp = 3;
    uint8_t* up_pin[kUp] = {nullptr, nullptr, nullptr};
    uint8_t* up_dev[kUp] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_up[kUp] = {nullptr, nullptr, nullptr};      // upload stream: slot k is on the device
    hipEvent_t ev_up_rgb[kUp] = {nullptr, nullptr, nullptr};  // ... its colour image is (sent first: the image side needs only that)
    bool up_recorded[kUp] = {false, false, false};
    hipStream_t up_stream = nullptr;
    hipEvent_t ev_up_begin = nullptr;  // fusion stream -> upload stream
    int up_cur = 0;
    // the last call waited on the host for a pose that came off the fusion's stream: everything enqueued before that call is
    // done, a slot of the ring can be refilled without an event from the fusion's stream (a marker on the stream a frame waits for)
    bool host_caught_up = false;
    // the NEXT call's host frame, handed in as a hint: staged and uploaded while the host would only wait for this
    // frame's pose (fusion_stage_host_next), its sensor-side preparation enqueued like a device-side hint's
    struct HostNext {
        const uint8_t* rgb = nullptr;
        const float* depth = nullptr;
        int slot = -1;
        bool pending = false;     // handed to the staging thread, not yet joined
        bool rgb_staged = false;  // its colour image is on its way (ev_up_rgb[slot] recorded)
        bool staged = false;      // all of it is on its way to (or on) the device in `slot`
    } host_next;
    // The copy of an announced frame into its pinned slot (2.15 MB at 640x480: ~150 us of one core) and the enqueue of its
    // upload run on a thread of their own, started when the call that announces the frame begins: on the calling thread the
    // copy was longer than the wait for the pose it was meant to hide in (announced frames ran 7 % behind device frames).
    struct Stager {
        std::thread thread;
        std::mutex mu;
        std::condition_variable cv;
        bool stop = false, busy = false;
        bool rgb_done = false;  // the job's colour image has been copied and its upload enqueued
        bool after_begin = true;  // the upload waits for ev_up_begin
        int slot = 0;
        const uint8_t* rgb = nullptr;
        const float* depth = nullptr;
        int rc = MMF_OK;
        std::string error;
    } stager;
};

static void identity16(float* m) {
    for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}

extern "C" int mmf_fusion_default_config(mmf_fusion_config* cfg) {
    MMF_REQUIRE(cfg != nullptr, "mmf_fusion_default_config: null argument");
    cfg->time_delta = 200;          // GUI/MainController.cpp:333 (timeDelta flag default)
    cfg->conf_global_init = 10.0f;  // GUI default confGlobalInit
    cfg->icp_weight = 10.0f;        // GUI default icpWeight
    cfg->depth_cutoff = 15.0f;      // GUI default depthCutoff (bilateral filter maxD)
    cfg->max_depth_processed = 20.0f;  // MultiMotionFusion.cpp:53
    cfg->rgb_only = 0;
    cfg->pyramid = 1;
    cfg->fast_odom = 0;
    cfg->so3 = 1;
    cfg->frame_to_frame_rgb = 0;
    cfg->outlier_coeff = 3.0f;      // GPUSetup::outlierCoefficient GUI default
    cfg->fill_in = 1;               // the global model is created with fill-in enabled
    cfg->max_surfels = 0;
    cfg->conf_object_init = 0.01f;  // GUI/MainController.cpp:333 (-confO)
    cfg->enable_multiple_models = 0;
    cfg->preallocated_models = 0;   // GUI/MainController.cpp:339 (-a)
    cfg->error_recording = 1;       // every Model is created with enableErrorRecording (MultiMotionFusion.cpp:70,128,944)
    cfg->pose_logging = 0;          // enablePoseLogging
    cfg->max_object_surfels = 0;
    cfg->batch_tracking = 1;
    return MMF_OK;
}

static void fusion_model_destroy(FusionModel* fm) {
    if (!fm) return;
    mmf_model_destroy(fm->model);
    mmf_odom_destroy(fm->odom);
    if (fm->ev_done) (void)hipEventDestroy(fm->ev_done);
    if (fm->own_lane) mmf_ctx_destroy(fm->lane);
    delete fm;
}

// std::make_shared<Model>(id, confidence, odom_cfg, enableFillIn, enableErrorRecording, ...) (Model.cpp:147-262)
static int fusion_model_create(mmf_fusion* f, int id, float conf, int fill_in, bool own_lane, FusionModel** out) {
    FusionModel* fm = new (std::nothrow) FusionModel();
    MMF_REQUIRE(fm != nullptr, "mmf_fusion: out of host memory");
    int rc = MMF_OK;
    // a rigid body another rank owns (ownership is by id: fusion_owner_of) is bookkeeping here: id, thresholds, pose,
    // statistics -- no surfel store, no odometry slab, no stream.  (The global model's odometry holds the sensor side of
    // every rank: it is always the real thing.)
    if (f->shard_world > 1 && id != 0 && id % f->shard_world != f->shard_rank) {
        fm->lane = f->ctx;
        rc = model_create_bookkeeping(f->ctx, f->width, f->height, f->cx, f->cy, f->fx, f->fy, (unsigned char)id, conf, &fm->model);
        if (rc == MMF_OK) rc = odom_create_bookkeeping(f->ctx, f->width, f->height, f->cx, f->cy, f->fx, f->fy, &fm->odom);
        if (rc != MMF_OK) {
            const std::string keep = g_last_error;
            fusion_model_destroy(fm);
            return fail(rc, keep);
        }
        fm->fill_in = fill_in;
        identity16(fm->last_pose);
        if (f->cfg.pose_logging) fm->pose_log.reserve(1000);  // Model.cpp:177
        *out = fm;
        return MMF_OK;
    }
    if (own_lane) {
        rc = mmf_ctx_create(f->ctx->device, nullptr, 1, &fm->lane);
        fm->own_lane = rc == MMF_OK;
    } else {
        fm->lane = f->ctx;
    }
    const int cap = id == 0 ? f->cfg.max_surfels : (f->cfg.max_object_surfels ? f->cfg.max_object_surfels : f->cfg.max_surfels);
    if (rc == MMF_OK)
        rc = mmf_model_create(fm->lane, f->width, f->height, f->cx, f->cy, f->fx, f->fy, (unsigned char)id, conf, cap, &fm->model);
    if (rc == MMF_OK)
        rc = mmf_odom_create(fm->lane, f->width, f->height, f->cx, f->cy, f->fx, f->fy, 0.10f,
                             std::sin(20.f * 3.14159254f / 180.f), &fm->odom);
    if (rc == MMF_OK && f->cfg.error_recording)  // Model::icpError / rgbError: images inside the odometry's slab
        fm->icp_error = fm->odom->icp_err, fm->rgb_error = fm->odom->rgb_err;
    if (rc == MMF_OK && hipEventCreateWithFlags(&fm->ev_done, hipEventDisableTiming) != hipSuccess)
        rc = fail(MMF_ERR_HIP, "mmf_fusion: hipEventCreate failed");
    if (rc != MMF_OK) {
        const std::string keep = g_last_error;
        fusion_model_destroy(fm);
        return fail(rc, keep);
    }
    // the prediction images (model slab) and the filtered depth (the fusion's buffer) are not written
    // between the init* calls of a frame and the end of its tracking
    fm->odom->alias_inputs = true;
    fm->fill_in = fill_in;
    identity16(fm->last_pose);
    if (f->cfg.pose_logging) fm->pose_log.reserve(1000);  // Model.cpp:177
    *out = fm;
    return MMF_OK;
}

// MultiMotionFusion::getNextModelID (MultiMotionFusion.cpp:983-999)
static int fusion_next_model_id(mmf_fusion* f, bool assign) {
    const int next = f->next_id;
    if (assign) {
        while (true) {
            f->next_id = (f->next_id + 1) & 255;
            bool occupied = false;
            for (FusionModel* m : f->models)
                if (f->next_id == (int)m->model->id) occupied = true;
            if (!occupied) break;
        }
    }
    return next;
}

extern "C" int mmf_fusion_create(mmf_ctx* c, int width, int height, float cx, float cy, float fx, float fy,
                                 const mmf_fusion_config* cfg, mmf_fusion** out) {
    MMF_REQUIRE(c && out, "mmf_fusion_create: null argument");
    MMF_HIP_TRY(hipSetDevice(c->device));
    mmf_fusion* f = new (std::nothrow) mmf_fusion();
    MMF_REQUIRE(f != nullptr, "mmf_fusion_create: out of host memory");
    f->ctx = c;
    if (cfg)
        f->cfg = *cfg;
    else
        mmf_fusion_default_config(&f->cfg);
    f->width = width, f->height = height;
    f->cx = cx, f->cy = cy, f->fx = fx, f->fy = fy;
    FusionModel* global = nullptr;
    // globalModel (MultiMotionFusion.cpp:69-71): id 0, fill-in, on the fusion's own stream
    int rc = fusion_model_create(f, fusion_next_model_id(f, true), f->cfg.conf_global_init, f->cfg.fill_in, false, &global);
    if (rc != MMF_OK) {
        delete f;
        return rc;
    }
    f->models.push_back(global);
    const size_t npix = (size_t)width * height;
    // from here on every failure leaves through mmf_fusion_destroy (it tolerates members that were never created)
    auto device_side = [&]() -> int {
        MMF_HIP_TRY(hipMalloc(&f->filtered[0], npix * 4));
        MMF_HIP_TRY(hipMalloc(&f->filtered[1], npix * 4));
        f->depth_filtered = f->filtered[0];
        // the side streams and events of the prefetch are created by its first call
        MMF_HIP_TRY(hipEventCreateWithFlags(&f->ev_inputs_free, hipEventDisableTiming));
        MMF_HIP_TRY(hipEventCreateWithFlags(&f->ev_frame_ready, hipEventDisableTiming));
        MMF_HIP_TRY(hipMalloc(&f->mask, npix));
        MMF_HIP_TRY(hipMemsetAsync(f->mask, 0, npix, c->stream));
        f->mask_is_zero = true;
        return MMF_OK;
    };
    rc = device_side();
    if (rc == MMF_OK)
        rc = mmf_fusion_preallocate_models(f, (unsigned)(f->cfg.preallocated_models > 0 ? f->cfg.preallocated_models : 0));
    if (rc != MMF_OK) {
        const std::string keep = g_last_error;
        mmf_fusion_destroy(f);
        return fail(rc, keep);
    }
    *out = f;
    return MMF_OK;
}

// preallocateModels(count) (MultiMotionFusion.cpp:125-131): object models created ahead of their first use
extern "C" int mmf_fusion_preallocate_models(mmf_fusion* f, unsigned count) {
    MMF_REQUIRE(f != nullptr, "mmf_fusion_preallocate_models: null fusion object");
    MMF_HIP_TRY(hipSetDevice(f->ctx->device));
    for (unsigned i = 0; i < count; ++i) {
        FusionModel* fm = nullptr;
        int rc = fusion_model_create(f, fusion_next_model_id(f, true), f->cfg.conf_object_init, 0, true, &fm);
        if (rc) return rc;
        f->preallocated.push_back(fm);
    }
    return MMF_OK;
}

extern "C" void mmf_fusion_destroy(mmf_fusion* f) {
    if (!f) return;
    (void)hipSetDevice(f->ctx->device);
    (void)hipStreamSynchronize(f->ctx->stream);
    if (f->side) (void)hipStreamSynchronize(f->side);
    if (f->side2) (void)hipStreamSynchronize(f->side2);
    for (auto* list : {&f->models, &f->preallocated, &f->inactive}) {
        for (FusionModel* fm : *list) fusion_model_destroy(fm);
        list->clear();
    }
    (void)hipFree(f->filtered[0]);
    (void)hipFree(f->filtered[1]);
    (void)hipFree(f->mask);
    (void)hipFree(f->mask_boxes);
    (void)hipFree(f->side_partials);
    (void)hipFree(f->side_ticket);
    if (f->stager.thread.joinable()) {
        {
            std::lock_guard<std::mutex> lock(f->stager.mu);
            f->stager.stop = true;
        }
        f->stager.cv.notify_all();
        f->stager.thread.join();
    }
    if (f->up_stream) (void)hipStreamSynchronize(f->up_stream);
    for (int i = 0; i < mmf_fusion::kUp; ++i) {
        if (f->up_pin[i]) (void)hipHostFree(f->up_pin[i]);
        (void)hipFree(f->up_dev[i]);
        if (f->ev_up[i]) (void)hipEventDestroy(f->ev_up[i]);
        if (f->ev_up_rgb[i]) (void)hipEventDestroy(f->ev_up_rgb[i]);
    }
    if (f->up_stream) (void)hipStreamDestroy(f->up_stream);
    if (f->ev_up_begin) (void)hipEventDestroy(f->ev_up_begin);
    (void)hipFree(f->so3_stage[0]);
    if (f->ev_inputs_free) (void)hipEventDestroy(f->ev_inputs_free);
    if (f->ev_frame_ready) (void)hipEventDestroy(f->ev_frame_ready);
    if (f->ev_prefetch_done) (void)hipEventDestroy(f->ev_prefetch_done);
    if (f->side) (void)hipStreamDestroy(f->side);
    if (f->ev_prefetch2_done) (void)hipEventDestroy(f->ev_prefetch2_done);
    if (f->side2) (void)hipStreamDestroy(f->side2);
    delete f;
}

extern "C" mmf_model* mmf_fusion_model(mmf_fusion* f) { return f ? f->models[0]->model : nullptr; }
extern "C" mmf_odom* mmf_fusion_odometry(mmf_fusion* f) { return f ? f->models[0]->odom : nullptr; }
extern "C" int mmf_fusion_tick(mmf_fusion* f) { return f ? f->tick : -1; }
extern "C" const float* mmf_fusion_depth_filtered(mmf_fusion* f) { return f ? f->depth_filtered : nullptr; }

// getModels() (MultiMotionFusion.h:107): the active list in its order, index 0 = the global model
extern "C" int mmf_fusion_num_models(mmf_fusion* f) { return f ? (int)f->models.size() : -1; }
extern "C" mmf_model* mmf_fusion_model_at(mmf_fusion* f, int index) {
    return (f && index >= 0 && index < (int)f->models.size()) ? f->models[index]->model : nullptr;
}
extern "C" mmf_odom* mmf_fusion_odometry_at(mmf_fusion* f, int index) {
    return (f && index >= 0 && index < (int)f->models.size()) ? f->models[index]->odom : nullptr;
}
extern "C" int mmf_fusion_num_inactive_models(mmf_fusion* f) { return f ? (int)f->inactive.size() : -1; }
extern "C" mmf_model* mmf_fusion_inactive_model_at(mmf_fusion* f, int index) {
    return (f && index >= 0 && index < (int)f->inactive.size()) ? f->inactive[index]->model : nullptr;
}
// The id the next spawned model will carry = the label a new segment must have in the id image.  The reference
// hands getNextModelID() to the segmentation (:148) but spawns the FRONT of preallocatedModels when there is one
// (:940-942), whose id was assigned at preallocation: with `-a N` its new segment is labelled with an id no model
// has.  Here the label and the spawned model always agree.
extern "C" int mmf_fusion_next_model_id(mmf_fusion* f) {
    if (!f) return -1;
    return f->preallocated.empty() ? f->next_id : (int)f->preallocated.front()->model->id;
}

// A model belongs to the rank its ID selects (id % world; the global model, id 0, to rank 0): fixed when the model is
// created and the same on every rank, whatever happens to the list around it -- a model that leaves the list (lost segment,
// scheduled deactivation) does not hand the models behind it to other ranks, which a rule by list position would.
static inline int fusion_owner_of(const mmf_fusion* f, const FusionModel* fm) { return (int)((unsigned)fm->model->id % (unsigned)f->shard_world); }
static inline bool fusion_owns_model(const mmf_fusion* f, const FusionModel* fm) { return fusion_owner_of(f, fm) == f->shard_rank; }
static inline bool fusion_owns(const mmf_fusion* f, size_t index) { return index < f->models.size() && fusion_owns_model(f, f->models[index]); }

extern "C" int mmf_fusion_set_shard(mmf_fusion* f, int rank, int world) {
    MMF_REQUIRE(f && world >= 1 && rank >= 0 && rank < world, "mmf_fusion_set_shard: bad argument");
    MMF_REQUIRE(f->models.size() == 1 && f->inactive.empty(), "mmf_fusion_set_shard: call it before the first object model is spawned");
    f->shard_rank = rank, f->shard_world = world;
    // models created ahead of their use (preallocateModels): the ones this rank will not own shrink to bookkeeping
    for (FusionModel*& fm : f->preallocated) {
        const int id = (int)fm->model->id;
        const bool is_light = fm->model->slab == nullptr, want_light = world > 1 && id % world != rank;
        if (is_light == want_light) continue;
        FusionModel* again = nullptr;
        const float conf = fm->model->conf_threshold;
        const int fill_in = fm->fill_in;
        fusion_model_destroy(fm);
        fm = nullptr;
        int rc = fusion_model_create(f, id, conf, fill_in, true, &again);
        if (rc) return rc;
        fm = again;
    }
    return MMF_OK;
}
extern "C" int mmf_fusion_owns_model(mmf_fusion* f, int index) {
    return (f && index >= 0 && fusion_owns(f, (size_t)index)) ? 1 : 0;
}
// the pose of a model another rank owns, as gathered from its owner (Model::overridePose semantics)
extern "C" int mmf_fusion_set_model_pose(mmf_fusion* f, int index, const float pose[16]) {
    MMF_REQUIRE(f && pose && index >= 0 && index < (int)f->models.size(), "mmf_fusion_set_model_pose: bad argument");
    std::memcpy(f->models[index]->model->pose, pose, sizeof(float) * 16);
    std::memcpy(f->models[index]->last_pose, pose, sizeof(float) * 16);
    return MMF_OK;
}
extern "C" int mmf_fusion_last_timings(mmf_fusion* f, double* tracking_s, double* frame_s) {
    MMF_REQUIRE(f != nullptr, "mmf_fusion_last_timings: null fusion object");
    if (tracking_s) *tracking_s = f->t_tracking_s;
    if (frame_s) *frame_s = f->t_frame_s;
    return MMF_OK;
}

static FusionModel* fusion_find(mmf_fusion* f, int id) {
    for (FusionModel* m : f->models)
        if ((int)m->model->id == id) return m;
    return nullptr;
}

// Model::getICPErrorTexture / getRGBErrorTexture (Model.h:232-236): R32F images written by the last level-0
// iteration of the model's tracking (reduce.cu:275,299; RGBDOdometry.cpp:369,408)
extern "C" int mmf_fusion_error_texture(mmf_fusion* f, int index, int which, float** dev_ptr) {
    MMF_REQUIRE(f && dev_ptr && index >= 0 && index < (int)f->models.size(), "mmf_fusion_error_texture: bad argument");
    *dev_ptr = which == 0 ? f->models[index]->icp_error : f->models[index]->rgb_error;
    MMF_REQUIRE(*dev_ptr != nullptr, "mmf_fusion_error_texture: error recording is off");
    return MMF_OK;
}

// getTextures() (MultiMotionFusion.h:124): the raw input images of the current frame as device images
extern "C" int mmf_fusion_texture(mmf_fusion* f, const char* name, const void** dev_ptr, size_t* bytes) {
    MMF_REQUIRE(f && name && dev_ptr && bytes, "mmf_fusion_texture: null argument");
    const size_t npix = (size_t)f->width * f->height;
    const std::string s(name);
    if (s == "RGB") *dev_ptr = f->frame_rgb, *bytes = npix * 3;                      // GPUTexture::RGB
    else if (s == "DEPTH_METRIC") *dev_ptr = f->frame_depth, *bytes = npix * 4;      // GPUTexture::DEPTH_METRIC
    else if (s == "DEPTH_METRIC_FILTERED") *dev_ptr = f->depth_filtered, *bytes = npix * 4;
    else if (s == "MASK") *dev_ptr = f->mask, *bytes = npix;
    else return fail(MMF_ERR_INVALID, "mmf_fusion_texture: unknown name '" + s + "'");
    return MMF_OK;
}

// ---- runtime setters the front end pushes every GUI tick (MultiMotionFusion.cpp:1064-1116, GUI/MainController.cpp:641-670)
#define MMF_FUSION_SETTER(name, field, type)                              \
    extern "C" int mmf_fusion_set_##name(mmf_fusion* f, type val) {       \
        MMF_REQUIRE(f != nullptr, "mmf_fusion_set_" #name ": null fusion object"); \
        f->cfg.field = val;                                               \
        return MMF_OK;                                                    \
    }
MMF_FUSION_SETTER(rgb_only, rgb_only, int)
MMF_FUSION_SETTER(icp_weight, icp_weight, float)
MMF_FUSION_SETTER(outlier_coefficient, outlier_coeff, float)
MMF_FUSION_SETTER(pyramid, pyramid, int)
MMF_FUSION_SETTER(fast_odom, fast_odom, int)
MMF_FUSION_SETTER(so3, so3, int)
MMF_FUSION_SETTER(frame_to_frame_rgb, frame_to_frame_rgb, int)
MMF_FUSION_SETTER(depth_cutoff, depth_cutoff, float)
MMF_FUSION_SETTER(enable_multiple_models, enable_multiple_models, int)
#undef MMF_FUSION_SETTER
// setConfidenceThreshold is declared by the reference (MultiMotionFusion.h:168) but has no definition: the
// threshold lives in each Model (Model.h:222-224)
extern "C" int mmf_fusion_set_confidence_threshold(mmf_fusion* f, float val) {
    MMF_REQUIRE(f != nullptr, "mmf_fusion_set_confidence_threshold: null fusion object");
    f->models[0]->model->conf_threshold = val;
    return MMF_OK;
}
extern "C" int mmf_fusion_set_tick(mmf_fusion* f, int val) {  // :1116
    MMF_REQUIRE(f != nullptr, "mmf_fusion_set_tick: null fusion object");
    f->tick = val;
    return MMF_OK;
}
extern "C" int mmf_fusion_get_config(mmf_fusion* f, mmf_fusion_config* out) {
    MMF_REQUIRE(f && out, "mmf_fusion_get_config: null argument");
    *out = f->cfg;
    return MMF_OK;
}
extern "C" int mmf_fusion_set_segmentation_callback(mmf_fusion* f, mmf_segmentation_fn fn, void* user) {
    MMF_REQUIRE(f != nullptr, "mmf_fusion_set_segmentation_callback: null fusion object");
    f->seg_fn = fn, f->seg_user = user;
    return MMF_OK;
}
// scheduleDeactivation (MultiMotionFusion.cpp: scheduled_model_deactivation, applied at the next frame :279-283)
extern "C" int mmf_fusion_schedule_deactivation(mmf_fusion* f, int id) {
    MMF_REQUIRE(f != nullptr && id > 0, "mmf_fusion_schedule_deactivation: bad argument (the global model stays)");
    f->scheduled_deactivation.push_back(id);
    return MMF_OK;
}

// Model::computeFusionWeight (Model.cpp:876-891)
static float fusion_weight(const float* pose, const float* last_pose, float multiplier) {
    float inv[16];
    inverse4f_host(pose, inv);  // getLastTransform() = getPose().inverse() * lastPose (Model.h:305)
    return mmf::host::compute_fusion_weight(inv, last_pose, multiplier);
}
extern "C" int mmf_compute_fusion_weight(const float pose[16], const float last_pose[16], float multiplier, float* out) {
    MMF_REQUIRE(pose && last_pose && out, "mmf_compute_fusion_weight: null argument");
    *out = fusion_weight(pose, last_pose, multiplier);
    return MMF_OK;
}

// Eigen::Quaternionf(rotation).coeffs() = (x, y, z, w), float32 (Eigen/src/Geometry/Quaternion.h,
// quaternionbase_assign_impl: branch on the trace, else on the largest diagonal element)
static void pose_to_log7(const float* T, float p[7]) {
    p[0] = T[3], p[1] = T[7], p[2] = T[11];
    const float R[3][3] = {{T[0], T[1], T[2]}, {T[4], T[5], T[6]}, {T[8], T[9], T[10]}};
    float q[4];  // x y z w
    float t = (R[0][0] + R[1][1]) + R[2][2];
    if (t > 0.f) {
        t = std::sqrt(t + 1.0f);
        q[3] = 0.5f * t;
        t = 0.5f / t;
        q[0] = (R[2][1] - R[1][2]) * t;
        q[1] = (R[0][2] - R[2][0]) * t;
        q[2] = (R[1][0] - R[0][1]) * t;
    } else {
        int i = 0;
        if (R[1][1] > R[0][0]) i = 1;
        if (R[2][2] > R[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(((R[i][i] - R[j][j]) - R[k][k]) + 1.0f);
        q[i] = 0.5f * t;
        t = 0.5f / t;
        q[3] = (R[k][j] - R[j][k]) * t;
        q[j] = (R[j][i] + R[i][j]) * t;
        q[k] = (R[k][i] + R[i][k]) * t;
    }
    p[3] = q[0], p[4] = q[1], p[5] = q[2], p[6] = q[3];
}

// every model's odometry reads the sensor-side images of the frame from the primary (global) odometry's buffers
static void odom_alias_sensor_side(mmf_odom* o, const mmf_odom* primary) {
    for (int i = 0; i < MMF_NUM_PYRS; ++i) {
        o->vmaps_curr[i] = primary->vmaps_curr[i], o->nmaps_curr[i] = primary->nmaps_curr[i];
        o->next_image[i] = primary->next_image[i], o->last_next_image[i] = primary->last_next_image[i];
        o->dIdx[i] = primary->dIdx[i], o->dIdy[i] = primary->dIdy[i];
        o->depth_pyr[i] = primary->depth_pyr[i];
    }
    o->depth_l0 = primary->depth_l0;
}

// where an OBJECT model's latest prediction is non-zero (device, PassBoxes::spl_nz of its last resolve), for a model-side
// preparation that may then leave the rest of the frame alone; null: unknown, or not such a model
static const int* fusion_pred_box(const FusionModel* fm, int side) {
    const mmf_model* m = fm->model;
    if (side != PREP_MODEL_SIDE || fm->fill_in || !fm->odom->sparse || !m->boxes || !m->spl_nz_known) return nullptr;
    return m->boxes->spl_nz[m->sgen & 1u];
}
// the covered thumbnail samples of a model's latest prediction (thumbnail_count_px)
static const int* fusion_thumb_count(const mmf_model* m) {
    return reinterpret_cast<const int*>(&m->totals[4 + (m->thumb_gen & 1)]);
}

// Model::combinedPredict(ACTIVE) + Model::performFillIn of one model (the body of predict(), :863-875)
static int fusion_predict_model(mmf_fusion* f, FusionModel* fm) {
    const mmf_fusion_config& g = f->cfg;
    if (fm->fill_in)  // combinedPredict + performFillIn in one pass
        return model_combined_predict(fm->model, g.max_depth_processed, f->tick, f->tick, g.time_delta, f->frame_rgb,
                                      f->depth_filtered, g.frame_to_frame_rgb, /*lost*/ 0);
    return mmf_model_combined_predict(fm->model, g.max_depth_processed, f->tick, f->tick, g.time_delta);
}

// The reference predicts twice per frame: behind the tracking (:675) and at the end (:821).  The first prediction's images
// feed loop closure (closeLoops: ferns, :682, and the model-to-model odometry, :714-760), which does not run inside this
// library; the segmentation (:412) comes before it and reads the previous frame's prediction (as a segmentation callback
// does here), a caller sees the images only between calls, and the fuse / clean passes read their own index maps.  Its images are overwritten by the second prediction before
// anything can read them, so it is not enqueued (bit-identical poses, maps and images: every oracle parity test runs this
// way, and tests/test_gpu_fusion.py compares the two).  MMF_MID_PREDICT=1 runs it as the reference does.
static std::atomic<int> g_mid_predict{-1};  // -1: MMF_MID_PREDICT decides (default off); 0 / 1: mmf_debug_set_mid_predict
extern "C" int mmf_debug_set_mid_predict(int on) {
    g_mid_predict.store(on < 0 ? -1 : (on ? 1 : 0));
    return MMF_OK;
}
static bool fusion_mid_predict() {
    return g_mid_predict.load() > 0;
}

// predictIndices -> fuse -> predictIndices -> clean of one model (:791-816 per model; models never read each
// other's surfels, so running the four passes model by model on the model's own stream gives the same maps as
// the reference's pass-by-pass loops over the list)
static int fusion_fuse_clean_model(mmf_fusion* f, FusionModel* fm, float weighting, bool indices_done = false) {
    const mmf_fusion_config& g = f->cfg;
    int rc = indices_done ? MMF_OK : mmf_model_predict_indices(fm->model, f->tick, g.max_depth_processed, g.time_delta);
    if (rc) return rc;
    // (the fuse's update pass projects the surfels for the predictIndices behind it: MMF_FUSE_INDEX=0 keeps the two apart)
    const bool merged = tunables().fuse_index;
    const IndexArgs ia = model_index_args(fm->model, f->tick, g.max_depth_processed, g.time_delta);
    rc = model_fuse(fm->model, f->tick, f->frame_rgb, f->mask, f->frame_depth, f->depth_filtered, g.max_depth_processed, weighting,
                    merged ? &ia : nullptr);
    if (rc) return rc;
    rc = model_predict_indices(fm->model, f->tick, g.max_depth_processed, g.time_delta, merged);
    if (rc) return rc;
    return mmf_model_clean(fm->model, f->tick, g.time_delta, g.max_depth_processed, f->depth_filtered, f->mask, g.outlier_coeff);
}

// getMaxDepth (:408): float operands, double arithmetic (1.2 is a double literal), returned as float
static float seg_max_depth(const mmf_segmentation_model& d) { return (float)((double)d.depth_mean + (double)d.depth_std * 1.2); }

// test / A-B hook: how the object models' passes go out (fusion_batch_mode)
static std::atomic<int> g_batch_passes{-1};  // -1: MMF_PASS_BATCH / the number of object models decide
extern "C" int mmf_debug_set_pass_batch(int mode) {
    g_batch_passes.store(mode < 0 ? -1 : (mode > 2 ? 2 : mode));
    return MMF_OK;
}
// 0: model by model; 1: one launch per pass, each covering the whole frame (*_batched_kernel); 2: one launch per pass,
// restricted to where the models are (pass_rect.hpp).  Neither the hook nor MMF_PASS_BATCH says: by the number of object
// models this GPU runs -- model by model on the models' own streams up to three of them, restricted launches from four on
// (measured, LABNOTES r5: 8 models 0.69-0.70 against 0.76-0.78 ms, 5 models 0.60-0.62 against 0.65-0.67, 4 models 0.575-0.58
// against 0.55-0.59)
constexpr int kRectPassObjects = 4;
static int fusion_batch_mode(const mmf_fusion* f) {
    int v = g_batch_passes.load();
    if (v < 0) v = tunables().pass_batch;
    if (v >= 0) return v;
    int objects = 0;
    for (size_t k = 1; k < f->models.size(); ++k) objects += fusion_owns(f, k) ? 1 : 0;
    return objects >= kRectPassObjects ? 2 : 0;
}
// the boxes of the ids of the frame's id image (pass_rect.hpp: mask_boxes_kernel), on `st`
static int fusion_note_mask_boxes(mmf_fusion* f, hipStream_t st) {
    if (!f->mask_boxes) {
        MMF_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&f->mask_boxes), 256 * 4 * sizeof(unsigned long long)));
        MMF_HIP_TRY(hipMemsetAsync(f->mask_boxes, 0, 256 * 4 * sizeof(unsigned long long), st));
    }
    ++f->mask_gen;
    const int tiles = ((f->width + 63) / 64) * ((f->height + 15) / 16);
    hipLaunchKernelGGL(mask_boxes_kernel, dim3(tiles), dim3(256), 0, st, f->mask, f->width, f->height, f->mask_boxes, f->mask_gen);
    MMF_HIP_TRY(hipGetLastError());
    return MMF_OK;
}
// `st` (the stream a batch of passes goes out on: objs[0]'s) waits for whatever the other models' own streams still hold
// (a stream that has drained holds nothing: no wait is enqueued)
static int fusion_lanes_join(const std::vector<FusionModel*>& objs, hipStream_t st) {
    for (size_t k = 1; k < objs.size(); ++k) {
        hipStream_t ls = objs[k]->lane->stream;
        if (ls == st) continue;
        if (hipStreamQuery(ls) == hipSuccess) continue;
        (void)hipGetLastError();
        MMF_HIP_TRY(hipEventRecord(objs[k]->ev_done, ls));
        MMF_HIP_TRY(hipStreamWaitEvent(st, objs[k]->ev_done, 0));
    }
    return MMF_OK;
}

static int lane_wait(FusionModel* fm, hipEvent_t ev) {
    MMF_HIP_TRY(hipStreamWaitEvent(fm->lane->stream, ev, 0));
    return MMF_OK;
}

// inactivateModel (:962-981) without the on-disk model database: the model leaves the active list and keeps its map
static void fusion_inactivate(mmf_fusion* f, FusionModel* fm) {
    f->models.erase(std::find(f->models.begin(), f->models.end(), fm));
    f->inactive.push_back(fm);
}

// spawnObjectModel (:938-947)
static int fusion_spawn(mmf_fusion* f, FusionModel** out) {
    FusionModel* fm = nullptr;
    if (!f->preallocated.empty()) {
        fm = f->preallocated.front();
        f->preallocated.erase(f->preallocated.begin());
    } else {
        int rc = fusion_model_create(f, fusion_next_model_id(f, true), f->cfg.conf_object_init, 0, true, &fm);
        if (rc) return rc;
    }
    // frameToModel.initFirstRGB(textures[RGB]) (:946): with the shared sensor-side images, the "last" image the new
    // model's SO3 pre-alignment reads next frame IS this frame's intensity pyramid (same kernels, same bits)
    *out = fm;
    return MMF_OK;
}

static int fusion_prefetch_impl(mmf_fusion* f, const uint8_t* rgb, const float* depth, int tick_at_use);
static int fusion_prefetch_image(mmf_fusion* f, const uint8_t* rgb, int tick_at_use, bool ahead);
static int fusion_stage_host_next(mmf_fusion* f);
// a stream waits for an event only when the event has not happened yet (a wait is a barrier packet on the queue either way)
static hipError_t fusion_wait_unless_done(hipStream_t s, hipEvent_t e) {
    if (hipEventQuery(e) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    return hipStreamWaitEvent(s, e, 0);
}
static int fusion_stage_host_rgb(mmf_fusion* f);

static int fusion_process_frame_impl(mmf_fusion* f, const mmf_frame* fr) {
    MMF_REQUIRE(f != nullptr && fr != nullptr, "mmf_fusion_process_frame: null argument");
    const uint8_t* rgb = fr->rgb;
    const float* depth = fr->depth;
    if (!rgb || !depth || fr->timestamp < 0)  // MultiMotionFusion.cpp:209-212
        return fail(MMF_ERR_INVALID, "invalid image data");
    mmf_ctx* c = f->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    const auto t_begin = std::chrono::steady_clock::now();
    // MMF_HOST_TRACE=1: where the calling thread is, in us after the call began, at six points of the call (averages over 100 calls)
    const bool host_trace = tunables().host_trace;
    auto stamp = [&](int i) {
        if (host_trace) f->trace_us[i] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    };
    f->t_tracking_s = 0;
    const mmf_fusion_config& g = f->cfg;
    const float weight_multiplier = fr->weight_multiplier;
    const bool have_init = fr->init_transforms != nullptr && fr->n_init_transforms > 0;  // odom_cfg.init == "kp"
    FusionModel* global = f->models[0];
    int rc = MMF_OK;
    bool prefetched = false;
    bool next_prefetched = false;  // mmf_frame::next_* has been enqueued
    if (f->pre_valid) {  // whatever was prefetched has to be complete before this frame touches the same buffers
        MMF_HIP_TRY(hipStreamWaitEvent(c->stream, f->ev_prefetch_done, 0));  // (recorded behind BOTH side streams' work)
        prefetched = f->pre_rgb == rgb && f->pre_depth == depth;
        f->pre_valid = false;
    }
    const bool track = f->tick > 1 && (fr->bootstrap || !fr->in_pose);  // :299 "regular execution"
    // a prefetched SO3 pre-alignment only counts for the frame it was computed for, and only when that frame is tracked
    const int so3_ready = f->so3_stage_ready;
    f->so3_stage_ready = -1;
    f->image_pre_rgb = nullptr;
    const OdomState* const so3_stage =
        (so3_ready >= 0 && prefetched && track && !(have_init && !fr->icp_refine)) ? f->so3_stage[so3_ready] : nullptr;
    for (FusionModel* fm : f->models) fm->odom->so3_prefetched = false, fm->odom->so3_stage = nullptr;
    if (prefetched) {  // the filter (:262) and the input-side preparation already ran on the side stream
        f->cur ^= 1;
        f->depth_filtered = f->filtered[f->cur];
        odom_adopt_gradients(global->odom);
    } else {
        rc = mmf_filter_depth(c, depth, f->width, f->height, g.depth_cutoff, f->depth_filtered);  // :262
        if (rc) return rc;
    }
    f->frame_rgb = rgb, f->frame_depth = depth;
    f->inputs_free_recorded = false;  // (set again by the branches below that record ev_inputs_free)
    f->host_caught_up = false;
    if (!g.enable_multiple_models && !f->mask_is_zero) {  // :268-275: everything is background
        MMF_HIP_TRY(hipMemsetAsync(f->mask, 0, (size_t)f->width * f->height, c->stream));
        f->mask_is_zero = true;
    }

    for (int id : f->scheduled_deactivation)  // :279-283
        if (FusionModel* fm = fusion_find(f, id)) fusion_inactivate(f, fm);
    f->scheduled_deactivation.clear();

    if (f->tick == 1) {  // :290-296
        if (fusion_owns(f, 0)) {
            rc = mmf_model_initialise(global->model, rgb, depth, f->depth_filtered, f->tick, g.max_depth_processed);
            if (rc) return rc;
        }
        rc = mmf_odom_init_first_rgb(global->odom, rgb, 0, 3);  // sensor side: every rank
        if (rc) return rc;
        MMF_HIP_TRY(hipEventRecord(f->ev_inputs_free, c->stream));
        f->inputs_free_recorded = true;
        MMF_HIP_TRY(hipEventRecord(f->ev_frame_ready, c->stream));
    } else {
        f->tracking_ok = 1;
        const size_t n_models = f->models.size();
        if (track) {
            MMF_REQUIRE(!have_init || !g.frame_to_frame_rgb, "ICP initialisation not supported in frame-to-frame mode");  // :370
            // generateCUDATextures (:302) + the sensor side of Model::initICP (Model.cpp:402-403: initICP, initRGB), once
            // for all models.  One model without pose initialisation: sensor side and model side share four launches.
            const auto t_track = std::chrono::steady_clock::now();
            for (size_t k = 0; k < n_models; ++k) {  // is last frame's end-of-frame preparation of a model still good?
                FusionModel* fm = f->models[k];
                float pose_now[16];
                mmf_model_get_pose(fm->model, pose_now);
                fm->spec_hit = fm->spec_valid && fusion_owns(f, k) && !have_init &&
                               std::memcmp(pose_now, fm->spec_pose, sizeof(pose_now)) == 0 &&
                               fm->model->tex_gen == fm->spec_tex_gen && fm->spec_f2f == g.frame_to_frame_rgb &&
                               fm->odom->prep_batched;
                fm->spec_valid = false;
                fm->odom->begin_spec_ok = fm->spec_hit;  // (the tracking's beginning rode that preparation: odom_begin_rider)
                fm->early_done = fm->early_fused = false;
            }
            const bool one_pass = n_models == 1 && !have_init && fusion_owns(f, 0) && !global->spec_hit;
            if (!prefetched && !one_pass) {
                float identity[16];
                identity16(identity);
                rc = odom_prepare_batched(global->odom, f->depth_filtered, g.max_depth_processed, rgb, 3, nullptr, nullptr,
                                          nullptr, 4, identity, nullptr, nullptr, nullptr, nullptr,
                                          PREP_INPUT_IMAGE | PREP_INPUT_DEPTH);
                if (rc) return rc;
                odom_adopt_gradients(global->odom);
            }
            // (waited for by the other models' streams only: one model and no segmentation = no marker on the model's stream)
            const bool lanes_wait = g.enable_multiple_models || n_models > 1;
            if (!one_pass && lanes_wait) MMF_HIP_TRY(hipEventRecord(f->ev_frame_ready, c->stream));
            // Round 3: the IMAGE side of the next frame -- intensity pyramid, gradients, SO3 pre-alignment: sixteen launches --
            // depends on nothing the chains read or write (the image ring and the gradients are double buffered, the
            // pre-alignment runs in a state of its own), so it need not wait behind the pose, where its enqueue was the longest
            // part of the host's tail (93 us) and its stream the last thing the next frame's chain waited for.
            //   MMF_EARLY_IMAGE=start (default): enqueued HERE, before this frame's chain: the GPU is still working off the
            //     last frame's tail then, and the image side runs beside that, not beside the latency-bound chain;
            //   =chain: after the chain's enqueue, while the host would only wait (runs beside the chain: +5..35 us on it);
            //   =off: at the end of the call with the depth side.
            // so3_stage: this frame's own pre-alignment ran ahead as well -- inside the chain it reads the LAST frame's level-2
            // image, the half of the image ring the next frame's pyramid is written to.
            // (-1, the default: `start` with one or two models on this GPU -- the GPU is still busy when the call begins --,
            // `chain` from three on: there the GPU waits for the chain's first launch when the call begins, and sixteen
            // launches enqueued in front of it are 50-80 us of that wait: 4 models 0.617 -> 0.558 ms, 8 models 0.70 -> 0.69)
            int owned_models = 0;
            for (size_t k = 0; k < n_models; ++k) owned_models += fusion_owns(f, k) ? 1 : 0;
            const int early_image = tunables().early_image >= 0 ? tunables().early_image : (owned_models >= 3 ? 1 : 2);
            const bool next_from_host = f->host_next.slot >= 0 && f->up_dev[0] != nullptr;  // (a frame still being uploaded)
            const bool image_early_any = fr->next_rgb && fr->next_depth && f->side2 && g.so3 && so3_stage != nullptr;
            const bool image_early_ok = image_early_any && !next_from_host;
            if (early_image == 2 && image_early_ok) {
                // the ring as it will be once this frame's chain is enqueued (RGBDOdometry.cpp:469-473; odom_enqueue_tracking)
                mmf_odom* go = global->odom;
                for (int i = 0; i < MMF_NUM_PYRS; ++i) std::swap(go->last_next_image[i], go->next_image[i]);
                rc = fusion_prefetch_image(f, fr->next_rgb, f->tick + 1, true);
                for (int i = 0; i < MMF_NUM_PYRS; ++i) std::swap(go->last_next_image[i], go->next_image[i]);
                if (rc) return rc;
            }
            stamp(6);
            // Several models in one chain (below): an object model's stream gets its next work behind that chain -- it waits for
            // an event recorded there -- so the wait for the frame's sensor side here would be a second barrier packet per stream
            // and two host calls per model at the point of the call where the GPU waits for the calling thread.
            int will_track = 0;
            for (size_t k = 0; k < n_models; ++k) will_track += fusion_owns(f, k) ? 1 : 0;
            // (only where the camera model leads the chain: its stream is the one the sensor side and the end-of-frame preparation
            // are ordered on; a rank of a sharded run that holds object models only keeps every wait)
            const bool will_batch = will_track > 1 && will_track <= kMaxBatch && !have_init && g.batch_tracking && fusion_owns(f, 0);
            std::vector<FusionModel*> tracked;
            for (size_t k = 0; k < n_models; ++k) {  // :312-387, enqueue only
                FusionModel* fm = f->models[k];
                fm->tracking = false;
                if (!fusion_owns(f, k)) continue;
                if (k > 0) {
                    if (!will_batch) {
                        rc = lane_wait(fm, f->ev_frame_ready);
                        if (rc) return rc;
                    }
                    odom_alias_sensor_side(fm->odom, global->odom);
                }
                bool do_icp = true;
                if (have_init) {  // initialise by track transformation (:316-376)
                    do_icp = fr->icp_refine != 0;
                    float pose[16], tnew[16];
                    mmf_model_get_pose(fm->model, pose);
                    const float* T = fr->init_transforms + 16 * (k < (size_t)fr->n_init_transforms ? k : 0);
                    if (k >= (size_t)fr->n_init_transforms) {
                        std::memcpy(tnew, pose, sizeof(pose));  // no transformation for this model: keep its pose
                    } else if (fm->model->id == 0) {
                        mmf::host::matmul4(pose, T, tnew);  // Tnew = model->getPose() * T (:331)
                    } else {
                        mmf::host::matmul4(T, pose, tnew);  // Tnew = T * model->getPose() (:334)
                    }
                    mmf_model_set_pose(fm->model, tnew);  // overridePose: pose = lastPose = Tnew (:350, Model.h:301-304)
                    std::memcpy(fm->last_pose, tnew, sizeof(tnew));
                    rc = fusion_predict_model(f, fm);  // :353-355
                    if (rc) return rc;
                    // Model::fuse(..., weightMultiplier) applies computeFusionWeight(weightMultiplier) (:359-360, Model.cpp:918)
                    rc = fusion_fuse_clean_model(f, fm, fusion_weight(tnew, fm->last_pose, weight_multiplier));  // :357-366
                    if (rc) return rc;
                }
                if (!do_icp) continue;  // no refinement, use the initial pose directly (:382-385)
                // Model::performTracking (Model.cpp:409-433) with Model::initICP (:390-407).  requiresFillIn (:380,
                // :877-895) is decided on the device: the preparation jobs pick their sources by the count of covered
                // thumbnail samples the prediction's resolve pass left behind (surfel_kernels.hpp: thumbnail_count_px)
                float pose[16];
                mmf_model_get_pose(fm->model, pose);
                std::memcpy(fm->last_pose, pose, sizeof(pose));  // lastPose = pose (Model.cpp:412)
                fm->tracking = true;
                tracked.push_back(fm);
            }
            stamp(7);
            if (so3_stage)  // every chain enqueued below starts from the prefetched pre-alignment
                for (FusionModel* fm : tracked) fm->odom->so3_prefetched = true, fm->odom->so3_stage = so3_stage;
            // ONE chain of launches for all tracked models (gridDim.y = model) when every level runs on the fused
            // producer path and no model went through a pose-initialisation round on its own stream; else one chain
            // per model on the model's stream.  Either way nothing waits here.
            const bool batched = tracked.size() > 1 && tracked.size() <= (size_t)kMaxBatch && !have_init && g.batch_tracking;
            const unsigned ext_gen = ++f->extent_seq;
            for (FusionModel* fm : tracked) fm->odom->sparse = fm != global && !fm->fill_in;  // an object model: extent.hpp, ChainGeom
            auto collect_prep = [&](PrepStages& stages, FusionModel* fm, int side) {
                const mmf_model* m = fm->model;
                const uint8_t* pi = (const uint8_t*)((g.frame_to_frame_rgb && fm->fill_in) ? m->fill_image : m->image);
                float pose[16];
                mmf_model_get_pose(fm->model, pose);
                odom_prepare_collect(stages, fm->odom, f->depth_filtered, g.max_depth_processed, rgb, 3, (const float*)m->vertexConf,
                                     (const float*)m->normalRadius, pi, 4, pose,
                                     fm->fill_in ? fusion_thumb_count(m) : nullptr, (const float*)m->fill_vertex,
                                     (const float*)m->fill_normal, (const uint8_t*)m->fill_image, side, (m->width / 20) * (m->height / 20),
                                     0.75f, fm->odom->sparse ? ext_gen : 0u, fusion_pred_box(fm, side));
            };
            bool batch_ok = batched;
            if (batched) {
                FusionModel* lead = tracked[0];
                hipStream_t st = lead->lane->stream;
                for (size_t k = 1; k < tracked.size(); ++k) {  // the other models' last work (previous frame's predict) precedes
                    // (a model prepared at the end of the last call, behind the join of all streams there, and untouched since:
                    // nothing enqueued below reads what its stream may still hold)
                    if (tracked[k]->spec_hit) continue;
                    MMF_HIP_TRY(hipEventRecord(tracked[k]->ev_done, tracked[k]->lane->stream));
                    MMF_HIP_TRY(hipStreamWaitEvent(st, tracked[k]->ev_done, 0));
                }
                PrepStages stages;
                stages.set_critical(true);  // the model's stream
                for (size_t k = 0; k < tracked.size(); ++k) {
                    if (tracked[k]->spec_hit) {
                        tracked[k]->odom->depth_l0 = f->depth_filtered;
                        continue;
                    }
                    collect_prep(stages, tracked[k], (k == 0 && tracked[k] == global && one_pass && !prefetched) ? PREP_ALL : PREP_MODEL_SIDE);
                }
                rc = stages.launch(st);
                if (rc) return rc;
                stamp(8);
                // (a PREP_ALL collect above prepared this frame's image side as well; NOT when the pending gradients are the
                // next frame's, from the image side enqueued ahead a few lines up)
                if (one_pass && !prefetched) odom_adopt_gradients(global->odom);
                batch_ok = odom_batchable(lead->odom, g.rgb_only, g.icp_weight, g.pyramid, g.fast_odom);
                if (batch_ok) {
                    TrackBatch tb;
                    tb.n = (int)tracked.size();
                    std::memset(&tb.bd, 0, sizeof(tb.bd));
                    for (int k = 0; k < tb.n; ++k) {
                        tb.o[k] = tracked[k]->odom;
                        tb.bd.d[k] = (long long)(reinterpret_cast<char*>(tracked[k]->odom->slab) - reinterpret_cast<char*>(lead->odom->slab));
                        float pose[16];
                        mmf_model_get_pose(tracked[k]->model, pose);
                        for (int r = 0; r < 3; ++r) {
                            for (int q = 0; q < 3; ++q) tb.poses.rot[k][r * 3 + q] = pose[r * 4 + q];
                            tb.poses.trans[k][r] = pose[r * 4 + 3];
                        }
                    }
                    lead->odom->exclusive_chain = true;  // one chain for all of them
                    rc = odom_enqueue_tracking(lead->odom, tb.poses.trans[0], tb.poses.rot[0], g.rgb_only, g.icp_weight, g.pyramid,
                                               g.fast_odom, g.so3, lead->icp_error, lead->rgb_error, &tb);
                    if (rc) return rc;
                    // the lanes continue after the chain.  With a segmentation every lane waits for ev_frame_ready before its
                    // passes anyway, and that event is recorded on this very stream (the camera model's) behind the chain (the mask's upload, below):
                    // an event and a wait per model here would be fourteen host calls and seven barrier packets for nothing
                    for (size_t k = 1; k < tracked.size() && !(g.enable_multiple_models && st == c->stream); ++k) {
                        MMF_HIP_TRY(hipEventRecord(tracked[k]->ev_done, st));
                        MMF_HIP_TRY(hipStreamWaitEvent(tracked[k]->lane->stream, tracked[k]->ev_done, 0));
                    }
                }
            }
            for (size_t k = 0; k < tracked.size() && !batch_ok; ++k) {
                FusionModel* fm = tracked[k];
                if (will_batch && fm != global) {  // (the wait skipped above)
                    rc = lane_wait(fm, f->ev_frame_ready);
                    if (rc) return rc;
                }
                if (fm->spec_hit) {  // the model side was prepared at the end of the last frame; the sensor side by the prefetch
                    fm->odom->depth_l0 = f->depth_filtered;  // (or just above)
                } else if (!batched) {  // (a failed batch has prepared every model already)
                    PrepStages stages;
                stages.set_critical(true);  // the model's stream
                    collect_prep(stages, fm, (fm == global && one_pass && !prefetched) ? PREP_ALL : PREP_MODEL_SIDE);
                    rc = stages.launch(fm->lane->stream);
                    if (rc) return rc;
                    if (fm == global && one_pass && !prefetched) odom_adopt_gradients(global->odom);  // (PREP_ALL: this frame's image side as well)
                } else if (k > 0) {  // prepared on the leader's stream
                    MMF_HIP_TRY(hipEventRecord(fm->ev_done, tracked[0]->lane->stream));
                    MMF_HIP_TRY(hipStreamWaitEvent(fm->lane->stream, fm->ev_done, 0));
                }
                float pose[16];
                mmf_model_get_pose(fm->model, pose);
                const float trans[3] = {pose[3], pose[7], pose[11]};
                const float rot[9] = {pose[0], pose[1], pose[2], pose[4], pose[5], pose[6], pose[8], pose[9], pose[10]};
                fm->odom->exclusive_chain = tracked.size() == 1;  // several chains side by side: no in-launch barriers
                // The frame's first projection, enqueued right behind this chain (below), carries the hand-over to the host and
                // the fusion weight on one extra workgroup (frame_rider.hpp): the chain's last launch is the solve alone (13 ->
                // 5.7 us on the stream a frame waits for).
                fm->odom->defer_publish = tracked.size() == 1 && !fr->bootstrap && !have_init && !g.rgb_only && f->tracking_ok;
                rc = odom_enqueue_tracking(fm->odom, trans, rot, g.rgb_only, g.icp_weight, g.pyramid, g.fast_odom, g.so3,
                                           fm->icp_error, fm->rgb_error);
                if (rc) return rc;
            }
            // One model on the context's stream, nothing between its tracking and its fusion that the host decides: the
            // frame's first projections -- predict() (:675) and the first predictIndices (:792) -- are enqueued right here,
            // behind the chain and the copy of its result, with the inverse pose read from the odometry's device state.
            // They run while the host picks the pose up and prepares the fusion passes (that turnaround used to be ~25 us
            // of idle GPU per frame).
            // (With several models it is one model per process in the sharded configuration: the segmentation between
            // tracking and fusion touches masks, thresholds and the list, none of which a projection reads.)
            static_assert(std::is_trivially_copyable<mmf_model>::value, "the speculation rollback below copies mmf_model by value");
            mmf_model early_snapshot;  // the model's host bookkeeping before the passes enqueued ahead of the pose (see `retrack` below)
            bool early_snapshot_valid = false;
            if (tracked.size() == 1 && !fr->bootstrap && !have_init && !g.rgb_only && f->tracking_ok) {
                FusionModel* fm = tracked[0];
                mmf_model* m = fm->model;
                early_snapshot = *m, early_snapshot_valid = true;
                m->abort_dev = &fm->odom->state->gn_fault;
                // "nothing enqueued so far reads the odometries' sensor-side buffers or the other filtered-depth buffer" holds
                // HERE, behind the one chain of this process; the passes enqueued next do not read them either.  No event
                // says so any more (see below: the host knows when it has the pose; a marker behind the chain cost the
                // model's stream ~4 us in front of the frame's first projection).
                m->t_inv_dev = fm->odom->state->pose_inv;
                m->rider = fm->odom->rider;
                fm->odom->rider = FrameRider();
                rc = fusion_mid_predict() ? fusion_predict_model(f, fm) : MMF_OK;
                if (rc == MMF_OK) rc = mmf_model_predict_indices(m, f->tick, g.max_depth_processed, g.time_delta);
                MMF_REQUIRE(rc != MMF_OK || m->rider.st == nullptr, "mmf_fusion_process_frame: the tracking result was not handed over");
                fm->early_done = rc == MMF_OK;
                // Without a segmentation the mask of the frame is known (all zeros) and nothing the host decides lies
                // between tracking and fusion: fuse -> predictIndices -> clean (:791-816) follow at once, with the pose and
                // Model::computeFusionWeight taken from the device state (odom_end, odom_fusion_weight_kernel).
                if (rc == MMF_OK && !g.enable_multiple_models) {
                    m->pose_dev = fm->odom->state->pose_out, m->weight_dev = &fm->odom->state->fusion_weight;
                    rc = fusion_fuse_clean_model(f, fm, weight_multiplier, true);
                    fm->early_fused = rc == MMF_OK;
                }
                m->t_inv_dev = m->pose_dev = m->weight_dev = nullptr;
                m->abort_dev = nullptr;
                if (rc) return rc;
            }
            // the sensor-side image ring (this frame's / last frame's intensity pyramid, RGBDOdometry.cpp:469-473) lives in
            // the global odometry and advances when its chain is enqueued: when its owner is another rank, the swap
            // happens here, whether or not this rank tracked anything (every rank's ring must advance with the global model's)
            const bool global_tracked = global->tracking;
            const bool global_tracks_somewhere = !(have_init && !fr->icp_refine);
            if (g.so3 && global_tracks_somewhere && !global_tracked)
                for (int i = 0; i < MMF_NUM_PYRS; ++i) std::swap(global->odom->last_next_image[i], global->odom->next_image[i]);
            if (early_image == 1 && image_early_ok && !tracked.empty()) {  // (see above)
                rc = fusion_prefetch_image(f, fr->next_rgb, f->tick + 1, true);
                if (rc) return rc;
            }
            stamp(0);
            // a host frame announced for the next call: staged and sent up now, while the GPU tracks and the host would only wait
            // (its colour image first; the depth image is joined where the depth side is enqueued, behind the pose)
            rc = fusion_stage_host_rgb(f);
            if (rc) return rc;
            // ... and its image side behind the upload (an announced HOST frame cannot have it at the start of the call: its
            // copy into pinned memory has only just begun then), beside the chain instead of behind the pose
            if (early_image != 0 && image_early_any && next_from_host && !tracked.empty() && f->host_next.rgb_staged) {
                MMF_HIP_TRY(fusion_wait_unless_done(f->side2, f->ev_up_rgb[f->host_next.slot]));
                rc = fusion_prefetch_image(f, fr->next_rgb, f->tick + 1, true);
                if (rc) return rc;
            }
            // The one-launch chain can give up (its count barrier needs every workgroup of a launch resident: another process on
            // the GPU can prevent that; OdomState::gn_fault).  Then nothing of the chain's result is valid and the passes enqueued
            // ahead of the pose have done nothing (MMF_SPECULATION_GUARD): the model's host bookkeeping goes back to where it
            // was, the process stops using that chain, and the frame's tracking is enqueued again -- the two-launch chain, from
            // the poses the frame started with (fm->last_pose) -- before the results are picked up a second time.
            auto retrack = [&]() -> int {
                std::vector<mmf_odom*> odoms;
                for (FusionModel* fm : tracked) odoms.push_back(fm->odom);
                odom_retrack_prepare(odoms.data(), (int)odoms.size(), g.so3);
                if (early_snapshot_valid) {
                    *tracked[0]->model = early_snapshot;
                    tracked[0]->early_done = tracked[0]->early_fused = false;
                    early_snapshot_valid = false;
                }
                for (FusionModel* fm : tracked) fm->odom->defer_publish = false, fm->odom->rider = FrameRider();
                if (batch_ok) {
                    FusionModel* lead = tracked[0];
                    TrackBatch tb;
                    tb.n = (int)tracked.size();
                    std::memset(&tb.bd, 0, sizeof(tb.bd));
                    for (int k = 0; k < tb.n; ++k) {
                        tb.o[k] = tracked[k]->odom;
                        tb.bd.d[k] = (long long)(reinterpret_cast<char*>(tracked[k]->odom->slab) - reinterpret_cast<char*>(lead->odom->slab));
                        for (int r = 0; r < 3; ++r) {
                            for (int q = 0; q < 3; ++q) tb.poses.rot[k][r * 3 + q] = tracked[k]->last_pose[r * 4 + q];
                            tb.poses.trans[k][r] = tracked[k]->last_pose[r * 4 + 3];
                        }
                    }
                    int rc2 = odom_enqueue_tracking(lead->odom, tb.poses.trans[0], tb.poses.rot[0], g.rgb_only, g.icp_weight, g.pyramid,
                                                    g.fast_odom, g.so3, lead->icp_error, lead->rgb_error, &tb);
                    if (rc2) return rc2;
                    for (size_t k = 1; k < tracked.size() && !(g.enable_multiple_models && lead->lane->stream == c->stream); ++k) {  // (as above)
                        MMF_HIP_TRY(hipEventRecord(tracked[k]->ev_done, lead->lane->stream));
                        MMF_HIP_TRY(hipStreamWaitEvent(tracked[k]->lane->stream, tracked[k]->ev_done, 0));
                    }
                    return MMF_OK;
                }
                for (FusionModel* fm : tracked) {
                    const float* p = fm->last_pose;
                    const float trans[3] = {p[3], p[7], p[11]};
                    const float rot[9] = {p[0], p[1], p[2], p[4], p[5], p[6], p[8], p[9], p[10]};
                    int rc2 = odom_enqueue_tracking(fm->odom, trans, rot, g.rgb_only, g.icp_weight, g.pyramid, g.fast_odom, g.so3,
                                                    fm->icp_error, fm->rgb_error);
                    if (rc2) return rc2;
                }
                return MMF_OK;
            };
            bool retracked = false;
            for (size_t k = 0; k < n_models; ++k) {  // the results, model by model
                FusionModel* fm = f->models[k];
                float pose[16];
                mmf_model_get_pose(fm->model, pose);
                if (fm->tracking) {
                    float trans[3], rot[9];
                    rc = odom_finish_tracking(fm->odom, trans, rot);
                    if (rc == kGnRetry && !retracked) {  // (at most once: the two-launch chain has no way to give up)
                        // every chain of the frame is void with it (a batch is one chain; chains side by side are two-launch chains)
                        for (FusionModel* t : tracked)
                            if (t->tracking && t != fm && t->odom->result_of) {  // drain what the other lanes' chains still publish
                                float tt[3], rr[9];
                                (void)odom_finish_tracking(t->odom, tt, rr);
                            }
                        rc = retrack();
                        if (rc) return rc;
                        retracked = true;
                        for (FusionModel* t : tracked) {  // poses already taken over from the void chain: back to the frame's start
                            mmf_model_set_pose(t->model, t->last_pose);
                            t->tracking = true;
                        }
                        k = (size_t)-1;  // pick the results up again, from the first model
                        continue;
                    }
                    if (rc == kGnRetry) return gn_retry_twice();
                    if (rc) return rc;
                    for (int r = 0; r < 3; ++r) {
                        for (int q = 0; q < 3; ++q) pose[r * 4 + q] = rot[r * 3 + q];
                        pose[r * 4 + 3] = trans[r];
                    }
                    mmf_model_set_pose(fm->model, pose);
                    fm->tracking = false;
                    if (fm->lane->stream == c->stream) f->host_caught_up = true;
                }
            }
            f->t_tracking_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_track).count();
            stamp(1);
            if (fr->bootstrap) {  // :397-400
                MMF_REQUIRE(fr->in_pose != nullptr, "mmf_fusion_process_frame: bootstrap needs in_pose");
                float pose[16], np[16];
                mmf_model_get_pose(global->model, pose);
                mmf::host::matmul4(pose, fr->in_pose, np);
                mmf_model_set_pose(global->model, np);  // overridePose
                std::memcpy(global->last_pose, np, sizeof(np));
            }
            // from here on nothing enqueued reads the odometries' sensor-side buffers or the other filtered-depth
            // buffer, and the HOST knows it: every chain's result has been received, so every chain -- and whatever its
            // stream held before it, e.g. this frame's own image-side preparation -- has run.  The next frame's side-stream
            // work, enqueued from here on, needs no event to wait for (inputs_free_recorded stays false).
            if (one_pass && lanes_wait) MMF_HIP_TRY(hipEventRecord(f->ev_frame_ready, c->stream));

            if (g.enable_multiple_models) {  // :407-622
                mmf_segmentation seg_cb;
                const mmf_segmentation* seg = fr->segmentation;
                if (!seg && f->seg_fn) {  // performSegmentation(frame) (:412)
                    std::memset(&seg_cb, 0, sizeof(seg_cb));
                    rc = f->seg_fn(f->seg_user, f, fr, &seg_cb);
                    if (rc) return fail(MMF_ERR_STATE, "mmf_fusion_process_frame: the segmentation callback failed");
                    seg = &seg_cb;
                }
                MMF_REQUIRE(seg && seg->mask, "mmf_fusion_process_frame: enableMultipleModels needs a segmentation "
                                              "(mmf_frame::segmentation or mmf_fusion_set_segmentation_callback)");
                // textures[MASK]->Upload(fullSegmentation) (:416)
                MMF_HIP_TRY(hipMemcpyAsync(f->mask, seg->mask, (size_t)f->width * f->height, hipMemcpyDeviceToDevice, c->stream));
                f->mask_is_zero = false;
                MMF_HIP_TRY(hipEventRecord(f->ev_frame_ready, c->stream));
                const int n_data = seg->model_data ? seg->n_models : 0;
                FusionModel* fresh = nullptr;
                if (seg->has_new_label) {  // :469-487
                    rc = fusion_spawn(f, &fresh);
                    if (rc) return rc;
                    if (n_data > 0)
                        fresh->model->max_depth = seg_max_depth(seg->model_data[n_data - 1]);
                }
                // Set max-depth (:585-586)
                for (size_t i = 1; i < f->models.size() && (int)i < n_data; ++i)
                    f->models[i]->model->max_depth = seg_max_depth(seg->model_data[i]);
                if (fresh && !fusion_owns_model(f, fresh)) {
                    f->models.push_back(fresh);  // another rank's model: bookkeeping only
                } else if (fresh) {  // :588-601: the first surfels of the new model, then it joins the list
                    rc = lane_wait(fresh, f->ev_frame_ready);
                    if (rc) return rc;
                    identity16(fresh->last_pose);
                    float pose[16];
                    mmf_model_get_pose(fresh->model, pose);
                    rc = mmf_model_predict_indices(fresh->model, f->tick, g.max_depth_processed, g.time_delta);
                    if (rc) return rc;
                    rc = mmf_model_fuse(fresh->model, f->tick, rgb, f->mask, depth, f->depth_filtered, g.max_depth_processed,
                                        fusion_weight(pose, fresh->last_pose, 100.f));  // fuse(..., 100) (:591-592)
                    if (rc) return rc;
                    // (the second predictIndices is commented out in the reference, :594)
                    rc = mmf_model_clean(fresh->model, f->tick, g.time_delta, g.max_depth_processed, f->depth_filtered, f->mask,
                                         g.outlier_coeff);
                    if (rc) return rc;
                    f->models.push_back(fresh);  // moveNewModelToList (:600)
                }
                // unseen models leave the list (:606-613); the confidence of object models rises (:616-620)
                std::vector<FusionModel*> lost;
                for (int i = 0; i < n_data; ++i) {
                    FusionModel* fm = fusion_find(f, (int)seg->model_data[i].id);
                    if (!fm || fm == fresh) continue;
                    if (seg->model_data[i].super_pixel_count <= 0 && ++fm->unseen > 0 && fm->model->id != 0) lost.push_back(fm);
                }
                for (FusionModel* fm : lost) fusion_inactivate(f, fm);
                for (size_t i = 1; i < f->models.size() && (int)i < n_data; ++i) {
                    const float old_conf = f->models[i]->model->conf_threshold;
                    const float avg = seg->model_data[i].avg_confidence;
                    const float m1 = old_conf < avg ? avg : old_conf;
                    f->models[i]->model->conf_threshold = 9.0f < m1 ? 9.0f : m1;
                }
            }
        } else {
            float pose[16];
            std::memcpy(pose, fr->in_pose, sizeof(pose));
            mmf_model_set_pose(global->model, pose);  // globalModel->overridePose(*inPose) (:670)
            std::memcpy(global->last_pose, pose, sizeof(pose));
            MMF_HIP_TRY(hipEventRecord(f->ev_inputs_free, c->stream));
            f->inputs_free_recorded = true;
            MMF_HIP_TRY(hipEventRecord(f->ev_frame_ready, c->stream));
        }

        stamp(2);
        // The OBJECT models' passes go out as one launch per pass for all of them (models_fuse_clean_batched) on the first
        // object's stream: ~9 launches per model and frame otherwise, and with seven objects the calling thread's launch rate
        // set the pace of this part of the frame.  The camera model keeps its own stream and kernels (riders, fill-in).
        std::vector<FusionModel*> objs;
        const bool fuse_now = !g.rgb_only && f->tracking_ok;
        const int pass_mode = fusion_batch_mode(f);
        for (size_t k = 1; k < f->models.size() && pass_mode != 0 && fuse_now && !fusion_mid_predict(); ++k) {
            FusionModel* fm = f->models[k];
            if (!fusion_owns(f, k) || fm->early_done || fm->early_fused || fm->fill_in || (int)objs.size() >= kMaxPassBatch) continue;
            objs.push_back(fm);
        }
        if (objs.size() < 2) objs.clear();
        for (size_t k = 0; k < f->models.size(); ++k) {  // predict() (:675), then :791-816, model by model
            FusionModel* fm = f->models[k];
            if (!fusion_owns(f, k)) continue;
            if (std::find(objs.begin(), objs.end(), fm) != objs.end()) continue;  // (batched below)
            if (k > 0) {
                rc = lane_wait(fm, f->ev_frame_ready);
                if (rc) return rc;
            }
            const bool early = fm->early_done;  // its predict + predictIndices are already enqueued
            fm->early_done = false;
            if (!early && fusion_mid_predict()) {
                rc = fusion_predict_model(f, fm);
                if (rc) return rc;
            }
            if (fuse_now && !fm->early_fused) {
                float pose[16];
                mmf_model_get_pose(fm->model, pose);
                rc = fusion_fuse_clean_model(f, fm, fusion_weight(pose, fm->last_pose, weight_multiplier), early);
                if (rc) return rc;
            }
            fm->early_fused = false;
        }
        if (!objs.empty()) {
            hipStream_t st = objs[0]->lane->stream;
            rc = lane_wait(objs[0], f->ev_frame_ready);
            if (rc) return rc;
            rc = fusion_lanes_join(objs, st);
            if (rc) return rc;
            mmf_model* ms[kMaxPassBatch];
            float wts[kMaxPassBatch];
            for (size_t k = 0; k < objs.size(); ++k) {
                float pose[16];
                mmf_model_get_pose(objs[k]->model, pose);
                ms[k] = objs[k]->model;
                wts[k] = fusion_weight(pose, objs[k]->last_pose, weight_multiplier);
            }
            if (pass_mode >= 2) {
                rc = fusion_note_mask_boxes(f, st);
                if (rc) return rc;
                rc = models_fuse_clean_rect(ms, (int)objs.size(), st, f->tick, g.time_delta, g.max_depth_processed, f->frame_rgb, f->mask,
                                            f->frame_depth, f->depth_filtered, g.outlier_coeff, wts, f->mask_boxes, f->mask_gen);
            } else {
                rc = models_fuse_clean_batched(ms, (int)objs.size(), st, f->tick, g.time_delta, g.max_depth_processed, f->frame_rgb, f->mask,
                                               f->frame_depth, f->depth_filtered, g.outlier_coeff, wts);
            }
            if (rc) return rc;
        }
    }
    stamp(3);
    {  // predict() (:821): the object models without fill-in as one batch, the others one by one
        std::vector<FusionModel*> objs;
        const int pass_mode = fusion_batch_mode(f);
        for (size_t k = 1; k < f->models.size() && pass_mode != 0; ++k) {
            FusionModel* fm = f->models[k];
            if (!fusion_owns(f, k) || fm->fill_in || !model_predict_batchable(fm->model) || (int)objs.size() >= kMaxPassBatch) continue;
            objs.push_back(fm);
        }
        if (objs.size() < 2) objs.clear();
        for (size_t k = 0; k < f->models.size(); ++k) {
            if (!fusion_owns(f, k)) continue;
            if (std::find(objs.begin(), objs.end(), f->models[k]) != objs.end()) continue;
            rc = fusion_predict_model(f, f->models[k]);
            if (rc) return rc;
        }
        if (!objs.empty()) {
            hipStream_t st = objs[0]->lane->stream;
            rc = fusion_lanes_join(objs, st);
            if (rc) return rc;
            mmf_model* ms[kMaxPassBatch];
            for (size_t k = 0; k < objs.size(); ++k) ms[k] = objs[k]->model;
            rc = pass_mode >= 2 ? models_combined_predict_rect(ms, (int)objs.size(), st, g.max_depth_processed, f->tick, f->tick, g.time_delta)
                                          : models_combined_predict_batched(ms, (int)objs.size(), st, g.max_depth_processed, f->tick, f->tick, g.time_delta);
            if (rc) return rc;
            // the other objects' streams continue behind the batch: whatever is enqueued on them next reads what it wrote
            MMF_HIP_TRY(hipEventRecord(objs[0]->ev_done, st));
            for (size_t k = 1; k < objs.size(); ++k) MMF_HIP_TRY(hipStreamWaitEvent(objs[k]->lane->stream, objs[0]->ev_done, 0));
        }
    }
    stamp(4);
    f->tick++;  // :825

    // :829-846: pose log (camera->world for the first model, object->world for the others)
    float global_pose[16];
    mmf_model_get_pose(global->model, global_pose);
    for (size_t k = 0; k < f->models.size(); ++k) {
        FusionModel* fm = f->models[k];
        if (fm->pose_log.capacity() == 0) continue;  // isLoggingPoses()
        // sharded: a model is logged by the rank that runs it -- the copies other ranks keep as bookkeeping hold poses that
        // arrive with the exchange, one to three frames late under mmf_shard_gather_poses_begin / _end (mmf_hip.h)
        if (f->shard_world > 1 && !fusion_owns(f, k)) continue;
        float T[16];
        if (k == 0) {
            std::memcpy(T, global_pose, sizeof(T));
        } else {
            float pose[16], inv[16];
            mmf_model_get_pose(fm->model, pose);
            inverse4f_host(pose, inv);
            mmf::host::matmul4(global_pose, inv, T);
        }
        PoseLogItem item;
        item.ts = fr->timestamp;
        pose_to_log7(T, item.p);
        fm->pose_log.push_back(item);
    }

    // the fusion's stream continues only after every lane: the next frame's filter, prefetch and mask upload
    // overwrite what the lanes read
    for (size_t k = 1; k < f->models.size(); ++k) {
        FusionModel* fm = f->models[k];
        if (!fusion_owns(f, k)) continue;
        MMF_HIP_TRY(hipEventRecord(fm->ev_done, fm->lane->stream));
        MMF_HIP_TRY(hipStreamWaitEvent(c->stream, fm->ev_done, 0));
    }
    if (fr->next_rgb && fr->next_depth && !next_prefetched) {  // (first frame, dictated pose, several lanes)
        rc = fusion_prefetch_impl(f, fr->next_rgb, fr->next_depth, f->tick);
        if (rc) return rc;
    }
    // next frame's model-side preparation, now (see FusionModel::spec_valid); behind the prefetch's enqueue: the side
    // streams have the longer way to go
    {
        FusionModel* only = nullptr;
        int owned = 0;
        for (size_t k = 0; k < f->models.size(); ++k)
            if (fusion_owns(f, k)) only = f->models[k], ++owned;
        if (owned == 1) {
            const mmf_model* m = only->model;
            hipStream_t st = only->lane->stream;
            // (requiresFillIn (:380, :877-895) of the next frame: decided on the device from the count this prediction's resolve left)
            PrepStages stages;
                stages.set_critical(true);  // the model's stream
            const uint8_t* pi = (const uint8_t*)((g.frame_to_frame_rgb && only->fill_in) ? m->fill_image : m->image);
            mmf_model_get_pose(only->model, only->spec_pose);
            odom_prepare_collect(stages, only->odom, f->depth_filtered, g.max_depth_processed, rgb, 3, (const float*)m->vertexConf,
                                 (const float*)m->normalRadius, pi, 4, only->spec_pose,
                                 only->fill_in ? fusion_thumb_count(m) : nullptr, (const float*)m->fill_vertex,
                                 (const float*)m->fill_normal, (const uint8_t*)m->fill_image, PREP_MODEL_SIDE,
                                 (m->width / 20) * (m->height / 20), 0.75f);
            // ... and the beginning of that tracking (odom_begin_kernel: the pose it starts from is this one, the pre-alignment
            // of the next frame sits staged since the start of this call) on one more workgroup of the preparation's last launch
            BeginRider rider;
            bool ride = false;
            if (only == global && !stages.empty() && fr->next_rgb && f->image_pre_rgb == fr->next_rgb && !g.rgb_only) {
                const OdomState* stage = (g.so3 && f->so3_stage_ready >= 0) ? f->so3_stage[f->so3_stage_ready] : nullptr;
                if (!g.so3 || stage != nullptr) {
                    if (stage) MMF_HIP_TRY(fusion_wait_unless_done(st, f->ev_prefetch2_done));  // (the staged pre-alignment is complete)
                    ride = odom_begin_rider(only->odom, only->spec_pose, g.rgb_only, g.icp_weight, g.pyramid, g.fast_odom, g.so3, stage, &rider);
                }
            }
            if (!ride) only->odom->begin_spec_valid = false;
            rc = stages.launch(st, ride ? &rider : nullptr);
            if (rc) return rc;
            only->spec_tex_gen = m->tex_gen;
            only->spec_f2f = g.frame_to_frame_rgb;
            only->spec_valid = true;
        } else if (owned >= std::max(2, tunables().spec_prep_all) && f->shard_world <= 1 && g.batch_tracking && owned <= kMaxBatch && tunables().spec_prep_all) {
            // Several models on this GPU: the same for all of them, in the launches they will share (the batched chain's
            // preparation: one set of stages, every model's jobs).  At the start of the next call these ~50 jobs were what the
            // calling thread enqueued first -- 130 us of it and as much of the GPU's -- while the GPU had nothing else to do; here
            // they queue up behind the frame's last passes.  A model whose pose or prediction changes before it is tracked
            // (a pose initialisation, a caller's predict()) is prepared again then (spec_hit).  From two models on (round 4: from
            // four -- with two or three the early preparation measured 6 % slower while every model's stream was joined by an
            // event in front of the chain; a model prepared HERE needs no such wait, and it is 4-7 % faster: LABNOTES r5).
            PrepStages stages;
            stages.set_critical(true);
            const unsigned ext_gen = ++f->extent_seq;
            for (size_t k = 0; k < f->models.size(); ++k) {
                if (!fusion_owns(f, k)) continue;
                FusionModel* fm = f->models[k];
                const mmf_model* m = fm->model;
                const uint8_t* pi = (const uint8_t*)((g.frame_to_frame_rgb && fm->fill_in) ? m->fill_image : m->image);
                fm->odom->sparse = fm != global && !fm->fill_in;
                mmf_model_get_pose(fm->model, fm->spec_pose);
                odom_prepare_collect(stages, fm->odom, f->depth_filtered, g.max_depth_processed, rgb, 3, (const float*)m->vertexConf,
                                     (const float*)m->normalRadius, pi, 4, fm->spec_pose,
                                     fm->fill_in ? fusion_thumb_count(m) : nullptr, (const float*)m->fill_vertex,
                                     (const float*)m->fill_normal, (const uint8_t*)m->fill_image, PREP_MODEL_SIDE,
                                     (m->width / 20) * (m->height / 20), 0.75f, fm->odom->sparse ? ext_gen : 0u, fusion_pred_box(fm, PREP_MODEL_SIDE));
                fm->spec_tex_gen = m->tex_gen;
                fm->spec_f2f = g.frame_to_frame_rgb;
                fm->spec_valid = true;
            }
            rc = stages.launch(c->stream);  // (every lane has joined this stream just above)
            if (rc) return rc;
        }
    }
    stamp(5);
    if (host_trace && ++f->trace_calls % 100 == 0) {
        std::fprintf(stderr, "host us from call start: next image side enqueued %.0f, lanes waiting %.0f, preparation enqueued %.0f (batched chains), chains enqueued %.0f, "
                             "poses here %.0f, segmentation handled %.0f, per-model passes enqueued %.0f, final predicts enqueued %.0f, end %.0f\n",
                     f->trace_us[6] / 100, f->trace_us[7] / 100, f->trace_us[8] / 100, f->trace_us[0] / 100, f->trace_us[1] / 100, f->trace_us[2] / 100,
                     f->trace_us[3] / 100, f->trace_us[4] / 100, f->trace_us[5] / 100);
        for (double& a : f->trace_us) a = 0;
    }
    f->t_frame_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return MMF_OK;
}

extern "C" int mmf_fusion_process_frame_ex(mmf_fusion* f, const mmf_frame* frame) { return fusion_process_frame_impl(f, frame); }

extern "C" int mmf_fusion_process_frame(mmf_fusion* f, const uint8_t* rgb, const float* depth, long long timestamp,
                                        const float* in_pose, float weight_multiplier, int bootstrap) {
    mmf_frame fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.rgb = rgb, fr.depth = depth, fr.timestamp = timestamp;
    fr.in_pose = in_pose, fr.weight_multiplier = weight_multiplier, fr.bootstrap = bootstrap;
    fr.icp_refine = 1;
    return fusion_process_frame_impl(f, &fr);
}

extern "C" int mmf_fusion_process_frame_init(mmf_fusion* f, const uint8_t* rgb, const float* depth, long long timestamp,
                                             const float* init_transform, int icp_refine, float weight_multiplier) {
    MMF_REQUIRE(init_transform != nullptr, "mmf_fusion_process_frame_init: null transformation");
    mmf_frame fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.rgb = rgb, fr.depth = depth, fr.timestamp = timestamp;
    fr.weight_multiplier = weight_multiplier;
    fr.init_transforms = init_transform, fr.n_init_transforms = 1, fr.icp_refine = icp_refine;
    return fusion_process_frame_impl(f, &fr);
}

// ---- processFrame(const FrameData&) with the frame still in host memory ------------------------------------------
// The reference uploads FrameData::rgb / depth to GL textures at :221, :261.  Here rgb, depth (and the optional id image)
// go through pinned staging buffers into device copies, three deep, on a stream of their own.
static int fusion_up_init(mmf_fusion* f) {
    if (f->up_pin[0]) return MMF_OK;
    const size_t total = (size_t)f->width * f->height * 8;  // 8 B/px: depth f32, rgb u8 x 3, id u8
    for (int i = 0; i < mmf_fusion::kUp; ++i) {
        MMF_HIP_TRY(hipHostMalloc((void**)&f->up_pin[i], total, hipHostMallocDefault));
        MMF_HIP_TRY(hipMalloc((void**)&f->up_dev[i], total));
        MMF_HIP_TRY(hipEventCreateWithFlags(&f->ev_up[i], hipEventDisableTiming));
        MMF_HIP_TRY(hipEventCreateWithFlags(&f->ev_up_rgb[i], hipEventDisableTiming));
    }
    MMF_HIP_TRY(hipStreamCreateWithFlags(&f->up_stream, hipStreamNonBlocking));
    MMF_HIP_TRY(hipEventCreateWithFlags(&f->ev_up_begin, hipEventDisableTiming));
    return MMF_OK;
}
// host -> pinned -> device of rgb, then depth, into `slot`, on the upload stream; ev_up_rgb[slot] / ev_up[slot] mark the
// arrivals.  The colour image goes first: the next frame's image side (intensity pyramid, gradients, SO3) needs only that
// and runs beside the chain, the depth side is enqueued behind the pose.  after_begin: see mmf_fusion::host_caught_up.
static int fusion_up_stage(mmf_fusion* f, int slot, const uint8_t* rgb_host, const float* depth_host, bool after_begin,
                           mmf_fusion::Stager* announce) {
    const size_t npix = (size_t)f->width * f->height;
    // the staging buffer's previous upload must have left it (round-2 advisor finding: nothing made sure of that)
    if (f->up_recorded[slot]) MMF_HIP_TRY(hipEventSynchronize(f->ev_up[slot]));
    std::memcpy(f->up_pin[slot] + npix * 4, rgb_host, npix * 3);
    // the device copy's last readers are frames enqueued before this call (ev_up_begin: recorded when the call began --
    // NOT now: by now this frame's whole tracking chain sits on the fusion's stream and the upload is meant to overlap it)
    if (after_begin) MMF_HIP_TRY(hipStreamWaitEvent(f->up_stream, f->ev_up_begin, 0));
    MMF_HIP_TRY(hipMemcpyAsync(f->up_dev[slot] + npix * 4, f->up_pin[slot] + npix * 4, npix * 3, hipMemcpyHostToDevice, f->up_stream));
    MMF_HIP_TRY(hipEventRecord(f->ev_up_rgb[slot], f->up_stream));
    if (announce) {
        {
            std::lock_guard<std::mutex> lock(announce->mu);
            announce->rgb_done = true;
        }
        announce->cv.notify_all();
    }
    std::memcpy(f->up_pin[slot], depth_host, npix * 4);
    MMF_HIP_TRY(hipMemcpyAsync(f->up_dev[slot], f->up_pin[slot], npix * 4, hipMemcpyHostToDevice, f->up_stream));
    MMF_HIP_TRY(hipEventRecord(f->ev_up[slot], f->up_stream));
    f->up_recorded[slot] = true;
    return MMF_OK;
}
// the staging thread: one job at a time (mmf_fusion::Stager)
static void fusion_stager_main(mmf_fusion* f) {
    mmf_fusion::Stager& s = f->stager;
    (void)hipSetDevice(f->ctx->device);
    std::unique_lock<std::mutex> lock(s.mu);
    for (;;) {
        s.cv.wait(lock, [&] { return s.stop || s.busy; });
        if (s.stop) return;
        const int slot = s.slot;
        const uint8_t* rgb = s.rgb;
        const float* depth = s.depth;
        const bool after_begin = s.after_begin;
        lock.unlock();
        const int rc = fusion_up_stage(f, slot, rgb, depth, after_begin, &s);
        const std::string err = rc ? std::string(mmf_last_error()) : std::string();  // (the error text is per thread)
        lock.lock();
        s.rc = rc, s.error = err, s.busy = false, s.rgb_done = true;
        s.cv.notify_all();
    }
}
// hands the announced frame to the staging thread (started on first use)
static int fusion_stage_host_begin(mmf_fusion* f, bool after_begin) {
    mmf_fusion::HostNext& hn = f->host_next;
    mmf_fusion::Stager& s = f->stager;
    if (!s.thread.joinable()) s.thread = std::thread(fusion_stager_main, f);
    {
        std::lock_guard<std::mutex> lock(s.mu);
        s.slot = hn.slot, s.rgb = hn.rgb, s.depth = hn.depth, s.busy = true, s.rgb_done = false, s.after_begin = after_begin;
    }
    s.cv.notify_all();
    hn.pending = true;
    return MMF_OK;
}
// the hinted next frame's upload has been enqueued when this returns (ev_up[slot] recorded): called before anything is told
// to wait for that event -- inside processFrame where the host would only wait for the pose, in the prefetch, or at the
// latest when the call returns
static int fusion_stage_host_next(mmf_fusion* f) {
    mmf_fusion::HostNext& hn = f->host_next;
    if (!hn.pending) return MMF_OK;
    hn.pending = false;
    mmf_fusion::Stager& s = f->stager;
    std::unique_lock<std::mutex> lock(s.mu);
    s.cv.wait(lock, [&] { return !s.busy; });
    if (s.rc) return fail(s.rc, s.error);
    hn.staged = hn.rgb_staged = true;
    return MMF_OK;
}
// ... only its colour image's (ev_up_rgb[slot] recorded): what the image side enqueued beside the chain waits for
static int fusion_stage_host_rgb(mmf_fusion* f) {
    mmf_fusion::HostNext& hn = f->host_next;
    if (!hn.pending || hn.rgb_staged) return MMF_OK;
    mmf_fusion::Stager& s = f->stager;
    std::unique_lock<std::mutex> lock(s.mu);
    s.cv.wait(lock, [&] { return s.rgb_done; });
    if (!s.busy && s.rc) return MMF_OK;  // (the join reports it)
    hn.rgb_staged = true;
    return MMF_OK;
}

// next_rgb_host / next_depth_host (both or neither): the frame the NEXT call will be given (same pointers, contents
// unchanged until then) -- the host-memory form of mmf_frame::next_*: it is staged and uploaded during this call and its
// sensor-side preparation overlaps this frame's fusion, so the next call starts with its frame on the device.
extern "C" int mmf_fusion_process_frame_host_next(mmf_fusion* f, const uint8_t* rgb_host, const float* depth_host,
                                                  const uint8_t* mask_host, int has_new_label, long long timestamp,
                                                  const float* in_pose, float weight_multiplier, int bootstrap,
                                                  const uint8_t* next_rgb_host, const float* next_depth_host) {
    MMF_REQUIRE(f != nullptr, "mmf_fusion_process_frame_host: null fusion object");
    if (!rgb_host || !depth_host || timestamp < 0) return fail(MMF_ERR_INVALID, "invalid image data");  // :209-212
    mmf_ctx* c = f->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    if (int rc = fusion_up_init(f)) return rc;
    const size_t npix = (size_t)f->width * f->height;
    const size_t o_depth = 0, o_rgb = npix * 4, o_mask = npix * 7;
    mmf_fusion::HostNext& hn = f->host_next;
    if (int rc = fusion_stage_host_next(f)) return rc;  // (nothing of an earlier announcement is still being staged)
    // what a refilled slot's upload has to wait for: the work enqueued before this call -- nothing, when the last call
    // ended with the host holding a pose from the fusion's stream (every lane joins that stream at the end of a frame)
    const bool host_knows = !tunables().host_up_events;
    const bool after_begin = !(host_knows && f->host_caught_up);
    if (after_begin) MMF_HIP_TRY(hipEventRecord(f->ev_up_begin, c->stream));
    int slot;
    if (hn.staged && hn.rgb == rgb_host && hn.depth == depth_host) {  // announced by the previous call: already on its way
        slot = hn.slot;
    } else {
        if (hn.staged || hn.pending) f->pre_rgb = nullptr, f->pre_depth = nullptr;  // what was prepared belongs to a frame that never came
        slot = f->up_cur;                                                          // (its device buffer is about to be reused)
        if (int rc = fusion_up_stage(f, slot, rgb_host, depth_host, after_begin, nullptr)) return rc;
    }
    hn = mmf_fusion::HostNext();
    f->up_cur = (slot + 1) % mmf_fusion::kUp;
    // (an announced frame arrived during the last call as a rule: then the fusion's stream gets no barrier packet)
    if (host_knows)
        MMF_HIP_TRY(fusion_wait_unless_done(c->stream, f->ev_up[slot]));
    else
        MMF_HIP_TRY(hipStreamWaitEvent(c->stream, f->ev_up[slot], 0));
    mmf_frame fr;
    std::memset(&fr, 0, sizeof(fr));
    fr.rgb = f->up_dev[slot] + o_rgb, fr.depth = (const float*)(f->up_dev[slot] + o_depth), fr.timestamp = timestamp;
    fr.in_pose = in_pose, fr.weight_multiplier = weight_multiplier, fr.bootstrap = bootstrap;
    fr.icp_refine = 1;
    mmf_segmentation seg;
    std::memset(&seg, 0, sizeof(seg));
    if (mask_host) {  // the id image is this frame's segmentation result: it cannot be announced a frame ahead
        std::memcpy(f->up_pin[slot] + o_mask, mask_host, npix);
        MMF_HIP_TRY(hipMemcpyAsync(f->up_dev[slot] + o_mask, f->up_pin[slot] + o_mask, npix, hipMemcpyHostToDevice, c->stream));
        seg.mask = f->up_dev[slot] + o_mask, seg.has_new_label = has_new_label;
        fr.segmentation = &seg;
    }
    if (next_rgb_host && next_depth_host) {
        hn.rgb = next_rgb_host, hn.depth = next_depth_host, hn.slot = f->up_cur;
        fr.next_rgb = f->up_dev[hn.slot] + o_rgb, fr.next_depth = (const float*)(f->up_dev[hn.slot] + o_depth);
        if (int rc = fusion_stage_host_begin(f, after_begin)) return rc;  // staged and sent up beside this call's own work
    }
    int rc = fusion_process_frame_impl(f, &fr);
    if (rc) {  // the staging thread may still be reading the caller's next_* buffers: not past this call's return
        const std::string why = mmf_last_error();
        (void)fusion_stage_host_next(f);
        return fail(rc, why);
    }
    return fusion_stage_host_next(f);  // (paths that enqueue no prefetch never reached the staging point)
}

extern "C" int mmf_fusion_process_frame_host(mmf_fusion* f, const uint8_t* rgb_host, const float* depth_host,
                                             const uint8_t* mask_host, int has_new_label, long long timestamp,
                                             const float* in_pose, float weight_multiplier, int bootstrap) {
    return mmf_fusion_process_frame_host_next(f, rgb_host, depth_host, mask_host, has_new_label, timestamp, in_pose,
                                              weight_multiplier, bootstrap, nullptr, nullptr);
}

// MultiMotionFusion::predict (:863-875) as a public call (the GUI re-predicts when a view changes)
extern "C" int mmf_fusion_predict(mmf_fusion* f) {
    MMF_REQUIRE(f != nullptr, "mmf_fusion_predict: null fusion object");
    MMF_REQUIRE(f->frame_rgb != nullptr, "mmf_fusion_predict: no frame has been processed yet");
    MMF_HIP_TRY(hipSetDevice(f->ctx->device));
    for (size_t k = 0; k < f->models.size(); ++k) {
        FusionModel* fm = f->models[k];
        if (!fusion_owns(f, k)) continue;
        int rc = fusion_predict_model(f, fm);
        if (rc) return rc;
        if (k > 0) {
            MMF_HIP_TRY(hipEventRecord(fm->ev_done, fm->lane->stream));
            MMF_HIP_TRY(hipStreamWaitEvent(f->ctx->stream, fm->ev_done, 0));
        }
    }
    return MMF_OK;
}

// The filter and the input-side preparation (vertex / normal maps, depth and intensity pyramids, gradients) of the
// NEXT frame, enqueued on a second stream so that they run while the current frame is still being fused.
// rgb / depth must stay unchanged until the mmf_fusion_process_frame call that consumes them (same pointers).
// `tick_at_use`: the tick of the processFrame call these buffers are for
static int fusion_prefetch_init(mmf_fusion* f) {
    if (f->side != nullptr) return MMF_OK;
    MMF_HIP_TRY(hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking));
    MMF_HIP_TRY(hipStreamCreateWithFlags(&f->side2, hipStreamNonBlocking));
    MMF_HIP_TRY(hipEventCreateWithFlags(&f->ev_prefetch_done, hipEventDisableTiming));
    MMF_HIP_TRY(hipEventCreateWithFlags(&f->ev_prefetch2_done, hipEventDisableTiming));
    MMF_HIP_TRY(hipMalloc(&f->side_partials, sizeof(float) * kMaxGrid * kPartialStride));
    MMF_HIP_TRY(hipMalloc(&f->side_ticket, sizeof(unsigned) * kTicketWords));
    MMF_HIP_TRY(hipMemsetAsync(f->side_ticket, 0, sizeof(unsigned) * kTicketWords, f->side2));
    MMF_HIP_TRY(hipMalloc(&f->so3_stage[0], 2 * sizeof(OdomState)));
    f->so3_stage[1] = f->so3_stage[0] + 1;
    MMF_HIP_TRY(hipMemsetAsync(f->so3_stage[0], 0, 2 * sizeof(OdomState), f->side2));
    return MMF_OK;
}

// The image side of the frame the call with tick `tick_at_use` will be given: intensity pyramid + gradients into the
// free halves of their double buffers, then the SO3 pre-alignment (this frame's against the last frame's level-2 image:
// no model, no pose) in the staging state of that tick's parity.  Second side stream; needs the streams to exist.
// ahead: enqueued at the start of a frame whose own image side was prepared on this same stream (nothing to wait for);
// else behind ev_inputs_free where a frame recorded it (the first frame's intensity pyramid and an untracked frame's
// preparation are written on the fusion's stream; after a tracked frame the host has the poses: everything is done)
static int fusion_prefetch_image(mmf_fusion* f, const uint8_t* rgb, int tick_at_use, bool ahead) {
    const mmf_fusion_config& g = f->cfg;
    mmf_odom* odom = f->models[0]->odom;
    float identity[16];
    identity16(identity);
    hipStream_t img_stream = f->side2;
    if (!ahead && f->inputs_free_recorded) MMF_HIP_TRY(hipStreamWaitEvent(img_stream, f->ev_inputs_free, 0));
    Enqueuer qi(img_stream);
    int rc = odom_prepare_batched(odom, nullptr, g.max_depth_processed, rgb, 3, nullptr, nullptr, nullptr, 4, identity, nullptr,
                                  nullptr, nullptr, nullptr, PREP_INPUT_IMAGE, img_stream, &qi);
    if (rc) return rc;
    f->so3_stage_ready = -1;
    if (g.so3 && tick_at_use > 1) {  // a model exists: the frame will be tracked, SO3 first
        const int s = tick_at_use & 1;
        rc = odom_prefetch_so3(odom, qi, f->side_partials, f->side_ticket, f->so3_stage[s]);
        if (rc) return rc;
        f->so3_stage_ready = s;
    }
    MMF_HIP_TRY(qi.flush());
    MMF_HIP_TRY(hipEventRecord(f->ev_prefetch2_done, img_stream));
    f->image_pre_rgb = rgb;
    return MMF_OK;
}

static int fusion_prefetch_impl(mmf_fusion* f, const uint8_t* rgb, const float* depth, int tick_at_use) {
    mmf_ctx* c = f->ctx;
    MMF_HIP_TRY(hipSetDevice(c->device));
    mmf_odom* odom = f->models[0]->odom;
    if (int rc0 = fusion_prefetch_init(f)) return rc0;
    f->pre_valid = false;  // an earlier prefetch is simply overwritten: same streams, same order
    if (f->host_next.slot >= 0 && f->up_dev[0] && rgb == f->up_dev[f->host_next.slot] + (size_t)f->width * f->height * 4) {
        // the hinted frame comes from host memory (mmf_fusion_process_frame_host_next): behind its upload
        if (int rc0 = fusion_stage_host_next(f)) return rc0;
        if (hipEventQuery(f->ev_up[f->host_next.slot]) != hipSuccess) {  // (as a rule it arrived long ago: no barrier packets)
            (void)hipGetLastError();
            MMF_HIP_TRY(hipStreamWaitEvent(f->side, f->ev_up[f->host_next.slot], 0));
            MMF_HIP_TRY(hipStreamWaitEvent(f->side2, f->ev_up[f->host_next.slot], 0));
        }
    }
    if (f->inputs_free_recorded) MMF_HIP_TRY(hipStreamWaitEvent(f->side, f->ev_inputs_free, 0));
    const mmf_fusion_config& g = f->cfg;
    float identity[16];
    identity16(identity);
    // depth chain (first side stream): filter, depth pyramid, vertex and normal maps -- what the chains' ICP term reads,
    // hence behind ev_inputs_free.  Enqueued before the image chain when that is still to come: it is five launches that
    // start with the 40 us filter, the image chain fifteen short ones.
    float* target = f->filtered[1 - f->cur];
    Enqueuer qd(f->side);
    int rc = filter_depth_on(c, qd, depth, f->width, f->height, g.depth_cutoff, target);
    if (rc) return rc;
    rc = odom_prepare_batched(odom, target, g.max_depth_processed, rgb, 3, nullptr, nullptr, nullptr, 4, identity, nullptr,
                              nullptr, nullptr, nullptr, PREP_INPUT_DEPTH, f->side, &qd);
    if (rc) return rc;
    MMF_HIP_TRY(qd.flush());
    if (f->image_pre_rgb != rgb) {  // (else: enqueued at the start of the frame)
        rc = fusion_prefetch_image(f, rgb, tick_at_use, false);
        if (rc) return rc;
    }
    // ONE event for the consumer: the first side stream takes in the second's (one wait less on the model's stream)
    MMF_HIP_TRY(hipStreamWaitEvent(f->side, f->ev_prefetch2_done, 0));
    MMF_HIP_TRY(hipEventRecord(f->ev_prefetch_done, f->side));
    f->pre_valid = true, f->pre_rgb = rgb, f->pre_depth = depth;
    return MMF_OK;
}

extern "C" int mmf_fusion_prefetch_frame(mmf_fusion* f, const uint8_t* rgb, const float* depth) {
    MMF_REQUIRE(f && rgb && depth, "mmf_fusion_prefetch_frame: null argument");
    return fusion_prefetch_impl(f, rgb, depth, f->tick);
}

extern "C" int mmf_fusion_get_pose(mmf_fusion* f, float pose[16]) {
    MMF_REQUIRE(f && pose, "mmf_fusion_get_pose: null argument");
    return mmf_model_get_pose(f->models[0]->model, pose);
}

static int model_reset_store(mmf_model* m) {
    if (int rc = model_resolve_count(m)) return rc;  // lets an in-flight count land before it is dropped
    m->count = 0, m->count_pending = false, m->count_bound = 0;
    if (m->slab) MMF_HIP_TRY(hipMemsetAsync(m->totals, 0, 16, m->ctx->stream));  // (no slab: another rank's model)
    identity16(m->pose);
    m->max_depth = FLT_MAX;
    return MMF_OK;
}

// start a new map: empty surfel stores, identity poses, tick = 1, only the global model active (what
// constructing a fresh MultiMotionFusion does, MultiMotionFusion.cpp:21-97); object models return to the
// preallocated pool
extern "C" int mmf_fusion_reset(mmf_fusion* f) {
    MMF_REQUIRE(f != nullptr, "mmf_fusion_reset: null fusion object");
    MMF_HIP_TRY(hipSetDevice(f->ctx->device));
    if (f->pre_valid) {  // a prefetched frame belongs to the sequence that ends here
        MMF_HIP_TRY(hipStreamWaitEvent(f->ctx->stream, f->ev_prefetch_done, 0));
        MMF_HIP_TRY(hipStreamWaitEvent(f->ctx->stream, f->ev_prefetch2_done, 0));
        f->pre_valid = false;
    }
    for (auto* list : {&f->models, &f->inactive})
        for (size_t k = (list == &f->models ? 1 : 0); k < list->size(); ++k) f->preallocated.push_back((*list)[k]);
    f->models.resize(1);
    f->inactive.clear();
    f->scheduled_deactivation.clear();
    std::vector<FusionModel*> all(f->preallocated);
    all.push_back(f->models[0]);
    for (FusionModel* fm : all) {
        int rc = model_reset_store(fm->model);
        if (rc) return rc;
        fm->model->conf_threshold = fm->model->id == 0 ? f->cfg.conf_global_init : f->cfg.conf_object_init;
        identity16(fm->last_pose);
        fm->odom->so3_prefetched = false, fm->odom->so3_stage = nullptr;
        fm->odom->have_tmp = false;
        fm->unseen = 0;
        fm->pose_log.clear();
    }
    for (FusionModel* fm : all) fm->spec_valid = false;
    f->so3_stage_ready = -1, f->image_pre_rgb = nullptr;
    f->tick = 1;
    return MMF_OK;
}

// MultiMotionFusion::exportPoses (:1020-1045): one file per pose-logging model, `poses-<id>.txt` in export_dir,
// lines `ts x y z qx qy qz qw` (operator<< formatting: 6 significant digits)
extern "C" int mmf_fusion_export_poses(mmf_fusion* f, const char* export_dir) {
    MMF_REQUIRE(f && export_dir, "mmf_fusion_export_poses: null argument");
    for (auto* list : {&f->models, &f->inactive})
        for (FusionModel* fm : *list) {
            if (fm->pose_log.capacity() == 0) continue;
            const std::string name = std::string(export_dir) + "poses-" + std::to_string((int)fm->model->id) + ".txt";
            FILE* fp = std::fopen(name.c_str(), "w");
            if (!fp) return fail(MMF_ERR_INVALID, "mmf_fusion_export_poses: cannot open " + name);
            for (const PoseLogItem& it : fm->pose_log) {
                std::fprintf(fp, "%lld", it.ts);
                for (int i = 0; i < 7; ++i) std::fprintf(fp, " %g", (double)it.p[i]);
                std::fprintf(fp, "\n");
            }
            std::fclose(fp);
        }
    return MMF_OK;
}

// the pose log of one active model (Model::getPoseLog): n entries of {ts, x y z qx qy qz qw}
extern "C" int mmf_fusion_pose_log(mmf_fusion* f, int index, long long* ts, float* p7, int max_entries, int* n_out) {
    MMF_REQUIRE(f && n_out && index >= 0 && index < (int)f->models.size(), "mmf_fusion_pose_log: bad argument");
    const std::vector<PoseLogItem>& log = f->models[index]->pose_log;
    *n_out = (int)log.size();
    for (int i = 0; i < (int)log.size() && i < max_entries; ++i) {
        if (ts) ts[i] = log[i].ts;
        if (p7) std::memcpy(p7 + 7 * i, log[i].p, sizeof(float) * 7);
    }
    return MMF_OK;
}
