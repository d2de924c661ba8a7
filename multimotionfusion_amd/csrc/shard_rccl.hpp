// shard_rccl.hpp -- the per-rigid-body shard's two exchanges below Python, for a C++ front-end that runs one
// process per GPU: rank 0's frame to every rank (ncclBroadcast, 8 B/px) and every rank's model poses to every rank
// (ncclAllGather, 18 floats per model slot), on the CONTEXT'S stream.  No J^T J crosses GPUs.  Textually included by
// mmf_hip.hip after the orchestrator.
//
// RCCL is bound at run time (dlopen of librccl.so.1 -- the copy already in the process when there is one, e.g. the one
// a PyTorch front-end brought), so libmmf_hip.so carries no link-time dependency on it and single-GPU users never load
// it.  A communicator handed in through mmf_shard_attach must come from that same copy.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static const RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) return nullptr;
    bool ok = true;
    auto sym = [&](const char* n) {
        void* p = dlsym(api.handle, n);
        ok = ok && p != nullptr;
        return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
        dlclose(api.handle);
        api.handle = nullptr;
        return nullptr;
    }
    return &api;
}

#define MMF_RCCL_TRY(api, expr)                                                                              \
    do {                                                                                                     \
        ncclResult_t r__ = (expr);                                                                           \
        if (r__ != ncclSuccess) return fail(MMF_ERR_HIP, std::string(#expr) + ": " + (api)->GetErrorString(r__)); \
    } while (0)

constexpr int kShardRecord = 18;  // 16 pose floats + lastICPError + lastICPCount (72 B per model)
constexpr int kShardFrameSlots = 8;  // frame broadcasts that can be posted ahead (mmf_shard_post_frame)
constexpr int kShardRing = 3;     // pose gathers that may be in flight (a front-end consumes them one or two frames later)

// One pose all-gather in flight: its own pinned send / receive buffers and device buffers, the event behind its last copy,
// and who was in which slot when it was enqueued (model ids: the list may change before the result is applied).
struct ShardGather {
    float *send_host = nullptr, *recv_host = nullptr;  // pinned
    float *send_dev = nullptr, *recv_dev = nullptr;
    hipEvent_t done = nullptr;
    int slots = 0;  // capacity
    int used_slots = 0;
    std::vector<int> ids;  // [rank * used_slots + j] -> model id, -1 = empty slot
    bool pending = false;
};

struct mmf_shard {
    mmf_ctx* ctx = nullptr;
    const RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 1;
    // Every collective of the shard runs on ONE stream of its own, in program order (one communicator, one stream: the
    // order RCCL sees is the same on every rank), next to the context's stream: the broadcast of a later frame and the
    // pose gather overlap the frame being processed.  Events tie the two streams together where data crosses.
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ctx = nullptr;                   // context stream -> comm stream
    hipEvent_t ev_frame[kShardFrameSlots] = {};   // comm stream: the broadcast posted into slot k is complete
    hipEvent_t ev_maps = nullptr;                  // comm stream: the map gather is complete
    ShardGather ring[kShardRing];
    unsigned long long begun = 0, ended = 0;  // gathers enqueued / applied
    float *maps_send = nullptr, *maps_recv = nullptr;  // step 3: per-model super-pixel maps
    size_t maps_floats = 0;                            // capacity of maps_send (maps_recv: x world)
};

extern "C" int mmf_shard_unique_id(char id[128]) {
    MMF_REQUIRE(id != nullptr, "mmf_shard_unique_id: null argument");
    const RcclApi* api = rccl_api();
    MMF_REQUIRE(api != nullptr, "mmf_shard: librccl.so.1 could not be loaded");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId uid;
    MMF_RCCL_TRY(api, api->GetUniqueId(&uid));
    std::memcpy(id, &uid, sizeof(uid));
    return MMF_OK;
}

static void shard_gather_free(ShardGather& g) {
    (void)hipFree(g.send_dev), (void)hipFree(g.recv_dev);
    if (g.send_host) (void)hipHostFree(g.send_host);
    if (g.recv_host) (void)hipHostFree(g.recv_host);
    if (g.done) (void)hipEventDestroy(g.done);
    g = ShardGather();
}
static int shard_gather_buffers(mmf_shard* s, ShardGather& g, int slots) {
    if (slots <= g.slots) return MMF_OK;
    hipEvent_t keep = g.done;
    g.done = nullptr;
    shard_gather_free(g);
    g.done = keep;
    if (!g.done) MMF_HIP_TRY(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    MMF_HIP_TRY(hipMalloc(&g.send_dev, sizeof(float) * kShardRecord * slots));
    MMF_HIP_TRY(hipMalloc(&g.recv_dev, sizeof(float) * kShardRecord * slots * s->world));
    MMF_HIP_TRY(hipHostMalloc(&g.send_host, sizeof(float) * kShardRecord * slots, hipHostMallocDefault));
    MMF_HIP_TRY(hipHostMalloc(&g.recv_host, sizeof(float) * kShardRecord * slots * s->world, hipHostMallocDefault));
    g.slots = slots;
    return MMF_OK;
}

// Who sends what: rank r's j-th slot carries the j-th model of the active list that r owns (fusion_owner_of: by model id).
// Every rank keeps the whole list as bookkeeping, so every rank derives the same table.  ids[r * slots + j] = list index
// or -1; returns the slots per rank (>= 1).
static int shard_slot_table(const mmf_fusion* f, int world, std::vector<int>& index) {
    std::vector<int> count((size_t)world, 0);
    for (const FusionModel* fm : f->models) ++count[(size_t)fusion_owner_of(f, fm)];
    int slots = 1;
    for (int c : count) slots = c > slots ? c : slots;
    index.assign((size_t)world * slots, -1);
    std::fill(count.begin(), count.end(), 0);
    for (size_t k = 0; k < f->models.size(); ++k) {
        const int r = fusion_owner_of(f, f->models[k]);
        index[(size_t)r * slots + count[(size_t)r]++] = (int)k;
    }
    return slots;
}

static int shard_make(mmf_ctx* c, int rank, int world, ncclComm_t comm, bool own, mmf_shard** out) {
    mmf_shard* s = new (std::nothrow) mmf_shard();
    MMF_REQUIRE(s != nullptr, "mmf_shard: out of host memory");
    s->ctx = c, s->api = rccl_api(), s->comm = comm, s->own_comm = own, s->rank = rank, s->world = world;
    *out = s;
    MMF_HIP_TRY(hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking));
    MMF_HIP_TRY(hipEventCreateWithFlags(&s->ev_ctx, hipEventDisableTiming));
    MMF_HIP_TRY(hipEventCreateWithFlags(&s->ev_maps, hipEventDisableTiming));
    for (hipEvent_t& e : s->ev_frame) MMF_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return MMF_OK;
}

// one communicator per process: ncclCommInitRank on the context's device with the id rank 0 made
extern "C" int mmf_shard_create(mmf_ctx* c, int rank, int world, const char id[128], mmf_shard** out) {
    MMF_REQUIRE(c && id && out && world >= 1 && rank >= 0 && rank < world, "mmf_shard_create: bad argument");
    const RcclApi* api = rccl_api();
    MMF_REQUIRE(api != nullptr, "mmf_shard: librccl.so.1 could not be loaded");
    MMF_HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    MMF_RCCL_TRY(api, api->CommInitRank(&comm, world, uid, rank));
    return shard_make(c, rank, world, comm, true, out);
}

// the caller's ncclComm_t (created with the RCCL copy this process has loaded); not destroyed by mmf_shard_destroy
extern "C" int mmf_shard_attach(mmf_ctx* c, int rank, int world, void* nccl_comm, mmf_shard** out) {
    MMF_REQUIRE(c && nccl_comm && out && world >= 1 && rank >= 0 && rank < world, "mmf_shard_attach: bad argument");
    MMF_REQUIRE(rccl_api() != nullptr, "mmf_shard: librccl.so.1 could not be loaded");
    return shard_make(c, rank, world, static_cast<ncclComm_t>(nccl_comm), false, out);
}

extern "C" void mmf_shard_destroy(mmf_shard* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
    if (s->own_comm && s->comm) (void)s->api->CommDestroy(s->comm);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    if (s->ev_ctx) (void)hipEventDestroy(s->ev_ctx);
    if (s->ev_maps) (void)hipEventDestroy(s->ev_maps);
    for (hipEvent_t e : s->ev_frame)
        if (e) (void)hipEventDestroy(e);
    for (ShardGather& g : s->ring) shard_gather_free(g);
    (void)hipFree(s->maps_send), (void)hipFree(s->maps_recv);
    delete s;
}

// the comm stream continues where the context's stream is now (what the caller has enqueued so far is visible to it)
static int shard_comm_after_ctx(mmf_shard* s) {
    MMF_HIP_TRY(hipEventRecord(s->ev_ctx, s->ctx->stream));
    MMF_HIP_TRY(hipStreamWaitEvent(s->comm_stream, s->ev_ctx, 0));
    return MMF_OK;
}

// Step 1 of a sharded frame: the root's RGB (u8 x 3), depth (f32) and id image (u8) reach the same-sized device buffers
// of every rank: three broadcasts in one group.  mask may be NULL (static scene).
//   mmf_shard_post_frame   starts the exchange on the shard's own stream, behind everything enqueued on the context's
//                          stream so far (the root's copies into the buffers; on the other ranks the last readers of the
//                          buffers): it overlaps whatever the context's stream does next.  slot (0 .. 7) names the exchange.
//   mmf_shard_wait_frame   the context's stream waits for the exchange posted into `slot` (no host synchronisation).
//   mmf_shard_broadcast_frame = post + wait: stream ordered on the context's stream, as before.
extern "C" int mmf_shard_post_frame(mmf_shard* s, uint8_t* rgb, float* depth, uint8_t* mask, int width, int height, int root, int slot) {
    MMF_REQUIRE(s && rgb && depth && width > 0 && height > 0 && root >= 0 && root < s->world && slot >= 0 && slot < kShardFrameSlots,
                "mmf_shard_post_frame: bad argument");
    MMF_HIP_TRY(hipSetDevice(s->ctx->device));
    if (int rc = shard_comm_after_ctx(s)) return rc;
    const size_t npix = (size_t)width * height;
    hipStream_t st = s->comm_stream;
    MMF_RCCL_TRY(s->api, s->api->GroupStart());
    ncclResult_t r0 = s->api->Broadcast(rgb, rgb, npix * 3, ncclUint8, root, s->comm, st);
    ncclResult_t r1 = s->api->Broadcast(depth, depth, npix, ncclFloat32, root, s->comm, st);
    ncclResult_t r2 = mask ? s->api->Broadcast(mask, mask, npix, ncclUint8, root, s->comm, st) : ncclSuccess;
    MMF_RCCL_TRY(s->api, s->api->GroupEnd());
    MMF_RCCL_TRY(s->api, r0);
    MMF_RCCL_TRY(s->api, r1);
    MMF_RCCL_TRY(s->api, r2);
    MMF_HIP_TRY(hipEventRecord(s->ev_frame[slot], st));
    return MMF_OK;
}
extern "C" int mmf_shard_wait_frame(mmf_shard* s, int slot) {
    MMF_REQUIRE(s && slot >= 0 && slot < kShardFrameSlots, "mmf_shard_wait_frame: bad argument");
    MMF_HIP_TRY(hipSetDevice(s->ctx->device));
    MMF_HIP_TRY(hipStreamWaitEvent(s->ctx->stream, s->ev_frame[slot], 0));
    return MMF_OK;
}
extern "C" int mmf_shard_broadcast_frame(mmf_shard* s, uint8_t* rgb, float* depth, uint8_t* mask, int width, int height, int root) {
    if (int rc = mmf_shard_post_frame(s, rgb, depth, mask, width, height, root, 0)) return rc;
    return mmf_shard_wait_frame(s, 0);
}

// Step 3a: every rank contributes {pose, lastICPError, lastICPCount} of the models it owns (shard_slot_table), one
// all-gather, and the poses of the models other ranks own are written into this rank's bookkeeping (as
// mmf_fusion_set_model_pose does).  Poses are host state, so the exchange is host -> device -> all-gather -> host; it is
// split in two so that it never stalls a frame:
//   _begin  fills a pinned record from the host state processFrame has just left, enqueues copy, all-gather and copy back
//           on the shard's own stream and records an event -- no synchronisation, nothing on the context's stream;
//   _end    waits for the OLDEST gather in flight (normally long complete: a front-end calls it one or two frames later,
//           like the frames it prefetches) and applies it.  With nothing in flight it returns at once.
// Up to kShardRing gathers may be in flight; a further _begin completes the oldest first.
extern "C" int mmf_shard_gather_poses_end(mmf_shard* s, mmf_fusion* f);
extern "C" int mmf_shard_gather_poses_begin(mmf_shard* s, mmf_fusion* f) {
    MMF_REQUIRE(s && f, "mmf_shard_gather_poses_begin: null argument");
    MMF_REQUIRE(f->shard_world == s->world && f->shard_rank == s->rank, "mmf_shard_gather_poses: the fusion object is not sharded "
                                                                         "like this communicator (mmf_fusion_set_shard)");
    MMF_HIP_TRY(hipSetDevice(s->ctx->device));
    if (s->begun - s->ended >= (unsigned long long)kShardRing)
        if (int rc = mmf_shard_gather_poses_end(s, f)) return rc;
    ShardGather& g = s->ring[s->begun % kShardRing];
    std::vector<int> index;
    const int slots = shard_slot_table(f, s->world, index);
    if (int rc = shard_gather_buffers(s, g, slots)) return rc;
    g.used_slots = slots;
    g.ids.assign(index.size(), -1);
    for (size_t i = 0; i < index.size(); ++i)
        if (index[i] >= 0) g.ids[i] = (int)f->models[(size_t)index[i]]->model->id;
    std::memset(g.send_host, 0, sizeof(float) * kShardRecord * slots);
    for (int j = 0; j < slots; ++j) {
        const int k = index[(size_t)s->rank * slots + j];
        if (k < 0) continue;
        const FusionModel* fm = f->models[(size_t)k];
        std::memcpy(g.send_host + j * kShardRecord, fm->model->pose, sizeof(float) * 16);
        g.send_host[j * kShardRecord + 16] = fm->odom->stats.lastICPError;
        g.send_host[j * kShardRecord + 17] = fm->odom->stats.lastICPCount;
    }
    hipStream_t st = s->comm_stream;  // host data in, host data out: nothing of the context's stream is involved
    MMF_HIP_TRY(hipMemcpyAsync(g.send_dev, g.send_host, sizeof(float) * kShardRecord * slots, hipMemcpyHostToDevice, st));
    MMF_RCCL_TRY(s->api, s->api->AllGather(g.send_dev, g.recv_dev, (size_t)kShardRecord * slots, ncclFloat32, s->comm, st));
    MMF_HIP_TRY(hipMemcpyAsync(g.recv_host, g.recv_dev, sizeof(float) * kShardRecord * slots * s->world, hipMemcpyDeviceToHost, st));
    MMF_HIP_TRY(hipEventRecord(g.done, st));
    g.pending = true;
    ++s->begun;
    return MMF_OK;
}

extern "C" int mmf_shard_gather_poses_end(mmf_shard* s, mmf_fusion* f) {
    MMF_REQUIRE(s && f, "mmf_shard_gather_poses_end: null argument");
    if (s->ended == s->begun) return MMF_OK;
    MMF_HIP_TRY(hipSetDevice(s->ctx->device));
    ShardGather& g = s->ring[s->ended % kShardRing];
    ++s->ended;
    if (!g.pending) return MMF_OK;
    g.pending = false;
    hipError_t e = hipErrorNotReady;
    for (int i = 0; i < 200000 && e == hipErrorNotReady; ++i) e = hipEventQuery(g.done);  // poll first, park late (wait_stream)
    if (e == hipErrorNotReady) e = hipEventSynchronize(g.done);
    MMF_HIP_TRY(e);
    for (int r = 0; r < s->world; ++r) {
        if (r == s->rank) continue;
        for (int j = 0; j < g.used_slots; ++j) {
            const int id = g.ids[(size_t)r * g.used_slots + j];
            FusionModel* fm = id >= 0 ? fusion_find(f, id) : nullptr;  // the model may have left the list meanwhile
            if (!fm) continue;
            const float* rec = g.recv_host + ((size_t)r * g.used_slots + j) * kShardRecord;
            std::memcpy(fm->model->pose, rec, sizeof(float) * 16);
            std::memcpy(fm->last_pose, rec, sizeof(float) * 16);
            fm->odom->stats.lastICPError = rec[16];
            fm->odom->stats.lastICPCount = rec[17];
        }
    }
    return MMF_OK;
}

// the blocking form: enqueue, wait, apply (every gather in flight, oldest first)
extern "C" int mmf_shard_gather_poses(mmf_shard* s, mmf_fusion* f) {
    if (int rc = mmf_shard_gather_poses_begin(s, f)) return rc;
    while (s->ended != s->begun)
        if (int rc = mmf_shard_gather_poses_end(s, f)) return rc;
    return MMF_OK;
}

// Step 3b (what Segmentation.cpp:214-223 reads of EVERY model): the per-model ICP-error image and the
// confidence channel of the splat's vertex image, averaged per super-pixel on the GPU that holds them
// (mmf_slic_downsample = Slic::downsample<float>, Slic.h:48-83) and all-gathered, so that (W/S) x (H/S) floats per map
// travel instead of two full-resolution images per model.  labels: the super-pixel index image (device, int32, W x H),
// the same on every rank (it is computed from the frame every rank holds).  out_dev (device): [n_models][2][nspix] in list
// order, {icp, confidence}; filled on every rank, stream ordered on the context's stream.
extern "C" int mmf_shard_gather_maps(mmf_shard* s, mmf_fusion* f, const int* labels, int spixel_size, float* out_dev) {
    MMF_REQUIRE(s && f && labels && out_dev, "mmf_shard_gather_maps: null argument");
    MMF_REQUIRE(f->shard_world == s->world && f->shard_rank == s->rank, "mmf_shard_gather_maps: the fusion object is not sharded "
                                                                         "like this communicator (mmf_fusion_set_shard)");
    MMF_REQUIRE(f->ctx == s->ctx, "mmf_shard_gather_maps: the communicator and the fusion object use different contexts");
    MMF_REQUIRE(spixel_size > 0 && f->width / spixel_size >= 1 && f->height / spixel_size >= 1, "mmf_shard_gather_maps: bad super-pixel size");
    MMF_HIP_TRY(hipSetDevice(s->ctx->device));
    const size_t nspix = (size_t)(f->width / spixel_size) * (size_t)(f->height / spixel_size);
    std::vector<int> index;
    const int slots = shard_slot_table(f, s->world, index);
    const size_t per_rank = (size_t)slots * 2 * nspix;
    if (per_rank > s->maps_floats) {
        MMF_HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        MMF_HIP_TRY(hipStreamSynchronize(s->comm_stream));
        (void)hipFree(s->maps_send), (void)hipFree(s->maps_recv);
        s->maps_send = s->maps_recv = nullptr, s->maps_floats = 0;
        MMF_HIP_TRY(hipMalloc(&s->maps_send, sizeof(float) * per_rank));
        MMF_HIP_TRY(hipMalloc(&s->maps_recv, sizeof(float) * per_rank * s->world));
        s->maps_floats = per_rank;
    }
    hipStream_t st = s->ctx->stream;
    MMF_HIP_TRY(hipMemsetAsync(s->maps_send, 0, sizeof(float) * per_rank, st));
    for (int j = 0; j < slots; ++j) {
        const int k = index[(size_t)s->rank * slots + j];
        if (k < 0) continue;
        FusionModel* fm = f->models[(size_t)k];
        MMF_REQUIRE(fm->icp_error && fm->model, "mmf_shard_gather_maps: the model has no error image (error_recording off)");
        float* dst = s->maps_send + (size_t)j * 2 * nspix;
        // (every public call of the fusion object returns with the context's stream ordered after the models' lanes)
        int rc = mmf_slic_downsample(s->ctx, labels, f->width, f->height, spixel_size, fm->icp_error, 1, 0, 0, 0.f, dst, nullptr);
        if (rc) return rc;
        rc = mmf_slic_downsample(s->ctx, labels, f->width, f->height, spixel_size, reinterpret_cast<const float*>(fm->model->vertexConf), 4,
                                 3, 0, 0.f, dst + nspix, nullptr);
        if (rc) return rc;
    }
    if (int rc = shard_comm_after_ctx(s)) return rc;  // the averages are computed on the context's stream
    MMF_RCCL_TRY(s->api, s->api->AllGather(s->maps_send, s->maps_recv, per_rank, ncclFloat32, s->comm, s->comm_stream));
    MMF_HIP_TRY(hipEventRecord(s->ev_maps, s->comm_stream));
    MMF_HIP_TRY(hipStreamWaitEvent(st, s->ev_maps, 0));
    for (int r = 0; r < s->world; ++r)
        for (int j = 0; j < slots; ++j) {
            const int k = index[(size_t)r * slots + j];
            if (k < 0) continue;
            MMF_HIP_TRY(hipMemcpyAsync(out_dev + (size_t)k * 2 * nspix, s->maps_recv + ((size_t)r * slots + j) * 2 * nspix,
                                       sizeof(float) * 2 * nspix, hipMemcpyDeviceToDevice, st));
        }
    return MMF_OK;
}
