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

struct mmf_shard {
    mmf_ctx* ctx = nullptr;
    const RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    int rank = 0, world = 1;
    float* send_dev = nullptr;  // slots * 18
    float* recv_dev = nullptr;  // world * slots * 18
    float* host = nullptr;      // pinned: send then recv
    int slots = 0;
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

static int shard_buffers(mmf_shard* s, int slots) {
    if (slots <= s->slots) return MMF_OK;
    (void)hipFree(s->send_dev), (void)hipFree(s->recv_dev);
    if (s->host) (void)hipHostFree(s->host);
    s->send_dev = s->recv_dev = s->host = nullptr;
    s->slots = 0;
    MMF_HIP_TRY(hipMalloc(&s->send_dev, sizeof(float) * kShardRecord * slots));
    MMF_HIP_TRY(hipMalloc(&s->recv_dev, sizeof(float) * kShardRecord * slots * s->world));
    MMF_HIP_TRY(hipHostMalloc(&s->host, sizeof(float) * kShardRecord * slots * s->world, hipHostMallocDefault));
    s->slots = slots;
    return MMF_OK;
}

static int shard_make(mmf_ctx* c, int rank, int world, ncclComm_t comm, bool own, mmf_shard** out) {
    mmf_shard* s = new (std::nothrow) mmf_shard();
    MMF_REQUIRE(s != nullptr, "mmf_shard: out of host memory");
    s->ctx = c, s->api = rccl_api(), s->comm = comm, s->own_comm = own, s->rank = rank, s->world = world;
    *out = s;
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
    if (s->own_comm && s->comm) (void)s->api->CommDestroy(s->comm);
    (void)hipFree(s->send_dev), (void)hipFree(s->recv_dev);
    if (s->host) (void)hipHostFree(s->host);
    delete s;
}

// Step 1 of a sharded frame: the root's RGB (u8 x 3), depth (f32) and id image (u8) reach the same-sized device
// buffers of every rank: three broadcasts in one group on the context's stream (asynchronous, stream ordered).
// mask may be NULL (static scene).
extern "C" int mmf_shard_broadcast_frame(mmf_shard* s, uint8_t* rgb, float* depth, uint8_t* mask, int width, int height, int root) {
    MMF_REQUIRE(s && rgb && depth && width > 0 && height > 0 && root >= 0 && root < s->world, "mmf_shard_broadcast_frame: bad argument");
    MMF_HIP_TRY(hipSetDevice(s->ctx->device));
    const size_t npix = (size_t)width * height;
    hipStream_t st = s->ctx->stream;
    MMF_RCCL_TRY(s->api, s->api->GroupStart());
    ncclResult_t r0 = s->api->Broadcast(rgb, rgb, npix * 3, ncclUint8, root, s->comm, st);
    ncclResult_t r1 = s->api->Broadcast(depth, depth, npix, ncclFloat32, root, s->comm, st);
    ncclResult_t r2 = mask ? s->api->Broadcast(mask, mask, npix, ncclUint8, root, s->comm, st) : ncclSuccess;
    MMF_RCCL_TRY(s->api, s->api->GroupEnd());
    MMF_RCCL_TRY(s->api, r0);
    MMF_RCCL_TRY(s->api, r1);
    MMF_RCCL_TRY(s->api, r2);
    return MMF_OK;
}

// Step 3: every rank contributes {pose, lastICPError, lastICPCount} of the models it owns (list index k lives in slot
// k / world of rank k % world), one all-gather, and the poses of the models other ranks own are written into this
// rank's bookkeeping (mmf_fusion_set_model_pose).  Synchronises the context's stream (poses are host state).
extern "C" int mmf_shard_gather_poses(mmf_shard* s, mmf_fusion* f) {
    MMF_REQUIRE(s && f, "mmf_shard_gather_poses: null argument");
    MMF_REQUIRE(f->shard_world == s->world && f->shard_rank == s->rank, "mmf_shard_gather_poses: the fusion object is not sharded "
                                                                         "like this communicator (mmf_fusion_set_shard)");
    MMF_HIP_TRY(hipSetDevice(s->ctx->device));
    const int n_models = (int)f->models.size();
    const int slots = (n_models + s->world - 1) / s->world;
    if (slots == 0) return MMF_OK;
    int rc = shard_buffers(s, slots);
    if (rc) return rc;
    hipStream_t st = s->ctx->stream;
    float* send = s->host;  // the first slots * 18 floats of the pinned buffer
    std::memset(send, 0, sizeof(float) * kShardRecord * slots);
    for (int j = 0; j < slots; ++j) {
        const int k = s->rank + j * s->world;
        if (k >= n_models) continue;
        std::memcpy(send + j * kShardRecord, f->models[k]->model->pose, sizeof(float) * 16);
        send[j * kShardRecord + 16] = f->models[k]->odom->stats.lastICPError;
        send[j * kShardRecord + 17] = f->models[k]->odom->stats.lastICPCount;
    }
    MMF_HIP_TRY(hipMemcpyAsync(s->send_dev, send, sizeof(float) * kShardRecord * slots, hipMemcpyHostToDevice, st));
    MMF_RCCL_TRY(s->api, s->api->AllGather(s->send_dev, s->recv_dev, (size_t)kShardRecord * slots, ncclFloat32, s->comm, st));
    MMF_HIP_TRY(hipMemcpyAsync(s->host, s->recv_dev, sizeof(float) * kShardRecord * slots * s->world, hipMemcpyDeviceToHost, st));
    MMF_HIP_TRY(hipStreamSynchronize(st));
    for (int r = 0; r < s->world; ++r) {
        if (r == s->rank) continue;
        for (int j = 0; j < slots; ++j) {
            const int k = r + j * s->world;
            if (k >= n_models) continue;
            const float* rec = s->host + ((size_t)r * slots + j) * kShardRecord;
            std::memcpy(f->models[k]->model->pose, rec, sizeof(float) * 16);
            std::memcpy(f->models[k]->last_pose, rec, sizeof(float) * 16);
            f->models[k]->odom->stats.lastICPError = rec[16];
            f->models[k]->odom->stats.lastICPCount = rec[17];
        }
    }
    return MMF_OK;
}
