"""Per-rigid-body sharding across GPUs (new design -- the reference is single
GPU and iterates its Model list serially, Core/MultiMotionFusion.cpp:312,793-816).

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  Models never read each other's surfels or pyramids, so the only exchange per frame
is
  1. the frame itself: rank 0 broadcasts RGB (u8 x3), depth (f32) and mask (u8) = 8 B/px, and
     every rank builds its own pyramids (cheaper than shipping them);
  2. an all_gather of each rank's 4x4 pose + {lastICPError, lastICPCount} (72 B per rank).
There is no cross-GPU reduction of JtJ: a model's 29-float sum stays on one GPU (a ring
all-reduce of 116 bytes would be pure latency on point-to-point xGMI links).
"""
import numpy as np
import torch
import torch.distributed as dist


def model_owner(model_id: int, world_size: int) -> int:
    """the model with id m lives on rank m mod G (the global model, id 0, on rank 0) -- by ID, not by position in the list:
    a model that leaves the list moves nobody else (csrc/fusion_orchestrator.hpp: fusion_owner_of)."""
    return model_id % world_size


def slot_table(model_ids, world_size: int):
    """Who sends what in the per-model exchanges (csrc/shard_rccl.hpp: shard_slot_table): rank r's j-th slot carries the
    j-th model of the active list that r owns.  Returns (slots per rank >= 1, table [world][slots] of list indices, -1 =
    empty).  Every rank derives the same table from the same list."""
    per_rank = [[k for k, m in enumerate(model_ids) if model_owner(int(m), world_size) == r] for r in range(world_size)]
    slots = max(1, max(len(p) for p in per_rank))
    return slots, [p + [-1] * (slots - len(p)) for p in per_rank]


def gather_maps(local_maps, model_ids, nspix: int, device):
    """Step 3b of a sharded frame (Segmentation.cpp:214-223 reads these of EVERY model): the super-pixel averages
    of each model's ICP-error image and vertex confidence, computed where the model lives, reach every rank.
    local_maps: {list index: float32 tensor [2, nspix]} for the models this rank owns.  Returns a tensor
    [len(model_ids), 2, nspix] in list order -- the torch.distributed twin of mmf_shard_gather_maps."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    slots, table = slot_table(model_ids, world)
    send = torch.zeros((slots, 2, nspix), dtype=torch.float32, device=device)
    for j, k in enumerate(table[rank]):
        if k >= 0:
            send[j].copy_(local_maps[k])
    if world == 1:
        parts = [send]
    else:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
    out = torch.zeros((len(model_ids), 2, nspix), dtype=torch.float32, device=device)
    for r in range(world):
        for j, k in enumerate(table[r]):
            if k >= 0:
                out[k].copy_(parts[r][j])
    return out


def local_models(num_models: int, rank: int, world_size: int):
    return [m for m in range(num_models) if model_owner(m, world_size) == rank]


def broadcast_frame(rgb: torch.Tensor, depth: torch.Tensor, mask: torch.Tensor, src: int = 0):
    """Replicate the frame held by `src` into the (pre-allocated) tensors of every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    # three collectives on one stream; sizes are fixed per frame so RCCL reuses its plan
    dist.broadcast(rgb, src=src)
    dist.broadcast(depth, src=src)
    dist.broadcast(mask, src=src)


def broadcast_frame_async(rgb: torch.Tensor, depth: torch.Tensor, mask: torch.Tensor, src: int = 0):
    """broadcast_frame without blocking: returns the work handles; `wait()` on them orders the current
    stream after the collectives (no host synchronisation with nccl)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return []
    return [dist.broadcast(t, src=src, async_op=True) for t in (rgb, depth, mask)]


def gather_poses_async(pose: np.ndarray, icp_error: float, icp_count: float, device):
    """Non-blocking gather_poses: returns (work handle, list of per-rank [18] tensors)."""
    rec = torch.zeros(18, dtype=torch.float32)
    rec[:16] = torch.from_numpy(np.asarray(pose, np.float32).reshape(16))
    rec[16], rec[17] = float(icp_error), float(icp_count)
    rec = rec.to(device, non_blocking=True)
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    return dist.all_gather(out, rec, async_op=True), out


def gather_poses(pose: np.ndarray, icp_error: float, icp_count: float, device) -> np.ndarray:
    """all_gather of [16 pose floats, lastICPError, lastICPCount] -> array [world, 18]."""
    rec = torch.zeros(18, dtype=torch.float32, device=device)
    rec[:16] = torch.from_numpy(np.asarray(pose, np.float32).reshape(16)).to(device)
    rec[16], rec[17] = float(icp_error), float(icp_count)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return rec.cpu().numpy()[None]
    out = [torch.empty_like(rec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, rec)
    return torch.stack(out).cpu().numpy()


class PoseRing:
    """torch.distributed twin of mmf_shard_gather_poses_begin / _end (csrc/shard_rccl.hpp): the pose exchange as a ring of up to
    three all-gathers in flight.  begin() snapshots the slot table of the model list AS IT IS THEN (who owns which slot) and posts
    the gather of this rank's records; end() waits for the OLDEST exchange in flight and returns {model id: [18] record} of the
    models OTHER ranks own -- keyed by id, so a model that has left the caller's list in the meantime is simply not applied,
    and one that moved in the list is found where it is now.  A fourth begin() first completes the oldest exchange (its records
    are returned by that begin()).  Documented staleness (include/mmf_hip.h): another rank's pose is up to two frames old."""
    RING = 3

    def __init__(self, device):
        self.device = device
        self.ring = []  # oldest first: (work handle or None, per-rank tensors, slots, ids[world][slots])
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0

    def begin(self, model_ids, records):
        """model_ids: the active list's ids in list order; records: {model id: 18 floats} of the models this rank owns."""
        applied = self.end() if len(self.ring) >= self.RING else {}
        slots, table = slot_table(model_ids, self.world)
        ids = [[int(model_ids[k]) if k >= 0 else -1 for k in table[r]] for r in range(self.world)]
        send = torch.zeros((slots, 18), dtype=torch.float32)
        for j, mid in enumerate(ids[self.rank]):
            if mid >= 0:
                send[j] = torch.as_tensor(np.asarray(records[mid], np.float32).reshape(18))
        send = send.to(self.device)
        if self.world == 1:
            work, parts = None, [send]
        else:
            parts = [torch.empty_like(send) for _ in range(self.world)]
            work = dist.all_gather(parts, send, async_op=True)
        self.ring.append((work, parts, slots, ids))
        return applied

    def end(self):
        if not self.ring:
            return {}
        work, parts, slots, ids = self.ring.pop(0)
        if work is not None:
            work.wait()
        out = {}
        for r in range(self.world):
            if r == self.rank:
                continue
            for j in range(slots):
                if ids[r][j] >= 0:
                    out[ids[r][j]] = parts[r][j].cpu().numpy().copy()
        return out

    def in_flight(self):
        return len(self.ring)


class Shard:
    """The C-level shard (mmf_shard_*): the same two exchanges -- frame broadcast, pose all-gather -- run by
    libmmf_hip.so on the context's stream over RCCL, for front-ends that are not PyTorch programs.  `unique_id` is the
    128-byte ncclUniqueId rank 0 made with Shard.unique_id(); ship it to the other ranks by any means."""

    @staticmethod
    def unique_id(lib):
        import ctypes as C
        from ._capi import check
        buf = C.create_string_buffer(128)
        check(lib.mmf_shard_unique_id(buf))
        return buf.raw

    def __init__(self, ctx, rank, world, unique_id):
        import ctypes as C
        from ._capi import check
        self.ctx, self.rank, self.world = ctx, rank, world
        h = C.c_void_p()
        check(ctx.lib.mmf_shard_create(ctx.handle, int(rank), int(world), unique_id, C.byref(h)))
        self.handle = h

    def broadcast_frame(self, rgb, depth, mask=None, root=0):
        from ._capi import check
        from .cudafuncs import _p
        h, w = depth.shape
        check(self.ctx.lib.mmf_shard_broadcast_frame(self.handle, _p(rgb), _p(depth), _p(mask), w, h, int(root)))

    def post_frame(self, rgb, depth, mask=None, root=0, slot=0):
        """start the exchange on the shard's own stream: it overlaps what the context's stream does next"""
        from ._capi import check
        from .cudafuncs import _p
        h, w = depth.shape
        check(self.ctx.lib.mmf_shard_post_frame(self.handle, _p(rgb), _p(depth), _p(mask), w, h, int(root), int(slot)))

    def wait_frame(self, slot=0):
        """the context's stream waits for the exchange posted into `slot` (the host does not)"""
        from ._capi import check
        check(self.ctx.lib.mmf_shard_wait_frame(self.handle, int(slot)))

    def gather_poses(self, fusion):
        from ._capi import check
        check(self.ctx.lib.mmf_shard_gather_poses(self.handle, fusion.handle))

    def gather_poses_begin(self, fusion):
        """enqueue the exchange on the context's stream; nothing waits (up to three may be in flight)"""
        from ._capi import check
        check(self.ctx.lib.mmf_shard_gather_poses_begin(self.handle, fusion.handle))

    def gather_poses_end(self, fusion):
        """wait for the oldest exchange in flight and write the other ranks' poses into the bookkeeping"""
        from ._capi import check
        check(self.ctx.lib.mmf_shard_gather_poses_end(self.handle, fusion.handle))

    def gather_maps(self, fusion, labels, spixel_size):
        """mmf_shard_gather_maps: float32 CUDA tensor [n_models, 2, nspix], {icp error, vertex confidence} per super-pixel"""
        from ._capi import check
        from .cudafuncs import _p
        n = len(fusion.getModels())
        nspix = (fusion.width // spixel_size) * (fusion.height // spixel_size)
        out = torch.empty((n, 2, nspix), dtype=torch.float32, device=labels.device)
        check(self.ctx.lib.mmf_shard_gather_maps(self.handle, fusion.handle, _p(labels), int(spixel_size), _p(out)))
        return out

    def close(self):
        if self.handle:
            self.ctx.lib.mmf_shard_destroy(self.handle)
        self.handle = None
