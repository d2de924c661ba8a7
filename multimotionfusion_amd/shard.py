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
    """model m lives on rank m mod G; the global model (id 0) on rank 0."""
    return model_id % world_size


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

    def gather_poses(self, fusion):
        from ._capi import check
        check(self.ctx.lib.mmf_shard_gather_poses(self.handle, fusion.handle))

    def close(self):
        if self.handle:
            self.ctx.lib.mmf_shard_destroy(self.handle)
        self.handle = None
