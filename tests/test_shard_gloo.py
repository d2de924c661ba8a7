"""CPU, world_size 2 over gloo: the per-model shard plumbing of multimotionfusion_amd.shard
(frame broadcast from rank 0, all_gather of poses) with the oracle standing in as the tracker."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimotionfusion_amd import shard, synth
    from oracle import oracle as orc
    w, h = 160, 120
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(2, seed=1)
    assert shard.local_models(5, rank, world) == [m for m in range(5) if m % world == rank]
    # rank 0 owns the sensor frame; the others start from garbage and must receive it
    if rank == 0:
        f = synth.render(poses[1], w, h, seed=1)
        rgb, depth = torch.from_numpy(f["rgb"].copy()), torch.from_numpy(f["depth"].copy())
    else:
        rgb, depth = torch.zeros(h, w, 3, dtype=torch.uint8), torch.full((h, w), -1.0)
    mask = torch.zeros(h, w, dtype=torch.uint8)
    shard.broadcast_frame(rgb, depth, mask, src=0)
    ref = synth.render(poses[1], w, h, seed=1)
    assert np.array_equal(rgb.numpy(), ref["rgb"]) and np.array_equal(depth.numpy(), ref["depth"])
    # each rank tracks its own model (here: the same scene from a rank-dependent start pose)
    fp = synth.render(poses[0], w, h, seed=0)
    o = orc.Odometry(w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    pose0 = poses[0].astype(np.float32)
    o.initICPModel(fp["vertex"], fp["normal"], pose0)
    o.initICP(depth.numpy(), 15.0)
    t, R = o.getIncrementalTransformation(pose0[:3, 3], pose0[:3, :3], False, 100.0, True, False, False)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3], pose[:3, 3] = R, t
    assert np.isfinite(t).all()
    pose[0, 3] += rank  # make the per-rank payload distinguishable
    s = o.stats()
    allp = shard.gather_poses(pose, s.lastICPError, s.lastICPCount, torch.device("cpu"))
    assert allp.shape == (world, 18)
    for r in range(world):
        assert abs(allp[r, 3] - (t[0] + r)) < 1e-5, (rank, r, allp[:, 3], t)  # every rank sees every rank's pose
        assert allp[r, 17] == s.lastICPCount
    # the non-blocking forms bench.py uses to overlap the next frame's broadcast with tracking
    rgb2 = rgb.clone() if rank == 0 else torch.zeros_like(rgb)
    depth2 = depth.clone() if rank == 0 else torch.zeros_like(depth)
    works = shard.broadcast_frame_async(rgb2, depth2, mask, src=0)
    assert len(works) == 3
    for wk in works:
        wk.wait()
    assert np.array_equal(rgb2.numpy(), ref["rgb"]) and np.array_equal(depth2.numpy(), ref["depth"])
    work, parts = shard.gather_poses_async(pose, s.lastICPError, s.lastICPCount, torch.device("cpu"))
    work.wait()
    assert np.array_equal(torch.stack(parts).numpy(), allp)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), allp)
    dist.barrier()
    dist.destroy_process_group()


def test_model_shard_world2(tmp_path, orc):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy")
    b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)  # both ranks hold the same gathered table


def _maps_worker(rank, world, port, out_dir):
    """step 3b over gloo: the per-model super-pixel maps reach every rank, keyed by model id -- with a hole in the id
    sequence (model 2 has left the list), which a rule by list position would hand to the wrong ranks"""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimotionfusion_amd import shard
    ids, nspix = [0, 1, 3, 4, 5], 40 * 30
    slots, table = shard.slot_table(ids, world)
    assert slots == 3 and table == [[0, 3, -1], [1, 2, 4]]  # ids 0, 4 on rank 0; ids 1, 3, 5 on rank 1

    def maps_of(model_id):  # what the owner would compute from its model: any function of the id will do here
        g = torch.Generator().manual_seed(1000 + model_id)
        return torch.rand((2, nspix), generator=g)

    local = {k: maps_of(m) for k, m in enumerate(ids) if shard.model_owner(m, world) == rank}
    assert sorted(local) == [k for k in table[rank] if k >= 0]
    got = shard.gather_maps(local, ids, nspix, torch.device("cpu"))
    want = torch.stack([maps_of(m) for m in ids])  # the unsharded run: every model's maps computed in one place
    assert got.shape == want.shape and torch.equal(got, want), rank
    np.save(os.path.join(out_dir, f"maps{rank}.npy"), got.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_model_maps_reach_every_rank_world2(tmp_path):
    port = _free_port()
    mp.spawn(_maps_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "maps0.npy"), np.load(tmp_path / "maps1.npy"))


def test_slot_table_follows_ids_not_positions():
    sys.path.insert(0, REPO)
    from multimotionfusion_amd import shard
    # three ranks; models 0..5, then model 1 leaves: nobody else changes owner
    before = shard.slot_table([0, 1, 2, 3, 4, 5], 3)
    after = shard.slot_table([0, 2, 3, 4, 5], 3)
    assert before == (2, [[0, 3], [1, 4], [2, 5]])
    assert after == (2, [[0, 2], [3, -1], [1, 4]])  # list indices shift, owners (id % 3) do not
    assert shard.slot_table([], 2) == (1, [[-1], [-1]])


def test_single_process_paths_are_noops():
    sys.path.insert(0, REPO)
    from multimotionfusion_amd import shard
    rgb = torch.ones(4, 4, 3, dtype=torch.uint8)
    shard.broadcast_frame(rgb, torch.ones(4, 4), torch.zeros(4, 4, dtype=torch.uint8))
    out = shard.gather_poses(np.eye(4), 0.5, 7.0, torch.device("cpu"))
    assert out.shape == (1, 18) and out[0, 16] == 0.5 and out[0, 17] == 7.0
    assert shard.model_owner(0, 8) == 0 and shard.model_owner(9, 8) == 1
