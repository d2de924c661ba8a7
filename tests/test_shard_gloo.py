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


def _ring_worker(rank, world, port, out_dir):
    """the pose exchange as a ring (begin / end, three in flight) over gloo, with a model leaving the list mid-ring"""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimotionfusion_amd import shard
    ring = shard.PoseRing(torch.device("cpu"))

    def rec(model_id, frame):  # what the owner would send: any function of (id, frame) will do
        v = np.zeros(18, np.float32)
        v[:16] = np.eye(4, dtype=np.float32).reshape(16)
        v[3], v[16], v[17] = 100.0 * model_id + frame, 0.5 * frame, 1000.0 + model_id
        return v

    lists = {0: [0, 1, 2, 3], 1: [0, 1, 2, 3], 2: [0, 1, 2, 3], 3: [0, 2, 3], 4: [0, 2, 3], 5: [0, 2, 3, 4]}  # model 1 leaves at frame 3
    applied_log = []
    known = {}  # this rank's bookkeeping of the OTHER ranks' models: id -> (frame of the record it holds)
    for frame in range(6):
        ids = lists[frame]
        mine = {m: rec(m, frame) for m in ids if shard.model_owner(m, world) == rank}
        got = ring.begin(ids, mine)  # a fourth begin completes the oldest exchange first
        assert ring.in_flight() <= shard.PoseRing.RING
        if frame < 3:
            assert got == {} and ring.in_flight() == frame + 1
        else:
            assert ring.in_flight() == 3
        for mid, r in got.items():
            if mid in ids:  # (mmf_shard_gather_poses_end: fusion_find(f, id) -- a model that has left is not applied)
                known[mid] = int(round(float(r[3]) - 100.0 * mid))
                assert r[17] == 1000.0 + mid and shard.model_owner(mid, world) != rank
            applied_log.append((frame, mid, mid in ids))
    # frames 3, 4, 5 completed the exchanges of frames 0, 1, 2: the records of model 1 (owner: rank 1) arrive at rank 0 after
    # it has left rank 0's list and are dropped there
    if rank == 0:
        assert [(f, m) for f, m, ok in applied_log if not ok] == [(3, 1), (4, 1), (5, 1)], applied_log
        assert known == {3: 2}  # rank 1's surviving model, as of frame 2 (three frames old at frame 5: two in flight + this one)
    else:
        assert all(ok for _, _, ok in applied_log) and known == {0: 2, 2: 2}
    while ring.in_flight():  # drain: the exchanges of frames 3, 4, 5 (lists without model 1; model 4 appears in the last)
        for mid, r in ring.end().items():
            known[mid] = int(round(float(r[3]) - 100.0 * mid))
    assert known == ({3: 5} if rank == 0 else {0: 5, 2: 5, 4: 5}), (rank, known)
    np.save(os.path.join(out_dir, f"ring{rank}.npy"), np.array(sorted(known.items())))
    dist.barrier()
    dist.destroy_process_group()


def test_pose_ring_world2(tmp_path):
    """begin / end with three in flight, and a model that leaves the list while its records are still in the ring"""
    port = _free_port()
    mp.spawn(_ring_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "ring0.npy").tolist() == [[3, 5]] and np.load(tmp_path / "ring1.npy").tolist() == [[0, 5], [2, 5], [4, 5]]
