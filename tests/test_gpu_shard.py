"""-m gpu: the shard below Python (mmf_shard_*: frame broadcast + pose all-gather over RCCL on the library's stream).
One rank on the single-GPU box (RCCL with world size 1 still runs both collectives); the two-rank case needs two
GPUs and skips otherwise -- RCCL does not place two ranks on one device.  No scaling curve is measured here."""
import numpy as np
import pytest
import torch

from multimotionfusion_amd import synth

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_shard_world_1_runs_both_collectives(gpu_ctx):
    from multimotionfusion_amd.fusion import MultiMotionFusion
    from multimotionfusion_amd.shard import Shard
    w, h, n = 320, 240, 4
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n, seed=29)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    sh = Shard(gpu_ctx, 0, 1, Shard.unique_id(gpu_ctx.lib))
    a = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    b = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    b.setShard(0, 1)
    keep = []
    for i, f in enumerate(frames):
        keep.append((dev(f["rgb"]), dev(f["depth"]), torch.zeros((h, w), dtype=torch.uint8, device="cuda")))
        a.processFrame(*keep[-1][:2], timestamp=i)
        sh.broadcast_frame(*keep[-1])
        b.processFrame(*keep[-1][:2], timestamp=i)
        sh.gather_poses(b)
        assert np.array_equal(a.getCurrPose(), b.getCurrPose()), i
        assert np.array_equal(keep[-1][0].cpu().numpy(), f["rgb"])  # the broadcast left the root's frame intact
    a.close()
    b.close()
    sh.close()


def _rank_main(rank, world, uid_path, out_path):
    import os
    import time
    torch.cuda.set_device(rank)
    from multimotionfusion_amd.cudafuncs import Context
    from multimotionfusion_amd.fusion import MultiMotionFusion
    from multimotionfusion_amd.shard import Shard
    ctx = Context(rank)
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as fp:
            fp.write(Shard.unique_id(ctx.lib))
        os.replace(uid_path + ".tmp", uid_path)
    while not os.path.exists(uid_path):
        time.sleep(0.05)
    uid = open(uid_path, "rb").read()
    sh = Shard(ctx, rank, world, uid)
    w, h, n = 320, 240, 5
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n, seed=1)
    objs = synth.make_objects(2, seed=2)
    traj = synth.object_trajectories(objs, n, seed=2)
    g = MultiMotionFusion(ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=1)
    g.setShard(rank, world)
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device=f"cuda:{rank}")
    depth = torch.zeros((h, w), dtype=torch.float32, device=f"cuda:{rank}")
    mask = torch.zeros((h, w), dtype=torch.uint8, device=f"cuda:{rank}")
    out = []
    for i in range(n):
        if rank == 0:  # only the root holds the sensor frame
            f = synth.render(poses[i], w, h, seed=i, objects=objs, object_poses=[t[i] for t in traj])
            rgb.copy_(torch.from_numpy(f["rgb"]))
            depth.copy_(torch.from_numpy(f["depth"]))
            mask.copy_(torch.from_numpy(np.where(f["ids"] < 2, f["ids"], 0).astype(np.uint8)))
        sh.broadcast_frame(rgb, depth, mask)
        g.processFrame(rgb, depth, timestamp=i, mask=mask, hasNewLabel=(i == 1))
        sh.gather_poses(g)
        out.append(np.stack([m.getPose() for m in g.getModels()]))
    np.save(out_path + f".{rank}.npy", np.concatenate([o.reshape(-1) for o in out]))
    g.close()
    sh.close()
    ctx.close()


def test_two_ranks_over_rccl(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL does not put two ranks on one device)")
    import torch.multiprocessing as mp
    uid, out = str(tmp_path / "uid"), str(tmp_path / "poses")
    mp.spawn(_rank_main, args=(2, uid, out), nprocs=2, join=True)
    a, b = np.load(out + ".0.npy"), np.load(out + ".1.npy")
    assert np.array_equal(a, b)  # after the all-gather both ranks hold every model's pose
    assert np.abs(a[-32:-16].reshape(4, 4) - np.eye(4)).max() > 1e-4  # the camera moved
