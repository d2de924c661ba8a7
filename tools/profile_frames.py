"""Steady-state frames of the bench workload for profiler runs (no extras):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python3 tools/profile_frames.py [frames] [WxH] [models] [prefetch]
models > 1: the moving-object scene with that many rigid-body models on the GPU (mask = ground-truth ids)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "640x480").split("x"))
models = int(sys.argv[3]) if len(sys.argv) > 3 else 1
prefetch = (sys.argv[4] if len(sys.argv) > 4 else "1") != "0"
headline = len(sys.argv) > 5 and sys.argv[5] == "headline"  # bench.py's N = 1 loop: 30 frames forwards and backwards
K = synth.intrinsics(W, H)
nf = 30 if headline else 10
poses = synth.trajectory(nf, seed=1)
objs = synth.make_objects(7, seed=2) if models > 1 else None
traj = synth.object_trajectories(objs, nf, seed=2) if objs else None
frames = [synth.render(p, W, H, seed=i, objects=objs, object_poses=[t[i] for t in traj] if traj else None) for i, p in enumerate(poses)]
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
rgb, depth = [up(f["rgb"]) for f in frames], [up(f["depth"]) for f in frames]
mask = [up(np.where(f["ids"] < models, f["ids"], 0).astype(np.uint8)) for f in frames]
ctx = Context(0)
sharded = bool(os.environ.get("MMF_PROFILE_SHARD"))  # like bench.py's sharded step: models are created when they are spawned
g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=int(models > 1),
                      preallocated_models=0 if sharded else models - 1)
if os.environ.get("MMF_PROFILE_SHARD"):  # "rank/world": this process runs only the models with index % world == rank (no collectives here)
    g.setShard(*(int(v) for v in os.environ["MMF_PROFILE_SHARD"].split("/")))


def frame_of(i):
    p = i % (2 * nf - 2)
    return p if p < nf else 2 * nf - 2 - p


import time  # noqa: E402
t_loop = time.perf_counter()
for i in range(n):
    if i == 40:
        torch.cuda.synchronize()
        t_loop = time.perf_counter()
    k = frame_of(i)
    if headline:  # bench.py's N = 1 loop: the 30-frame sequence forwards and backwards, no reset
        kn = frame_of(i + 1)
        g.processFrame(rgb[k], depth[k], timestamp=i, next=(rgb[kn], depth[kn]) if prefetch else None)  # (prefetch 0: nothing on the side streams)
        continue
    if models > 1:
        kn = frame_of(i + 1)
        g.processFrame(rgb[k], depth[k], timestamp=i, mask=mask[k], hasNewLabel=1 <= i < models,
                       next=(rgb[kn], depth[kn]) if prefetch else None)
    else:
        g.processFrame(rgb[k], depth[k], timestamp=i)
        if prefetch:
            kn = frame_of(i + 1)
            g.prefetchFrame(rgb[kn], depth[kn])
torch.cuda.synchronize()
if n > 40:
    print("ms per frame (after 40): %.4f" % ((time.perf_counter() - t_loop) / (n - 40) * 1e3))
print("frames", n, "models", len(g.getModels()), "surfels", [m.lastCount() for m in g.getModels()])
g.close()
ctx.close()
