"""GPU box: how the one-launch chain walks the object models of the multi-model loop (tools/profile_frames.py's scene):
per frame and model whether it was walked by its extents, the level-0 ICP rectangle of the chain's last launch, the most
passes a level-0 rectangle took, and how often a chain was given up (mmf_gn_chain_status).
    python tools/mm_sparse_probe.py [models] [frames]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402

models = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, nf = 640, 480, 10
K = synth.intrinsics(W, H)
poses = synth.trajectory(nf, seed=1)
objs = synth.make_objects(7, seed=2)
traj = synth.object_trajectories(objs, nf, seed=2)
frames = [synth.render(p, W, H, seed=i, objects=objs, object_poses=[t[i] for t in traj]) for i, p in enumerate(poses)]
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
rgb, depth = [up(f["rgb"]) for f in frames], [up(f["depth"]) for f in frames]
mask = [up(np.where(f["ids"] < models, f["ids"], 0).astype(np.uint8)) for f in frames]
for k in range(1, models):
    ys, xs = np.nonzero(frames[0]["ids"] == k)
    print(f"object {k}: mask box in frame 0 x {xs.min()}..{xs.max()} y {ys.min()}..{ys.max()} ({len(xs)} px)")
ctx = Context(0)
g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=models - 1)


def frame_of(i):
    p = i % (2 * nf - 2)
    return p if p < nf else 2 * nf - 2 - p


def status():
    rec, in_use = C.c_int(0), C.c_int(0)
    ctx.lib.mmf_gn_chain_status(C.byref(rec), C.byref(in_use))
    return rec.value, in_use.value


for i in range(n):
    k, kn = frame_of(i), frame_of(i + 1)
    g.processFrame(rgb[k], depth[k], timestamp=i, mask=mask[k], hasNewLabel=1 <= i < models, next=(rgb[kn], depth[kn]))
    line = []
    for m in range(len(g.getModels())):
        od = g.getModelOdometry(m)
        outside, by_ext = od.sparseWalk()
        line.append(f"E{od.sparseRect[:2]}p{od.sparseRect[4]}{'' if od.sparseRect[5] else '!'}n{od.sparseRect[6:9]}" if by_ext else "d")
    print(f"frame {i:3d} recoveries/in-use {status()} :", " ".join(line))
g.close()
ctx.close()
