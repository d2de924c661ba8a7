"""A/B of the next-frame prefetch inside ONE process (box-to-box spread is larger than the effect):
none / prefetchFrame after processFrame.   python tools/prefetch_ab.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402

W, H, N = 640, 480, 50
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
K = synth.intrinsics(W, H)
poses = synth.trajectory(N, seed=1)
frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
rgb = [torch.from_numpy(f["rgb"]).cuda() for f in frames]
depth = [torch.from_numpy(f["depth"]).cuda() for f in frames]
ctx = Context(0)


def run(mode):
    mmf = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], icp_weight=10.0)
    def step(i):
        k = i % N
        if i and k == 0:
            mmf.reset()
        kn = (i + 1) % N
        mmf.processFrame(rgb[k], depth[k], timestamp=i)
        if mode == "after" and kn != 0:
            mmf.prefetchFrame(rgb[kn], depth[kn])
        return mmf.getCurrPose()
    for i in range(20):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20, 20 + steps):
        p = step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    mmf.close()
    return dt * 1e3, p


for rep in range(int(os.environ.get("AB_REPS", "3"))):
    out = []
    for mode in ("none", "after"):
        ms, p = run(mode)
        out.append(f"{mode} {ms:.4f} ms ({1e3 / ms:.0f} fps)")
    print(" | ".join(out), flush=True)
