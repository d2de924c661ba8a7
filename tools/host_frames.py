"""Frames/s of processFrame with HOST frames announced one call ahead (mmf_fusion_process_frame_host_next), and with device
frames + next-frame hint for comparison, in one process.   python tools/host_frames.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import HostFrame, MultiMotionFusion  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, nf = 640, 480, 30
K = synth.intrinsics(W, H)
poses = synth.trajectory(nf, seed=1)
frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
ctx = Context(0)
dev = [(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda()) for f in frames]
host = [HostFrame(f["rgb"], f["depth"]) for f in frames]
import gc  # noqa: E402
gc.collect()
gc.disable()
for mode in ("device", "host", "device", "host"):
    g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"])
    t0 = None
    for i in range(n + 40):
        if i == 40:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        k = i % nf
        if i and k == 0:
            g.reset()
        last = (i + 1) % nf == 0
        if mode == "host":
            g.processFrameHost(host[k], timestamp=i, next=None if last else host[k + 1])
        else:
            nxt = None if last else dev[k + 1]
            g.processFrame(dev[k][0], dev[k][1], timestamp=i, next=nxt)
    torch.cuda.synchronize()
    print("%-6s frames: %.0f frames/s" % (mode, n / (time.perf_counter() - t0)))
    g.close()
