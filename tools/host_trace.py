"""Host-side time line of processFrame on device frames (MMF_HOST_TRACE=1 prints it every 100 calls).
   MMF_HOST_TRACE=1 [MMF_EARLY_IMAGE=chain] python tools/host_trace.py [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, nf = 640, 480, 30
K = synth.intrinsics(W, H)
poses = synth.trajectory(nf, seed=1)
frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
ctx = Context(0)
dev = [(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda()) for f in frames]
g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"])
t0 = None
for i in range(n + 40):
    if i == 40:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    k = i % nf
    if i and k == 0:
        g.reset()
    g.processFrame(dev[k][0], dev[k][1], timestamp=i, next=None if (i + 1) % nf == 0 else dev[k + 1])
torch.cuda.synchronize()
print("%.0f frames/s" % (n / (time.perf_counter() - t0)))
