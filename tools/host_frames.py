"""GPU box: frames/s of the drop-in call with HOST frames (mmf_fusion_process_frame_host[_next]) against device frames, one
process, same sequence (forwards and backwards, as bench.py): announced / unannounced, caller buffers page-locked by the
library on first sight (default) or staged through pinned copies (MMF_HOST_REGISTER=0).

    python tools/host_frames.py [steps]
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import HostFrame, MultiMotionFusion  # noqa: E402

W, H, N = 640, 480, 30
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
K = synth.intrinsics(W, H)
poses = synth.trajectory(N, seed=1)
frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
host = [HostFrame(f["rgb"], f["depth"]) for f in frames]
d_rgb = [torch.from_numpy(f["rgb"]).cuda() for f in frames]
d_depth = [torch.from_numpy(f["depth"]).cuda() for f in frames]
ctx = Context(0)


def pp(i):
    p = i % (2 * N - 2)
    return p if p < N else 2 * N - 2 - p


def run(kind):
    g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], icp_weight=10.0)
    t1 = 0.0
    for i in range(20 + steps):
        if i == 20:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        k, kn = pp(i), pp(i + 1)
        if kind == "device":
            g.processFrame(d_rgb[k], d_depth[k], timestamp=i, next=(d_rgb[kn], d_depth[kn]))
        elif kind == "device, no hint":
            g.processFrame(d_rgb[k], d_depth[k], timestamp=i)
        elif kind == "host, announced":
            g.processFrameHost(host[k], timestamp=i, next=host[kn])
        else:
            g.processFrameHost(host[k], timestamp=i)
    torch.cuda.synchronize()
    fps = steps / (time.perf_counter() - t1)
    g.close()
    return fps


for rep in range(2):
    print("  ".join(f"{kind}: {run(kind):.0f}" for kind in ("device", "host, announced", "device, no hint", "host, unannounced")), flush=True)
ctx.close()
