"""Surfel passes on a FIXED map, for same-input A/B runs of kernel variants under rocprofv3 (experiment tool).

    python tools/surfel_probe.py build [WxH] [frames]    # shipped library: run the sequence, save the map to /tmp
    rocprofv3 --kernel-trace --stats ... -- python3 tools/surfel_probe.py run [rounds]   # any library (MMF_HIP_LIB)

`run` restores the saved map before every round, so each round (predictIndices, fuse, predictIndices, clean,
combinedPredict on the same frame and pose) does identical work whatever an experimental variant writes."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402

STATE = "/tmp/surfel_probe_state.npz"
MAXD, CUTOFF, TIME_DELTA, CONF = 20.0, 3.0, 200, 10.0
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731

if sys.argv[1] == "build":
    from multimotionfusion_amd.fusion import MultiMotionFusion
    W, H = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "640x480").split("x"))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    K = synth.intrinsics(W, H)
    poses = synth.trajectory(n + 1, seed=1)
    frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
    ctx = Context(0)
    g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"])
    for i in range(n):
        g.processFrame(up(frames[i]["rgb"]), up(frames[i]["depth"]), timestamp=i)
    m = g.getModels()[0]
    surfels = m.downloadMap()
    np.savez(STATE, surfels=surfels, pose=np.asarray(g.getCurrPose(), np.float32), rgb=frames[n]["rgb"], depth=frames[n]["depth"],
             tick=g.getTick(), size=np.array([W, H]))
    print("saved", surfels.shape[0], "surfels, tick", g.getTick())
    g.close()
    ctx.close()
else:
    from multimotionfusion_amd.model import Model, filterDepth
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    st = np.load(STATE)
    W, H = (int(v) for v in st["size"])
    K = synth.intrinsics(W, H)
    ctx = Context(0)
    m = Model(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], 0, CONF)
    d_rgb, d_raw = up(st["rgb"]), up(st["depth"])
    d_fil = filterDepth(ctx, d_raw, CUTOFF)
    d_mask = up(np.zeros((H, W), np.uint8))
    tick = int(st["tick"]) + 1
    surfels = st["surfels"]
    counts = []
    for r in range(rounds):
        m.uploadMap(surfels)
        m.overridePose(st["pose"])
        m.predictIndices(tick, MAXD, TIME_DELTA)
        m.fuse(tick, d_rgb, d_mask, d_raw, d_fil, MAXD, 1.0)
        m.predictIndices(tick, MAXD, TIME_DELTA)
        m.clean(tick, TIME_DELTA, MAXD, d_fil, d_mask, 3.0)
        m.combinedPredict(MAXD, tick, tick, TIME_DELTA)
        counts.append(m.lastCount())
    torch.cuda.synchronize()
    print("rounds", rounds, "surfels in", surfels.shape[0], "out", counts[0], counts[-1])
    m.close()
    ctx.close()
