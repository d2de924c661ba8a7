import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from multimotionfusion_amd import synth
from multimotionfusion_amd.cudafuncs import Context
from multimotionfusion_amd.fusion import MultiMotionFusion
from oracle.fusion import OracleFusion
from oracle import oracle as orc
orc.build()
w, h, n = 320, 240, 4
K = synth.intrinsics(w, h)
ctx = Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for seed in range(40, 52):
    poses = synth.trajectory(n, seed=seed)
    frames = [synth.render(p, w, h, seed=i + seed) for i, p in enumerate(poses)]
    g = MultiMotionFusion(ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], rgb_only=1)
    o = OracleFusion(w, h, K, rgb_only=True)
    worst, its = 0.0, []
    for i, f in enumerate(frames):
        g.processFrame(dev(f["rgb"]), dev(f["depth"]), timestamp=i)
        o.process_frame(f["rgb"], f["depth"])
        worst = max(worst, float(np.abs(g.getCurrPose() - o.pose).max()))
        its.append((g.getFrameOdometry().iterations_run, o.models[0].odom.stats().iterations_run))
    print("seed", seed, "worst |dpose| %.2e" % worst, "iterations (product, oracle) per frame", its)
    g.close()
