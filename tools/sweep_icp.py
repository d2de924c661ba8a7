"""Tuning sweep (GPU box): time the ICP reduction kernel for each launch geometry and level."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multimotionfusion_amd import synth
from multimotionfusion_amd.cudafuncs import Context
from multimotionfusion_amd.odometry import RGBDOdometry

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
K = synth.intrinsics(W, H)
poses = synth.trajectory(2, seed=1)
fp, fc = synth.render(poses[0], W, H, seed=0), synth.render(poses[1], W, H, seed=1)
ctx = Context(0)
g = RGBDOdometry(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"])
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pose = poses[0].astype(np.float32)
g.initFirstRGB(dev(fp["rgb"]))
g.initICPModel(dev(fp["vertex"]), dev(fp["normal"]), 15.0, pose)
g.initRGBModel(dev(fp["rgb"]))
g.buildDepthPyramid(dev(fc["depth"]))
g.initICP(depthCutoff=15.0)
g.initRGB(dev(fc["rgb"]))
g.getIncrementalTransformation(pose[:3, 3], pose[:3, :3], False, 10.0, True, False, True)
print("inliers", g.lastICPCount)
for variant in (20256, 10256, 1020256, 2020256, 1040256, 2040256, 2040128, 2020128, 2020064, 1010256, 2010256, 2010128, 2010064):
    row = []
    for lvl in range(3):
        us = min(g.timeIcpKernel(lvl, 300, variant) for _ in range(3))
        n = (W >> lvl) * (H >> lvl)
        row.append(f"L{lvl} {us:7.2f} us {(48*n+116)/us/1e3:7.1f} GB/s")
    print(f"V={variant//1000000} PX={(variant//10000)%100} BLOCK={variant%10000:4d}: " + " | ".join(row), flush=True)
