"""PMC run (GPU box): launch the level-0 ICP producer kernel a few times so rocprofv3 --pmc can count its
memory-side traffic.  rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/pmc_icp.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from multimotionfusion_amd import synth
from multimotionfusion_amd.cudafuncs import Context
from multimotionfusion_amd.odometry import RGBDOdometry

W, H = 640, 480
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
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # GEN*1000000 + PX*10000 + BLOCK, 0 = shipped default
for lvl in range(3):
    print(lvl, g.timeIcpKernel(lvl, 20, variant))
