"""debug aid: poses of the multi-model sequence under the current MMF_ICP_VARIANT, saved for comparison"""
import sys, os
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from multimotionfusion_amd.cudafuncs import Context
from multimotionfusion_amd.fusion import MultiMotionFusion
import test_gpu_multimodel as T
dev = T.dev
w, h = 320, 240
K, poses, traj, frames, objs = T.scene(w, h, 4, 3)
ctx = Context(0)
g = MultiMotionFusion(ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1)
known = [0]; keep = []; out = {}
for i, f in enumerate(frames):
    spawn = 1 <= i <= 3
    if spawn: known.append(i)
    mask = T.gt_mask(f["ids"], known)
    data = T.model_data(mask, f["depth"], known) if i > 0 else None
    t = (dev(f["rgb"]), dev(f["depth"]), dev(mask)); keep.append(t)
    g.processFrame(t[0], t[1], timestamp=i, mask=t[2], hasNewLabel=spawn, modelData=data)
    for k, a in enumerate(g.getModels()):
        out[f"{i}_{k}"] = a.getPose()
np.savez(sys.argv[1], **out)
