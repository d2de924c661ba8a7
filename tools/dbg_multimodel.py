"""debug aid: per-frame pose differences GPU vs oracle for the multi-model sequence of tests/test_gpu_multimodel.py"""
import sys, os
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import oracle as orc
from oracle.fusion import OracleFusion
from multimotionfusion_amd.cudafuncs import Context
from multimotionfusion_amd.fusion import MultiMotionFusion
import test_gpu_multimodel as T
dev = T.dev
w, h = 320, 240
ctx = Context(0)
for seed, with_data, sync in ((22, True, True), (23, True, True), (24, True, True), (22, False, False), (23, False, False)):
    K, poses, traj, frames, objs = T.scene(w, h, 6, 3, seed=seed)
    g = MultiMotionFusion(ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1)
    o = OracleFusion(w, h, K, enable_multiple_models=True)
    known = [0]
    keep = []
    for i, f in enumerate(frames):
        spawn = 1 <= i <= 3
        if spawn: known.append(i)
        mask = T.gt_mask(f["ids"], known)
        data = T.model_data(mask, f["depth"], known) if with_data and i > 0 else None
        t = (dev(f["rgb"]), dev(f["depth"]), dev(mask)); keep.append(t)
        g.processFrame(t[0], t[1], timestamp=i, mask=t[2], hasNewLabel=spawn, modelData=data)
        o.process_frame(f["rgb"], f["depth"], mask=mask, has_new_label=spawn, model_data=data)
        gm = g.getModels()
        print(f"seed={seed} data={with_data} sync={sync} frame {i}:", [f"{np.abs(a.getPose()-b.pose).max():.1e}/{a.lastCount()-b.surfels.shape[0]}" for a, b in zip(gm, o.models)],
              [round(a.confidenceThreshold(),3) for a in gm], flush=True)
        if sync:
            for a, b in zip(gm, o.models):
                a.uploadMap(b.surfels); a.overridePose(b.pose)
            g.predict()
    g.close()
