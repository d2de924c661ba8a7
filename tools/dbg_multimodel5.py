import sys, os
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle.fusion import OracleFusion
from multimotionfusion_amd.cudafuncs import Context
from multimotionfusion_amd.fusion import MultiMotionFusion
import test_gpu_multimodel as T
dev = T.dev
w, h = 320, 240
K, poses, traj, frames, objs = T.scene(w, h, 4, 3)
ctx = Context(0)
for label, kw in (("default", {}), ("icp only", dict(icp_weight=100.0)), ("no so3", dict(so3=False)), ("no pyramid", dict(pyramid=False)),
                  ("fast", dict(fast_odom=True)), ("rgb only", dict(rgb_only=True))):
    g = MultiMotionFusion(ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1)
    o = OracleFusion(w, h, K, enable_multiple_models=True)
    known = [0]; keep = []
    for i, f in enumerate(frames[:3]):
        spawn = 1 <= i <= 3
        if spawn: known.append(i)
        mask = T.gt_mask(f["ids"], known)
        data = T.model_data(mask, f["depth"], known) if i > 0 else None
        t = (dev(f["rgb"]), dev(f["depth"]), dev(mask)); keep.append(t)
        if i == 2:
            for key, v in kw.items():
                setattr(o, key, v)
                {"icp_weight": g.setIcpWeight, "so3": g.setSo3, "pyramid": g.setPyramid, "fast_odom": g.setFastOdom, "rgb_only": g.setRgbOnly}[key](v)
        g.processFrame(t[0], t[1], timestamp=i, mask=t[2], hasNewLabel=spawn, modelData=data)
        o.process_frame(f["rgb"], f["depth"], mask=mask, has_new_label=spawn, model_data=data)
    gm = g.getModels()
    st, so = g.getModelOdometry(1), o.models[1].odom.stats()
    print(label, [f"{np.abs(a.getPose()-b.pose).max():.1e}" for a, b in zip(gm, o.models)], "iters", st.iterations_run, so.iterations_run,
          "icp", st.lastICPCount, so.lastICPCount, "rgb", st.lastRGBCount, so.lastRGBCount, flush=True)
    g.close()
