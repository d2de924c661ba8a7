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
g = MultiMotionFusion(ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1)
o = OracleFusion(w, h, K, enable_multiple_models=True)
known = [0]; keep = []
for i, f in enumerate(frames[:3]):
    spawn = 1 <= i <= 3
    if spawn: known.append(i)
    mask = T.gt_mask(f["ids"], known)
    data = T.model_data(mask, f["depth"], known) if i > 0 else None
    t = (dev(f["rgb"]), dev(f["depth"]), dev(mask)); keep.append(t)
    g.processFrame(t[0], t[1], timestamp=i, mask=t[2], hasNewLabel=spawn, modelData=data)
    o.process_frame(f["rgb"], f["depth"], mask=mask, has_new_label=spawn, model_data=data)
for k in (0, 1):
    od, oo = g.getModelOdometry(k), o.models[k].odom
    for lvl in range(3):
        rows = h >> lvl
        for name in ("vmaps_curr", "nmaps_curr", "vmaps_g_prev", "nmaps_g_prev", "last_depth", "last_image", "next_image", "dIdx", "dIdy", "cloud", "depth_pyr"):
            if name == "depth_pyr" and lvl == 0: continue
            a, b = od.download(name, lvl), oo.buffer(name, lvl)
            a = np.asarray(a); b = np.asarray(b).reshape(a.shape)
            if a.dtype.kind == "f":
                nan_ne = int((np.isnan(a) != np.isnan(b)).sum())
                ok = ~np.isnan(a) & ~np.isnan(b)
                # only x-plane NaN marks validity
                if name.startswith(("vmaps", "nmaps")):
                    va, vb = ~np.isnan(a[:rows]), ~np.isnan(b[:rows])
                    nan_ne = int((va != vb).sum())
                    ok = np.tile(va & vb, (3, 1))
                d = np.abs(a[ok] - b[ok]).max() if ok.any() else 0
                print(k, lvl, name, "nan mismatch", nan_ne, "max diff", d, "n", int(ok.sum()))
            else:
                print(k, lvl, name, "diff", int((a != b).sum()))
