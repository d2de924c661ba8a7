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
K, poses, traj, frames, objs = T.scene(w, h, 5, 3)
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
    gm = g.getModels()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(gm, o.models)):
        sa, sb = a.downloadMap(), b.surfels
        print(i, k, "pose", np.abs(a.getPose() - b.pose).max(), "surfels equal", sa.shape == sb.shape and np.array_equal(sa.view(np.uint32), sb.view(np.uint32)),
              "conf", a.confidenceThreshold(), b.conf)
        for name, ref in (("vertexConf", b.vertexConf), ("normalRadius", b.normalRadius), ("image", b.image)):
            got = a.texture(name).cpu().numpy()
            ne = (got.view(np.uint32) != np.ascontiguousarray(ref).view(np.uint32)) if got.dtype != np.uint8 else (got != ref)
            print("    ", name, "diff px", int(ne.any(-1).sum()), "valid", int((ref[..., 2] > 0).sum()) if name == "vertexConf" else "")
        st = g.getModelOdometry(k); so = b.odom.stats()
        print("     stats icp", st.lastICPError, so.lastICPError, st.lastICPCount, so.lastICPCount, "rgb", st.lastRGBError, so.lastRGBError, st.lastRGBCount, so.lastRGBCount,
              "\n     eigA gpu", np.linalg.eigvalsh(np.array(st.lastA).reshape(6, 6)).round(2), "\n     eigA orc", np.linalg.eigvalsh(np.array(so.lastA).reshape(6, 6)).round(2),
              "\n     relA", np.abs(np.array(st.lastA).ravel() - np.array(so.lastA).ravel()).max() / np.abs(np.array(so.lastA)).max(), "relb", np.abs(np.array(st.lastb).ravel() - np.array(so.lastb).ravel()).max() / (np.abs(np.array(so.lastb)).max() + 1e-30),
              "so3", st.lastSO3Error, so.lastSO3Error, st.lastSO3Count, so.lastSO3Count, "iters", st.iterations_run, so.iterations_run)
g.close()
