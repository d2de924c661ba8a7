import sys, os
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from oracle import oracle as orc
from oracle.fusion import OracleFusion
from multimotionfusion_amd.cudafuncs import Context, icpStep, CameraModel
import test_gpu_multimodel as T
dev = T.dev
w, h = 320, 240
K, poses, traj, frames, objs = T.scene(w, h, 4, 3)
ctx = Context(0)
o = OracleFusion(w, h, K, enable_multiple_models=True)
known = [0]
for i, f in enumerate(frames[:3]):
    spawn = 1 <= i <= 3
    if spawn: known.append(i)
    mask = T.gt_mask(f["ids"], known)
    data = T.model_data(mask, f["depth"], known) if i > 0 else None
    if i == 2:
        pose_before = o.models[1].pose.copy()
    o.process_frame(f["rgb"], f["depth"], mask=mask, has_new_label=spawn, model_data=data)
m = o.models[1]
R = pose_before[:3, :3]; t = pose_before[:3, 3]
Rinv = np.linalg.inv(R.astype(np.float64)).astype(np.float32)
rng = np.random.default_rng(0)
for trial in range(4):
    # trial 0: the initial pose; others: small perturbations (what later GN iterations look like)
    dR = np.eye(3, dtype=np.float32); dt = np.zeros(3, np.float32)
    if trial:
        from multimotionfusion_amd import synth
        dR = synth.rodrigues(rng.normal(size=3) * 3e-3).astype(np.float32); dt = (rng.normal(size=3) * 3e-3).astype(np.float32)
    Rc = (dR @ R).astype(np.float32); tc = (t + dt).astype(np.float32)
    for lvl in (2, 1, 0):
        s = 1 << lvl
        vg, ng, vc, nc = (np.array(m.odom.buffer(n, lvl)) for n in ("vmaps_g_prev", "nmaps_g_prev", "vmaps_curr", "nmaps_curr"))
        out, _ = orc.icp_step(Rc, tc, vc, nc, Rinv, t, K["fx"] / s, K["fy"] / s, K["cx"] / s, K["cy"] / s, vg, ng, 0.10, T_ANG := float(np.float32(np.sin(20.0 * 3.14159254 / 180.0))))
        Ao, bo, ro = orc.unpack_se3(out)
        Ag, bg, rg = icpStep(ctx, Rc, tc, dev(vc), dev(nc), Rinv, t, CameraModel(K["fx"] / s, K["fy"] / s, K["cx"] / s, K["cy"] / s), dev(vg), dev(ng), 0.10, T_ANG)
        print(trial, lvl, "count", ro[1], rg[1], "res", ro[0], rg[0], "relA", np.abs(Ao - Ag).max() / np.abs(Ao).max(), "relb", np.abs(bo - bg).max() / (np.abs(bo).max() + 1e-20), flush=True)
