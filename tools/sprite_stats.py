"""Sprite-size statistics of a map as combinedPredict would rasterise it (numpy restatement of splat_setup's bounding
box): how uneven is the splat's work across surfels, waves (64) and workgroups (256)?
    python tools/sprite_stats.py [WxH] [frames] [objects: 0/1]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "640x480").split("x"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
objects = len(sys.argv) > 3 and sys.argv[3] == "1"
K = synth.intrinsics(W, H)
nf = 10
poses = synth.trajectory(nf, seed=1)
objs = synth.make_objects(7, seed=2) if objects else None
traj = synth.object_trajectories(objs, nf, seed=2) if objs else None
frames = [synth.render(p, W, H, seed=i, objects=objs, object_poses=[t[i] for t in traj] if traj else None) for i, p in enumerate(poses)]
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
ctx = Context(0)
g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=int(objects))
zero = up(np.zeros((H, W), np.uint8))
for i in range(n):
    p = i % (2 * nf - 2)
    k = p if p < nf else 2 * nf - 2 - p
    if objects:
        g.processFrame(up(frames[k]["rgb"]), up(frames[k]["depth"]), timestamp=i, mask=zero, hasNewLabel=False)
    else:
        g.processFrame(up(frames[k]["rgb"]), up(frames[k]["depth"]), timestamp=i)
s = g.getBackgroundModel().downloadMap()
pose = np.asarray(g.getCurrPose(), np.float64)
tick = g.getTick()
g.close()
ctx.close()
Ti = np.linalg.inv(pose)
P = s[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
N = s[:, 8:11].astype(np.float64) @ Ti[:3, :3].T
N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-12)
rad = s[:, 11].astype(np.float64)
conf, ts = s[:, 3], s[:, 7]
ok = (P[:, 2] > 0) & (P[:, 2] < 20) & (conf >= 10.0) & (tick - ts <= 200)
x1 = np.stack([N[:, 1] - N[:, 2], -N[:, 0], N[:, 0]], 1)
x1 = x1 / np.maximum(np.linalg.norm(x1, axis=1, keepdims=True), 1e-12) * rad[:, None] * 1.41421356
y1 = np.cross(N, x1)
px, py = [], []
for q in (P + x1, P + y1, P - y1, P - x1):
    px.append(K["fx"] * q[:, 0] / q[:, 2] + K["cx"])
    py.append(K["fy"] * q[:, 1] / q[:, 2] + K["cy"])
px, py = np.stack(px), np.stack(py)
size = np.maximum(np.maximum(px.max(0) - px.min(0), py.max(0) - py.min(0)), 1.0)
size = np.where(ok, np.minimum(size, max(W, H)), 0.0)
area = np.ceil(size) ** 2
print("surfels", len(s), "drawn", int(ok.sum()), "fragments %.2f M" % (area.sum() / 1e6))
print("sprite side: median %.1f p90 %.1f p99 %.1f max %.0f" % (np.median(size[ok]), np.percentile(size[ok], 90), np.percentile(size[ok], 99), size.max()))
for grp, name in ((64, "wave"), (256, "workgroup")):
    m = len(area) // grp * grp
    a = area[:m].reshape(-1, grp)
    rows = np.ceil(size[:m]).reshape(-1, grp)
    tot = a.sum(1)
    print("%s: fragments mean %.0f p99 %.0f max %.0f (max / mean = %.1f); rows-balanced cost = sum over lanes' rows x widest row: mean %.0f max %.0f" % (
        name, tot.mean(), np.percentile(tot, 99), tot.max(), tot.max() / max(tot.mean(), 1), (rows.sum(1) / 64 * rows.max(1)).mean() if grp == 64 else 0,
        (rows.sum(1) / 64 * rows.max(1)).max() if grp == 64 else 0))
# bounding boxes of the sprites of 256 consecutive surfels (would an LDS tile hold a workgroup's depth test?)
m = len(area) // 256 * 256
cxp = (K["fx"] * P[:m, 0] / np.maximum(P[:m, 2], 1e-6) + K["cx"]).reshape(-1, 256)
cyp = (K["fy"] * P[:m, 1] / np.maximum(P[:m, 2], 1e-6) + K["cy"]).reshape(-1, 256)
okm = ok[:m].reshape(-1, 256)
bw, bh = [], []
for r in range(cxp.shape[0]):
    if okm[r].sum() == 0:
        continue
    bw.append(cxp[r][okm[r]].max() - cxp[r][okm[r]].min() + 8)
    bh.append(cyp[r][okm[r]].max() - cyp[r][okm[r]].min() + 8)
ba = np.array(bw) * np.array(bh)
print("workgroup bounding boxes (px^2): median %.0f p75 %.0f p90 %.0f; <= 2048: %.0f %%, <= 4096: %.0f %%, <= 8192: %.0f %%" % (
    np.median(ba), np.percentile(ba, 75), np.percentile(ba, 90), 100 * (ba <= 2048).mean(), 100 * (ba <= 4096).mean(), 100 * (ba <= 8192).mean()))
