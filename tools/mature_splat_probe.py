"""GPU box: combinedPredict on a DEEP store (one frame's map + three occluded copies, ~740 k stable surfels at 640x480) with the
plain and with the early depth test (splat_kernel<true>: mmf_debug_set_splat_bound), HIP events around the pass, median of 15.

    python tools/mature_splat_probe.py [WxH]
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.model import Model, filterDepth  # noqa: E402

W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "640x480").split("x"))
MAXD, CUTOFF, TIME_DELTA, CONF = 20.0, 3.0, 200, 10.0
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
K = synth.intrinsics(W, H)
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402
poses = synth.trajectory(14, seed=1)
frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
ctx = Context(0)
g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"])
for i in range(12):  # the map of a short sequence (it lives in the first camera's frame)
    g.processFrame(up(frames[i]["rgb"]), up(frames[i]["depth"]), timestamp=i)
base = g.getModels()[0].downloadMap()
pose_now = np.asarray(g.getCurrPose(), np.float32)
tick = g.getTick()
g.close()
m = Model(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], 0, CONF)
layers = []
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for step in (0.0, 0.02, 0.04, 0.06)[:NL]:
    cp = base.copy()
    cp[:, :3] += step * cp[:, :3] / np.maximum(np.linalg.norm(cp[:, :3], axis=1, keepdims=True), 1e-6)
    cp[:, 3] = np.maximum(cp[:, 3], 20.0)
    cp[:, 7] = float(tick)
    layers.append(cp)
deep = np.concatenate(layers)[: 1024 * 1024 - 310000]
m.uploadMap(deep)
m.overridePose(pose_now)
ref = None
for mode, name in ((0, "plain"), (1, "early-z"), (0, "plain"), (1, "early-z")):
    ctx.lib.mmf_debug_set_splat_bound(mode)
    ts = []
    for rep in range(18):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.combinedPredict(MAXD, tick, tick, TIME_DELTA)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    vc = m.texture("vertexConf").cpu().numpy()
    if ref is None:
        ref = vc
    same = np.array_equal(vc.view(np.uint32), ref.view(np.uint32))
    print(f"{name:8s}: combinedPredict {np.median(ts[3:]):7.1f} us (min {min(ts):.1f}) on {deep.shape[0]} surfels, covered {float((vc[..., 2] > 0).mean()):.3f}, "
          f"images equal to the first run: {same}")
if os.environ.get("MMF_HIP_LIB") and "count" in os.environ["MMF_HIP_LIB"]:  # a -DMMF_SPLAT_COUNT build
    import ctypes as C
    raw = C.CDLL(os.environ["MMF_HIP_LIB"])
    out = (C.c_ulonglong * 4)()
    raw.mmf_debug_splat_counts(out, 1)
    ctx.lib.mmf_debug_set_splat_bound(1)
    m.combinedPredict(MAXD, tick, tick, TIME_DELTA)
    raw.mmf_debug_splat_counts(out, 1)
    print("early-z pass: fragments in sprites %.2f M, past the disc-depth bound %.2f M, drawn (atomics) %.2f M" % (out[0] / 1e6, out[1] / 1e6, out[2] / 1e6))
ctx.lib.mmf_debug_set_splat_bound(-1)
m.close()
ctx.close()

# sprite statistics of the store at that pose (numpy restatement of splat_setup's bounding box)
Ti = np.linalg.inv(pose_now.astype(np.float64))
P = deep[:, :3].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]
N = deep[:, 8:11].astype(np.float64) @ Ti[:3, :3].T
N /= np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-12)
rad = deep[:, 11].astype(np.float64)
ok = (P[:, 2] > 0) & (P[:, 2] < MAXD) & (deep[:, 3] >= CONF)
x1 = np.stack([N[:, 1] - N[:, 2], -N[:, 0], N[:, 0]], 1)
x1 = x1 / np.maximum(np.linalg.norm(x1, axis=1, keepdims=True), 1e-12) * rad[:, None] * 1.41421356
y1 = np.cross(N, x1)
pxs, pys = [], []
for q in (P + x1, P + y1, P - y1, P - x1):
    pxs.append(K["fx"] * q[:, 0] / q[:, 2] + K["cx"])
    pys.append(K["fy"] * q[:, 1] / q[:, 2] + K["cy"])
pxs, pys = np.stack(pxs), np.stack(pys)
size = np.maximum(np.maximum(pxs.max(0) - pxs.min(0), pys.max(0) - pys.min(0)), 1.0)
size = np.where(ok, np.minimum(size, max(W, H)), 0.0)
print("drawn", int(ok.sum()), "of", len(deep), "; fragments %.1f M (%.1f per pixel); sprite side median %.1f p90 %.1f p99 %.1f; radius median %.4f m at depth %.2f m"
      % ((np.ceil(size) ** 2).sum() / 1e6, (np.ceil(size) ** 2).sum() / (W * H), np.median(size[ok]), np.percentile(size[ok], 90), np.percentile(size[ok], 99),
         np.median(rad[ok]), np.median(P[ok, 2])))
cx = np.floor(K["fx"] * P[ok, 0] / P[ok, 2] + K["cx"]).astype(int)
cy = np.floor(K["fy"] * P[ok, 1] / P[ok, 2] + K["cy"]).astype(int)
inimg = (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H)
cnt = np.zeros((H, W), int)
np.add.at(cnt, (cy[inimg], cx[inimg]), 1)
print("pixels holding a surfel centre: %.3f" % float((cnt > 0).mean()))
