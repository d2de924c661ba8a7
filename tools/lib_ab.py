"""A/B of two builds of the library on the SAME box (box-to-box spread exceeds most effects): alternates
subprocesses running the processFrame loop with MMF_HIP_LIB pointing at either .so.
    python tools/lib_ab.py build/libmmf_head.so build/libmmf_new.so [reps] [steps]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
import numpy as np, torch
sys.path.insert(0, %r)
from multimotionfusion_amd import synth
from multimotionfusion_amd.cudafuncs import Context
from multimotionfusion_amd.fusion import MultiMotionFusion
W, H, N, steps = 640, 480, 50, int(sys.argv[1])
K = synth.intrinsics(W, H)
poses = synth.trajectory(N, seed=1)
frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
rgb = [torch.from_numpy(f["rgb"]).cuda() for f in frames]
depth = [torch.from_numpy(f["depth"]).cuda() for f in frames]
ctx = Context(0)
mmf = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], icp_weight=10.0)
pre = hasattr(mmf, "prefetchFrame") and os.environ.get("AB_PREFETCH", "1") != "0"
def step(i):
    k = i %% N
    if i and k == 0: mmf.reset()
    mmf.processFrame(rgb[k], depth[k], timestamp=i)
    if pre and (i + 1) %% N: mmf.prefetchFrame(rgb[(i + 1) %% N], depth[(i + 1) %% N])
    return mmf.getCurrPose()
for i in range(40): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(40, 40 + steps): step(i)
torch.cuda.synchronize()
print("%%.4f" %% ((time.perf_counter() - t0) / steps * 1e3))
''' % ROOT

libs = [os.path.abspath(p) for p in sys.argv[1:3]]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = sys.argv[4] if len(sys.argv) > 4 else "400"
for rep in range(reps):
    row = []
    for lib in libs:
        env = dict(os.environ, MMF_HIP_LIB=lib)
        out = subprocess.run([sys.executable, "-c", CHILD, steps], env=env, capture_output=True, text=True)
        row.append(f"{os.path.basename(lib)} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-200:]} ms")
    print(" | ".join(row), flush=True)
