"""Fast / slow mode of the headline loop against where the buffers of a MultiMotionFusion object landed: several objects in
one process, frames/s of each, and the addresses of its main buffers.   python tools/mode_probe.py [objects] [frames]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd._capi import check  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402

nobj = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
W, H, nf = 640, 480, 30
K = synth.intrinsics(W, H)
poses = synth.trajectory(nf, seed=1)
frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
ctx = Context(0)
dev = [(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda()) for f in frames]
import gc  # noqa: E402
gc.collect()
gc.disable()
keep = []
if os.environ.get("MMF_PROBE_PREHEAT"):  # seconds of dense work first: do the clocks explain the two modes?
    a = torch.randn(4096, 4096, device="cuda")
    t_end = time.perf_counter() + float(os.environ["MMF_PROBE_PREHEAT"])
    while time.perf_counter() < t_end:
        for _ in range(20):
            a = (a @ a) * 1e-3
        torch.cuda.synchronize()
    del a
for k_obj in range(nobj):
    g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"])
    t0 = None
    for i in range(n + 40):
        if i == 40:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        k = i % nf
        if i and k == 0:
            g.reset()
        g.processFrame(dev[k][0], dev[k][1], timestamp=i, next=None if (i + 1) % nf == 0 else dev[k + 1])
    torch.cuda.synchronize()
    fps = n / (time.perf_counter() - t0)
    od, m = g.getFrameOdometry(), g.getBackgroundModel()
    addr = {}
    for name in ("vmaps_curr", "prev_packed", "cloud4", "last_depth", "dIdx"):
        p, b = C.c_void_p(), C.c_size_t()
        check(ctx.lib.mmf_odom_buffer(od.handle, name.encode(), 0, C.byref(p), C.byref(b)))
        addr[name] = p.value
    for name in ("vertexConf", "index"):
        addr[name] = m.texture(name).data_ptr()
    print("%5.0f frames/s  " % fps + "  ".join("%s %x" % (k2, v) for k2, v in addr.items()), flush=True)
    if k_obj % 2 == 0:
        keep.append(torch.empty(int(1.3e6) * (k_obj + 1), dtype=torch.uint8, device="cuda"))  # shift what the next object gets
    g.close()
