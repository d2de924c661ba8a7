"""Diagnostic (GPU box): where does one gn_iter_kernel launch (csrc/gn_fused.hpp) spend its time?

Builds an instrumented copy of the library (-DMMF_STAMPS: thread 0 of every workgroup stamps the 100 MHz constant clock at
each phase boundary), runs getIncrementalTransformation and reads the stamps of the LAST gn_iter launch (finest level,
last iteration; workgroup 0 is skipped: gn_final_kernel overwrites its solve stamps).  Read the shares, not the totals:
a stamp makes its workgroup wait for what the shipped kernel lets overlap.

    python tools/gn_iter_probe.py [width height]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = "/tmp/libmmf_stamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC",
                "-shared", "-DMMF_STAMPS", "-o", LIB, os.path.join(ROOT, "multimotionfusion_amd/csrc/mmf_hip.hip")],
               check=True)
os.environ["MMF_HIP_LIB"] = LIB

import numpy as np  # noqa: E402
import torch  # noqa: E402
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.odometry import RGBDOdometry  # noqa: E402

# slot order along a workgroup's timeline
ORDER = [5, 0, 3, 4, 1, 2, 14, 15, 8, 9, 6, 10, 11, 12, 13]
NAMES = ["index math + image loads issued", "records landed (barrier)", "record tree + combine", "(lane 0 starts)",
         "6x6 LDLT", "rodrigues + resultRt", "pose compose + K R K^-1", "pose -> LDS -> SGPRs", "windows, warp, ICP projection",
         "gathers landed + accept", "count: wave + block sum, arrive", "ICP rows", "count barrier (poll)",
         "photometric rows + reduce + record"]


def probe(W, H):
    K = synth.intrinsics(W, H)
    poses = synth.trajectory(2, seed=1)
    fp, fc = synth.render(poses[0], W, H, seed=0), synth.render(poses[1], W, H, seed=1)
    ctx = Context(0)
    raw = C.CDLL(LIB)
    g = RGBDOdometry(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    pose = poses[0].astype(np.float32)
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    for rep in range(3):
        g.initFirstRGB(dev(fp["rgb"]))
        g.initICPModel(dev(fp["vertex"]), dev(fp["normal"]), 15.0, pose)
        g.initRGBModel(dev(fp["rgb"]))
        g.buildDepthPyramid(dev(fc["depth"]))
        g.initICP(depthCutoff=15.0)
        g.initRGB(dev(fc["rgb"]))
        if rep == 2:
            assert raw.mmf_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
        g.getIncrementalTransformation(pose[:3, 3], pose[:3, :3], False, 10.0, True, False, True)
    torch.cuda.synchronize()
    raw.mmf_debug_set_stamps(C.c_void_p(0))
    raw_s = stamps.cpu().numpy().reshape(-1, 16)
    nb = int(np.count_nonzero(raw_s[:, 13]))  # workgroups of the last launch (the host picks the geometry per level)
    s = raw_s[1:nb][:, ORDER]
    t0 = s[:, 0].min()
    ph = np.diff(s, axis=1) * 0.01  # us
    print(f"{W}x{H}: {nb} workgroups; first start .. last end {(s[:, -1].max() - t0) * 0.01:.2f} us, start spread "
          f"{(s[:, 0].max() - t0) * 0.01:.2f} us, mean workgroup lifetime {((s[:, -1] - s[:, 0]).mean()) * 0.01:.2f} us")
    for n, mean, mx in zip(NAMES, ph.mean(axis=0), ph.max(axis=0)):
        print(f"    {n:42s} mean {mean:5.2f}  max {mx:5.2f} us")
    print(f"    cumulative (mean, from the first start): " + ", ".join(f"{v:.2f}" for v in ((s - t0).mean(axis=0) * 0.01)))
    arrive = (s[:, ORDER.index(10)] - t0) * 0.01
    release = (s[:, ORDER.index(12)] - t0) * 0.01
    q = lambda v: ", ".join(f"{np.percentile(v, p):.2f}" for p in (0, 50, 90, 99, 100))  # noqa: E731
    print(f"    arrival at the count barrier (us after the first start; min, p50, p90, p99, max): {q(arrive)}; release: {q(release)}")
    for k, name in ((ORDER.index(3), "records landed"), (ORDER.index(8), "pose known"), (ORDER.index(6), "gathers landed"), (ORDER.index(13), "end")):
        print(f"    {name} (min, p50, p90, p99, max): {q((s[:, k] - t0) * 0.01)}")
    if nb > 256:  # blocks b and b + 256 plausibly share a CU
        idx = np.arange(1, nb)
        shared = (idx < nb - 256) | (idx >= 256)
        print(f"    arrival, workgroups that share a CU (by index) vs alone: {arrive[shared].mean():.2f} vs {arrive[~shared].mean():.2f} us; "
              f"start: {((s[:, 0] - t0) * 0.01)[shared].mean():.2f} vs {((s[:, 0] - t0) * 0.01)[~shared].mean():.2f}")
    g.close()
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        probe(int(sys.argv[1]), int(sys.argv[2]))
    else:
        probe(640, 480)
        probe(160, 120)
