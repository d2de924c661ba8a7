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

# stamp slots (thread 0 = the solver wave, thread 64 = the first pixel wave) and what they mark
SLOTS = [(5, "solver: start"), (0, "solver: state loaded"), (3, "solver: barrier A passed (records summed by the pixel waves)"),
         (4, "solver: totals + combined system"), (1, "solver: lane 0 starts"), (2, "solver: 6x6 LDLT done"),
         (14, "solver: rodrigues + resultRt done"), (15, "solver: pose compose + K R K^-1 done"),
         (8, "solver: barrier B passed (pose in LDS)"), (9, "pixel: warp done, photometric gathers issued"),
         (6, "pixel: ICP projection + gathers issued, accept done"), (7, "pixel: barrier C passed (counts in LDS)"),
         (10, "solver: arrived at the count barrier"), (11, "pixel: ICP rows done"), (12, "solver: barrier D passed (sigma known)"),
         (13, "solver: record stored")]


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
    s = raw_s[1:nb]
    t0 = s[:, 5].min()
    print(f"{W}x{H}: {nb} workgroups; first start .. last end {(s[:, 13].max() - t0) * 0.01:.2f} us, start spread "
          f"{(s[:, 5].max() - t0) * 0.01:.2f} us")
    q = lambda v: " ".join(f"{np.percentile(v, p):6.2f}" for p in (0, 50, 90, 100))  # noqa: E731
    print("    us after the first workgroup's start:                               min    p50    p90    max")
    for slot, name in sorted(SLOTS, key=lambda sn: np.median(s[:, sn[0]])):
        if np.count_nonzero(s[:, slot]) == 0:
            continue
        print(f"    {name:66s} {q((s[:, slot] - t0) * 0.01)}")
    g.close()
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) >= 3:
        probe(int(sys.argv[1]), int(sys.argv[2]))
    else:
        probe(640, 480)
        probe(160, 120)
