"""Diagnostic (GPU box): where does one rgb_step_kernel launch spend its time?

Builds an instrumented copy of the library (-DMMF_STAMPS: thread 0 of every workgroup stamps the
100 MHz constant clock at each phase boundary), runs one getIncrementalTransformation and reads the
stamps of the LAST rgb_step launch (finest level, last iteration).  Read the shares, not the total:
the stamps force the loads to land where the shipped kernel lets them overlap.

    python tools/rgb_step_probe.py [width height]
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

NAMES = ["(unused)", "(unused)", "records + state + gathers", "rows", "reduce+arrive"]


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
    s = stamps.cpu().numpy().reshape(-1, 16)
    nb = (W * H + 1023) // 1024
    s = s[:nb]
    t0 = s[:, 0].min()
    solve = s[:, [6, 1, 2, 14, 15, 7]].copy()  # inside the finishing lane's solve (slots 1, 2, 14, 15)
    s[:, 1] = s[:, 0]
    s[:, 2] = s[:, 0]  # phase 2 spans start .. gathers landed + state consumed
    ph = np.diff(s[:, :6], axis=1) * 0.01  # us
    last = int(np.argmax(s[:, 7]))
    print(f"{W}x{H}: {nb} workgroups; first start .. last arrive {(s[:, 5].max() - t0) * 0.01:.2f} us, "
          f"start spread {(s[:, 0].max() - t0) * 0.01:.2f} us")
    print("  mean per-workgroup phase (us): " + ", ".join(f"{n} {v:.2f}" for n, v in list(zip(NAMES[:5], ph.mean(axis=0)))[2:]))
    print(f"  finishing workgroup {last}: record sums {(s[last, 6] - s[last, 5]) * 0.01:.2f} us, solve + stores issued "
          f"{(s[last, 7] - s[last, 6]) * 0.01:.2f} us; kernel first start .. finish {(s[last, 7] - t0) * 0.01:.2f} us")
    sv = np.diff(solve[last]) * 0.01
    print("  solve of the finishing lane (us): " + ", ".join(f"{n} {v:.2f}" for n, v in zip(
        ["unpack + combine A, b", "6x6 LDLT", "rodrigues + resultRt", "pose compose + next K R K^-1", "stores issued"], sv)))
    # correspondence-pass workgroups of the last producer launch: slots 8..13 of workgroups [0, N/1024)
    full = stamps.cpu().numpy().reshape(-1, 16)
    r = full[:nb, 8:14]  # the correspondence workgroups take the first block indices
    r0 = r[:, 0].min()
    rp = np.diff(r, axis=1) * 0.01
    print(f"  residual workgroups of the last producer launch: first start .. last end {(r[:, 5].max() - r0) * 0.01:.2f} us; "
          "mean phase (us): " + ", ".join(f"{n} {v:.2f}" for n, v in zip(
              ["image loads", "warp", "gathers", "evaluate+store", "sum+publish"], rp.mean(axis=0)))
          + f"; max phase: " + ", ".join(f"{v:.2f}" for v in rp.max(axis=0)))
    g.close()
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 2:
        probe(int(sys.argv[1]), int(sys.argv[2]))
    else:
        probe(640, 480)
        probe(160, 120)
