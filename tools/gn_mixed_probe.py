"""Diagnostic (GPU box): phase stamps of the LAST level-0 launch of a multi-model one-launch chain (gn_iter_mixed_kernel), split into
the camera model's workgroups (dense walk) and the object models' (walked by their extents).  Builds an instrumented library
(-DMMF_STAMPS); MMF_DBG_NO_ERR=1 in that build leaves the error images out so that the last launch is an ordinary one.
    python tools/gn_mixed_probe.py [models]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = "/tmp/libmmf_stamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC",
                "-shared", "-DMMF_STAMPS", "-o", LIB, os.path.join(ROOT, "multimotionfusion_amd/csrc/mmf_hip.hip")] +
               os.environ.get("MMF_PROBE_FLAGS", "").split(), check=True)
os.environ["MMF_HIP_LIB"] = LIB
if os.environ.get("MMF_PROBE_ERR", "") != "1":
    os.environ["MMF_DBG_NO_ERR"] = "1"

import numpy as np  # noqa: E402
import torch  # noqa: E402
from multimotionfusion_amd import synth  # noqa: E402
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402

SLOTS = [(5, "solver: start"), (0, "solver: state loaded"), (3, "solver: previous sums in LDS"), (4, "solver: combined system"),
         (8, "solver: barrier B passed (pose in LDS)"), (9, "pixel: photometric gathers issued"), (6, "pixel: accept done"),
         (7, "pixel: barrier C passed (counts in LDS)"), (10, "solver: arrived at the count barrier"), (11, "pixel: ICP rows done"),
         (12, "solver: barrier D passed (sigma known)"), (13, "solver: sums added")]
models = int(sys.argv[1]) if len(sys.argv) > 1 else 2
W, H, nf = 640, 480, 10
K = synth.intrinsics(W, H)
poses = synth.trajectory(nf, seed=1)
objs = synth.make_objects(7, seed=2)
traj = synth.object_trajectories(objs, nf, seed=2)
frames = [synth.render(p, W, H, seed=i, objects=objs, object_poses=[t[i] for t in traj]) for i, p in enumerate(poses)]
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
rgb, depth = [up(f["rgb"]) for f in frames], [up(f["depth"]) for f in frames]
mask = [up(np.where(f["ids"] < models, f["ids"], 0).astype(np.uint8)) for f in frames]
ctx = Context(0)
raw = C.CDLL(LIB)
g = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], enable_multiple_models=1, preallocated_models=models - 1)
stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
n = models + 12
for i in range(n):
    k = i % nf
    if i == n - 1:
        torch.cuda.synchronize()
        assert raw.mmf_debug_set_stamps(C.c_void_p(stamps.data_ptr())) == 0
    g.processFrame(rgb[k], depth[k], timestamp=i, mask=mask[k], hasNewLabel=1 <= i < models)
torch.cuda.synchronize()
raw.mmf_debug_set_stamps(C.c_void_p(0))
s = stamps.cpu().numpy().reshape(-1, 16)
nb = int(np.count_nonzero(s[:, 13]))
print(f"{models} models: {nb} workgroups stamped their last slot; by extent:", [g.getModelOdometry(m).sparseWalk()[1] for m in range(models)])
s = s[:nb]
t0 = s[:, 0][s[:, 0] > 0].min()  # the first "state loaded" of the launch
q = lambda v: " ".join(f"{np.percentile(v, p):6.2f}" for p in (0, 50, 90, 100))  # noqa: E731
cam = -(-(W * H // 5) // 192)  # the camera model's workgroups in a launch that carries object models: 192 pixel lanes x 5 pixels
obj_first = os.environ.get("MMF_GN_OBJ_FIRST", "1") != "0"  # (tunables.hpp: the object models' workgroups are dispatched first)
cam_wgs, obj_wgs = (s[nb - cam:], s[:nb - cam]) if obj_first else (s[:cam], s[cam:])
for name, sel in ((f"camera model ({cam} workgroups, dispatched {'last' if obj_first else 'first'})", cam_wgs), ("object models", obj_wgs)):
    if len(sel) == 0:
        continue
    print(f"  {name}: us after the launch's first 'state loaded':                     min    p50    p90    max")
    for slot, what in sorted(SLOTS, key=lambda sn: np.median(sel[:, sn[0]])):
        if np.count_nonzero(sel[:, slot]) == 0:
            continue
        print(f"    {what:66s} {q((sel[:, slot] - t0) * 0.01)}")
if os.environ.get("MMF_PROBE_DUMP"):  # per workgroup, in dispatch order: start, pose in LDS, arrival, sums added
    print("block  state_loaded  B  accept  arrived  sums_added   (us)")
    for b in range(nb):
        print(f"{b:5d} " + " ".join(f"{(s[b, k] - t0) * 0.01:7.2f}" for k in (0, 8, 6, 10, 13)))
g.close()
ctx.close()
