"""Generates the small golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no vectors for this path and cannot run here (CUDA + OpenGL), so these are
outputs of oracle/mmf_oracle.c on seeded synthetic inputs: they pin the ORACLE, not the
reference ("parity unpinned").  Re-run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from helpers import ANGLE_THRESH, DIST_THRESH, frame_pair  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()

# 1. one ICP reduction on a 64x48 pair
w, h = 64, 48
K, prev, cur, fp, fc = frame_pair(w, h)
vm = orc.create_vmap(fc["depth"], K["fx"], K["fy"], K["cx"], K["cy"], 15.0)
nm = orc.create_nmap(vm)
vg, ng = orc.copy_maps(fp["vertex"], fp["normal"])
pose = prev.astype(np.float32)
vg, ng = orc.transform_maps(vg, ng, pose[:3, :3], pose[:3, 3])
Rp = pose[:3, :3]
tp = pose[:3, 3]
Rpi = np.linalg.inv(Rp).astype(np.float32)
intr = np.array([K["fx"], K["fy"], K["cx"], K["cy"]], np.float32)
out, err = orc.icp_step(Rp, tp, vm, nm, Rpi, tp, *intr, vg, ng, DIST_THRESH, ANGLE_THRESH, want_err=True)
np.savez_compressed(os.path.join(HERE, "icp_pair_64x48.npz"), Rcurr=Rp, tcurr=tp, Rprev_inv=Rpi, tprev=tp, intr=intr,
                    vmap_curr=vm, nmap_curr=nm, vmap_g_prev=vg, nmap_g_prev=ng, out29=out, err_map=err)

# 2. a whole getIncrementalTransformation on a 160x120 pair
w, h = 160, 120
K, prev, cur, fp, fc = frame_pair(w, h)
o = orc.Odometry(w, h, K["cx"], K["cy"], K["fx"], K["fy"])
pose = prev.astype(np.float32)
o.initFirstRGB(fp["rgb"])
o.initICPModel(fp["vertex"], fp["normal"], pose)
o.initRGBModel(fp["rgb"])
o.initICP(fc["depth"], 15.0)
o.initRGB(fc["rgb"])
t, R = o.getIncrementalTransformation(pose[:3, 3], pose[:3, :3], False, 10.0, True, False, True)
s = o.stats()
np.savez_compressed(os.path.join(HERE, "odometry_160x120.npz"), rgb_prev=fp["rgb"], rgb_cur=fc["rgb"],
                    vertex_prev=fp["vertex"], normal_prev=fp["normal"], depth_cur=fc["depth"], pose_prev=pose,
                    pose_gt=cur.astype(np.float32), trans=t, rot=R, lastA=np.array(s.lastA), lastb=np.array(s.lastb),
                    lastICPCount=s.lastICPCount, lastICPError=s.lastICPError, lastRGBCount=s.lastRGBCount)
print("wrote fixtures to", HERE)
