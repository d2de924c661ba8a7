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

# 3. the surfel path on a 96x72 two-frame cycle: bilateral filter, initialise, predictIndices, fuse,
#    predictIndices, clean, combinedPredict (the per-frame order of MultiMotionFusion::processFrame)
from multimotionfusion_amd import synth  # noqa: E402

w, h = 96, 72
MAXD, CUTOFF, TIME_DELTA, CONF = 20.0, 15.0, 200, 10.0
K = synth.intrinsics(w, h)
poses = synth.trajectory(2, seed=5)
f0, f1 = synth.render(poses[0], w, h, seed=0), synth.render(poses[1], w, h, seed=1)
mask = np.zeros((h, w), np.uint8)
fil0, fil1 = orc.bilateral_filter(f0["depth"], CUTOFF), orc.bilateral_filter(f1["depth"], CUTOFF)
s0 = orc.surfel_initialise(f0["rgb"], f0["depth"], fil0, K, 1, MAXD)
pose1 = poses[1].astype(np.float32)
index, vc, ct, nr = orc.predict_indices(s0, pose1, K, w, h, MAXD, 2, TIME_DELTA)
s_upd, new = orc.fuse(s0, f1["rgb"], f1["depth"], fil1, mask, index, vc, nr, pose1, K, 2, 1.0, 0, MAXD)
index2, vc2, ct2, nr2 = orc.predict_indices(s_upd, pose1, K, w, h, MAXD, 2, TIME_DELTA)
s1 = orc.clean(s_upd, new, pose1, K, w, h, 2, TIME_DELTA, CONF, 3.0, 0, index2, vc2, ct2, fil1, mask)
s1c = s1.copy()
s1c[:, 3] = 20.0  # confident, so the splat shows them
image, vcp, nrp, tm = orc.combined_predict(s1c, pose1, K, w, h, MAXD, CONF, 2, 2, TIME_DELTA)
np.savez_compressed(os.path.join(HERE, "surfel_cycle_96x72.npz"), rgb0=f0["rgb"], depth0=f0["depth"], rgb1=f1["rgb"],
                    depth1=f1["depth"], pose0=poses[0].astype(np.float32), pose1=pose1,
                    intr=np.array([K["fx"], K["fy"], K["cx"], K["cy"]], np.float32), filtered1=fil1,
                    surfels_init=s0, index_after_fuse=index2, surfels_final=s1, splat_vertexConf=vcp,
                    splat_image=image)
print("wrote surfel fixture:", s0.shape[0], "->", s1.shape[0], "surfels")

# 4. SuperPoint on a 64x48 image (weights from the seeded generator, not stored: 5 MB) and the super-pixel
#    resampling of a 160x128 map
import hashlib  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tests"))
from helpers import slic_like_labels  # noqa: E402

rng = np.random.default_rng(2024)
img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
weights = orc.sp_random_weights(seed=4)
wsum = hashlib.sha256(b"".join(a.tobytes() for w, b in weights for a in (w, b))).hexdigest()
semi, desc = orc.sp_forward(orc.sp_input(img), weights)
heat = orc.sp_heatmap(semi)
xy, conf = orc.sp_keypoints(heat)
kdesc = orc.sp_sample_descriptors(desc, xy, 48, 64)
labels = slic_like_labels(160, 128, 16, seed=3, empty_every=7)
smap = rng.random((128, 160), dtype=np.float32)
sdepth = smap * 3.0
sdepth[rng.random((128, 160)) < 0.3] = 0.0
srgb = rng.integers(0, 256, (128, 160, 3), dtype=np.uint8)
np.savez_compressed(os.path.join(HERE, "superpoint_slic.npz"), image=img, weights_seed=4, weights_sha256=wsum, semi=semi,
                    desc=desc, heat=heat, xy=xy, conf=conf, kdesc=kdesc, labels=labels, smap=smap, sdepth=sdepth, srgb=srgb,
                    low=orc.slic_downsample(labels, 16, smap), low_depth=orc.slic_downsample(labels, 16, sdepth, threshold=0.02),
                    low_rgb=orc.slic_downsample_rgb(labels, 16, srgb))
print("wrote superpoint_slic.npz")
