"""Synthetic RGB-D scene of SURVEY.md section 8(d): analytic ray-cast of a room (5 planes) with
spheres, procedural texture, known SE3 camera motion.  Pure numpy; used by tests/, bench.py and
the golden-fixture generator (no dataset exists in the build environment).

Pose convention: `pose` is the 4x4 camera-to-world transform (what Model::getPose() holds in the
reference: tranformMaps(pose) lifts the predicted camera-frame maps to the global frame,
RGBDOdometry.cpp:164-172).
"""
import numpy as np

DEFAULT_INTRINSICS = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)  # GUI/MainController.cpp:147-148

# room: x in [-2.5, 2.5], y in [-1.5, 1.5], front wall at z = 3.5 (camera starts at the origin, looks +z)
_PLANES = [  # (normal pointing into the room, offset d with n.p + d = 0)
    (np.array([1.0, 0.0, 0.0]), 2.5),
    (np.array([-1.0, 0.0, 0.0]), 2.5),
    (np.array([0.0, 1.0, 0.0]), 1.5),
    (np.array([0.0, -1.0, 0.0]), 1.5),
    (np.array([0.0, 0.0, -1.0]), 3.5),
]
_SPHERES = [  # (centre, radius)
    (np.array([-0.8, 0.6, 2.2]), 0.45),
    (np.array([0.9, -0.3, 2.6]), 0.55),
    (np.array([0.1, 0.9, 1.8]), 0.30),
]


def intrinsics(width=640, height=480):
    s = width / 640.0
    return dict(fx=528.0 * s, fy=528.0 * s, cx=320.0 * s, cy=240.0 * s)


def rodrigues(rvec):
    rvec = np.asarray(rvec, np.float64)
    th = np.linalg.norm(rvec)
    if th < 1e-12:
        return np.eye(3)
    k = rvec / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_pose(rvec=(0, 0, 0), t=(0, 0, 0)):
    T = np.eye(4)
    T[:3, :3] = rodrigues(rvec)
    T[:3, 3] = t
    return T


def trajectory(n_frames, seed=1, trans_mm=5.0, rot_deg=0.5):
    """Camera poses with uniform increments of +-trans_mm / +-rot_deg per frame."""
    rng = np.random.RandomState(seed)
    poses = [np.eye(4)]
    for _ in range(n_frames - 1):
        dt = rng.uniform(-trans_mm, trans_mm, 3) * 1e-3
        dr = np.deg2rad(rng.uniform(-rot_deg, rot_deg, 3))
        poses.append(poses[-1] @ make_pose(dr, dt))
    return poses


def _hash01(ix, iy, seed):
    """Cheap integer hash -> [0,1) per pixel (fixed-seed dropout / texture noise)."""
    h = (ix.astype(np.uint64) * np.uint64(73856093)) ^ (iy.astype(np.uint64) * np.uint64(19349663)) ^ np.uint64(
        (seed * 83492791) & 0xFFFFFFFF)
    h = (h * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    return (h & np.uint64(0xFFFFFF)).astype(np.float64) / float(1 << 24)


def _texture(p, prim):
    """Procedural albedo in [0,1] x 3 from world position and primitive id."""
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    base = 0.5 + 0.17 * np.sin(7.0 * x + 1.3 * prim) + 0.15 * np.sin(9.0 * y + 0.7 * prim) + 0.13 * np.sin(
        11.0 * z + 2.1)
    checker = ((np.floor(x * 6) + np.floor(y * 6) + np.floor(z * 6)) % 2) * 0.25
    is_sphere = prim >= len(_PLANES)
    v = np.where(is_sphere, 0.35 + checker + 0.1 * np.sin(13.0 * x), base)
    is_object = prim >= len(_PLANES) + len(_SPHERES)  # moving bodies: finer pattern (they are 0.2-0.3 m across)
    # smooth, band-limited pattern: a point-sampled checker has no usable image gradient for the photometric term
    fine = 0.2 * np.sin(31.0 * x + 0.9 * prim) * np.sin(27.0 * y + 1.7) + 0.15 * np.sin(23.0 * z + 19.0 * x + 0.3 * prim)
    v = np.where(is_object, 0.5 + fine + 0.1 * np.sin(11.0 * y - 7.0 * z), v)
    r = np.clip(v, 0.04, 1.0)
    g = np.clip(v * 0.9 + 0.08 * np.sin(5.0 * x + 3.0 * y), 0.04, 1.0)
    b = np.clip(v * 0.8 + 0.1 * np.cos(4.0 * z - 2.0 * y), 0.04, 1.0)
    return np.stack([r, g, b], -1)


def make_objects(n, seed=2, kinds=("box",)):
    """n rigid objects placed in front of the start camera, between 1.3 and 2.2 m, on a jittered grid so that they do
    not overlap in the image ("1-8 spheres/boxes").  Each is a dict {kind, centre (world, at frame 0),
    size (radius | half extents), orient (4x4 rotation about the centre)}; object k carries the id k + 1.
    Default: boxes turned so that three faces are visible -- a lone sphere leaves the rotation about its centre
    unobservable to the point-to-plane ICP (three zero eigenvalues in J^T J), which no rigid tracker survives."""
    rng = np.random.RandomState(seed)
    cols = 4 if n > 2 else max(n, 1)
    rows = (n + cols - 1) // cols
    objs = []
    for k in range(n):
        gx, gy = k % cols, k // cols
        z = 1.3 + 0.9 * rng.uniform()
        # spread over +-32 deg horizontally, +-20 deg vertically
        ax = np.deg2rad(-30.0 + 60.0 * (gx + 0.5) / cols + rng.uniform(-3, 3))
        ay = np.deg2rad(-17.0 + 34.0 * (gy + 0.5) / max(rows, 1) + rng.uniform(-3, 3)) if rows > 1 else np.deg2rad(rng.uniform(-8, 8))
        c = np.array([np.tan(ax) * z, np.tan(ay) * z, z])
        kind = kinds[k % len(kinds)]
        R0 = make_pose(np.deg2rad([25.0 + 20.0 * rng.uniform(), 30.0 + 25.0 * rng.uniform(), 40.0 * rng.uniform()]))
        orient = make_pose(t=c) @ R0 @ make_pose(t=-c)
        if kind == "sphere":
            objs.append(dict(kind="sphere", centre=c, size=np.array([0.13 + 0.06 * rng.uniform()] * 3), orient=orient))
        else:
            objs.append(dict(kind="box", centre=c, size=0.12 + 0.06 * rng.uniform(size=3), orient=orient))
    return objs


def object_trajectories(objects, n_frames, seed=2, trans_mm=3.0, rot_deg=0.3):
    """Per object and frame the 4x4 that moves the object's frame-0 geometry to its place at that frame (rotation
    about the object's own centre): uniform increments of +-trans_mm / +-rot_deg per frame, seed 2 + id."""
    out = []
    for k, ob in enumerate(objects):
        rng = np.random.RandomState(seed + k + 1)
        C, Ci = make_pose(t=ob["centre"]), make_pose(t=-ob["centre"])
        Ts = [np.eye(4)]
        for _ in range(n_frames - 1):
            dt = rng.uniform(-trans_mm, trans_mm, 3) * 1e-3
            dr = np.deg2rad(rng.uniform(-rot_deg, rot_deg, 3))
            Ts.append(C @ make_pose(dr, dt) @ Ci @ Ts[-1])
        out.append(Ts)
    return out


def _object_hit(ob, T, t_cam, d_w):
    """Ray (t_cam + s d_w) against object `ob` moved by T: (s, world normal, object-frame point), s = inf on a miss."""
    T = T @ ob.get("orient", np.eye(4))
    Ti = np.linalg.inv(T)
    o = Ti[:3, :3] @ t_cam + Ti[:3, 3]
    d = d_w @ Ti[:3, :3].T
    c, h = ob["centre"], ob["size"]
    if ob["kind"] == "sphere":
        oc = o - c
        a = np.sum(d * d, -1)
        b = 2.0 * (d @ oc)
        cc = oc @ oc - h[0] * h[0]
        disc = b * b - 4 * a * cc
        with np.errstate(invalid="ignore"):
            s = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
        s = np.where(s > 1e-6, s, np.inf)
        p = o + d * np.where(np.isfinite(s), s, 0.0)[..., None]
        n_o = (p - c) / h[0]
    else:  # axis-aligned box in the object frame: slab test
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            t0 = ((c - h) - o) * inv
            t1 = ((c + h) - o) * inv
        tn, tf = np.minimum(t0, t1), np.maximum(t0, t1)
        axis = np.argmax(tn, -1)
        s_near, s_far = np.max(tn, -1), np.min(tf, -1)
        s = np.where((s_near <= s_far) & (s_near > 1e-6), s_near, np.inf)
        p = o + d * np.where(np.isfinite(s), s, 0.0)[..., None]
        n_o = np.zeros(d.shape)
        sign = -np.sign(np.take_along_axis(d, axis[..., None], -1))[..., 0]
        np.put_along_axis(n_o, axis[..., None], sign[..., None], -1)
    return s, n_o @ T[:3, :3].T, p


def render(pose, width=640, height=480, seed=0, noise=True, dropout=0.03, K=None, depth_noise=5e-5, objects=None,
           object_poses=None):
    """Ray-cast the scene from `pose`.

    Returns dict with
      depth   float32 [H,W] metres (z-depth), 0 = invalid
      rgb     uint8   [H,W,3]
      vertex  float32 [H,W,4] camera-frame point + confidence (the splat prediction format)
      normal  float32 [H,W,4] camera-frame normal + radius
      ids     uint8   [H,W] ground-truth object id (0 = static scene, k + 1 = objects[k])
    objects / object_poses: moving rigid bodies (make_objects) and, per object, the 4x4 of this frame
    (object_trajectories): config 4 / 5 of BASELINE.json ("8 rigid objects, mask = ground-truth ids").
    `depth`/`rgb` carry sensor noise and dropout; `vertex`/`normal` are the clean model prediction.
    Dropout zeroes 8x8 pixel patches (invalid depth comes in blobs on real sensors; the
    reference's depth pyramid averages zeros in, cudafuncs.cu:356, so isolated zero pixels would
    corrupt every coarse level).  depth_noise: sigma = depth_noise * z^2 metres; the default
    stands for depth AFTER the reference's bilateral filter (MultiMotionFusion.cpp:897-904).
    """
    K = K or intrinsics(width, height)
    pose = np.asarray(pose, np.float64)
    R, t = pose[:3, :3], pose[:3, 3]
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    d_cam = np.stack([(u - K["cx"]) / K["fx"], (v - K["cy"]) / K["fy"], np.ones_like(u)], -1)
    d_w = d_cam @ R.T
    best_t = np.full(u.shape, np.inf)
    best_n = np.zeros(u.shape + (3,))
    best_prim = np.zeros(u.shape, np.int64)
    for i, (n, off) in enumerate(_PLANES):
        denom = d_w @ n
        num = -(t @ n + off)
        with np.errstate(divide="ignore", invalid="ignore"):
            tt = np.where(denom < -1e-9, num / denom, np.inf)
        hit = (tt > 1e-6) & (tt < best_t)
        best_t = np.where(hit, tt, best_t)
        best_n = np.where(hit[..., None], n, best_n)
        best_prim = np.where(hit, i, best_prim)
    for j, (c, rad) in enumerate(_SPHERES):
        oc = t - c
        a = np.sum(d_w * d_w, -1)
        b = 2.0 * (d_w @ oc)
        cc = oc @ oc - rad * rad
        disc = b * b - 4 * a * cc
        with np.errstate(invalid="ignore"):
            tt = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
        hit = (tt > 1e-6) & (tt < best_t)
        p = t + d_w * np.where(np.isfinite(tt), tt, 0.0)[..., None]
        n = (p - c) / rad
        best_t = np.where(hit, tt, best_t)
        best_n = np.where(hit[..., None], np.nan_to_num(n), best_n)
        best_prim = np.where(hit, len(_PLANES) + j, best_prim)

    ids = np.zeros(u.shape, np.uint8)
    p_tex = None
    if objects:  # moving rigid bodies: the texture is evaluated in the object's own frame, so it moves with it
        p_tex = t + d_w * np.where(np.isfinite(best_t), best_t, 0.0)[..., None]
        for k, ob in enumerate(objects):
            T = np.eye(4) if object_poses is None else np.asarray(object_poses[k], np.float64)
            tt, n_w, p_o = _object_hit(ob, T, t, d_w)
            hit = tt < best_t
            best_t = np.where(hit, tt, best_t)
            best_n = np.where(hit[..., None], np.nan_to_num(n_w), best_n)
            best_prim = np.where(hit, len(_PLANES) + len(_SPHERES) + k, best_prim)
            p_tex = np.where(hit[..., None], p_o, p_tex)
            ids = np.where(hit, np.uint8(k + 1), ids)

    valid = np.isfinite(best_t)
    z = np.where(valid, best_t, 0.0)  # d_cam.z == 1 so the ray parameter is the z-depth
    p_w = t + d_w * z[..., None]
    p_c = d_cam * z[..., None]
    n_c = best_n @ R  # R^T n

    vertex = np.concatenate([p_c, np.ones_like(z)[..., None]], -1).astype(np.float32)
    radius = (z / K["fx"] * np.sqrt(2.0)).astype(np.float32)
    normal = np.concatenate([n_c, radius[..., None]], -1).astype(np.float32)
    vertex[~valid] = 0
    normal[~valid] = 0

    albedo = _texture(p_w if p_tex is None else p_tex, best_prim)
    iy, ix = np.indices(u.shape)
    rgb_f = albedo * 255.0
    depth = z.copy()
    if noise:
        rng = np.random.RandomState(0x4D4D46 + seed)
        depth = depth + rng.normal(0.0, 1.0, depth.shape) * depth_noise * depth * depth
        rgb_f = rgb_f + (_hash01(ix, iy, seed + 17)[..., None] * 8.0 - 4.0)
    if dropout > 0:
        depth = np.where(_hash01(ix // 8, iy // 8, 7) < dropout, 0.0, depth)
    depth = np.where(valid, depth, 0.0)
    rgb = np.clip(np.rint(rgb_f), 1, 255).astype(np.uint8)
    return dict(depth=depth.astype(np.float32), rgb=rgb, vertex=vertex, normal=normal, ids=ids)


def rotation_angle(Ra, Rb):
    c = (np.trace(np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))
