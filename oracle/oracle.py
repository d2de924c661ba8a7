"""ctypes wrapper of the CPU oracle (oracle/mmf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker / reported CPU baseline.  PARITY UNPINNED (see mmf_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_DIR, "liboracle.so")


def build(force=False, march=None, out=None, extra=None, contract=None):
    """make the oracle library (gcc).  `march`/`out` let bench.py build a -march=native copy; `extra` adds compiler
    flags (-DORC_LIBM_EXP: the C library's expf instead of the shared mmf_expf); `contract` = "fast" with an FMA `march`
    builds the contracting variant that tests/test_oracle_contraction.py compares the checker with."""
    target = out or "liboracle.so"
    path = os.path.join(_DIR, target)
    src_m = max(os.path.getmtime(os.path.join(_DIR, f))
                for f in ("mmf_oracle.c", "mmf_oracle_surfel.c", "mmf_oracle_match.c", "mmf_oracle_superpoint.c", "mmf_oracle_slic.c", "mmf_oracle_pose.c", "mmf_oracle.h",
                          "Makefile"))
    if not force and os.path.exists(path) and os.path.getmtime(path) >= src_m:
        return path
    cmd = ["make", "-C", _DIR, f"OUT={target}", "-B"]
    if march:
        cmd.append(f"MARCH={march}")
    if extra:
        cmd.append(f"EXTRA={extra}")
    if contract:
        cmd.append(f"CONTRACT={contract}")
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return path


class use_lib:
    """`with use_lib(path):` -- every wrapper below that takes the default library takes `path` instead (tests that run the
    same steps through two builds of the oracle)."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global LIB
        self.saved, LIB = LIB, self.path
        return lib(self.path)

    def __exit__(self, *exc):
        global LIB
        LIB = self.saved
        return False


class _Dataterm(C.Structure):
    _fields_ = [("zero_x", C.c_int16), ("zero_y", C.c_int16), ("one_x", C.c_int16), ("one_y", C.c_int16),
                ("diff", C.c_float), ("valid", C.c_uint8), ("pad_", C.c_uint8 * 3)]


class OdomStats(C.Structure):
    _fields_ = [("lastICPError", C.c_float), ("lastICPCount", C.c_float), ("lastRGBError", C.c_float),
                ("lastRGBCount", C.c_float), ("lastSO3Error", C.c_float), ("lastSO3Count", C.c_float),
                ("lastA", C.c_double * 36), ("lastb", C.c_double * 6), ("iterations_run", C.c_int),
                ("so3_iterations_run", C.c_int)]


_lib_cache = {}


def lib(path=None):
    path = path or LIB
    if path in _lib_cache:
        return _lib_cache[path]
    if not os.path.exists(path):
        build()
    l = C.CDLL(path)
    l.orc_odom_create.restype = C.c_void_p
    l.orc_odom_buffer_f32.restype = C.POINTER(C.c_float)
    l.orc_odom_buffer_u8.restype = C.POINTER(C.c_uint8)
    l.orc_odom_buffer_i16.restype = C.POINTER(C.c_int16)
    l.orc_omp_threads.restype = C.c_int
    _lib_cache[path] = l
    return l


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pu8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _pi16(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


def _cf(x):
    return C.c_float(float(x))


# ---- map kernels --------------------------------------------------------------------------------
def create_vmap(depth, fx, fy, cx, cy, cutoff):
    depth = _f(depth)
    rows, cols = depth.shape
    out = np.zeros((3 * rows, cols), np.float32)
    lib().orc_create_vmap(_pf(depth), cols, rows, _cf(fx), _cf(fy), _cf(cx), _cf(cy), _cf(cutoff), _pf(out))
    return out


def create_nmap(vmap):
    vmap = _f(vmap)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    out = np.zeros_like(vmap)
    lib().orc_create_nmap(_pf(vmap), cols, rows, _pf(out))
    return out


def transform_maps(vmap, nmap, R, t):
    vmap, nmap = _f(vmap), _f(nmap)
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    vd, nd = vmap.copy(), nmap.copy()
    R, t = _f(np.reshape(R, 9)), _f(np.reshape(t, 3))
    lib().orc_transform_maps(_pf(vmap), _pf(nmap), cols, rows, _pf(R), _pf(t), _pf(vd), _pf(nd))
    return vd, nd


def copy_maps(v_rgba, n_rgba):
    v_rgba, n_rgba = _f(v_rgba), _f(n_rgba)
    rows, cols = v_rgba.shape[:2]
    vd = np.zeros((3 * rows, cols), np.float32)
    nd = np.zeros((3 * rows, cols), np.float32)
    lib().orc_copy_maps(_pf(v_rgba), _pf(n_rgba), cols, rows, _pf(vd), _pf(nd))
    return vd, nd


def resize_map(m, normalize):
    m = _f(m)
    rows, cols = m.shape[0] // 3, m.shape[1]
    out = np.zeros((3 * (rows // 2), cols // 2), np.float32)
    lib().orc_resize_map(_pf(m), cols, rows, int(normalize), _pf(out))
    return out


def pyrdown_gauss_f(src):
    src = _f(src)
    rows, cols = src.shape
    out = np.zeros((rows // 2, cols // 2), np.float32)
    lib().orc_pyrdown_gauss_f(_pf(src), cols, rows, _pf(out))
    return out


def pyrdown_uchar_gauss(src):
    src = np.ascontiguousarray(src, np.uint8)
    rows, cols = src.shape
    out = np.zeros((rows // 2, cols // 2), np.uint8)
    lib().orc_pyrdown_uchar_gauss(_pu8(src), cols, rows, _pu8(out))
    return out


def vertices_to_depth(v_rgba, cutoff):
    v_rgba = _f(v_rgba)
    rows, cols = v_rgba.shape[:2]
    out = np.zeros((rows, cols), np.float32)
    lib().orc_vertices_to_depth(_pf(v_rgba), cols, rows, _cf(cutoff), _pf(out))
    return out


def image_to_intensity(img):
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols, ch = img.shape
    out = np.zeros((rows, cols), np.uint8)
    lib().orc_image_to_intensity(_pu8(img), ch, cols, rows, _pu8(out))
    return out


def derivative_images(src):
    src = np.ascontiguousarray(src, np.uint8)
    rows, cols = src.shape
    dx = np.zeros((rows, cols), np.int16)
    dy = np.zeros((rows, cols), np.int16)
    lib().orc_derivative_images(_pu8(src), cols, rows, _pi16(dx), _pi16(dy))
    return dx, dy


def project_to_cloud(depth, fx, fy, cx, cy):
    depth = _f(depth)
    rows, cols = depth.shape
    out = np.zeros((rows, cols, 3), np.float32)
    lib().orc_project_to_cloud(_pf(depth), cols, rows, _cf(fx), _cf(fy), _cf(cx), _cf(cy), _pf(out))
    return out


# ---- reductions -----------------------------------------------------------------------------------
def icp_step(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, fx, fy, cx, cy, vmap_g_prev, nmap_g_prev,
             dist_thres, angle_thres, want_err=False):
    """Returns (out29 float64, err_map or None)."""
    vc, nc, vp, np_ = _f(vmap_curr), _f(nmap_curr), _f(vmap_g_prev), _f(nmap_g_prev)
    rows, cols = vc.shape[0] // 3, vc.shape[1]
    out = np.zeros(29, np.float64)
    err = np.zeros((rows, cols), np.float32) if want_err else None
    Rc, tc, Rp, tp = _f(np.reshape(Rcurr, 9)), _f(np.reshape(tcurr, 3)), _f(np.reshape(Rprev_inv, 9)), _f(
        np.reshape(tprev, 3))
    lib().orc_icp_step(_pf(Rc), _pf(tc), _pf(vc), _pf(nc), _pf(Rp), _pf(tp), _cf(fx), _cf(fy), _cf(cx), _cf(cy),
                       _pf(vp), _pf(np_), _cf(dist_thres), _cf(angle_thres), cols, rows,
                       out.ctypes.data_as(C.POINTER(C.c_double)), _pf(err) if want_err else None)
    return out, err


def icp_step_omp_f32(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, fx, fy, cx, cy, vmap_g_prev,
                     nmap_g_prev, dist_thres, angle_thres, libpath=None):
    vc, nc, vp, np_ = _f(vmap_curr), _f(nmap_curr), _f(vmap_g_prev), _f(nmap_g_prev)
    rows, cols = vc.shape[0] // 3, vc.shape[1]
    out = np.zeros(29, np.float32)
    Rc, tc, Rp, tp = _f(np.reshape(Rcurr, 9)), _f(np.reshape(tcurr, 3)), _f(np.reshape(Rprev_inv, 9)), _f(
        np.reshape(tprev, 3))
    lib(libpath).orc_icp_step_omp_f32(_pf(Rc), _pf(tc), _pf(vc), _pf(nc), _pf(Rp), _pf(tp), _cf(fx), _cf(fy),
                                      _cf(cx), _cf(cy), _pf(vp), _pf(np_), _cf(dist_thres), _cf(angle_thres), cols,
                                      rows, _pf(out))
    return out


def omp_threads(libpath=None):
    return lib(libpath).orc_omp_threads()


def rgb_residual(min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, max_depth_delta, kt, krkinv,
                 want_err=False):
    """Returns (corres uint8[rows,cols,16], sigma_sum, count, err_map or None)."""
    dIdx, dIdy = np.ascontiguousarray(dIdx, np.int16), np.ascontiguousarray(dIdy, np.int16)
    ld, nd = _f(last_depth), _f(next_depth)
    li, ni = np.ascontiguousarray(last_image, np.uint8), np.ascontiguousarray(next_image, np.uint8)
    rows, cols = ni.shape
    corres = np.zeros((rows, cols, 16), np.uint8)
    sigma, count = C.c_int(0), C.c_int(0)
    err = np.zeros((rows, cols), np.float32) if want_err else None
    ktv, kk = _f(np.reshape(kt, 3)), _f(np.reshape(krkinv, 9))
    lib().orc_rgb_residual(_cf(min_scale), _pi16(dIdx), _pi16(dIdy), _pf(ld), _pf(nd), _pu8(li), _pu8(ni),
                           corres.ctypes.data_as(C.POINTER(_Dataterm)), _cf(max_depth_delta), _pf(ktv), _pf(kk),
                           cols, rows, C.byref(sigma), C.byref(count), _pf(err) if want_err else None)
    return corres, sigma.value, count.value, err


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale):
    corres = np.ascontiguousarray(corres, np.uint8)
    cloud = _f(cloud)
    dIdx, dIdy = np.ascontiguousarray(dIdx, np.int16), np.ascontiguousarray(dIdy, np.int16)
    rows, cols = dIdx.shape
    out = np.zeros(29, np.float64)
    lib().orc_rgb_step(corres.ctypes.data_as(C.POINTER(_Dataterm)), _cf(sigma), _pf(cloud), _cf(fx), _cf(fy),
                       _pi16(dIdx), _pi16(dIdy), _cf(sobel_scale), cols, rows,
                       out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def so3_step(last_image, next_image, image_basis, kinv, krlr):
    li, ni = np.ascontiguousarray(last_image, np.uint8), np.ascontiguousarray(next_image, np.uint8)
    rows, cols = ni.shape
    out = np.zeros(11, np.float64)
    B, ki, kr = _f(np.reshape(image_basis, 9)), _f(np.reshape(kinv, 9)), _f(np.reshape(krlr, 9))
    lib().orc_so3_step(_pu8(li), _pu8(ni), _pf(B), _pf(ki), _pf(kr), cols, rows,
                       out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def unpack_se3(out29):
    out29 = np.ascontiguousarray(out29, np.float64)
    A, b, r = np.zeros(36, np.float32), np.zeros(6, np.float32), np.zeros(2, np.float32)
    lib().orc_unpack_se3(out29.ctypes.data_as(C.POINTER(C.c_double)), _pf(A), _pf(b), _pf(r))
    return A.reshape(6, 6), b, r


def unpack_so3(out11):
    out11 = np.ascontiguousarray(out11, np.float64)
    A, b, r = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)
    lib().orc_unpack_so3(out11.ctypes.data_as(C.POINTER(C.c_double)), _pf(A), _pf(b), _pf(r))
    return A.reshape(3, 3), b, r


# ---- whole odometry object --------------------------------------------------------------------
class Odometry:
    """Oracle restatement of class RGBDOdometry (same method names as the product mirror)."""

    def __init__(self, width, height, cx, cy, fx, fy, distThresh=0.10,
                 angleThresh=float(np.sin(20.0 * 3.14159254 / 180.0))):
        self.width, self.height = width, height
        self.h = C.c_void_p(lib().orc_odom_create(width, height, _cf(cx), _cf(cy), _cf(fx), _cf(fy),
                                                  _cf(distThresh), _cf(angleThresh)))

    def initICP(self, depth_l0, depthCutoff):
        d = _f(depth_l0)
        lib().orc_odom_init_icp(self.h, _pf(d), _cf(depthCutoff))

    def initICPFromPrediction(self, vert_rgba, norm_rgba):
        v, n = _f(vert_rgba), _f(norm_rgba)
        lib().orc_odom_init_icp_from_prediction(self.h, _pf(v), _pf(n))

    def initICPModel(self, vert_rgba, norm_rgba, pose):
        v, n, p = _f(vert_rgba), _f(norm_rgba), _f(np.reshape(pose, 16))
        lib().orc_odom_init_icp_model(self.h, _pf(v), _pf(n), _pf(p))

    def initRGB(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        lib().orc_odom_init_rgb(self.h, _pu8(rgb), rgb.shape[2])

    def initRGBModel(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        lib().orc_odom_init_rgb_model(self.h, _pu8(rgb), rgb.shape[2])

    def initFirstRGB(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        lib().orc_odom_init_first_rgb(self.h, _pu8(rgb), rgb.shape[2])

    def getIncrementalTransformation(self, trans, rot, rgbOnly, icpWeight, pyramid, fastOdom, so3, want_err=False):
        t = _f(np.reshape(trans, 3)).copy()
        r = _f(np.reshape(rot, 9)).copy()
        icp_err = np.zeros((self.height, self.width), np.float32) if want_err else None
        rgb_err = np.zeros((self.height, self.width), np.float32) if want_err else None
        lib().orc_odom_get_incremental_transformation(self.h, _pf(t), _pf(r), int(bool(rgbOnly)), _cf(icpWeight),
                                                      int(bool(pyramid)), int(bool(fastOdom)), int(bool(so3)),
                                                      _pf(icp_err) if want_err else None,
                                                      _pf(rgb_err) if want_err else None)
        self.icp_err, self.rgb_err = icp_err, rgb_err
        return t, r.reshape(3, 3)

    def stats(self):
        s = OdomStats()
        lib().orc_odom_get_stats(self.h, C.byref(s))
        return s

    def buffer(self, name, level):
        cols, rows = self.width >> level, self.height >> level
        if name in ("last_image", "next_image", "last_next_image"):
            p = lib().orc_odom_buffer_u8(self.h, name.encode(), level)
            return np.ctypeslib.as_array(p, (rows, cols)).copy()
        if name in ("dIdx", "dIdy"):
            p = lib().orc_odom_buffer_i16(self.h, name.encode(), level)
            return np.ctypeslib.as_array(p, (rows, cols)).copy()
        p = lib().orc_odom_buffer_f32(self.h, name.encode(), level)
        if name == "cloud":
            return np.ctypeslib.as_array(p, (rows, cols, 3)).copy()
        planes = 3 if name.startswith(("vmaps", "nmaps")) else 1
        return np.ctypeslib.as_array(p, (planes * rows, cols)).copy()

    def close(self):
        if self.h:
            lib().orc_odom_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- surfel path (mmf_oracle_surfel.c) ------------------------------------------------------------------
def _pu32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _surf(a):
    return np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, 12))


def inverse4f(m):
    m = _f(np.reshape(m, 16))
    out = np.zeros(16, np.float32)
    lib().orc_inverse4f(_pf(m), _pf(out))
    return out.reshape(4, 4)


def matmul4f(a, b):
    """Eigen Matrix4f product, float32, terms summed in k order."""
    a, b = _f(np.reshape(a, 16)), _f(np.reshape(b, 16))
    out = np.zeros(16, np.float32)
    lib().orc_matmul4f(_pf(a), _pf(b), _pf(out))
    return out.reshape(4, 4)


def jacobi_svd3f(a):
    """Eigen::JacobiSVD<Matrix3f>(a, ComputeFullU | ComputeFullV) -> U, singular values, V."""
    a = _f(np.reshape(a, 9))
    U, sv, V = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
    lib().orc_jacobi_svd3f(_pf(a), _pf(U), _pf(sv), _pf(V))
    return U.reshape(3, 3), sv, V.reshape(3, 3)


def rodrigues2(R):
    """Model::rodrigues2 (Model.cpp:1301-1342)."""
    R = _f(np.reshape(R, 9))
    out = np.zeros(3, np.float32)
    lib().orc_rodrigues2(_pf(R), _pf(out))
    return out


def compute_fusion_weight(pose, last_pose, weight_multiplier):
    """Model::computeFusionWeight (Model.cpp:876-891)."""
    pose, last_pose = _f(np.reshape(pose, 16)), _f(np.reshape(last_pose, 16))
    f = lib().orc_compute_fusion_weight
    f.restype = C.c_float
    return float(f(_pf(pose), _pf(last_pose), _cf(weight_multiplier)))


def bilateral_filter(depth, maxD):
    depth = _f(depth)
    rows, cols = depth.shape
    out = np.zeros_like(depth)
    lib().orc_bilateral_filter(_pf(depth), cols, rows, _cf(maxD), _pf(out))
    return out


def surfel_initialise(rgb, depth_raw, depth_filtered, K, time, maxDepth):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    dr, df = _f(depth_raw), _f(depth_filtered)
    rows, cols = dr.shape
    out = np.zeros((rows * cols, 12), np.float32)
    lib().orc_surfel_initialise.restype = C.c_int
    n = lib().orc_surfel_initialise(_pu8(rgb), _pf(dr), _pf(df), cols, rows, _cf(K["cx"]), _cf(K["cy"]), _cf(K["fx"]),
                                    _cf(K["fy"]), int(time), _cf(maxDepth), _pf(out))
    return out[:n].copy()


def predict_indices(surfels, pose, K, cols, rows, maxDepth, time, timeDelta):
    s = _surf(surfels)
    pose = _f(np.reshape(pose, 16))
    index = np.zeros((rows, cols), np.uint32)
    vc, ct, nr = (np.zeros((rows, cols, 4), np.float32) for _ in range(3))
    lib().orc_predict_indices(_pf(s), s.shape[0], _pf(pose), _cf(K["cx"]), _cf(K["cy"]), _cf(K["fx"]), _cf(K["fy"]),
                              cols, rows, _cf(maxDepth), int(time), int(timeDelta), _pu32(index), _pf(vc), _pf(ct),
                              _pf(nr))
    return index, vc, ct, nr


def combined_predict(surfels, pose, K, cols, rows, maxDepth, confThreshold, time, maxTime, timeDelta):
    s = _surf(surfels)
    pose = _f(np.reshape(pose, 16))
    image = np.zeros((rows, cols, 4), np.uint8)
    vc, nr = np.zeros((rows, cols, 4), np.float32), np.zeros((rows, cols, 4), np.float32)
    tm = np.zeros((rows, cols), np.uint16)
    lib().orc_combined_predict(_pf(s), s.shape[0], _pf(pose), _cf(K["cx"]), _cf(K["cy"]), _cf(K["fx"]), _cf(K["fy"]),
                               cols, rows, _cf(maxDepth), _cf(confThreshold), int(time), int(maxTime), int(timeDelta),
                               _pu8(image), _pf(vc), _pf(nr), tm.ctypes.data_as(C.POINTER(C.c_uint16)))
    return image, vc, nr, tm


def synthesize_depth(surfels, pose, K, cols, rows, maxDepth, confThreshold, time, maxTime, timeDelta):
    s = _surf(surfels)
    pose = _f(np.reshape(pose, 16))
    depth = np.zeros((rows, cols), np.float32)
    lib().orc_synthesize_depth(_pf(s), s.shape[0], _pf(pose), _cf(K["cx"]), _cf(K["cy"]), _cf(K["fx"]), _cf(K["fy"]),
                               cols, rows, _cf(maxDepth), _cf(confThreshold), int(time), int(maxTime), int(timeDelta),
                               _pf(depth))
    return depth


def fuse(surfels, rgb, depth_raw, depth_filtered, mask, index, vertConf, normRad, pose, K, time, weighting, maskID,
         maxDepth):
    """Returns (updated surfels, new unstable surfels in draw order)."""
    s = _surf(surfels).copy()
    rgb = np.ascontiguousarray(rgb, np.uint8)
    dr, df = _f(depth_raw), _f(depth_filtered)
    mask = np.ascontiguousarray(mask, np.uint8)
    index = np.ascontiguousarray(index, np.uint32)
    vc, nr = _f(vertConf), _f(normRad)
    pose = _f(np.reshape(pose, 16))
    rows, cols = dr.shape
    new = np.zeros((rows * cols, 12), np.float32)
    lib().orc_fuse.restype = C.c_int
    n = lib().orc_fuse(_pf(s), s.shape[0], _pu8(rgb), _pf(dr), _pf(df), _pu8(mask), _pu32(index), _pf(vc), _pf(nr),
                       _pf(pose), _cf(K["cx"]), _cf(K["cy"]), _cf(K["fx"]), _cf(K["fy"]), cols, rows, int(time),
                       _cf(weighting), C.c_uint8(maskID), _cf(maxDepth), _pf(new))
    return s, new[:n].copy()


def clean(surfels, new_unstable, pose, K, cols, rows, time, timeDelta, confThreshold, outlierCoeff, maskID, index,
          vertConf, colorTime, depth_filtered, mask):
    s, nu = _surf(surfels), _surf(new_unstable)
    pose = _f(np.reshape(pose, 16))
    index = np.ascontiguousarray(index, np.uint32)
    vc, ct, df = _f(vertConf), _f(colorTime), _f(depth_filtered)
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros((s.shape[0] + nu.shape[0] + 1, 12), np.float32)
    lib().orc_clean.restype = C.c_int
    n = lib().orc_clean(_pf(s), s.shape[0], _pf(nu), nu.shape[0], _pf(pose), _cf(K["cx"]), _cf(K["cy"]), _cf(K["fx"]),
                        _cf(K["fy"]), cols, rows, int(time), int(timeDelta), _cf(confThreshold), _cf(outlierCoeff),
                        C.c_uint8(maskID), _pu32(index), _pf(vc), _pf(ct), _pf(df), _pu8(mask), _pf(out))
    return out[:n].copy()


def fill_in(vertex_pred, normal_pred, image_pred, depth_filtered, rgb, K, passthrough_geom, passthrough_rgb):
    vp, npd = _f(vertex_pred), _f(normal_pred)
    ip = np.ascontiguousarray(image_pred, np.uint8)
    df = _f(depth_filtered)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    rows, cols = df.shape
    vo, no, io = np.zeros_like(vp), np.zeros_like(npd), np.zeros_like(ip)
    lib().orc_fill_in(_pf(vp), _pf(npd), _pu8(ip), _pf(df), _pu8(rgb), cols, rows, _cf(K["cx"]), _cf(K["cy"]),
                      _cf(K["fx"]), _cf(K["fy"]), int(passthrough_geom), int(passthrough_rgb), _pf(vo), _pf(no),
                      _pu8(io))
    return vo, no, io


def requires_fill_in(image_pred, ratio=0.75):
    ip = np.ascontiguousarray(image_pred, np.uint8)
    rows, cols = ip.shape[:2]
    lib().orc_requires_fill_in.restype = C.c_int
    return bool(lib().orc_requires_fill_in(_pu8(ip), cols, rows, _cf(ratio)))


def match_descriptors(query, train, max_distance=0.0):
    """cv::BFMatcher(NORM_L2, crossCheck=True).match(query, train) + the distance gate of PointTracker.cpp:108.
    Returns (train_idx [nq] int32, -1 = unmatched; distance [nq] float32)."""
    q, t = _f(query), _f(train)
    nq, nt = q.shape[0], t.shape[0]
    dim = q.shape[1] if nq else (t.shape[1] if nt else 0)
    idx = np.full(nq, -1, np.int32)
    dist = np.zeros(nq, np.float32)
    lib().orc_match_descriptors(_pf(q), nq, _pf(t), nt, dim, _cf(max_distance), idx.ctypes.data_as(C.POINTER(C.c_int)),
                                _pf(dist))
    return idx, dist


# ---- SuperPoint (mmf_oracle_superpoint.c) ----------------------------------------------------------------
SP_LAYERS = (("conv1a", 1, 64, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
             ("conv3a", 64, 128, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
             ("convPa", 128, 256, 3), ("convPb", 256, 65, 1), ("convDa", 128, 256, 3), ("convDb", 256, 256, 1))


def sp_random_weights(seed=0, scale=1.0):
    """He-initialised random SuperPointNet weights in PyTorch layout: [(w [Cout,Cin,k,k], b [Cout])] * 12.
    There is no network access for the MagicLeap checkpoint, so tests and the bench run on these."""
    rng = np.random.default_rng(seed)
    out = []
    for _, cin, cout, k in SP_LAYERS:
        std = scale * np.sqrt(2.0 / (cin * k * k))
        out.append((rng.normal(0, std, (cout, cin, k, k)).astype(np.float32),
                    rng.normal(0, 0.05, cout).astype(np.float32)))
    return out


def _sp_weight_ptrs(weights):
    flat = []
    for w, b in weights:
        flat += [np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)]
    arr = (C.POINTER(C.c_float) * 24)(*[_pf(a) for a in flat])
    return arr, flat


def sp_conv(x, w, b, relu):
    """x [H,W,Cin] channels-last, w [Cout,Cin,k,k] -> [H,W,Cout]"""
    x, w, b = _f(x), _f(w), _f(b)
    H, W, cin = x.shape
    cout, taps = w.shape[0], w.shape[2] * w.shape[3]
    out = np.empty((H, W, cout), np.float32)
    lib().orc_sp_conv(_pf(x), H, W, cin, cin, _pf(w), _pf(b), cout, taps, int(relu), _pf(out))
    return out


def sp_input(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((H, W), np.float32)
    lib().orc_sp_input(_pu8(img), H, W, ch, _pf(out))
    return out


def sp_forward(inp, weights):
    """inp [H,W] float32 in [0,1] -> (semi [H/8,W/8,65], desc [H/8,W/8,256] L2-normalised)"""
    inp = _f(inp)
    H, W = inp.shape
    semi = np.empty((H // 8, W // 8, 65), np.float32)
    desc = np.empty((H // 8, W // 8, 256), np.float32)
    arr, keep = _sp_weight_ptrs(weights)
    lib().orc_sp_forward.restype = C.c_int
    rc = lib().orc_sp_forward(_pf(inp), H, W, arr, _pf(semi), _pf(desc))
    if rc:
        raise ValueError("SuperPoint needs an image whose sides are multiples of 8")
    del keep
    return semi, desc


def sp_heatmap(semi):
    semi = _f(semi)
    Hc, Wc = semi.shape[:2]
    heat = np.empty((Hc * 8, Wc * 8), np.float32)
    lib().orc_sp_heatmap(_pf(semi), Hc, Wc, _pf(heat))
    return heat


def sp_keypoints(heat, conf_thresh=0.015, nms_dist=4, border=4):
    heat = _f(heat)
    H, W = heat.shape
    xy = np.empty((H * W, 2), np.int32)
    conf = np.empty(H * W, np.float32)
    lib().orc_sp_keypoints.restype = C.c_int
    n = lib().orc_sp_keypoints(_pf(heat), H, W, _cf(conf_thresh), nms_dist, border, H * W,
                               xy.ctypes.data_as(C.POINTER(C.c_int)), _pf(conf))
    return xy[:n].copy(), conf[:n].copy()


def sp_sample_descriptors(desc, xy, H, W):
    desc = _f(desc)
    xy = np.ascontiguousarray(xy, np.int32)
    Hc, Wc = desc.shape[:2]
    out = np.empty((xy.shape[0], 256), np.float32)
    lib().orc_sp_sample_descriptors(_pf(desc), Hc, Wc, xy.ctypes.data_as(C.POINTER(C.c_int)), xy.shape[0], H, W, _pf(out))
    return out


def sp_get_features(img, weights, conf_thresh=0.015, nms_dist=4, border=4):
    """SuperPoint::getFeatures as MultiMotionFusion.cpp:233 consumes it: (coordinates [n,2] float64 normalised
    to [0,1), descriptors [n,256] float64), strongest keypoint first."""
    inp = sp_input(img)
    H, W = inp.shape
    semi, desc = sp_forward(inp, weights)
    xy, _ = sp_keypoints(sp_heatmap(semi), conf_thresh, nms_dist, border)
    d = sp_sample_descriptors(desc, xy, H, W)
    return xy.astype(np.float64) / np.array([W, H], np.float64), d.astype(np.float64)


# ---- super-pixel resampling (mmf_oracle_slic.c) ---------------------------------------------------------------
def _pi32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def slic_counts(labels, nspix):
    labels = np.ascontiguousarray(labels, np.int32)
    counts = np.empty(nspix, np.int32)
    lib().orc_slic_counts(_pi32(labels), labels.size, nspix, _pi32(counts))
    return counts


def slic_downsample(labels, S, image, channel=0, threshold=None):
    """Slic::downsample<float>(image, channel) / downsampleThresholded<float>(image, threshold) -> [H/S, W/S]"""
    labels = np.ascontiguousarray(labels, np.int32)
    image = _f(image)
    H, W = labels.shape
    ch = 1 if image.ndim == 2 else image.shape[2]
    out = np.empty((H // S, W // S), np.float32)
    lib().orc_slic_downsample(_pi32(labels), W, H, S, _pf(image), ch, channel, int(threshold is not None),
                              _cf(0.0 if threshold is None else threshold), _pf(out))
    return out


def slic_downsample_rgb(labels, S, rgb):
    labels = np.ascontiguousarray(labels, np.int32)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    H, W = labels.shape
    out = np.empty((H // S, W // S, 3), np.uint8)
    lib().orc_slic_downsample_rgb(_pi32(labels), W, H, S, _pu8(rgb), rgb.shape[2], _pu8(out))
    return out


def slic_upsample_u8(labels, small):
    labels = np.ascontiguousarray(labels, np.int32)
    small = np.ascontiguousarray(small, np.uint8)
    out = np.empty(labels.shape, np.uint8)
    lib().orc_slic_upsample_u8(_pi32(labels), labels.size, _pu8(small), _pu8(out))
    return out
