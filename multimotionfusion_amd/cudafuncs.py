"""Host-side mirror of the reference's device entry points (Core/Cuda/cudafuncs.cuh:64-193),
same names and argument meaning, over the C ABI of include/mmf_hip.h.

Device arrays are torch CUDA tensors (torch is only the allocator here): a DeviceArray2D<T> of
the reference is a 2-D contiguous tensor, a vertex/normal map is float32 [3*rows, cols].
Every function enqueues on the context's stream; the *Step functions return host results and
therefore synchronise, exactly like the reference (reduce.cu:452-456).
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import check, fptr, mmf_camera


class Context:
    """Owns an mmf_ctx bound to torch's current stream on `device`."""

    def __init__(self, device=0, use_torch_stream=True):
        if not torch.cuda.is_available():
            raise RuntimeError("multimotionfusion_amd needs a HIP device (gfx950); none is visible")
        self.lib = _capi.load()
        self.device = int(device)
        torch.cuda.set_device(self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream if use_torch_stream else 0
        h = C.c_void_p()
        check(self.lib.mmf_ctx_create(self.device, C.c_void_p(stream), 0 if use_torch_stream else 1, C.byref(h)))
        self.handle = h
        self._children = []  # weakrefs of objects holding device memory of this context

    def synchronize(self):
        check(self.lib.mmf_ctx_synchronize(self.handle))

    def device_name(self):
        buf = C.create_string_buffer(64)
        check(self.lib.mmf_ctx_device_name(self.handle, buf, 64))
        return buf.value.decode()

    def close(self):
        if self.handle:
            for ref in self._children:  # children dereference the context when destroyed
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            self.lib.mmf_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _step(t):
    return t.stride(0) * t.element_size()


def _f32(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    assert a.size == n
    return a


def CameraModel(fx, fy, cx, cy):
    return mmf_camera(fx, fy, cx, cy)


def icpStep(ctx, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, intr, vmap_g_prev, nmap_g_prev, distThres,
            angleThres, icpErrorSurface=None):
    """cudafuncs.cuh:64-82. Returns (A[6,6], b[6], residual[2]) as float32 numpy arrays."""
    rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
    Rc, tc, Rp, tp = _f32(Rcurr, 9), _f32(tcurr, 3), _f32(Rprev_inv, 9), _f32(tprev, 3)
    A = np.zeros(36, np.float32)
    b = np.zeros(6, np.float32)
    res = np.zeros(2, np.float32)
    err = icpErrorSurface
    check(ctx.lib.mmf_icp_step(ctx.handle, fptr(Rc), fptr(tc), _p(vmap_curr), _step(vmap_curr), _p(nmap_curr),
                               _step(nmap_curr), fptr(Rp), fptr(tp), C.byref(intr), _p(vmap_g_prev),
                               _step(vmap_g_prev), _p(nmap_g_prev), _step(nmap_g_prev), distThres, angleThres, cols,
                               rows, fptr(A), fptr(b), fptr(res), _p(err), _step(err) if err is not None else 0))
    return A.reshape(6, 6), b, res


def computeRgbResidual(ctx, minScale, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, corresImg,
                       maxDepthDelta, kt, krkinv, rgbErrorSurface=None):
    """cudafuncs.cuh:113-132. corresImg: uint8 tensor [rows, cols, 16]. Returns (sigmaSum, count)."""
    rows, cols = nextImage.shape
    ktv, kk = _f32(kt, 3), _f32(krkinv, 9)
    sigma, count = C.c_int(0), C.c_int(0)
    err = rgbErrorSurface
    check(ctx.lib.mmf_compute_rgb_residual(ctx.handle, minScale, _p(dIdx), _step(dIdx), _p(dIdy), _step(dIdy),
                                           _p(lastDepth), _step(lastDepth), _p(nextDepth), _step(nextDepth),
                                           _p(lastImage), _step(lastImage), _p(nextImage), _step(nextImage),
                                           _p(corresImg), maxDepthDelta, fptr(ktv), fptr(kk), cols, rows,
                                           C.byref(sigma), C.byref(count), _p(err),
                                           _step(err) if err is not None else 0))
    return sigma.value, count.value


def rgbStep(ctx, corresImg, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale):
    """cudafuncs.cuh:84-97. cloud: float32 [rows, cols, 3]. Returns (A[6,6], b[6])."""
    rows, cols = dIdx.shape
    A = np.zeros(36, np.float32)
    b = np.zeros(6, np.float32)
    check(ctx.lib.mmf_rgb_step(ctx.handle, _p(corresImg), sigma, _p(cloud), fx, fy, _p(dIdx), _step(dIdx), _p(dIdy),
                               _step(dIdy), sobelScale, cols, rows, fptr(A), fptr(b)))
    return A.reshape(6, 6), b


def so3Step(ctx, lastImage, nextImage, imageBasis, kinv, krlr):
    """cudafuncs.cuh:99-110. Returns (A[3,3], b[3], residual[2])."""
    rows, cols = nextImage.shape
    B, ki, kr = _f32(imageBasis, 9), _f32(kinv, 9), _f32(krlr, 9)
    A = np.zeros(9, np.float32)
    b = np.zeros(3, np.float32)
    res = np.zeros(2, np.float32)
    check(ctx.lib.mmf_so3_step(ctx.handle, _p(lastImage), _step(lastImage), _p(nextImage), _step(nextImage), fptr(B),
                               fptr(ki), fptr(kr), cols, rows, fptr(A), fptr(b), fptr(res)))
    return A.reshape(3, 3), b, res


def createVMap(ctx, intr, depth, vmap, depthCutoff):
    rows, cols = depth.shape
    check(ctx.lib.mmf_create_vmap(ctx.handle, C.byref(intr), _p(depth), _step(depth), cols, rows, _p(vmap),
                                  _step(vmap), depthCutoff))


def createNMap(ctx, vmap, nmap):
    rows, cols = vmap.shape[0] // 3, vmap.shape[1]
    check(ctx.lib.mmf_create_nmap(ctx.handle, _p(vmap), _step(vmap), cols, rows, _p(nmap), _step(nmap)))


def tranformMaps(ctx, vmap_src, nmap_src, Rmat, tvec, vmap_dst, nmap_dst):
    rows, cols = vmap_src.shape[0] // 3, vmap_src.shape[1]
    R, t = _f32(Rmat, 9), _f32(tvec, 3)
    check(ctx.lib.mmf_transform_maps(ctx.handle, _p(vmap_src), _p(nmap_src), _step(vmap_src), cols, rows, fptr(R),
                                     fptr(t), _p(vmap_dst), _p(nmap_dst), _step(vmap_dst)))


def copyMaps(ctx, vmap_src, nmap_src, vmap_dst, nmap_dst):
    rows, cols = vmap_dst.shape[0] // 3, vmap_dst.shape[1]
    check(ctx.lib.mmf_copy_maps(ctx.handle, _p(vmap_src), _p(nmap_src), cols, rows, _p(vmap_dst), _p(nmap_dst),
                                _step(vmap_dst)))


def resizeVMap(ctx, inp, out):
    rows, cols = inp.shape[0] // 3, inp.shape[1]
    check(ctx.lib.mmf_resize_vmap(ctx.handle, _p(inp), _step(inp), cols, rows, _p(out), _step(out)))


def resizeNMap(ctx, inp, out):
    rows, cols = inp.shape[0] // 3, inp.shape[1]
    check(ctx.lib.mmf_resize_nmap(ctx.handle, _p(inp), _step(inp), cols, rows, _p(out), _step(out)))


def imageBGRToIntensity(ctx, img, dst):
    rows, cols, ch = img.shape
    check(ctx.lib.mmf_image_bgr_to_intensity(ctx.handle, _p(img), _step(img), ch, cols, rows, _p(dst), _step(dst)))


def verticesToDepth(ctx, vmap_src, dst, cutOff):
    rows, cols = dst.shape
    check(ctx.lib.mmf_vertices_to_depth(ctx.handle, _p(vmap_src), cols, rows, cutOff, _p(dst), _step(dst)))


def projectToPointCloud(ctx, depth, cloud, intrinsics, level):
    rows, cols = depth.shape
    check(ctx.lib.mmf_project_to_point_cloud(ctx.handle, _p(depth), _step(depth), cols, rows, C.byref(intrinsics),
                                             level, _p(cloud)))


def pyrDownGaussF(ctx, src, dst):
    rows, cols = src.shape
    check(ctx.lib.mmf_pyr_down_gauss_f(ctx.handle, _p(src), _step(src), cols, rows, _p(dst), _step(dst)))


def pyrDownUcharGauss(ctx, src, dst):
    rows, cols = src.shape
    check(ctx.lib.mmf_pyr_down_uchar_gauss(ctx.handle, _p(src), _step(src), cols, rows, _p(dst), _step(dst)))


def computeDerivativeImages(ctx, src, dx, dy):
    rows, cols = src.shape
    check(ctx.lib.mmf_compute_derivative_images(ctx.handle, _p(src), _step(src), cols, rows, _p(dx), _step(dx),
                                                _p(dy), _step(dy)))
