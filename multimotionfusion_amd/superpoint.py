"""SuperPoint keypoints on the device: the host-side mirror of the `SuperPoint` class the reference takes from
the super_point_inference package (Core/MultiMotionFusion.h:46,366; MultiMotionFusion.cpp:78,233), over the C
ABI -- no fallback.

    kp = SuperPoint(ctx, weights, max_width=640, max_height=480)
    coordinates, descriptors = kp.getFeatures(img)        # [n,2] float64 in [0,1), [n,256] float64
"""
import ctypes as C
import weakref

import numpy as np
import torch

from .cudafuncs import Context, _p, check

LAYERS = ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa",
          "convDb")


SHAPES = ((1, 64, 3), (64, 64, 3), (64, 64, 3), (64, 64, 3), (64, 128, 3), (128, 128, 3), (128, 128, 3), (128, 128, 3),
          (128, 256, 3), (256, 65, 1), (128, 256, 3), (256, 256, 1))  # (Cin, Cout, k) of LAYERS


def random_weights(seed=0):
    """He-initialised weights of the SuperPointNet architecture (there is no network access for the MagicLeap
    checkpoint): what bench.py and tools/ run the network on."""
    rng = np.random.default_rng(seed)
    return [(rng.normal(0, np.sqrt(2.0 / (ci * k * k)), (co, ci, k, k)).astype(np.float32),
             rng.normal(0, 0.05, co).astype(np.float32)) for ci, co, k in SHAPES]


def forward_flops(width, height):
    """multiply-adds * 2 of one forward pass (convolutions only)"""
    total, h, w = 0, height, width
    for i, (ci, co, k) in enumerate(SHAPES):
        if i in (2, 4, 6):
            h, w = h // 2, w // 2
        total += 2 * h * w * ci * co * k * k
    return total


def load_weights(path):
    """The 12 (weight, bias) pairs of a SuperPointNet checkpoint: a TorchScript archive (what the reference's
    `-model SuperPointNet.pt` points at) or a plain state dict.  File plumbing only."""
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except Exception:
        sd = torch.load(path, map_location="cpu")
        if hasattr(sd, "state_dict"):
            sd = sd.state_dict()
    return [(sd[f"{n}.weight"].float().numpy(), sd[f"{n}.bias"].float().numpy()) for n in LAYERS]


class SuperPoint:
    def __init__(self, ctx: Context, weights, max_width=640, max_height=480, max_keypoints=4096, conf_thresh=0.015,
                 nms_dist=4, border=4):
        if isinstance(weights, str):
            weights = load_weights(weights)
        assert len(weights) == 12
        flat = []
        for w, b in weights:
            flat += [np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)]
        arr = (C.c_void_p * 24)(*[a.ctypes.data for a in flat])
        self.ctx, self.lib = ctx, ctx.lib
        self.max_keypoints = int(max_keypoints)
        self.conf_thresh, self.nms_dist, self.border = float(conf_thresh), int(nms_dist), int(border)
        h = C.c_void_p()
        check(self.lib.mmf_superpoint_create(ctx.handle, arr, int(max_width), int(max_height), self.max_keypoints, C.byref(h)))
        self.handle = h
        self._size = None
        ctx._children.append(weakref.ref(self))

    def _image(self, img):
        if not isinstance(img, torch.Tensor):
            img = torch.from_numpy(np.ascontiguousarray(img, np.uint8)).cuda(self.ctx.device)
        assert img.dtype == torch.uint8 and img.is_cuda and img.dim() in (2, 3)
        img = img.contiguous()
        return img, img.shape[0], img.shape[1], 1 if img.dim() == 2 else img.shape[2]

    def forward(self, img):
        """Runs the network; returns (semi [H/8,W/8,65], desc [H/8,W/8,256], heat [H,W]) as numpy arrays."""
        img, H, W, ch = self._image(img)
        check(self.lib.mmf_superpoint_forward(self.handle, _p(img), W, H, ch))
        out = []
        for which, shape in enumerate(((H // 8, W // 8, 65), (H // 8, W // 8, 256), (H, W))):
            a = np.empty(shape, np.float32)
            check(self.lib.mmf_superpoint_download(self.handle, which, a.ctypes.data, a.size))
            out.append(a)
        return tuple(out)

    def enqueue(self, img):
        """forward pass only, asynchronous (bench)"""
        img, H, W, ch = self._image(img)
        check(self.lib.mmf_superpoint_forward(self.handle, _p(img), W, H, ch))

    def keypoints(self, img):
        """(xy [n,2] int32 pixels, conf [n] float32, desc [n,256] float32), strongest first"""
        img, H, W, ch = self._image(img)
        xy = np.empty((self.max_keypoints, 2), np.int32)
        conf = np.empty(self.max_keypoints, np.float32)
        desc = np.empty((self.max_keypoints, 256), np.float32)
        n = C.c_int(0)
        check(self.lib.mmf_superpoint_get_features(self.handle, _p(img), W, H, ch, self.conf_thresh, self.nms_dist, self.border,
                                                   xy.ctypes.data, conf.ctypes.data, desc.ctypes.data, C.byref(n)))
        return xy[:n.value].copy(), conf[:n.value].copy(), desc[:n.value].copy()

    def getFeatures(self, img):
        """SuperPoint::getFeatures as MultiMotionFusion.cpp:233 consumes it: (coordinates [n,2] float64 normalised
        by (width, height), descriptors [n,256] float64)."""
        img, H, W, _ = self._image(img)
        xy, _, desc = self.keypoints(img)
        return xy.astype(np.float64) / np.array([W, H], np.float64), desc.astype(np.float64)

    def close(self):
        if self.handle:
            self.lib.mmf_superpoint_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def conv(ctx: Context, x: torch.Tensor, w, b, relu=True, pool=False, nt=0):
    """One layer by itself: x [H,W,Cin] float32 CUDA tensor (channels last), w [Cout,Cin,k,k], b [Cout] numpy."""
    assert x.dtype == torch.float32 and x.is_cuda and x.dim() == 3
    x = x.contiguous()
    w, b = np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)
    H, W, cin = x.shape
    cout, taps = w.shape[0], w.shape[2] * w.shape[3]
    out = torch.empty((H // 2, W // 2, cout) if pool else (H, W, cout), dtype=torch.float32, device=x.device)
    check(ctx.lib.mmf_superpoint_conv(ctx.handle, _p(x), H, W, cin, w.ctypes.data, b.ctypes.data, cout, taps, int(relu),
                                      int(pool), int(nt), _p(out)))
    return out
