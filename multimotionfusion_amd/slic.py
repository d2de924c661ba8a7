"""Super-pixel resampling on the device: the host-side mirror of Slic::downsample / downsampleThresholded /
upsample (Core/Segmentation/Slic.h:48-146, Slic.cpp:82-112) as Segmentation.cpp:177-178,218-221,683 uses them,
over the C ABI -- no fallback.  The label image is gSLICr's segmentation mask (an input)."""
import torch

from .cudafuncs import Context, _p, check


def _labels(labels):
    assert labels.dtype == torch.int32 and labels.is_cuda and labels.dim() == 2
    return labels.contiguous()


def downsample(ctx: Context, labels, spixelSize, image, channel=0, threshold=None, with_counts=False):
    """Slic::downsample<float>(image, channel), or downsampleThresholded<float>(image, threshold) when
    `threshold` is given.  image [H,W] or [H,W,C] float32 CUDA -> [H/S, W/S] float32 (and spixelCounts)."""
    labels = _labels(labels)
    assert image.dtype == torch.float32 and image.is_cuda
    image = image.contiguous()
    H, W = labels.shape
    ch = 1 if image.dim() == 2 else image.shape[2]
    out = torch.empty((H // spixelSize, W // spixelSize), dtype=torch.float32, device=labels.device)
    counts = torch.empty(out.shape, dtype=torch.int32, device=labels.device) if with_counts else None
    check(ctx.lib.mmf_slic_downsample(ctx.handle, _p(labels), W, H, int(spixelSize), _p(image), ch, int(channel),
                                      int(threshold is not None), float(threshold or 0.0), _p(out), _p(counts)))
    return (out, counts) if with_counts else out


def downsample_rgb(ctx: Context, labels, spixelSize, rgb):
    """Slic::downsample(): [H/S, W/S, 3] u8 integer means of input channels (2, 1, 0)"""
    labels = _labels(labels)
    assert rgb.dtype == torch.uint8 and rgb.is_cuda and rgb.dim() == 3
    rgb = rgb.contiguous()
    H, W = labels.shape
    out = torch.empty((H // spixelSize, W // spixelSize, 3), dtype=torch.uint8, device=labels.device)
    check(ctx.lib.mmf_slic_downsample_rgb(ctx.handle, _p(labels), W, H, int(spixelSize), _p(rgb), rgb.shape[2], _p(out)))
    return out


def upsample_u8(ctx: Context, labels, small):
    """Slic::upsample<unsigned char>(map): full[i] = map[labels[i]]"""
    labels = _labels(labels)
    assert small.dtype == torch.uint8 and small.is_cuda
    small = small.contiguous()
    H, W = labels.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=labels.device)
    check(ctx.lib.mmf_slic_upsample_u8(ctx.handle, _p(labels), W, H, _p(small), small.numel(), _p(out)))
    return out
