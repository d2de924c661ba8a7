"""Keypoint descriptor matching on the device (PointTracker::addKeypoints' cv::BFMatcher call,
Core/Utils/PointTracker.cpp:100-114), through the C ABI -- no fallback."""
import torch

from .cudafuncs import Context, _p, check


def matchDescriptors(ctx: Context, query: torch.Tensor, train: torch.Tensor, min_feature_distance: float = 0.0):
    """cv::BFMatcher(cv::NORM_L2, crossCheck=True).match(query, train) followed by the distance gate
    (`min_feature_distance < epsilon or distance <= min_feature_distance`).

    query [nq, dim], train [nt, dim]: float32 CUDA tensors, dim a multiple of 8.
    Returns (trainIdx [nq] int32, -1 = unmatched; distance [nq] float32), both on the device."""
    assert query.dtype == torch.float32 and train.dtype == torch.float32 and query.is_cuda and train.is_cuda
    query, train = query.contiguous(), train.contiguous()
    nq, nt = query.shape[0], train.shape[0]
    dim = query.shape[1] if query.dim() == 2 else train.shape[1]
    idx = torch.empty(nq, dtype=torch.int32, device=query.device)
    dist = torch.empty(nq, dtype=torch.float32, device=query.device)
    check(ctx.lib.mmf_match_descriptors(ctx.handle, _p(query) if nq else None, nq, _p(train) if nt else None, nt, int(dim),
                                        float(min_feature_distance), _p(idx) if nq else None, _p(dist) if nq else None))
    return idx, dist
