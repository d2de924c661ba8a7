"""Times the SuperPoint forward pass (network + heat map) and getFeatures on synthetic input.
usage: python tools/superpoint_bench.py [width height [reps]]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.superpoint import SuperPoint, forward_flops, random_weights  # noqa: E402

def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    ctx = Context(0)
    sp = SuperPoint(ctx, random_weights(), max_width=W, max_height=H)
    rng = np.random.default_rng(1)
    img = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
    for _ in range(5):
        sp.enqueue(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        sp.enqueue(img)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = forward_flops(W, H)
    print(f"forward {W}x{H}: {ms * 1e3:.1f} us, {fl / 1e9:.2f} GFLOP, {fl / ms / 1e9:.2f} TFLOP/s "
          f"({fl / ms / 1e9 / 157.3 * 100:.1f}% of the f32 MFMA peak)")
    t0 = time.perf_counter()
    for _ in range(10):
        xy, conf, desc = sp.keypoints(img)
    print(f"getFeatures: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms, {len(xy)} keypoints")
    sp.close()
    ctx.close()


if __name__ == "__main__":
    main()
