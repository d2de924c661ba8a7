"""Replay a .klg sequence through the HIP path and write the camera trajectory in the reference's pose-log
format (poses-<id>.txt of MultiMotionFusion::exportPoses), for diffing against a reference run.
usage: python tools/replay_klg.py seq.klg [out_dir] [width height fx fy cx cy]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimotionfusion_amd.cudafuncs import Context  # noqa: E402
from multimotionfusion_amd.fusion import MultiMotionFusion  # noqa: E402
from multimotionfusion_amd.klg import KlgLogReader, write_pose_log  # noqa: E402


def main():
    file = sys.argv[1]
    out_dir = sys.argv[2] if len(sys.argv) > 2 else "."
    W, H, fx, fy, cx, cy = (float(a) for a in sys.argv[3:9]) if len(sys.argv) > 8 else (640, 480, 528, 528, 320, 240)
    W, H = int(W), int(H)
    reader = KlgLogReader(file, W, H)
    ctx = Context(0)
    mmf = MultiMotionFusion(ctx, W, H, cx, cy, fx, fy)
    log = []
    while reader.hasMore():  # MainController.cpp:547-715: one processFrame per log frame
        ts, depth, rgb = reader.getNext()
        mmf.processFrame(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), timestamp=ts)
        log.append((ts, mmf.getCurrPose()))
    write_pose_log(os.path.join(out_dir, "poses-0.txt"), log)
    print(f"{len(log)} frames -> {os.path.join(out_dir, 'poses-0.txt')}")


if __name__ == "__main__":
    main()
