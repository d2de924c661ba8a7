#!/usr/bin/env python
"""bench.py -- frames/sec at 640x480 of the dense-tracking hot path on MI355X, plus the ICP
JtJ-reduce roofline figure and a CPU baseline (BASELINE.json metric).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one MultiMotionFusion::processFrame for one rigid-body model (static scene):
bilateral depth filter, dense tracking (pyramids, SO3 pre-alignment, 4/5/10 ICP+RGB Gauss-Newton
iterations against the splat prediction), splat, index map, fuse, index map, clean, splat and
fill-in -- all inputs already resident in HBM.  With N GPUs every rank tracks its own model
on the same broadcast frame (per-object shard, weak scaling); value = model-frames/s over all
ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

W, H = 640, 480  # BASELINE.json configs[1]; MMF_BENCH_SIZE=1280x960 rehearses config 5's frame size
if os.environ.get("MMF_BENCH_SIZE"):
    W, H = (int(v) for v in os.environ["MMF_BENCH_SIZE"].lower().split("x"))
ICP_WEIGHT = 10.0  # GUI default (GUI/MainController.cpp:333-345)
DEPTH_CUTOFF = 15.0
N_FRAMES = 30  # frames of the synthetic sequence; the map is reset when the sequence wraps


def icp_step_bytes(n_px):
    return 48 * n_px + 116  # 12 floats per pixel + one JtJJtrSE3


def cpu_baseline(frames, K, poses):
    """Naive OpenMP CPU run of the same ICP reduction (oracle = "port"), bounded sample."""
    from oracle import oracle as orc
    try:
        libpath = orc.build(march="native", out="liboracle_native.so")
    except Exception:
        libpath = orc.build()
    prev, cur = poses[0], poses[1]
    o = orc.Odometry(W, H, K["cx"], K["cy"], K["fx"], K["fy"])
    o.initICPModel(frames[0]["vertex"], frames[0]["normal"], prev.astype(np.float32))
    o.initICP(frames[1]["depth"], DEPTH_CUTOFF)
    Rp = prev[:3, :3].astype(np.float32)
    tp = prev[:3, 3].astype(np.float32)
    Rpi = np.linalg.inv(Rp).astype(np.float32)
    per_level = []
    reps_by_level = (4000, 8000, 16000)  # ~2 s of wall time on all host cores (a few hundred core-seconds)
    for lvl in range(3):
        d = 1 << lvl
        args = (Rp, tp, o.buffer("vmaps_curr", lvl), o.buffer("nmaps_curr", lvl), Rpi, tp, K["fx"] / d, K["fy"] / d,
                K["cx"] / d, K["cy"] / d, o.buffer("vmaps_g_prev", lvl), o.buffer("nmaps_g_prev", lvl), 0.10,
                float(np.sin(20.0 * 3.14159254 / 180.0)))
        for _ in range(3):
            orc.icp_step_omp_f32(*args, libpath=libpath)
        ts = []
        for _ in range(reps_by_level[lvl]):
            t0 = time.perf_counter()
            orc.icp_step_omp_f32(*args, libpath=libpath)
            ts.append(time.perf_counter() - t0)
        per_level.append(float(np.median(ts)))
    schedule_s = 10 * per_level[0] + 5 * per_level[1] + 4 * per_level[2]
    return {
        "value": 1.0 / schedule_s,
        "unit": "frames/s",
        "cores": orc.omp_threads(libpath),
        "kind": "port",
        "sample": ("oracle icp_step (OpenMP, f32 accumulators) on the same 640x480 synthetic frame pair: median of "
                   "4000/8000/16000 reps at L0/L1/L2 (~2 s of wall time on every host core); value = 1/(10*t0+5*t1+4*t2), the ICP reduction schedule of one frame "
                   "only (no RGB term, no pyramids)"),
        "ms_per_step_l0": per_level[0] * 1e3,
        "gbps_l0": icp_step_bytes(W * H) / per_level[0] / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-reps", type=int, default=200)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MMF_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks
    # (ranks share devices round robin); the driver's runs use nccl = RCCL, one rank per GPU
    backend = os.environ.get("MMF_BENCH_BACKEND", "nccl")
    ndev = max(1, torch.cuda.device_count())
    if backend != "nccl":
        local_rank %= ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device"

    from multimotionfusion_amd import shard, synth
    from multimotionfusion_amd.cudafuncs import Context
    from multimotionfusion_amd.odometry import RGBDOdometry

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    K = synth.intrinsics(W, H)
    # every rank tracks one rigid-body model against the SAME sensor frames (broadcast by rank 0)
    poses = synth.trajectory(N_FRAMES, seed=1)
    frames = [synth.render(p, W, H, seed=i) for i, p in enumerate(poses)]
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    d_rgb = [up(f["rgb"]) for f in frames]
    d_depth = [up(f["depth"]) for f in frames]
    # N > 1: three frame buffers -- frame k is processed, frame k + 1 (already received) is prefetched on the
    # side streams, frame k + 2 is being broadcast by rank 0 (RCCL's own stream)
    NB = 3
    rgb_in = [torch.empty_like(d_rgb[0]) for _ in range(NB)]
    depth_in = [torch.empty_like(d_depth[0]) for _ in range(NB)]
    mask_in = [torch.zeros(H, W, dtype=torch.uint8, device=dev) for _ in range(NB)]
    pending = {}
    posted = set()

    ctx = Context(local_rank)
    from multimotionfusion_amd.fusion import MultiMotionFusion
    mmf = MultiMotionFusion(ctx, W, H, K["cx"], K["cy"], K["fx"], K["fy"], icp_weight=ICP_WEIGHT)
    odom = mmf.getFrameOdometry()
    state = {"frame": 0}
    # next-frame prefetch on side streams: on at N = 1; at N > 1 only on request (MMF_BENCH_PREFETCH=2) -- the
    # collective's own streams share the device with it there and that combination could not be measured on a
    # one-GPU box (two ranks sharing one GPU time-slice pathologically with the extra streams)
    _pf = os.environ.get("MMF_BENCH_PREFETCH", "1")
    PREFETCH = _pf == "2" or (_pf != "0" and world == 1)

    def step(i):
        """One processFrame: bilateral filter, tracking (SO3 + 4/5/10 ICP+RGB GN iterations against
        the splat prediction), splat, index map, fuse, index map, clean, splat + fill-in."""
        k = state["frame"] % len(frames)
        if state["frame"] and k == 0:  # sequence wrapped: start a fresh map (the trajectory jumps back)
            mmf.reset()
        state["frame"] += 1
        if world > 1:  # rank 0's sensor frame reaches every model owner (RCCL broadcast over xGMI)
            def post(n):  # start the broadcast of sequence frame n into buffer n % NB (once)
                if n in posted:
                    return
                posted.add(n)
                b, kk = n % NB, n % len(frames)
                if rank == 0:
                    rgb_in[b].copy_(d_rgb[kk])
                    depth_in[b].copy_(d_depth[kk])
                pending[n] = shard.broadcast_frame_async(rgb_in[b], depth_in[b], mask_in[b], src=0)

            def arrived(n):  # the compute stream waits for the collective; the host does not
                for w in pending.pop(n, []):
                    w.wait()
            n = state["frame"] - 1
            post(n)
            post(n + 1)
            arrived(n)
            if PREFETCH:  # the next frame has to be in place before this frame's tracking is enqueued: the side
                arrived(n + 1)  # streams of the prefetch start right after it
            post(n + 2)
            posted.discard(n - 1)
            mmf.processFrame(rgb_in[n % NB], depth_in[n % NB], timestamp=i)
            if PREFETCH and (n + 1) % len(frames) != 0:
                mmf.prefetchFrame(rgb_in[(n + 1) % NB], depth_in[(n + 1) % NB])
        else:  # inputs already resident in HBM
            mmf.processFrame(d_rgb[k], d_depth[k], timestamp=i)
            kn = state["frame"] % len(frames)
            if PREFETCH and kn != 0:  # the next frame's filter, pyramids and SO3 pre-alignment overlap this frame's fusion
                mmf.prefetchFrame(d_rgb[kn], d_depth[kn])
        pose = mmf.getCurrPose()
        if world > 1:  # every rank learns every model's pose (18 floats per rank), without a host round trip
            od = mmf.getFrameOdometry()
            state["poses"] = shard.gather_poses_async(pose, od.lastICPError, od.lastICPCount, dev)
        return pose

    def fence():
        if world > 1:
            if state.get("poses") is not None:
                state["poses"][0].wait()
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    last_pose = None
    for i in range(args.steps):
        last_pose = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # accuracy sanity of the last tracked frame against the known trajectory (relative to frame 0)
    k = (state["frame"] - 1) % len(frames)
    gt = np.linalg.inv(poses[0]) @ poses[k]
    t_err = float(np.linalg.norm(last_pose[:3, 3] - gt[:3, 3]))
    odom = mmf.getFrameOdometry()
    n_surfels = mmf.getBackgroundModel().lastCount()

    result = None
    if rank == 0:
        # roofline of the dominant kernel: level-0 ICP JtJ reduction, HIP events on the launch stream
        n0 = W * H
        us = odom.timeIcpKernel(0, args.roofline_reps)
        achieved = icp_step_bytes(n0) / (us * 1e-6) / 1e9
        # traffic: memory-side bytes per level-0 launch from the separate rocprofv3 --pmc passes kept in
        # profiles/r01_pmc_icp_{fetch,write}_size.csv (this kernel, tools/pmc_icp.py): FETCH_SIZE
        # 7348.5 KiB, doubled because gfx950 tallies 128-byte read requests at 64 bytes (guide, HBM
        # section; calibrated there for 16-B-per-lane streams -- the 16-B-per-lane build of this kernel,
        # profiles/r01_pmc_icp_px4_*.csv, reads 7305.25 KiB, so the same factor is applied here) +
        # WRITE_SIZE 75 KiB (600 partial records).  Algorithmic bytes are 14.75 MB: no wasted re-reads.
        traffic = (2 * 7348.5 + 75.0) * 1024
        roofline = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                    "traffic": traffic, "kernel": f"icp_kernel2<2,1,256,packed> level 0 ({W}x{H})", "us_per_launch": us,
                    "bytes_per_launch": icp_step_bytes(n0),
                    "us_per_launch_l1": odom.timeIcpKernel(1, args.roofline_reps),
                    "us_per_launch_l2": odom.timeIcpKernel(2, args.roofline_reps)}
        # the two surfel projections named in the north star, timed the same way (HIP events on the stream,
        # back to back; both are idempotent on the current map): achieved = algorithmic bytes of
        # SURVEY.md 8(d) / launch-pair time.  Reported beside the contract's roofline object, not in it.
        model = mmf.getBackgroundModel()
        tick = mmf.getTick()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def timed(fn, reps=50):
            for _ in range(3):
                fn()
            ev0.record()
            for _ in range(reps):
                fn()
            ev1.record()
            ev1.synchronize()
            return ev0.elapsed_time(ev1) * 1e3 / reps

        us_idx = timed(lambda: model.predictIndices(tick, 20.0, 200))
        us_spl = timed(lambda: model.combinedPredict(20.0, tick, tick, 200))
        b_idx, b_spl = 48 * n_surfels + 52 * n0, 48 * n_surfels + 38 * n0
        surfel_passes = {
            "predictIndices": {"us": us_idx, "bytes": b_idx, "GBps": b_idx / us_idx / 1e3, "frac": b_idx / us_idx / 1e3 / 8000.0,
                               "kernels": "index_map_kernel + index_resolve_kernel"},
            "combinedPredict": {"us": us_spl, "bytes": b_spl, "GBps": b_spl / us_spl / 1e3, "frac": b_spl / us_spl / 1e3 / 8000.0,
                                "kernels": "splat_kernel + splat_resolve_kernel"},
        }
        # the keypoint descriptor matcher, the one MFMA kernel beside the path:
        # 1024 x 1024 descriptors of 256 floats, 2 * nq * nt * dim flops on v_mfma_f32_32x32x2_f32
        from multimotionfusion_amd.matcher import matchDescriptors
        gen = torch.Generator(device="cpu").manual_seed(0)
        dq = torch.nn.functional.normalize(torch.randn(1024, 256, generator=gen), dim=1).to(dev)
        dt = torch.nn.functional.normalize(torch.randn(1024, 256, generator=gen), dim=1).to(dev)
        matchDescriptors(ctx, dq, dt, 0.7)  # sizes the workspace
        m_idx = torch.empty(1024, dtype=torch.int32, device=dev)
        m_dist = torch.empty(1024, dtype=torch.float32, device=dev)
        from multimotionfusion_amd.cudafuncs import _p

        def match_raw():  # the C entry point with preallocated outputs: the Python wrapper's allocations would dominate
            ctx.lib.mmf_match_descriptors(ctx.handle, _p(dq), 1024, _p(dt), 1024, 256, 0.7, _p(m_idx), _p(m_dist))

        us_match = timed(match_raw, reps=100)
        flops = 2.0 * 1024 * 1024 * 256
        matcher = {"us": us_match, "nq": 1024, "nt": 1024, "dim": 256, "TFLOPs": flops / us_match / 1e6,
                   "frac_f32_mfma_peak": flops / us_match / 1e6 / 157.3, "launches": 3,
                   "kernels": "row_norms_kernel (MFMA; also resets the arg-min keys) + match_tile64_kernel (MFMA, LDS-shared 64x64 tiles) + match_cross_check_kernel"}
        # the SuperPoint keypoint network (north star: "the only true dense contractions ... on MFMA"): one
        # forward pass (12 convolutions + normalisation + heat map) on an image of the bench size, random-init
        # weights, f32 operands on v_mfma_f32_32x32x2_f32; flops = convolution multiply-adds * 2
        from multimotionfusion_amd.superpoint import SuperPoint, forward_flops, random_weights
        kp = SuperPoint(ctx, random_weights(0), max_width=W, max_height=H)
        us_sp = timed(lambda: kp.enqueue(d_rgb[0]), reps=30)
        fl_sp = float(forward_flops(W, H))
        superpoint = {"us": us_sp, "GFLOP": fl_sp / 1e9, "TFLOPs": fl_sp / us_sp / 1e6,
                      "frac_f32_mfma_peak": fl_sp / us_sp / 1e6 / 157.3, "launches": 13, "dtype": "f32",
                      "kernels": "sp_conv1a (grey + 1->64, VALU) + 8 x sp_conv_mfma_kernel + 1 x sp_conv_mfma_pair_kernel (implicit GEMM, fused ReLU / "
                                 "2x2 max pool) + sp_l2_normalize + sp_heatmap",
                      "weights": "random-init SuperPointNet architecture"}
        kp.close()
        # super-pixel resampling of a per-model map for the segmentation: the ICP-error
        # map of the frame into 16-pixel super-pixels (a regular grid stands in for gSLICr's mask)
        from multimotionfusion_amd import slic
        S = 16
        yy, xx = np.mgrid[0:H, 0:W]
        labels = torch.from_numpy(((yy // S).clip(0, H // S - 1) * (W // S) + (xx // S).clip(0, W // S - 1)).astype(np.int32)).to(dev)
        err_map = torch.rand((H, W), device=dev)
        slic_out = torch.empty((H // S, W // S), dtype=torch.float32, device=dev)

        def slic_raw():
            ctx.lib.mmf_slic_downsample(ctx.handle, _p(labels), W, H, S, _p(err_map), 1, 0, 0, 0.0, _p(slic_out), None)

        slic_raw()
        us_slic = timed(slic_raw, reps=50)
        slic_info = {"us": us_slic, "superpixels": (H // S) * (W // S), "bytes_in": 8 * n0, "bytes_out": 4 * (H // S) * (W // S),
                     "kernels": "slic_reset + slic_census (atomics) + slic_sum (wave per super-pixel, pixel-order sums) + "
                                "slic_finish", "replaces": "two 4.9 MB / 1.2 MB texture downloads per model + CPU loops"}
        result = {
            "metric": f"frames/sec @ {W}x{H} (dense ICP+RGB tracking); ICP JtJ-reduce achieved HBM GB/s vs peak",
            "value": world * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{W}x{H} synthetic RGB-D sequence through MultiMotionFusion::processFrame, static "
                                   "scene (no segmentation): bilateral filter, dense ICP+RGB odometry (SO3 + 4/5/10 "
                                   "Gauss-Newton iterations, icpWeight 10) against the surfel splat, index map, fuse, "
                                   "clean, splat + fill-in; one rigid-body model per GPU; the next frame's depth filter, input "
                                   "pyramids and SO3 pre-alignment run on two side streams during the current frame's fusion",
                       "width": W, "height": H, "models_per_gpu": 1, "parallelism": f"model-shard x{world}"},
            "roofline": roofline,
            "surfel_passes": surfel_passes,
            "matcher": matcher,
            "superpoint": superpoint,
            "slic_downsample": slic_info,
            "device": ctx.device_name(),
            "last_frame_translation_error_m": t_err,
            "icp_inliers_last": odom.lastICPCount,
            "surfels": n_surfels,
        }
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only
            result["cpu_baseline"] = cpu_baseline(frames, K, poses)
    fence()
    if rank == 0:
        print(json.dumps(result), flush=True)
    mmf.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
