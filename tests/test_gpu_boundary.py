"""-m gpu: the drop-in boundary exercised the way the reference's front-end uses it.  tests/cpp/main_controller_sequence.cpp
mirrors GUI/MainController.cpp:547-715 (processFrame(FrameData) with the upload inside, getModelToModel, the block of
per-tick setters, getTextures / getModels / getIndexMap, setTick, exportPoses) and drives a stand-alone Model through
performTracking / fuse / clean with the reference's argument lists; it is compiled with g++ against the C++ shims
(multimotionfusion_amd/cpp/) and libmmf_hip.so and run on the device."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_main_controller_call_sequence_runs_on_the_device(tmp_path):
    pkg = os.path.join(REPO, "multimotionfusion_amd")
    exe = tmp_path / "main_controller_sequence"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-isystem", "/opt/rocm/include",
                    os.path.join(REPO, "tests", "cpp", "main_controller_sequence.cpp"), "-o", str(exe), f"-L{pkg}", "-lmmf_hip",
                    "-lamdhip64", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)
    out = str(tmp_path) + "/"
    r = subprocess.run([str(exe), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "main controller sequence: ok" in r.stdout and "invalid image data" in r.stderr
    lines = open(out + "poses-0.txt").read().splitlines()
    assert len(lines) == 8 and lines[0].split()[0] == "1000" and all(len(l.split()) == 8 for l in lines)
    assert len(open(out + "poses-1.txt").read().splitlines()) == 6


def test_host_frame_upload_gives_the_same_bits_as_device_frames(gpu_ctx):
    """processFrame(const FrameData&) stages the frame through pinned double buffers and uploads it inside the call
    (MultiMotionFusion.cpp:221, 261): same poses and surfels as handing over device-resident frames."""
    import numpy as np
    import torch
    from multimotionfusion_amd import synth
    from multimotionfusion_amd.fusion import MultiMotionFusion
    w, h, n = 320, 240, 5
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n, seed=19)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    a = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    b = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    keep = []
    for i, f in enumerate(frames):
        keep.append((torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda()))
        a.processFrame(*keep[-1], timestamp=i)
        b.processFrameHost(f["rgb"], f["depth"], timestamp=i)
        assert np.array_equal(a.getCurrPose(), b.getCurrPose()), i
    assert np.array_equal(a.getBackgroundModel().downloadMap().view(np.uint32), b.getBackgroundModel().downloadMap().view(np.uint32))
    assert np.array_equal(b.getTexture("RGB").cpu().numpy(), frames[-1]["rgb"])
    assert np.array_equal(b.getTexture("DEPTH_METRIC").cpu().numpy(), frames[-1]["depth"])
    a.close()
    b.close()


def test_host_frames_announced_one_call_ahead_give_the_same_bits(gpu_ctx):
    """mmf_fusion_process_frame_host_next: the next call's host frame is staged, uploaded and prepared during this call.
    Same poses and surfels as device-resident frames -- also when an announced frame never comes (another one is passed), when
    nothing is announced for a frame, and when the same host arrays are refilled between calls (a reader's double buffer)."""
    import numpy as np
    import torch
    from multimotionfusion_amd import synth
    from multimotionfusion_amd.fusion import MultiMotionFusion
    w, h, n = 320, 240, 9
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n, seed=19)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    a = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    b = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    # the front-end's two frame buffers, refilled in turn
    buf = [(np.empty_like(frames[0]["rgb"]), np.empty_like(frames[0]["depth"])) for _ in range(2)]
    decoy = (frames[0]["rgb"].copy(), frames[0]["depth"].copy())

    def fill(slot, i):
        buf[slot][0][...] = frames[i]["rgb"]
        buf[slot][1][...] = frames[i]["depth"]

    keep = []
    fill(0, 0)
    for i in range(n):
        keep.append((torch.from_numpy(frames[i]["rgb"]).cuda(), torch.from_numpy(frames[i]["depth"]).cuda()))
        a.processFrame(*keep[-1], timestamp=i)
        cur = buf[i % 2]
        if i + 1 < n and i not in (3, 5):
            fill((i + 1) % 2, i + 1)  # the reader has the next frame already
            nxt = buf[(i + 1) % 2]
        elif i == 3:
            nxt = decoy  # announced, but the next call brings another frame
        else:
            nxt = None  # nothing announced
        b.processFrameHost(cur[0], cur[1], timestamp=i, next=nxt)
        if i + 1 < n and i in (3, 5):
            fill((i + 1) % 2, i + 1)
        assert np.array_equal(a.getCurrPose(), b.getCurrPose()), i
    assert np.array_equal(a.getBackgroundModel().downloadMap().view(np.uint32), b.getBackgroundModel().downloadMap().view(np.uint32))
    assert np.array_equal(b.getTexture("RGB").cpu().numpy(), frames[-1]["rgb"])
    a.close()
    b.close()


def test_announced_host_frames_around_a_dictated_pose(gpu_ctx):
    """The upload of an announced frame waits for an event from the fusion's stream only when the last call did not end
    with the host holding a tracked pose (a dictated pose: no wait for the GPU in that call) -- both ways give the bits of
    device-resident frames; frames handed in as HostFrame objects (addresses taken once)."""
    import numpy as np
    import torch
    from multimotionfusion_amd import synth
    from multimotionfusion_amd.fusion import HostFrame, MultiMotionFusion
    w, h, n = 320, 240, 10
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(n, seed=29)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    host = [HostFrame(f["rgb"], f["depth"]) for f in frames]
    dev = [(torch.from_numpy(f["rgb"]).cuda(), torch.from_numpy(f["depth"]).cuda()) for f in frames]
    a = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    b = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    for i in range(n):
        dictated = a.getCurrPose().copy() if i in (4, 5) else None  # two calls in a row that track nothing
        a.processFrame(*dev[i], timestamp=i, inPose=dictated, next=dev[i + 1] if i + 1 < n else None)
        b.processFrameHost(host[i], timestamp=i, inPose=dictated, next=host[i + 1] if i + 1 < n else None)
        assert np.array_equal(a.getCurrPose(), b.getCurrPose()), i
    assert np.array_equal(a.getBackgroundModel().downloadMap().view(np.uint32), b.getBackgroundModel().downloadMap().view(np.uint32))
    a.close()
    b.close()


def test_runtime_setters_take_effect_at_the_next_frame(gpu_ctx):
    """setFastOdom / setPyramid / setSo3 / setIcpWeight / setRgbOnly change the NEXT processFrame like a GUI checkbox
    (MultiMotionFusion.cpp:1064-1116): iteration counts and the tracker's mode follow them."""
    import numpy as np
    import torch
    from multimotionfusion_amd import synth
    from multimotionfusion_amd.fusion import MultiMotionFusion
    w, h = 320, 240
    K = synth.intrinsics(w, h)
    poses = synth.trajectory(6, seed=23)
    frames = [synth.render(p, w, h, seed=i) for i, p in enumerate(poses)]
    g = MultiMotionFusion(gpu_ctx, w, h, K["cx"], K["cy"], K["fx"], K["fy"])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    g.processFrame(dev(frames[0]["rgb"]), dev(frames[0]["depth"]), timestamp=0)
    g.processFrame(dev(frames[1]["rgb"]), dev(frames[1]["depth"]), timestamp=1)
    assert g.getFrameOdometry().iterations_run == 19 and g.getFrameOdometry().so3_iterations_run > 0
    g.setFastOdom(True)
    g.processFrame(dev(frames[2]["rgb"]), dev(frames[2]["depth"]), timestamp=2)
    assert g.getFrameOdometry().iterations_run == 12  # 3 + 5 + 4 (RGBDOdometry.cpp:312-314)
    g.setPyramid(False)
    g.setSo3(False)
    g.processFrame(dev(frames[3]["rgb"]), dev(frames[3]["depth"]), timestamp=3)
    od = g.getFrameOdometry()
    assert od.iterations_run == 3 and od.so3_iterations_run == 0
    g.setFastOdom(False)
    g.setPyramid(True)
    g.setIcpWeight(100.0)  # depth only: the photometric term is off (RGBDOdometry.cpp:221-222)
    n_before = g.getBackgroundModel().lastCount()
    g.processFrame(dev(frames[4]["rgb"]), dev(frames[4]["depth"]), timestamp=4)
    assert g.getFrameOdometry().iterations_run == 19 and g.getConfig().icp_weight == 100.0
    g.setRgbOnly(True)  # 2.5D photometric tracking only, and no fusion (MultiMotionFusion.cpp:791)
    n_before = g.getBackgroundModel().lastCount()
    g.processFrame(dev(frames[5]["rgb"]), dev(frames[5]["depth"]), timestamp=5)
    assert g.getBackgroundModel().lastCount() == n_before and g.getTick() == 7
    gt = np.linalg.inv(poses[0]) @ poses[4]
    g.setTick(20)
    assert g.getTick() == 20
    g.close()
    del gt
