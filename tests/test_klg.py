"""Sequence I/O (multimotionfusion_amd/klg.py): the .klg layout of GUI/Tools/KlgLogReader.cpp:20-130 and the
pose-log lines of MultiMotionFusion::exportPoses (Core/MultiMotionFusion.cpp:1020-1045)."""
import struct
import zlib

import numpy as np
import pytest

from multimotionfusion_amd.klg import KlgLogReader, pose_7d, quaternion_xyzw, write_klg, write_pose_log

W, H = 64, 48


def frames(n, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        d16 = rng.integers(0, 5000, (H, W)).astype(np.uint16)
        yy, xx = np.mgrid[0:H, 0:W]
        rgb = np.stack([(xx * 4 + i) % 256, (yy * 5) % 256, ((xx + yy) * 2) % 256], 2).astype(np.uint8)
        out.append((1000 + 33 * i, d16, rgb))
    return out


@pytest.mark.parametrize("compress", [True, False])
def test_round_trip_raw_colour(tmp_path, compress):
    f = frames(4)
    path = str(tmp_path / "a.klg")
    write_klg(path, f, compress_depth=compress)
    r = KlgLogReader(path, W, H)
    assert r.getNumFrames() == 4
    got = []
    while r.hasMore():  # like the reference's main loop: the last frame is never served (KlgLogReader.cpp:112)
        got.append(r.getNext())
    assert len(got) == 3
    for (ts, depth, rgb), (ts0, d16, rgb0) in zip(got, f):
        assert ts == ts0
        assert depth.dtype == np.float32 and np.array_equal(depth, d16.astype(np.float32) * np.float32(0.001))
        assert np.array_equal(rgb, rgb0)
    ts, depth, _ = r.getPrevious()  # re-reads the frame served last
    assert ts == f[2][0] and np.array_equal(depth, got[2][1])
    r.close()


def test_layout_is_the_reference_layout(tmp_path):
    ts, d16, rgb = frames(1)[0]
    path = str(tmp_path / "b.klg")
    with open(path, "wb") as fp:  # written by hand, field by field (KlgLogReader.cpp:57-62)
        comp = zlib.compress(d16.tobytes())
        fp.write(struct.pack("<i", 2))
        for _ in range(2):
            fp.write(struct.pack("<q", ts) + struct.pack("<i", len(comp)) + struct.pack("<i", 0) + comp)
    r = KlgLogReader(path, W, H, flipColors=True)
    t, depth, colour = r.getNext()
    assert t == ts and np.array_equal(depth, d16.astype(np.float32) * np.float32(0.001))
    assert not colour.any()  # imageSize 0: black (KlgLogReader.cpp:82)
    r.close()


def test_jpeg_colour_and_flip(tmp_path):
    f = frames(2)
    path = str(tmp_path / "c.klg")
    write_klg(path, f, jpeg_quality=95)
    r = KlgLogReader(path, W, H)
    _, _, rgb = r.getNext()
    assert np.abs(rgb.astype(int) - f[0][2].astype(int)).mean() < 6.0  # lossy
    r.close()
    r = KlgLogReader(path, W, H, flipColors=True)
    _, _, bgr = r.getNext()
    assert np.array_equal(bgr, rgb[:, :, ::-1])
    r.close()


def test_fast_forward_and_rewind(tmp_path):
    f = frames(6)
    path = str(tmp_path / "d.klg")
    write_klg(path, f)
    r = KlgLogReader(path, W, H)
    r.fastForward(3)
    assert r.getNext()[0] == f[3][0]
    assert r.rewind() is False  # file pointers are stacked: KlgLogReader.cpp:114
    r.close()
    r = KlgLogReader(path, W, H)
    assert r.rewind() is True and r.getNext()[0] == f[0][0]
    r.close()


def test_truncated_file_raises(tmp_path):
    path = str(tmp_path / "e.klg")
    write_klg(path, frames(2))
    data = open(path, "rb").read()
    open(path, "wb").write(data[:len(data) // 3])
    r = KlgLogReader(path, W, H)
    with pytest.raises(IOError):
        r.getNext()
    r.close()


def rot(axis, angle):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def test_quaternion_matches_every_branch():
    for axis, angle in (((1, 2, 3), 0.3), ((1, 0, 0), 3.0), ((0, 1, 0), 3.1), ((0, 0, 1), 2.9), ((1, 1, 0), np.pi)):
        R = rot(axis, angle)
        x, y, z, w = quaternion_xyzw(R)
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.abs(Rq - R).max() < 1e-5
        assert abs(x * x + y * y + z * z + w * w - 1) < 1e-5


def test_pose_log_lines(tmp_path):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = rot((0, 0, 1), np.pi / 2)
    T[:3, 3] = (0.5, -1.25, 3.0)
    assert np.allclose(pose_7d(T), [0.5, -1.25, 3.0, 0, 0, np.sqrt(0.5), np.sqrt(0.5)], atol=1e-6)
    path = str(tmp_path / "poses-0.txt")
    write_pose_log(path, [(123456789012, np.eye(4)), (123456789045, T)])
    lines = open(path).read().splitlines()
    assert lines[0] == "123456789012 0 0 0 0 0 0 1"           # operator<< prints 6 significant digits
    assert lines[1] == "123456789045 0.5 -1.25 3 0 0 0.707107 0.707107"
