"""Sequence I/O around the path: the `.klg` log reader of the reference
(GUI/Tools/KlgLogReader.cpp:20-130) and the pose-log writer of MultiMotionFusion::exportPoses
(Core/MultiMotionFusion.cpp:1020-1045), so that recorded sequences can be replayed through the HIP path and
the trajectories diffed against a reference run.  Host-side file plumbing only: no arithmetic of the path.

.klg layout: int32 numFrames, then per frame int64 timestamp, int32 depthSize, int32 imageSize, depthSize
bytes (uint16 millimetres, zlib-compressed unless depthSize == 2 * pixels), imageSize bytes (RGB u8, JPEG
unless imageSize == 3 * pixels; imageSize == 0: black).
"""
import io
import struct
import zlib

import numpy as np


class KlgLogReader:
    def __init__(self, file, width=640, height=480, flipColors=False):
        self.file, self.width, self.height, self.flipColors = file, width, height, flipColors
        self.numPixels = width * height
        self.fp = open(file, "rb")
        head = self.fp.read(4)
        if len(head) != 4:
            raise ValueError("Could not open log-file: " + file)  # KlgLogReader.cpp:33
        self.numFrames = struct.unpack("<i", head)[0]
        self.currentFrame = 0
        self.filePointers = []
        self.timestamp, self.depth, self.rgb = 0, None, None

    def getNumFrames(self):
        return self.numFrames

    def hasMore(self):
        return self.currentFrame + 1 < self.numFrames  # KlgLogReader.cpp:112 (the last frame is never served)

    def _read(self, n):
        b = self.fp.read(n)
        if len(b) != n:
            raise IOError("truncated .klg file")  # CHECK_THROW
        return b

    def _header(self):
        self.timestamp, depth_size, rgb_size = struct.unpack("<qii", self._read(16))
        return depth_size, rgb_size

    def getNext(self):
        """-> (timestamp, depth float32 metres [H,W], rgb u8 [H,W,3])"""
        self.filePointers.append(self.fp.tell())
        return self._core()

    def getPrevious(self):
        assert self.filePointers
        self.fp.seek(self.filePointers.pop())
        return self._core()

    def _core(self):
        depth_size, rgb_size = self._header()
        depth_bytes = self._read(depth_size)
        rgb_bytes = self._read(rgb_size) if rgb_size > 0 else b""
        if depth_size != self.numPixels * 2:
            depth_bytes = zlib.decompress(depth_bytes)
        d16 = np.frombuffer(depth_bytes, np.uint16, self.numPixels).reshape(self.height, self.width)
        self.depth = d16.astype(np.float32) * np.float32(0.001)  # convertTo(CV_32FC1, 0.001)
        if rgb_size == 0:
            self.rgb = np.zeros((self.height, self.width, 3), np.uint8)
        elif rgb_size != self.numPixels * 3:
            from PIL import Image  # the reference decodes with its JPEGLoader (libjpeg)
            self.rgb = np.asarray(Image.open(io.BytesIO(rgb_bytes)).convert("RGB"), np.uint8).copy()
        else:
            self.rgb = np.frombuffer(rgb_bytes, np.uint8).reshape(self.height, self.width, 3).copy()
        if self.flipColors:
            self.rgb = self.rgb[:, :, ::-1].copy()
        self.currentFrame += 1
        return self.timestamp, self.depth, self.rgb

    def fastForward(self, frame):
        while self.currentFrame < frame and self.hasMore():
            self.filePointers.append(self.fp.tell())
            depth_size, rgb_size = self._header()
            self.fp.seek(depth_size + max(rgb_size, 0), 1)
            self.currentFrame += 1

    def rewind(self):
        if not self.filePointers:  # KlgLogReader.cpp:114-128
            self.fp.seek(4)
            self.currentFrame = 0
            return True
        return False

    def close(self):
        self.fp.close()


def write_klg(file, frames, compress_depth=True, jpeg_quality=None):
    """frames: iterable of (timestamp, depth [H,W] metres or uint16 mm, rgb [H,W,3] u8).  Writes what the
    reference's loggers write: zlib depth, raw or JPEG colour."""
    frames = list(frames)
    with open(file, "wb") as fp:
        fp.write(struct.pack("<i", len(frames)))
        for ts, depth, rgb in frames:
            d16 = depth if depth.dtype == np.uint16 else np.rint(np.asarray(depth, np.float64) * 1000.0).astype(np.uint16)
            db = np.ascontiguousarray(d16).tobytes()
            if compress_depth:
                db = zlib.compress(db)
            if jpeg_quality is None:
                rb = np.ascontiguousarray(rgb, np.uint8).tobytes()
            else:
                from PIL import Image
                buf = io.BytesIO()
                Image.fromarray(np.ascontiguousarray(rgb, np.uint8)).save(buf, format="JPEG", quality=jpeg_quality)
                rb = buf.getvalue()
            fp.write(struct.pack("<qii", int(ts), len(db), len(rb)))
            fp.write(db)
            fp.write(rb)


def quaternion_xyzw(R):
    """Eigen::Quaternionf(rotation matrix).coeffs() = (x, y, z, w): Eigen's branch on the trace / the largest
    diagonal element (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl), in float32."""
    R = np.asarray(R, np.float32)
    t = np.float32(R[0, 0] + R[1, 1] + R[2, 2])
    q = np.zeros(4, np.float32)  # x y z w
    if t > 0:
        t = np.sqrt(np.float32(t + np.float32(1.0)))
        q[3] = np.float32(0.5) * t
        t = np.float32(0.5) / t
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(np.float32(R[i, i] - R[j, j] - R[k, k] + np.float32(1.0)))
        q[i] = np.float32(0.5) * t
        t = np.float32(0.5) / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def pose_7d(T):
    """Model.cpp:653-660: (x, y, z, qx, qy, qz, qw) of a 4x4 pose"""
    T = np.asarray(T, np.float32)
    return np.concatenate([T[:3, 3], quaternion_xyzw(T[:3, :3])]).astype(np.float32)


def write_pose_log(filename, poses):
    """MultiMotionFusion::exportPoses (MultiMotionFusion.cpp:1020-1045): one line per frame,
    `ts x y z qx qy qz qw`, floats through operator<< (6 significant digits).  poses: [(ts, 4x4)]"""
    with open(filename, "w") as fs:
        for ts, T in poses:
            fs.write(str(int(ts)))
            for v in pose_7d(T):
                fs.write(" " + "%g" % float(v))
            fs.write("\n")
