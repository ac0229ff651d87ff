"""cv.VideoCapture(deviceId) for a V4L2 camera (reference camera_sync/camera.py:98-118, 150-154).

The reference opens the capture device with OpenCV, switches autofocus off, sets the focus and the frame size, and
`takePic()` returns what `.read()` yields.  This module does the same against the kernel's V4L2 interface directly
(`/dev/video<deviceId>`, ioctls + memory-mapped streaming buffers; plain Python, no OpenCV): the camera's raw frames
— packed YUYV 4:2:2 where the device offers it at the size asked for (the format OpenCV's V4L2 backend prefers too),
MJPG otherwise — go to the GPU as they come and are converted there (camsync_ingest_yuyv / camsync_decode_jpeg_batch),
so the frame the caller gets is the BGR frame VideoCapture would have produced on the CPU.

Like VideoCapture, a device that cannot be opened does not raise: `isOpened()` is False and `read()` returns
`(False, None)`.  Only Linux has V4L2; elsewhere the capture is never opened.

The structure layouts and ioctl numbers below are those of <linux/videodev2.h> on 64-bit Linux;
tests/test_v4l2.py compiles a probe against that header and compares every one of them.  No camera exists where this
package is built and tested, so the streaming loop itself has only been exercised against the ioctl contract, not
against hardware.
"""
import ctypes as C
import mmap
import os
import select
import sys

import numpy as np

from .. import _hip

# cv.CAP_PROP_* identifiers the reference uses (camera.py:108-118)
CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FOCUS, CAP_PROP_AUTOFOCUS = 3, 4, 28, 39

V4L2_BUF_TYPE_VIDEO_CAPTURE = 1
V4L2_MEMORY_MMAP = 1
V4L2_FIELD_ANY = 0
V4L2_CAP_VIDEO_CAPTURE = 0x00000001
V4L2_CAP_STREAMING = 0x04000000
V4L2_CAP_DEVICE_CAPS = 0x80000000
V4L2_CID_FOCUS_ABSOLUTE = 0x009A090A
V4L2_CID_FOCUS_AUTO = 0x009A090C


def fourcc(s):
    return ord(s[0]) | (ord(s[1]) << 8) | (ord(s[2]) << 16) | (ord(s[3]) << 24)


V4L2_PIX_FMT_YUYV = fourcc("YUYV")
V4L2_PIX_FMT_MJPEG = fourcc("MJPG")


class v4l2_capability(C.Structure):
    _fields_ = [("driver", C.c_uint8 * 16), ("card", C.c_uint8 * 32), ("bus_info", C.c_uint8 * 32), ("version", C.c_uint32),
                ("capabilities", C.c_uint32), ("device_caps", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class v4l2_pix_format(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("pixelformat", C.c_uint32), ("field", C.c_uint32),
                ("bytesperline", C.c_uint32), ("sizeimage", C.c_uint32), ("colorspace", C.c_uint32), ("priv", C.c_uint32),
                ("flags", C.c_uint32), ("ycbcr_enc", C.c_uint32), ("quantization", C.c_uint32), ("xfer_func", C.c_uint32)]


class _fmt_union(C.Union):
    # (the kernel's union holds pointers in another member, hence its 8-byte alignment)
    _fields_ = [("pix", v4l2_pix_format), ("raw_data", C.c_uint8 * 200), ("_align", C.c_uint64)]


class v4l2_format(C.Structure):
    _fields_ = [("type", C.c_uint32), ("fmt", _fmt_union)]


class v4l2_requestbuffers(C.Structure):
    _fields_ = [("count", C.c_uint32), ("type", C.c_uint32), ("memory", C.c_uint32), ("capabilities", C.c_uint32),
                ("flags", C.c_uint8), ("reserved", C.c_uint8 * 3)]


class _timeval(C.Structure):
    _fields_ = [("tv_sec", C.c_long), ("tv_usec", C.c_long)]


class v4l2_timecode(C.Structure):
    _fields_ = [("type", C.c_uint32), ("flags", C.c_uint32), ("frames", C.c_uint8), ("seconds", C.c_uint8),
                ("minutes", C.c_uint8), ("hours", C.c_uint8), ("userbits", C.c_uint8 * 4)]


class _buf_m(C.Union):
    _fields_ = [("offset", C.c_uint32), ("userptr", C.c_ulong), ("planes", C.c_void_p), ("fd", C.c_int32)]


class v4l2_buffer(C.Structure):
    _fields_ = [("index", C.c_uint32), ("type", C.c_uint32), ("bytesused", C.c_uint32), ("flags", C.c_uint32),
                ("field", C.c_uint32), ("timestamp", _timeval), ("timecode", v4l2_timecode), ("sequence", C.c_uint32),
                ("memory", C.c_uint32), ("m", _buf_m), ("length", C.c_uint32), ("reserved2", C.c_uint32),
                ("request_fd", C.c_int32)]


class v4l2_control(C.Structure):
    _fields_ = [("id", C.c_uint32), ("value", C.c_int32)]


def _ioc(direction, nr, size):
    return (direction << 30) | (size << 16) | (ord("V") << 8) | nr


_R, _W = 2, 1
VIDIOC_QUERYCAP = _ioc(_R, 0, C.sizeof(v4l2_capability))
VIDIOC_G_FMT = _ioc(_R | _W, 4, C.sizeof(v4l2_format))
VIDIOC_S_FMT = _ioc(_R | _W, 5, C.sizeof(v4l2_format))
VIDIOC_REQBUFS = _ioc(_R | _W, 8, C.sizeof(v4l2_requestbuffers))
VIDIOC_QUERYBUF = _ioc(_R | _W, 9, C.sizeof(v4l2_buffer))
VIDIOC_QBUF = _ioc(_R | _W, 15, C.sizeof(v4l2_buffer))
VIDIOC_DQBUF = _ioc(_R | _W, 17, C.sizeof(v4l2_buffer))
VIDIOC_STREAMON = _ioc(_W, 18, C.sizeof(C.c_int))
VIDIOC_STREAMOFF = _ioc(_W, 19, C.sizeof(C.c_int))
VIDIOC_S_CTRL = _ioc(_R | _W, 28, C.sizeof(v4l2_control))


class V4L2Capture:
    """The part of cv.VideoCapture the reference uses: set(), read(), isOpened(), release()."""

    def __init__(self, deviceId, resolution=(640, 480), device=0, buffers=4, timeout_s=2.0):
        self._fd = -1
        self._gpu = device
        self._w, self._h = int(resolution[0]), int(resolution[1])
        self._nbuf = buffers
        self._timeout = timeout_s
        self._maps = []
        self._streaming = False
        self._fmt = 0
        self._ioctl = None
        path = deviceId if isinstance(deviceId, str) else "/dev/video%d" % int(deviceId)
        if not sys.platform.startswith("linux"):
            return
        try:
            import fcntl
            self._ioctl = fcntl.ioctl
            self._fd = os.open(path, os.O_RDWR)
            cap = v4l2_capability()
            self._ioctl(self._fd, VIDIOC_QUERYCAP, cap)
            caps = cap.device_caps if cap.capabilities & V4L2_CAP_DEVICE_CAPS else cap.capabilities
            if not (caps & V4L2_CAP_VIDEO_CAPTURE) or not (caps & V4L2_CAP_STREAMING):
                raise OSError("not a streaming video capture device")
        except OSError:
            self.release()

    # -- VideoCapture's surface ----------------------------------------------------------
    def isOpened(self):
        return self._fd >= 0

    def set(self, prop, value):
        """CAP_PROP_FRAME_WIDTH / _HEIGHT (take effect when streaming starts, at the first read), CAP_PROP_AUTOFOCUS,
        CAP_PROP_FOCUS (V4L2_CID_FOCUS_AUTO / _ABSOLUTE).  False if the device refuses, as VideoCapture.set."""
        if self._fd < 0:
            return False
        if prop == CAP_PROP_FRAME_WIDTH or prop == CAP_PROP_FRAME_HEIGHT:
            if self._streaming:
                self._stop()
            if prop == CAP_PROP_FRAME_WIDTH:
                self._w = int(value)
            else:
                self._h = int(value)
            return True
        cid = {CAP_PROP_AUTOFOCUS: V4L2_CID_FOCUS_AUTO, CAP_PROP_FOCUS: V4L2_CID_FOCUS_ABSOLUTE}.get(prop)
        if cid is None:
            return False
        try:
            self._ioctl(self._fd, VIDIOC_S_CTRL, v4l2_control(cid, int(value)))
            return True
        except OSError:
            return False

    def read(self):
        """(True, BGR frame [H,W,3] uint8) or (False, None): the next frame of the camera, converted on the GPU."""
        if self._fd < 0:
            return False, None
        try:
            if not self._streaming:
                self._start()
            ready, _, _ = select.select([self._fd], [], [], self._timeout)
            if not ready:
                return False, None
            buf = v4l2_buffer(type=V4L2_BUF_TYPE_VIDEO_CAPTURE, memory=V4L2_MEMORY_MMAP)
            self._ioctl(self._fd, VIDIOC_DQBUF, buf)
            raw = bytes(self._maps[buf.index][:buf.bytesused])
            self._ioctl(self._fd, VIDIOC_QBUF, buf)
        except OSError:
            return False, None
        return self._convert(raw)

    def release(self):
        if self._fd >= 0:
            self._stop()
            try:
                os.close(self._fd)
            except OSError:
                pass
        self._fd = -1

    # -- streaming -------------------------------------------------------------------------
    def _negotiate(self):
        """Ask for YUYV at the size wanted, MJPG if the driver answers with another format; the driver may also
        adjust the size (VIDIOC_S_FMT returns what it will deliver)."""
        for want in (V4L2_PIX_FMT_YUYV, V4L2_PIX_FMT_MJPEG):
            f = v4l2_format(type=V4L2_BUF_TYPE_VIDEO_CAPTURE)
            f.fmt.pix.width, f.fmt.pix.height = self._w, self._h
            f.fmt.pix.pixelformat, f.fmt.pix.field = want, V4L2_FIELD_ANY
            try:
                self._ioctl(self._fd, VIDIOC_S_FMT, f)
            except OSError:
                continue
            if f.fmt.pix.pixelformat in (V4L2_PIX_FMT_YUYV, V4L2_PIX_FMT_MJPEG):
                self._fmt = f.fmt.pix.pixelformat
                self._w, self._h = int(f.fmt.pix.width), int(f.fmt.pix.height)
                return
        raise OSError("the device offers neither YUYV nor MJPG")

    def _start(self):
        self._negotiate()
        req = v4l2_requestbuffers(count=self._nbuf, type=V4L2_BUF_TYPE_VIDEO_CAPTURE, memory=V4L2_MEMORY_MMAP)
        self._ioctl(self._fd, VIDIOC_REQBUFS, req)
        if req.count < 1:
            raise OSError("the driver granted no buffers")
        for i in range(req.count):
            buf = v4l2_buffer(index=i, type=V4L2_BUF_TYPE_VIDEO_CAPTURE, memory=V4L2_MEMORY_MMAP)
            self._ioctl(self._fd, VIDIOC_QUERYBUF, buf)
            self._maps.append(mmap.mmap(self._fd, buf.length, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE, offset=buf.m.offset))
            self._ioctl(self._fd, VIDIOC_QBUF, buf)
        self._ioctl(self._fd, VIDIOC_STREAMON, C.c_int(V4L2_BUF_TYPE_VIDEO_CAPTURE))
        self._streaming = True

    def _stop(self):
        if self._streaming:
            try:
                self._ioctl(self._fd, VIDIOC_STREAMOFF, C.c_int(V4L2_BUF_TYPE_VIDEO_CAPTURE))
            except OSError:
                pass
        for m in self._maps:
            m.close()
        self._maps = []
        if self._streaming:
            try:  # give the buffers back
                self._ioctl(self._fd, VIDIOC_REQBUFS, v4l2_requestbuffers(count=0, type=V4L2_BUF_TYPE_VIDEO_CAPTURE, memory=V4L2_MEMORY_MMAP))
            except OSError:
                pass
        self._streaming = False

    def _convert(self, raw):
        """One raw frame as the camera delivered it -> (True, BGR) on the GPU; (False, None) for a damaged frame."""
        try:
            if self._fmt == V4L2_PIX_FMT_YUYV:
                if len(raw) < self._w * self._h * 2:
                    return False, None
                yuyv = np.frombuffer(raw, np.uint8, self._w * self._h * 2).reshape(self._h, self._w, 2)
                ctx = _hip.get_context(max(self._h, 32), max(self._w, 32), 1, self._gpu)
                return True, ctx.ingest_yuyv(yuyv)
            h, w, _ = _hip.jpeg_info(raw)
            ctx = _hip.get_context(max(h, 32), max(w, 32), 1, self._gpu)
            return True, ctx.decode_jpeg([raw])[0]
        except (ValueError, _hip.CamsyncError):
            return False, None
