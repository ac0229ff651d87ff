"""Frame ingest for the mirror of camera_sync.camera (reference camera_sync/camera.py:98-118, 150-154).

The reference opens a cv.VideoCapture and `takePic()` returns what `.read()` yields: BGR frames that OpenCV has
converted on the CPU from what the camera delivers.  A UVC camera delivers packed YUYV 4:2:2; this module takes such
raw frames from any byte stream (a file of raw frames, a pipe from a capture tool such as `v4l2-ctl --stream-to=-`, a
V4L2 device node opened for read() I/O) and converts them on the GPU (camsync_ingest_yuyv).  It offers VideoCapture's
two methods the reference uses, `read()` and `release()`, so it can stand in as `Camera.captureStream`.  Opening camera
hardware itself (device enumeration, focus and resolution controls, camera.py:102-118) stays outside this package.
"""
import numpy as np

from .. import _hip


class RawYUYVStream:
    """VideoCapture-like source of raw YUYV frames of a fixed resolution (width, height), converted to BGR on the GPU."""

    def __init__(self, source, resolution, device=0):
        self._own = isinstance(source, (str, bytes))
        self._f = open(source, "rb", buffering=0) if self._own else source
        self._w, self._h = int(resolution[0]), int(resolution[1])
        if self._w % 8:
            raise ValueError("frame width must be a multiple of 8")
        self._device = device
        self._frame_bytes = self._w * self._h * 2
        # page-locked staging: the raw frame goes to the GPU without the driver's bounce copy
        self._pin = _hip.PinnedFrames((1, self._h, self._w, 2))

    def isOpened(self):
        return self._f is not None

    def _fill(self, view):
        got = 0
        while got < self._frame_bytes:
            n = self._f.readinto(view[got:])
            if not n:
                return False
            got += n
        return True

    def read(self):
        """(True, BGR frame [H,W,3] uint8) or (False, None) at the end of the stream, like cv.VideoCapture.read()."""
        if self._f is None:
            return False, None
        raw = self._pin.array.reshape(-1)
        if not self._fill(memoryview(raw)):
            return False, None
        ctx = _hip.get_context(self._h, self._w, 1, self._device)
        return True, ctx.ingest_yuyv(self._pin.array)[0]

    def read_batch(self, n):
        """Up to n frames at once: BGR [k,H,W,3] (k <= n, 0 at the end of the stream); one copy and one launch."""
        buf = np.empty((n, self._h, self._w, 2), np.uint8)
        k = 0
        while k < n and self._f is not None and self._fill(memoryview(buf[k].reshape(-1))):
            k += 1
        if k == 0:
            return np.empty((0, self._h, self._w, 3), np.uint8)
        ctx = _hip.get_context(self._h, self._w, k, self._device)
        return ctx.ingest_yuyv(buf[:k])

    def release(self):
        if self._f is not None and self._own:
            self._f.close()
        self._f = None
        self._pin = None
