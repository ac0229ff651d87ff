"""Frame ingest for the mirror of camera_sync.camera (reference camera_sync/camera.py:98-118, 150-154).

The reference opens a cv.VideoCapture and `takePic()` returns what `.read()` yields: BGR frames that OpenCV has
converted on the CPU from what the camera delivers.  A UVC camera delivers packed YUYV 4:2:2; this module takes such
raw frames from any byte stream (a file of raw frames, a pipe from a capture tool such as `v4l2-ctl --stream-to=-`, a
V4L2 device node opened for read() I/O) and converts them on the GPU (camsync_ingest_yuyv).  It offers VideoCapture's
two methods the reference uses, `read()` and `release()`, so it can stand in as `Camera.captureStream`.  Cameras that
deliver MJPG instead (what VideoCapture negotiates where YUYV is not offered at the resolution asked for) send one
baseline JPEG per frame: `MJPEGStream` splits such a byte stream into frames and decodes them on the GPU
(camsync_decode_jpeg_batch), and `imread` decodes a stored JPEG photo the same way (cv.imread of the reference's
pics/*.jpg).  The camera device itself is opened by `v4l2.V4L2Capture` (camera.py:102-118), which feeds the same two conversions.
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


def _jpeg_end(buf, start):
    """Index one past the EOI marker of the JPEG that starts at buf[start], or -1 if the buffer ends before it.
    Header segments are skipped by their lengths; in entropy-coded data a 0xFF is followed by 0x00 (a stuffed byte), a
    restart marker or the marker that ends the scan."""
    n = len(buf)
    i = start + 2
    while True:
        if i + 2 > n:
            return -1
        if buf[i] != 0xFF:
            raise ValueError("damaged JPEG stream (no marker where one is due)")
        m = buf[i + 1]
        if m == 0xFF:
            i += 1
            continue
        if m == 0xD9:
            return i + 2
        if m == 0xD8 or m == 0x01 or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        if i + 4 > n:
            return -1
        i += 2 + ((buf[i + 2] << 8) | buf[i + 3])
        if m == 0xDA:  # entropy-coded data up to the next marker that is not a restart marker
            while True:
                j = buf.find(b"\xff", i)
                if j < 0 or j + 1 >= n:
                    return -1
                if buf[j + 1] == 0 or 0xD0 <= buf[j + 1] <= 0xD7 or buf[j + 1] == 0xFF:
                    i = j + (1 if buf[j + 1] == 0xFF else 2)
                    continue
                i = j
                break


class MJPEGStream:
    """VideoCapture-like source of MJPEG frames (concatenated baseline JPEGs, as a UVC camera in MJPG mode or
    `v4l2-ctl --stream-to=-` delivers them) from a file name or any object with read(n); frames are decoded on the GPU."""

    def __init__(self, source, device=0, chunk=1 << 20):
        self._own = isinstance(source, (str, bytes))
        self._f = open(source, "rb") if self._own else source
        self._device = device
        self._chunk = chunk
        self._buf = bytearray()

    def isOpened(self):
        return self._f is not None

    def _next_frame(self):
        """The bytes of the next JPEG of the stream, or None at its end."""
        while self._f is not None:
            start = self._buf.find(b"\xff\xd8")
            if start >= 0:
                end = _jpeg_end(self._buf, start)
                if end > 0:
                    frame = bytes(self._buf[start:end])
                    del self._buf[:end]
                    return frame
            more = self._f.read(self._chunk)
            if not more:
                return None
            self._buf += more
        return None

    def _decode(self, frames):
        h, w, _ = _hip.jpeg_info(frames[0])
        ctx = _hip.get_context(max(h, 32), max(w, 32), len(frames), self._device)
        return ctx.decode_jpeg(frames)

    def read(self):
        """(True, BGR frame [H,W,3] uint8) or (False, None) at the end of the stream, like cv.VideoCapture.read()."""
        frame = self._next_frame()
        if frame is None:
            return False, None
        return True, self._decode([frame])[0]

    def read_batch(self, n):
        """Up to n frames at once (they must be of one size): BGR [k,H,W,3]; the frames are decoded side by side."""
        frames = []
        while len(frames) < n:
            f = self._next_frame()
            if f is None:
                break
            frames.append(f)
        if not frames:
            return np.empty((0, 0, 0, 3), np.uint8)
        return self._decode(frames)

    def release(self):
        if self._f is not None and self._own:
            self._f.close()
        self._f = None


def imread(path, device=0):
    """cv.imread for a baseline JPEG file: BGR [H,W,3] uint8, decoded on the GPU."""
    with open(path, "rb") as f:
        data = f.read()
    h, w, _ = _hip.jpeg_info(data)
    return _hip.get_context(max(h, 32), max(w, 32), 1, device).decode_jpeg([data])[0]
