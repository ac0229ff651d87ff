"""ctypes binding of libcamsync_hip.so (include/camsync.h).

This is the only way the package reaches the GPU.  There is no CPU fallback:
if the library is missing, or no GPU is visible, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _dict4x4_50 as _dict

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcamsync_hip.so")

N_STAGES = 9
E_ARG, E_HIP, E_CAPACITY, E_NODEVICE, E_NUMERIC = -1, -2, -3, -4, -5  # include/camsync.h
STAGE_NAMES = ("gray_threshold", "contour_starts", "trace", "quad_fit", "identify", "select", "subpix", "pnp", "total")


class CamsyncError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libcamsync_hip error {code}: {msg}")
        self.code = code


# camsync_allreduce_fn: int (*)(void* user, double* d_buf, int n, void* hip_stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
CALIB_WORK_DOUBLES = 192
E_COMM = -6


class Params(C.Structure):
    """camsync_params — cv.aruco.DetectorParameters as aruco.py:249-254 sets it."""

    _fields_ = [
        ("adaptiveThreshWinSizeMin", C.c_int32),
        ("adaptiveThreshWinSizeMax", C.c_int32),
        ("adaptiveThreshWinSizeStep", C.c_int32),
        ("adaptiveThreshConstant", C.c_double),
        ("minMarkerPerimeterRate", C.c_double),
        ("maxMarkerPerimeterRate", C.c_double),
        ("polygonalApproxAccuracyRate", C.c_double),
        ("minCornerDistanceRate", C.c_double),
        ("minDistanceToBorder", C.c_int32),
        ("markerBorderBits", C.c_int32),
        ("minMarkerDistanceRate", C.c_double),
        ("minGroupDistance", C.c_float),
        ("cornerRefinementMethod", C.c_int32),
        ("cornerRefinementWinSize", C.c_int32),
        ("relativeCornerRefinmentWinSize", C.c_float),
        ("cornerRefinementMaxIterations", C.c_int32),
        ("cornerRefinementMinAccuracy", C.c_double),
        ("perspectiveRemovePixelPerCell", C.c_int32),
        ("perspectiveRemoveIgnoredMarginPerCell", C.c_double),
        ("maxErroneousBitsInBorderRate", C.c_double),
        ("minOtsuStdDev", C.c_double),
        ("errorCorrectionRate", C.c_double),
    ]


_lib = None
_lib_lock = threading.Lock()


def load_library() -> C.CDLL:
    """Load libcamsync_hip.so; raises if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `make -C cameracalibration_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback"
            )
        lib = C.CDLL(LIB_PATH)
        vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
        lib.camsync_device_count.restype = i32
        lib.camsync_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32]
        lib.camsync_destroy.argtypes = [vp]
        lib.camsync_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
        lib.camsync_host_free.argtypes = [vp]
        lib.camsync_destroy.restype = None
        lib.camsync_last_error.argtypes = [vp]
        lib.camsync_last_error.restype = C.c_char_p
        lib.camsync_default_params.argtypes = [C.POINTER(Params)]
        lib.camsync_default_params.restype = None
        lib.camsync_set_params.argtypes = [vp, C.POINTER(Params)]
        lib.camsync_get_params.argtypes = [vp, C.POINTER(Params)]
        lib.camsync_set_dictionary.argtypes = [vp, vp, i32, i32, i32]
        lib.camsync_gray_threshold.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
        lib.camsync_detect_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, vp]
        lib.camsync_debug_counts.argtypes = [vp, i32, vp]
        lib.camsync_debug_lists.argtypes = [vp, vp, vp]
        lib.camsync_debug_planes.argtypes = [vp, i32, i32, vp]
        lib.camsync_debug_starts.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
        lib.camsync_debug_quads.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
        lib.camsync_debug_contours.argtypes = [vp, i32, i32, vp, vp, vp, vp]
        lib.camsync_debug_points.argtypes = [vp, i32, i32, vp]
        lib.camsync_debug_limit_capacity.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.camsync_debug_restore_capacity.argtypes = [vp]
        lib.camsync_debug_identify_all.argtypes = [vp, i32]
        lib.camsync_debug_shrink_lists.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
        lib.camsync_debug_regrown.argtypes = [vp]
        lib.camsync_last_extent.argtypes = [vp, vp, vp]
        lib.camsync_fuse_moving_markers.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, vp, vp]
        lib.camsync_pnp_batch.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]
        lib.camsync_pnp_batch_noalloc.argtypes = lib.camsync_pnp_batch.argtypes
        lib.camsync_ippe_square_batch.argtypes = [vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp]
        lib.camsync_detect_pnp_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32,
                                                 i32, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.camsync_detect_pnp_frames.argtypes = lib.camsync_detect_pnp_batch.argtypes
        lib.camsync_detect_pnp_resident.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32,
                                                    i32, vp, vp, vp, vp, vp, vp]
        lib.camsync_chess_candidates.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
        lib.camsync_corner_subpix.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, i32, dbl, vp]
        lib.camsync_optimal_new_camera_matrix.argtypes = [vp, vp, i32, i32, i32, dbl, i32, i32, vp, vp]
        lib.camsync_undistort_batch.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]
        lib.camsync_ingest_yuyv.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32]
        lib.camsync_jpeg_info.argtypes = [vp, C.c_size_t, vp, vp, vp]
        lib.camsync_decode_jpeg_batch.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32]
        lib.camsync_calibrate.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
        lib.camsync_calibrate_dist.argtypes = [vp, vp, vp, i32, i32, C.c_int64, i32, i32, i32, i32, ALLREDUCE_FN, vp, vp,
                                               vp, vp, vp, vp, vp, vp]
        lib.camsync_calib_begin.argtypes = [vp, vp, vp, i32, i32, i32]
        lib.camsync_calib_focal_acc.argtypes = [vp, i32, i32, vp]
        lib.camsync_calib_set_intrinsics.argtypes = [vp, vp, i32]
        lib.camsync_calib_accumulate.argtypes = [vp, dbl, vp, i32]
        lib.camsync_calib_solve.argtypes = [vp, i32, vp]
        lib.camsync_calib_step.argtypes = [vp, vp, vp]
        lib.camsync_calib_accept.argtypes = [vp]
        lib.camsync_calib_get.argtypes = [vp, vp, vp, vp]
        lib.camsync_calib_n_intrinsics.argtypes = [vp]
        lib.camsync_enable_timers.argtypes = [vp, i32]
        lib.camsync_get_timers.argtypes = [vp, vp, vp]
        lib.camsync_reset_timers.argtypes = [vp]
        _lib = lib
        return lib


def _ptr(a):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _cam(K, dist, B):
    """Normalise camera matrix / distortion to contiguous f64 and report layout."""
    K = np.ascontiguousarray(K, dtype=np.float64)
    per_frame = 1 if K.ndim == 3 else 0
    K = K.reshape(-1, 9)
    if per_frame and K.shape[0] != B:
        raise ValueError("per-frame camera matrices must have one entry per frame")
    if dist is None:
        d = np.zeros((K.shape[0], 0), dtype=np.float64)
    else:
        d = np.ascontiguousarray(dist, dtype=np.float64).reshape(K.shape[0], -1)
    if d.shape[1] > 8:
        if np.any(d[:, 8:] != 0):
            raise ValueError("thin-prism / tilt distortion terms are not supported")
        d = np.ascontiguousarray(d[:, :8])
    return K, d, per_frame


def jpeg_info(data):
    """(height, width, components) of a JPEG file / MJPEG frame, from its header (host only)."""
    buf = data if isinstance(data, np.ndarray) else np.frombuffer(bytes(data), np.uint8)
    h = C.c_int32(0)
    w = C.c_int32(0)
    n = C.c_int32(0)
    if load_library().camsync_jpeg_info(_ptr(buf), C.c_size_t(buf.size), C.byref(h), C.byref(w), C.byref(n)):
        raise ValueError("not a JPEG, or its headers are damaged")
    return h.value, w.value, n.value


class PinnedFrames:
    """A uint8 numpy array in page-locked host memory (camsync_host_alloc): frames written here are copied to the GPU
    at the link's rate and without the driver's staging copy.  Keep the object alive as long as `.array` is used."""

    def __init__(self, shape):
        lib = load_library()
        self._lib = lib
        self._p = C.c_void_p()
        n = int(np.prod(shape))
        rc = lib.camsync_host_alloc(C.byref(self._p), C.c_size_t(n))
        if rc:
            raise CamsyncError(rc, "camsync_host_alloc(%d bytes) failed" % n)
        buf = (C.c_uint8 * n).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=np.uint8).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            self._lib.camsync_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One camsync_ctx: a GPU, a stream and the detection scratch."""

    def __init__(self, max_h: int, max_w: int, max_batch: int = 8, device: int = 0, dictionary=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.camsync_create(C.byref(self._h), device, max_h, max_w, max_batch)
        if rc != 0:
            msg = self._lib.camsync_last_error(self._h).decode() if self._h else (
                "no usable GPU" if rc == E_NODEVICE else "bad context size (max_h, max_w >= 32, max_batch >= 1)")
            if self._h:
                self._lib.camsync_destroy(self._h)
                self._h = C.c_void_p()
            raise CamsyncError(rc, msg)
        self.max_h, self.max_w, self.max_batch, self.device = max_h, max_w, max_batch, device
        codes = _dict.CODES if dictionary is None else np.ascontiguousarray(dictionary, dtype=np.uint16)
        self._check(self._lib.camsync_set_dictionary(self._h, _ptr(codes), len(codes), _dict.MARKER_SIZE,
                                                     _dict.MAX_CORRECTION_BITS))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.camsync_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CamsyncError(rc, self._lib.camsync_last_error(self._h).decode())

    # -- parameters ----------------------------------------------------
    def default_params(self) -> Params:
        p = Params()
        self._lib.camsync_default_params(C.byref(p))
        return p

    def set_params(self, p: Params):
        self._check(self._lib.camsync_set_params(self._h, C.byref(p)))

    def get_params(self) -> Params:
        p = Params()
        self._check(self._lib.camsync_get_params(self._h, C.byref(p)))
        return p

    # -- detection -----------------------------------------------------
    @staticmethod
    def _frames(frames):
        f = np.ascontiguousarray(frames, dtype=np.uint8)
        if f.ndim == 2:
            f = f[None, :, :, None]
        elif f.ndim == 3:
            # (H,W,C) single frame or (B,H,W) gray batch: channel counts are 1 or 3
            f = f[None] if f.shape[-1] in (1, 3) else f[..., None]
        if f.ndim != 4:
            raise ValueError("frames must be [B,H,W,C], [H,W,C], [B,H,W] or [H,W]")
        return f

    def gray_threshold(self, frames):
        f = self._frames(frames)
        B, H, W, Cn = f.shape
        gray = np.empty((B, H, W), np.uint8)
        binp = np.empty((B, H, W), np.uint8)
        self._check(self._lib.camsync_gray_threshold(self._h, _ptr(f), 0, B, H, W, Cn, _ptr(gray), _ptr(binp)))
        return gray, binp

    def last_extent(self):
        """(markers, rejected): the largest unclamped per-frame counts of the last detection call."""
        m = C.c_int32(0)
        r = C.c_int32(0)
        self._check(self._lib.camsync_last_extent(self._h, C.byref(m), C.byref(r)))
        return int(m.value), int(r.value)

    def _wider(self, max_markers, max_rejected=None):
        """Output widths for a second call when a frame of the last one had more markers (rejected candidates) than
        the outputs hold -- cv.aruco.detectMarkers returns all of them (aruco.py:256-264) -- or None."""
        m, r = self.last_extent()
        if m <= max_markers and (max_rejected is None or r <= max_rejected):
            return None
        return max(m, max_markers), (None if max_rejected is None else max(r, max_rejected))

    def detect(self, frames, max_markers: int = 64, max_rejected: int = 1024):
        """Returns (ids [B,M], corners [B,M,4,2], counts [B], rejected [B,R,4,2], rej_counts [B]).  M and R are at
        least max_markers / max_rejected, and wider when a frame holds more (the call is then made a second time)."""
        f = self._frames(frames)
        B, H, W, Cn = f.shape
        ids = np.full((B, max_markers), -1, np.int32)
        corners = np.zeros((B, max_markers, 4, 2), np.float32)
        counts = np.zeros(B, np.int32)
        rej = np.zeros((B, max_rejected, 4, 2), np.float32)
        rejc = np.zeros(B, np.int32)
        self._check(self._lib.camsync_detect_batch(self._h, _ptr(f), 0, B, H, W, Cn, max_markers, _ptr(ids),
                                                   _ptr(corners), _ptr(counts), max_rejected, _ptr(rej), _ptr(rejc)))
        wider = self._wider(max_markers, max_rejected)
        if wider:
            return self.detect(f, *wider)
        return ids, corners, counts, rej, rejc

    def debug_counts(self, frame=0):
        out = np.zeros(8, np.int32)
        self._check(self._lib.camsync_debug_counts(self._h, frame, _ptr(out)))
        return dict(candidates=int(out[0]), contours=int(out[1]), points=int(out[2]), markers=int(out[3]),
                    overflow=int(out[4]), quads=int(out[5]), long_candidates=int(out[6]), queue_head=int(out[7]))

    LIST_NAMES = ("starts", "long_starts", "contours", "points", "store_tasks", "segments", "copy_tasks", "seeds")

    def debug_lists(self):
        """{list: (entries used by the last chunk, capacity)} of the chunk-wide scratch lists."""
        used = np.zeros(8, np.uint32); cap = np.zeros(8, np.uint32)
        self._check(self._lib.camsync_debug_lists(self._h, _ptr(used), _ptr(cap)))
        return {n: (int(u), int(k)) for n, u, k in zip(self.LIST_NAMES, used, cap)}

    def debug_planes(self, frame, scale, H, W):
        """Threshold bits [H,W] (0/1) of `frame` at `scale` from the last detection chunk."""
        out = np.zeros((H, W), np.uint8)
        self._check(self._lib.camsync_debug_planes(self._h, frame, scale, _ptr(out)))
        return out

    def debug_starts(self, frames, cap=1 << 22):
        """Contour-start candidates of `frames` (threshold + start stages only): sorted uint64 array, see camsync.h."""
        f = self._frames(frames)
        B, H, W, Cn = f.shape
        out = np.zeros(cap, np.uint64)
        n = self._lib.camsync_debug_starts(self._h, _ptr(f), B, H, W, Cn, cap, _ptr(out))
        if n < 0:
            self._check(n)
        assert n <= cap
        return np.sort(out[:n])

    def debug_shrink_lists(self, starts=0, contours=0, points=0, tasks=0, seeds=0):
        self._check(self._lib.camsync_debug_shrink_lists(self._h, int(starts), int(contours), int(points), int(tasks), int(seeds)))

    def debug_regrown(self):
        return int(self._lib.camsync_debug_regrown(self._h))

    def debug_identify_all(self, on=True):
        """Identify every quad candidate in the following calls (debug_quads then reports an id for each)."""
        self._check(self._lib.camsync_debug_identify_all(self._h, 1 if on else 0))

    def debug_quads(self, frame=0, cap=1024):
        c = np.zeros((cap, 4, 2), np.float32)
        sc, key, ln, idv, rot = (np.zeros(cap, np.int32) for _ in range(5))
        n = self._lib.camsync_debug_quads(self._h, frame, cap, _ptr(c), _ptr(sc), _ptr(key), _ptr(ln), _ptr(idv), _ptr(rot))
        if n < 0:
            self._check(n)
        n = min(n, cap)
        return dict(corners=c[:n], scale=sc[:n], scan_key=key[:n], contour_len=ln[:n], id=idv[:n], rot=rot[:n])

    def debug_contours(self, frame=0, cap=8192):
        sc, key, hole, ln = (np.zeros(cap, np.int32) for _ in range(4))
        n = self._lib.camsync_debug_contours(self._h, frame, cap, _ptr(sc), _ptr(key), _ptr(hole), _ptr(ln))
        if n < 0:
            self._check(n)
        n = min(n, cap)
        return dict(scale=sc[:n], start_key=key[:n], is_hole=hole[:n], length=ln[:n])

    def debug_points(self, frame=0, cap=1 << 20):
        xy = np.zeros((cap, 2), np.int32)
        n = self._lib.camsync_debug_points(self._h, frame, cap, _ptr(xy))
        if n < 0:
            self._check(n)
        return xy[:min(n, cap)]

    def debug_limit_capacity(self, contours=0, points=0, tasks=0):
        """Testing aid: lower the list capacities (0 = unchanged) so that one small frame overflows them."""
        self._check(self._lib.camsync_debug_limit_capacity(self._h, contours, points, tasks))

    def debug_restore_capacity(self):
        self._check(self._lib.camsync_debug_restore_capacity(self._h))

    # -- pose ----------------------------------------------------------
    def pnp(self, obj, img, npts, K, dist, noalloc=False):
        """camsync_pnp_batch, or with noalloc=True camsync_pnp_batch_noalloc (no device allocation; B <= 256,
        B * N <= 16384)."""
        obj = np.ascontiguousarray(obj, dtype=np.float32)
        img = np.ascontiguousarray(img, dtype=np.float32)
        B, N = obj.shape[0], obj.shape[1]
        npts = np.ascontiguousarray(npts, dtype=np.int32)
        K, d, per_frame = _cam(K, dist, B)
        rvec = np.zeros((B, 3)); tvec = np.zeros((B, 3)); ok = np.zeros(B, np.uint8)
        mae = np.zeros(B); rmse = np.zeros(B)
        fn = self._lib.camsync_pnp_batch_noalloc if noalloc else self._lib.camsync_pnp_batch
        self._check(fn(self._h, _ptr(obj), _ptr(img), _ptr(npts), B, N, _ptr(K), _ptr(d),
                       d.shape[1], per_frame, _ptr(rvec), _ptr(tvec), _ptr(ok), _ptr(mae), _ptr(rmse)))
        return rvec, tvec, ok, mae, rmse

    def ippe_square(self, corners, size, K, dist):
        corners = np.ascontiguousarray(corners, dtype=np.float32).reshape(-1, 4, 2)
        B = corners.shape[0]
        size = np.ascontiguousarray(np.broadcast_to(np.asarray(size, dtype=np.float64), (B,)))
        K, d, per_frame = _cam(K, dist, B)
        rvec = np.zeros((B, 3)); tvec = np.zeros((B, 3)); ok = np.zeros(B, np.uint8)
        self._check(self._lib.camsync_ippe_square_batch(self._h, _ptr(corners), _ptr(size), B, _ptr(K), _ptr(d),
                                                        d.shape[1], per_frame, _ptr(rvec), _ptr(tvec), _ptr(ok)))
        return rvec, tvec, ok

    def fuse_moving_markers(self, n_frames, moving_ids, moving_sizes, want_poses=False):
        """camsync_fuse_moving_markers on the frames of the detect_pnp / detect_pnp_frames call just before it: per
        moving marker the mean world corners [M,4,3] over the cameras that located it, and how many did [M]
        (with want_poses also the markers' aruco2cam rvec / tvec [B,M,3] and the located flags [B,M])."""
        ids = np.ascontiguousarray(moving_ids, dtype=np.int32)
        sizes = np.ascontiguousarray(moving_sizes, dtype=np.float64)
        M = len(ids)
        fused = np.zeros((M, 4, 3)); seen = np.zeros(M, np.int32)
        rv = np.zeros((n_frames, M, 3)) if want_poses else None
        tv = np.zeros((n_frames, M, 3)) if want_poses else None
        loc = np.zeros((n_frames, M), np.uint8) if want_poses else None
        self._check(self._lib.camsync_fuse_moving_markers(self._h, int(n_frames), _ptr(ids), _ptr(sizes), M, _ptr(fused), _ptr(seen),
                                                          _ptr(rv) if want_poses else None, _ptr(tv) if want_poses else None,
                                                          _ptr(loc) if want_poses else None))
        return (fused, seen, rv, tv, loc) if want_poses else (fused, seen)

    def detect_pnp(self, frames, anchor_ids, anchor_obj, K, dist, max_markers: int = 64):
        f = self._frames(frames)
        B, H, W, Cn = f.shape
        anchor_ids = np.ascontiguousarray(anchor_ids, dtype=np.int32)
        anchor_obj = np.ascontiguousarray(anchor_obj, dtype=np.float32).reshape(len(anchor_ids), 4, 3)
        K0, K, d, per_frame = (K,) + _cam(K, dist, B)
        ids = np.full((B, max_markers), -1, np.int32)
        corners = np.zeros((B, max_markers, 4, 2), np.float32)
        counts = np.zeros(B, np.int32)
        rvec = np.zeros((B, 3)); tvec = np.zeros((B, 3)); ok = np.zeros(B, np.uint8)
        mae = np.zeros(B); rmse = np.zeros(B)
        self._check(self._lib.camsync_detect_pnp_batch(
            self._h, _ptr(f), 0, B, H, W, Cn, _ptr(anchor_ids), _ptr(anchor_obj), len(anchor_ids), _ptr(K), _ptr(d),
            d.shape[1], per_frame, max_markers, _ptr(ids), _ptr(corners), _ptr(counts), _ptr(rvec), _ptr(tvec),
            _ptr(ok), _ptr(mae), _ptr(rmse)))
        wider = self._wider(max_markers)
        if wider:
            return self.detect_pnp(f, anchor_ids, anchor_obj, K0, dist, wider[0])
        return ids, corners, counts, rvec, tvec, ok, mae, rmse

    def detect_pnp_frames(self, frames, anchor_ids, anchor_obj, K, dist, max_markers: int = 64):
        """camsync_detect_pnp_frames: `frames` is a sequence of B separate [H,W,C] uint8 host arrays of one shape
        (one per camera); K [B,3,3] / dist [B,n] per frame, or one camera for all.  Results as detect_pnp."""
        fs = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        fs = [f[..., None] if f.ndim == 2 else f for f in fs]
        B = len(fs)
        H, W, Cn = fs[0].shape
        if any(f.shape != (H, W, Cn) for f in fs):
            raise ValueError("all frames of one call must have the same shape")
        ptrs = (C.c_void_p * B)(*[f.ctypes.data for f in fs])
        anchor_ids = np.ascontiguousarray(anchor_ids, dtype=np.int32)
        anchor_obj = np.ascontiguousarray(anchor_obj, dtype=np.float32).reshape(len(anchor_ids), 4, 3)
        K0, K, d, per_frame = (K,) + _cam(K, dist, B)
        ids = np.full((B, max_markers), -1, np.int32)
        corners = np.zeros((B, max_markers, 4, 2), np.float32)
        counts = np.zeros(B, np.int32)
        rvec = np.zeros((B, 3)); tvec = np.zeros((B, 3)); ok = np.zeros(B, np.uint8)
        mae = np.zeros(B); rmse = np.zeros(B)
        self._check(self._lib.camsync_detect_pnp_frames(
            self._h, ptrs, 0, B, H, W, Cn, _ptr(anchor_ids), _ptr(anchor_obj), len(anchor_ids), _ptr(K), _ptr(d),
            d.shape[1], per_frame, max_markers, _ptr(ids), _ptr(corners), _ptr(counts), _ptr(rvec), _ptr(tvec),
            _ptr(ok), _ptr(mae), _ptr(rmse)))
        wider = self._wider(max_markers)
        if wider:
            return self.detect_pnp_frames(fs, anchor_ids, anchor_obj, K0, dist, wider[0])
        return ids, corners, counts, rvec, tvec, ok, mae, rmse

    def detect_pnp_resident(self, d_frames_ptr, B, H, W, Cn, anchor_ids, anchor_obj, K, dist, max_markers,
                            d_ids, d_corners, d_counts, d_rvec, d_tvec, d_ok):
        """All d_* are raw device pointers (ints), e.g. torch tensor .data_ptr()."""
        anchor_ids = np.ascontiguousarray(anchor_ids, dtype=np.int32)
        anchor_obj = np.ascontiguousarray(anchor_obj, dtype=np.float32)
        K, d, _ = _cam(K, dist, 1)
        self._check(self._lib.camsync_detect_pnp_resident(
            self._h, C.c_void_p(d_frames_ptr), B, H, W, Cn, _ptr(anchor_ids), _ptr(anchor_obj), len(anchor_ids),
            _ptr(K), _ptr(d), d.shape[1], max_markers, C.c_void_p(d_ids), C.c_void_p(d_corners), C.c_void_p(d_counts),
            C.c_void_p(d_rvec), C.c_void_p(d_tvec), C.c_void_p(d_ok)))

    # -- chessboard front end ------------------------------------------
    def chess_candidates(self, frames, scale=1, max_cand=1024):
        """X-corner candidates per frame, strongest first: list of (xy int32 [n,2], strength int32 [n])."""
        f = self._frames(frames)
        B, H, W, Cn = f.shape
        xy = np.zeros((B, max_cand, 2), np.int32); st = np.zeros((B, max_cand), np.int32); cnt = np.zeros(B, np.int32)
        self._check(self._lib.camsync_chess_candidates(self._h, _ptr(f), 0, B, H, W, Cn, scale, max_cand, _ptr(xy), _ptr(st), _ptr(cnt)))
        return [(xy[b, :cnt[b]].copy(), st[b, :cnt[b]].copy()) for b in range(B)]

    def corner_subpix(self, frames, pts, win, max_iter, eps, npts=None):
        """cv.cornerSubPix for a batch: pts [B,N,2] -> refined [B,N,2] float32."""
        f = self._frames(frames)
        B, H, W, Cn = f.shape
        p = np.ascontiguousarray(pts, dtype=np.float32).reshape(B, -1, 2)
        N = p.shape[1]
        n = None if npts is None else np.ascontiguousarray(npts, dtype=np.int32)
        out = np.zeros_like(p)
        self._check(self._lib.camsync_corner_subpix(self._h, _ptr(f), 0, B, H, W, Cn, _ptr(p), _ptr(n), N, int(win), int(max_iter),
                                                    float(eps), _ptr(out)))
        return out

    # -- frame ingest ----------------------------------------------------
    def ingest_yuyv(self, yuyv, d_yuyv_ptr=None, d_bgr_ptr=None, shape=None):
        """Packed YUYV 4:2:2 frames (what a UVC camera delivers) -> BGR, as cv.VideoCapture.read() converts them.

        yuyv: uint8 host array [B,H,W,2] (or [H,W,2]); or pass d_yuyv_ptr (raw device pointer) with shape=(B,H,W).
        With d_bgr_ptr (raw device pointer to B*H*W*3 bytes) the frames stay on the GPU and None is returned;
        otherwise the BGR frames come back as a host array of the input's batch shape."""
        if d_yuyv_ptr is None:
            src = np.ascontiguousarray(yuyv, dtype=np.uint8)
            one = src.ndim == 3
            f = src[None] if one else src
            if f.ndim != 4 or f.shape[3] != 2:
                raise ValueError("yuyv frames must be [B,H,W,2] or [H,W,2] uint8")
            B, H, W = f.shape[:3]
            in_ptr, in_dev = _ptr(f), 0
        else:
            B, H, W = shape
            one = False
            in_ptr, in_dev = C.c_void_p(d_yuyv_ptr), 1
        if d_bgr_ptr is not None:
            self._check(self._lib.camsync_ingest_yuyv(self._h, in_ptr, in_dev, B, H, W, C.c_void_p(d_bgr_ptr), 1))
            return None
        out = np.empty((B, H, W, 3), np.uint8)
        self._check(self._lib.camsync_ingest_yuyv(self._h, in_ptr, in_dev, B, H, W, _ptr(out), 0))
        return out[0] if one else out

    def decode_jpeg(self, jpegs, d_bgr_ptr=None):
        """Baseline JPEG files / MJPEG frames (a sequence of bytes objects, all of one image size) -> BGR frames as
        cv.imread / cv.VideoCapture.read() decode them.  Returns a host array [B,H,W,3], or with d_bgr_ptr (raw device
        pointer to B*H*W*3 bytes) leaves the frames on the GPU and returns their shape."""
        bufs = [np.frombuffer(bytes(j), np.uint8) for j in jpegs]
        B = len(bufs)
        if not B:
            raise ValueError("no frames")
        H, W, _ = jpeg_info(bufs[0])
        ptrs = (C.c_void_p * B)(*[b.ctypes.data for b in bufs])
        sizes = (C.c_size_t * B)(*[b.size for b in bufs])
        if d_bgr_ptr is not None:
            self._check(self._lib.camsync_decode_jpeg_batch(self._h, ptrs, sizes, B, H, W, C.c_void_p(d_bgr_ptr), 1))
            return (B, H, W, 3)
        out = np.empty((B, H, W, 3), np.uint8)
        self._check(self._lib.camsync_decode_jpeg_batch(self._h, ptrs, sizes, B, H, W, _ptr(out), 0))
        return out

    # -- undistortion --------------------------------------------------
    def undistort(self, frames, K, dist, new_K=None):
        """cv.undistort for a batch [B,H,W,C] (or one frame) of one camera; returns the same shape."""
        src = np.asarray(frames)
        f = self._frames(src)
        B, H, W, Cn = f.shape
        K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
        d = np.zeros(0) if dist is None else np.ascontiguousarray(dist, dtype=np.float64).ravel()
        if d.size > 8:
            if np.any(d[8:] != 0):
                raise ValueError("thin-prism / tilt distortion terms are not supported")
            d = np.ascontiguousarray(d[:8])
        nk = None if new_K is None else np.ascontiguousarray(new_K, dtype=np.float64).reshape(3, 3)
        out = np.empty_like(f)
        self._check(self._lib.camsync_undistort_batch(self._h, _ptr(f), 0, B, H, W, Cn, _ptr(K), _ptr(d), d.size,
                                                      _ptr(nk), _ptr(out)))
        return out.reshape(src.shape)

    # -- calibration ---------------------------------------------------
    def calibrate(self, obj, img, image_size, rational=False, max_iter=30):
        obj = np.ascontiguousarray(obj, dtype=np.float32)
        img = np.ascontiguousarray(img, dtype=np.float32).reshape(obj.shape[0], obj.shape[1], 2)
        V, P = obj.shape[:2]
        W, H = image_size
        K = np.zeros((3, 3)); dist = np.zeros(8 if rational else 5)
        rvecs = np.zeros((V, 3)); tvecs = np.zeros((V, 3))
        rms = C.c_double(0); iters = C.c_int32(0)
        self._check(self._lib.camsync_calibrate(self._h, _ptr(obj), _ptr(img), V, P, W, H, 1 if rational else 0, max_iter,
                                                _ptr(K), _ptr(dist), _ptr(rvecs), _ptr(tvecs), C.byref(rms), C.byref(iters)))
        return rms.value, K, dist, rvecs, tvecs, iters.value

    def calibrate_dist(self, obj, img, n_views_total, image_size, rational=False, max_iter=30, allreduce=None, d_work=None):
        """camsync_calibrate_dist: this rank's views (possibly none) of a calibration over `n_views_total` views.
        `allreduce(d_buf_ptr, n, hip_stream)` must sum n device doubles at d_buf_ptr over the ranks in place (see
        include/camsync.h); None = single rank.  d_work: device pointer of CALIB_WORK_DOUBLES doubles, or None."""
        obj = np.ascontiguousarray(obj, dtype=np.float32)
        V = obj.shape[0]
        P = obj.shape[1]
        img = np.ascontiguousarray(img, dtype=np.float32).reshape(V, P, 2)
        W, H = image_size
        K = np.zeros((3, 3)); dist = np.zeros(8 if rational else 5)
        rvecs = np.zeros((max(V, 1), 3)); tvecs = np.zeros((max(V, 1), 3))
        rms = C.c_double(0); iters = C.c_int32(0)
        failure = []

        def _cb(user, d_buf, n, stream):
            try:
                allreduce(int(d_buf), int(n), int(stream or 0))
                return 0
            except BaseException as e:  # an exception must not unwind through the C frames
                failure.append(e)
                return 1

        cb = ALLREDUCE_FN(_cb) if allreduce is not None else ALLREDUCE_FN()
        rc = self._lib.camsync_calibrate_dist(self._h, _ptr(obj) if V else None, _ptr(img) if V else None, V, P,
                                              int(n_views_total), W, H, 1 if rational else 0, max_iter, cb, None,
                                              C.c_void_p(d_work) if d_work else None, _ptr(K), _ptr(dist), _ptr(rvecs),
                                              _ptr(tvecs), C.byref(rms), C.byref(iters))
        if failure:
            raise failure[0]
        self._check(rc)
        return rms.value, K, dist, rvecs[:V], tvecs[:V], iters.value

    def calib_begin(self, obj, img, rational=False):
        obj = np.ascontiguousarray(obj, dtype=np.float32)
        V = obj.shape[0]
        P = obj.shape[1] if V else (img.shape[1] if hasattr(img, "shape") and len(img.shape) > 1 else 49)
        img = np.ascontiguousarray(img, dtype=np.float32).reshape(V, P, 2)
        self._check(self._lib.camsync_calib_begin(self._h, _ptr(obj) if V else None, _ptr(img) if V else None, V, P,
                                                  1 if rational else 0))
        return self._lib.camsync_calib_n_intrinsics(self._h)

    def calib_focal_acc(self, W, H):
        acc = np.zeros(6)
        self._check(self._lib.camsync_calib_focal_acc(self._h, W, H, _ptr(acc)))
        return acc

    def calib_set_intrinsics(self, intr, init_extrinsics):
        intr = np.ascontiguousarray(intr, dtype=np.float64)
        assert intr.size == 12
        self._check(self._lib.camsync_calib_set_intrinsics(self._h, _ptr(intr), 1 if init_extrinsics else 0))

    def calib_accumulate(self, lam, ni):
        acc = np.zeros(ni * ni + ni + 1)
        self._check(self._lib.camsync_calib_accumulate(self._h, float(lam), _ptr(acc), 0))
        return acc

    def calib_accumulate_device(self, lam, d_acc_ptr):
        self._check(self._lib.camsync_calib_accumulate(self._h, float(lam), C.c_void_p(d_acc_ptr), 1))

    def calib_solve(self, acc, ni):
        acc = np.ascontiguousarray(acc, dtype=np.float64)
        delta = np.zeros(12)
        rc = self._lib.camsync_calib_solve(_ptr(acc), ni, _ptr(delta))
        if rc != 0:
            raise CamsyncError(rc, "intrinsic block is singular")
        return delta

    def calib_step(self, delta):
        delta = np.ascontiguousarray(delta, dtype=np.float64)
        err = C.c_double(0)
        self._check(self._lib.camsync_calib_step(self._h, _ptr(delta), C.byref(err)))
        return err.value

    def calib_accept(self):
        self._check(self._lib.camsync_calib_accept(self._h))

    def calib_get(self, V):
        intr = np.zeros(12); rv = np.zeros((max(V, 1), 3)); tv = np.zeros((max(V, 1), 3))
        self._check(self._lib.camsync_calib_get(self._h, _ptr(intr), _ptr(rv), _ptr(tv)))
        return intr, rv[:V], tv[:V]

    # -- timers --------------------------------------------------------
    def enable_timers(self, on=True):
        self._check(self._lib.camsync_enable_timers(self._h, 1 if on else 0))

    def reset_timers(self):
        self._check(self._lib.camsync_reset_timers(self._h))

    def timers(self):
        ms = np.zeros(N_STAGES); n = np.zeros(N_STAGES, np.int64)
        self._check(self._lib.camsync_get_timers(self._h, _ptr(ms), _ptr(n)))
        return {STAGE_NAMES[i]: (float(ms[i]), int(n[i])) for i in range(N_STAGES)}


def optimal_new_camera_matrix(K, dist, image_size, alpha, new_size=None):
    """cv.getOptimalNewCameraMatrix(K, dist, (w,h), alpha, new_size) -> (new_K (3,3), roi (x, y, w, h)).  Host only."""
    lib = load_library()
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
    d = np.zeros(0) if dist is None else np.ascontiguousarray(dist, dtype=np.float64).ravel()
    if d.size > 8:
        if np.any(d[8:] != 0):
            raise ValueError("thin-prism / tilt distortion terms are not supported")
        d = np.ascontiguousarray(d[:8])
    w, h = int(image_size[0]), int(image_size[1])
    nw, nh = (int(new_size[0]), int(new_size[1])) if new_size is not None else (w, h)
    new_K = np.zeros((3, 3)); roi = np.zeros(4, np.int32)
    rc = lib.camsync_optimal_new_camera_matrix(_ptr(K), _ptr(d), d.size, w, h, float(alpha), nw, nh, _ptr(new_K), _ptr(roi))
    if rc != 0:
        raise CamsyncError(rc, "bad arguments to camsync_optimal_new_camera_matrix")
    return new_K, tuple(int(v) for v in roi)


_default_ctx = {}
_ctx_lock = threading.Lock()


def get_context(h: int, w: int, batch: int = 1, device: int = 0) -> Context:
    """A cached context large enough for (batch, h, w) frames on `device`."""
    h, w = max(h, 32), max(w, 32)
    with _ctx_lock:
        key = device
        ctx = _default_ctx.get(key)
        if ctx is None or ctx.max_h < h or ctx.max_w < w or ctx.max_batch < batch:
            if ctx is not None:
                ctx.close()
            ctx = Context(max(h, ctx.max_h if ctx else 0), max(w, ctx.max_w if ctx else 0),
                          max(batch, ctx.max_batch if ctx else 0), device)
            _default_ctx[key] = ctx
        return ctx
