"""ctypes access to the CPU oracle (oracle/_build/libcamsync_oracle.so).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# (CAMSYNC_ORACLE_LIB: another build of the oracle, e.g. the sanitizer build of `make -C oracle asan`)
LIB_PATH = os.environ.get("CAMSYNC_ORACLE_LIB") or os.path.join(ORACLE_DIR, "_build", "libcamsync_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("adaptiveThreshWinSizeMin", C.c_int32), ("adaptiveThreshWinSizeMax", C.c_int32),
        ("adaptiveThreshWinSizeStep", C.c_int32), ("adaptiveThreshConstant", C.c_double),
        ("minMarkerPerimeterRate", C.c_double), ("maxMarkerPerimeterRate", C.c_double),
        ("polygonalApproxAccuracyRate", C.c_double), ("minCornerDistanceRate", C.c_double),
        ("minDistanceToBorder", C.c_int32), ("markerBorderBits", C.c_int32),
        ("minMarkerDistanceRate", C.c_double), ("minGroupDistance", C.c_float),
        ("cornerRefinementMethod", C.c_int32), ("cornerRefinementWinSize", C.c_int32),
        ("relativeCornerRefinmentWinSize", C.c_float), ("cornerRefinementMaxIterations", C.c_int32),
        ("cornerRefinementMinAccuracy", C.c_double), ("perspectiveRemovePixelPerCell", C.c_int32),
        ("perspectiveRemoveIgnoredMarginPerCell", C.c_double), ("maxErroneousBitsInBorderRate", C.c_double),
        ("minOtsuStdDev", C.c_double), ("errorCorrectionRate", C.c_double),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_calibrate.restype = C.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params():
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    return p


def dict_codes():
    from cameracalibration_amd._dict4x4_50 import CODES
    return np.ascontiguousarray(CODES)


def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_p(bgr), h, w, _p(out))
    return out


def adaptive_threshold(gray, win, c=7.0):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_adaptive_threshold(_p(gray), h, w, int(win), C.c_double(c), _p(out))
    return out


def find_contours(binimg, max_contours=1 << 20, max_pts=1 << 23):
    binimg = np.ascontiguousarray(binimg, np.uint8)
    h, w = binimg.shape
    starts = np.zeros(max_contours, np.int32); lens = np.zeros(max_contours, np.int32)
    pts = np.zeros((max_pts, 2), np.int32); total = C.c_int32(0)
    n = lib().orc_find_contours(_p(binimg), h, w, _p(starts), _p(lens), max_contours, _p(pts), max_pts, C.byref(total))
    assert n <= max_contours and total.value <= max_pts
    return [pts[starts[i]:starts[i] + lens[i]] for i in range(n)]


def quad_candidates(binimg, params=None, cap=4096):
    binimg = np.ascontiguousarray(binimg, np.uint8)
    h, w = binimg.shape
    p = params or default_params()
    quads = np.zeros((cap, 4, 2), np.float32); key = np.zeros(cap, np.int32); clen = np.zeros(cap, np.int32)
    n = lib().orc_quad_candidates(_p(binimg), h, w, C.byref(p), _p(quads), _p(key), _p(clen), cap)
    assert n <= cap
    return quads[:n], key[:n], clen[:n]


def detect(img, params=None, max_markers=256, max_rejected=1024):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    p = params or default_params()
    codes = dict_codes()
    ids = np.zeros(max_markers, np.int32); cor = np.zeros((max_markers, 4, 2), np.float32)
    rej = np.zeros((max_rejected, 4, 2), np.float32); nrej = C.c_int32(0)
    n = lib().orc_detect(_p(img), h, w, ch, C.byref(p), _p(codes), len(codes), _p(ids), _p(cor), max_markers,
                         _p(rej), max_rejected, C.byref(nrej))
    assert n >= 0
    if n > max_markers or nrej.value > max_rejected:  # (the oracle has no limits: again with room for everything)
        return detect(img, params, max(n, max_markers), max(nrej.value, max_rejected))
    return ids[:n].copy(), cor[:n].copy(), rej[:min(nrej.value, max_rejected)].copy()


def identify(gray, corners, params=None):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    c = np.ascontiguousarray(corners, np.float32)
    p = params or default_params()
    codes = dict_codes()
    idv = C.c_int32(-1); rot = C.c_int32(0); bits = np.zeros(36, np.uint8)
    ok = lib().orc_identify(_p(gray), h, w, _p(c), C.byref(p), _p(codes), len(codes), C.byref(idv), C.byref(rot), _p(bits))
    return bool(ok), idv.value, rot.value, bits.reshape(6, 6)


def corner_subpix(gray, pts, win, max_iter=100, eps=0.01):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    p = np.ascontiguousarray(pts, np.float32).copy()
    lib().orc_corner_subpix(_p(gray), h, w, _p(p), p.size // 2, int(win), int(max_iter), C.c_double(eps))
    return p


def rodrigues_vec2mat(r):
    r = np.ascontiguousarray(r, np.float64).ravel()
    R = np.zeros((3, 3))
    lib().orc_rodrigues_vec2mat(_p(r), _p(R), None)
    return R


def rodrigues_mat2vec(R):
    R = np.ascontiguousarray(R, np.float64)
    r = np.zeros(3)
    lib().orc_rodrigues_mat2vec(_p(R), _p(r))
    return r


def project_points(obj, rvec, tvec, K, dist):
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3)
    rvec = np.ascontiguousarray(rvec, np.float64).ravel(); tvec = np.ascontiguousarray(tvec, np.float64).ravel()
    K = np.ascontiguousarray(K, np.float64); d = np.ascontiguousarray(dist, np.float64).ravel()
    out = np.zeros((len(obj), 2))
    lib().orc_project_points(_p(obj), len(obj), _p(rvec), _p(tvec), _p(K), _p(d), len(d), _p(out))
    return out


def solve_pnp(obj, img, K, dist, refine=True):
    """obj/img are taken as float32 values (like the reference passes them) and widened."""
    obj = np.ascontiguousarray(np.asarray(obj, np.float32).astype(np.float64)).reshape(-1, 3)
    img = np.ascontiguousarray(np.asarray(img, np.float32).astype(np.float64)).reshape(-1, 2)
    K = np.ascontiguousarray(K, np.float64); d = np.ascontiguousarray(dist, np.float64).ravel()
    rv = np.zeros(3); tv = np.zeros(3)
    ok = lib().orc_solve_pnp(_p(obj), _p(img), len(obj), _p(K), _p(d), len(d), 1 if refine else 0, _p(rv), _p(tv))
    return bool(ok), rv, tv


def ippe_square(corners, size, K, dist):
    c = np.ascontiguousarray(np.asarray(corners, np.float32).astype(np.float64)).reshape(4, 2)
    K = np.ascontiguousarray(K, np.float64); d = np.ascontiguousarray(dist, np.float64).ravel()
    rv = np.zeros(3); tv = np.zeros(3)
    ok = lib().orc_ippe_square(_p(c), C.c_double(size), _p(K), _p(d), len(d), _p(rv), _p(tv))
    return bool(ok), rv, tv


def calibrate(obj, img, image_size, rational=False, max_iter=30):
    obj = np.ascontiguousarray(np.asarray(obj, np.float32).astype(np.float64))
    V, P = obj.shape[:2]
    img = np.ascontiguousarray(np.asarray(img, np.float32).astype(np.float64)).reshape(V, P, 2)
    W, H = image_size
    K = np.zeros((3, 3)); dist = np.zeros(8 if rational else 5)
    rv = np.zeros((V, 3)); tv = np.zeros((V, 3)); iters = C.c_int32(0)
    rms = lib().orc_calibrate(_p(obj), _p(img), V, P, W, H, 1 if rational else 0, max_iter, _p(K), _p(dist), _p(rv), _p(tv),
                              C.byref(iters))
    return rms, K, dist, rv, tv, iters.value


class CalibShare:
    """A share of views held by the oracle's orc_calib_* pieces; same interface as
    cameracalibration_amd.calibrate_dist.HipCalibWorker, so the distributed driver can be run on CPU."""

    def __init__(self, obj, img, rational=False):
        L = lib()
        L.orc_calib_begin.restype = C.c_void_p
        L.orc_calib_step.restype = C.c_double
        for f in (L.orc_calib_end, L.orc_calib_focal_acc, L.orc_calib_set_intrinsics, L.orc_calib_accumulate,
                  L.orc_calib_step, L.orc_calib_accept, L.orc_calib_get):
            f.argtypes = None
        obj = np.ascontiguousarray(np.asarray(obj, np.float32).astype(np.float64))
        self.V, self.points_per_view = obj.shape[0], obj.shape[1]
        img = np.ascontiguousarray(np.asarray(img, np.float32).astype(np.float64)).reshape(self.V, self.points_per_view, 2)
        self.ni = 12 if rational else 9
        self._h = C.c_void_p(L.orc_calib_begin(_p(obj), _p(img), self.V, self.points_per_view, 1 if rational else 0))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_calib_end(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def focal_acc(self, W, H):
        a = np.zeros(6)
        lib().orc_calib_focal_acc(self._h, int(W), int(H), _p(a))
        return a

    def set_intrinsics(self, intr, init_extrinsics):
        intr = np.ascontiguousarray(intr, np.float64)
        lib().orc_calib_set_intrinsics(self._h, _p(intr), 1 if init_extrinsics else 0)

    def accumulate(self, lam):
        a = np.zeros(self.ni * self.ni + self.ni + 1)
        lib().orc_calib_accumulate(self._h, C.c_double(lam), _p(a))
        return a

    def step(self, delta):
        d = np.ascontiguousarray(delta, np.float64)
        return float(lib().orc_calib_step(self._h, _p(d)))

    def accept(self):
        lib().orc_calib_accept(self._h)

    def get(self):
        intr = np.zeros(12); rv = np.zeros((max(self.V, 1), 3)); tv = np.zeros((max(self.V, 1), 3))
        lib().orc_calib_get(self._h, _p(intr), _p(rv), _p(tv))
        return intr, rv[:self.V], tv[:self.V]


def optimal_new_camera_matrix(K, dist, image_size, alpha, new_size=None):
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3); d = np.ascontiguousarray(dist, np.float64).ravel()[:8].copy()
    w, h = image_size
    nw, nh = new_size if new_size is not None else (w, h)
    new_K = np.zeros((3, 3)); roi = np.zeros(4, np.int32)
    lib().orc_optimal_new_camera_matrix(_p(K), _p(d), len(d), int(w), int(h), C.c_double(alpha), int(nw), int(nh), _p(new_K), _p(roi))
    return new_K, tuple(int(v) for v in roi)


def undistort(img, K, dist, new_K=None):
    src = np.ascontiguousarray(img, np.uint8)
    h, w = src.shape[:2]
    c = 1 if src.ndim == 2 else src.shape[2]
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3); d = np.ascontiguousarray(dist, np.float64).ravel()[:8].copy()
    nk = None if new_K is None else np.ascontiguousarray(new_K, np.float64).reshape(3, 3)
    out = np.empty_like(src)
    rc = lib().orc_undistort(_p(src), h, w, c, _p(K), _p(d), len(d), _p(nk) if nk is not None else None, _p(out))
    assert rc == 0
    return out


def chess_candidates(gray, scale=1, max_cand=4096):
    g = np.ascontiguousarray(gray, np.uint8)
    h, w = g.shape
    xy = np.zeros((max_cand, 2), np.int32); st = np.zeros(max_cand, np.int32)
    n = lib().orc_chess_candidates(_p(g), h, w, int(scale), max_cand, _p(xy), _p(st))
    assert n <= max_cand
    return xy[:n], st[:n]


def yuyv_to_bgr(yuyv):
    """Packed YUYV [H,W,2] uint8 -> BGR [H,W,3] (oracle/ingest.cpp)."""
    src = np.ascontiguousarray(yuyv, dtype=np.uint8)
    h, w = src.shape[:2]
    out = np.empty((h, w, 3), np.uint8)
    rc = lib().orc_yuyv_to_bgr(_p(src), h, w, _p(out))
    if rc:
        raise ValueError("orc_yuyv_to_bgr: bad arguments")
    return out


def jpeg_decode(data):
    """One JPEG file / MJPEG frame (bytes) -> BGR [H,W,3] as libjpeg-turbo decodes it (oracle/jpeg.cpp)."""
    buf = np.frombuffer(bytes(data), np.uint8)
    h = C.c_int32(0); w = C.c_int32(0); nc = C.c_int32(0)
    rc = lib().orc_jpeg_info(_p(buf), C.c_size_t(len(buf)), C.byref(h), C.byref(w), C.byref(nc))
    if rc:
        raise ValueError("orc_jpeg_info: %d" % rc)
    out = np.empty((h.value, w.value, 3), np.uint8)
    rc = lib().orc_jpeg_decode(_p(buf), C.c_size_t(len(buf)), _p(out), C.c_size_t(out.size))
    if rc:
        raise ValueError("orc_jpeg_decode: %d" % rc)
    return out
