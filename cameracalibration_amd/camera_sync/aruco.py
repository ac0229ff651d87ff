"""camera_sync.aruco (reference: camera_sync/aruco.py): ArUco detection and pose on the GPU.

Function names, arguments, return shapes and error behaviour follow the reference module;
the OpenCV calls inside them (cvtColor + ArucoDetector.detectMarkers, solvePnP +
solvePnPRefineLM + projectPoints, solvePnP IPPE_SQUARE) are entry points of
libcamsync_hip.so.  The *Batch functions at the end are additions: same results per
frame, one launch sequence per batch.
"""
from itertools import combinations

import numpy as np

from .. import _dict4x4_50 as _dict
from .. import _hip
from .camera import Camera
from .position import Point, Position
from .referential import Transform

DICT_4X4_50 = 0  # value of cv.aruco.DICT_4X4_50, the only dictionary the reference uses (aruco.py:233)


class Aruco:
    """A marker of known size, optionally located in the world (aruco.py:12-79).
    Corner order: top-left, top-right, bottom-right, bottom-left; y decreases downwards on the sheet."""

    corners: list = None

    @staticmethod
    def getCornersFromTopLeft(topLeft, size):
        tl = np.asarray(topLeft)
        out = np.stack([tl, tl, tl, tl]).copy()
        out[1, 0] += size
        out[2, 0] += size
        out[2, 1] -= size
        out[3, 1] -= size
        return out

    @staticmethod
    def getCornersFromPose(pose: Transform, size: float):
        h = size / 2
        return [pose.apply(c) for c in ([-h, -h, 0.0], [h, -h, 0.0], [h, h, 0.0], [-h, h, 0.0])]

    @property
    def isLocated(self) -> bool:
        return self.corners is not None

    def __init__(self, id: int, size: float, topLeft: Position = None):
        self.id = id
        self.size = size
        if topLeft is not None:
            self.corners = [Position(*c) for c in Aruco.getCornersFromTopLeft(topLeft.coords, self.size)]

    def getCornersAsList(self):
        return np.array([self.corners[i].coords for i in range(4)], dtype=np.float32)

    def updatePose(self, aruco2world: Transform):
        self.corners = Aruco.getCornersFromPose(aruco2world, self.size)

    def getCenter(self) -> Position:
        c0 = self.corners[0]
        return Position(c0.x + self.size / 2, c0.y - self.size / 2, c0.z)


def estimate_pose_single_markers(corners, marker_size, mtx, distortion):
    """Pose of one marker from its 4 image corners, first IPPE_SQUARE solution (aruco.py:82-119).
    The marker frame has its origin at the centre, corners (-s/2, s/2), (s/2, s/2), (s/2, -s/2), (-s/2, -s/2).
    Returns rvec, tvec as (3,1) float64."""
    c = np.asarray(corners).astype(np.float32).reshape(1, 4, 2)
    ctx = _hip.get_context(8, 8, 1)
    rvec, tvec, _ = ctx.ippe_square(c, float(marker_size), mtx, distortion)
    return rvec.reshape(3, 1), tvec.reshape(3, 1)


def getArucosFromPaper(pap_v: int = 1) -> dict:
    """Marker sheets of the project (aruco.py:122-177); sizes and offsets in millimetres."""
    dic = {}
    if pap_v == 1:
        for i in (1, 2, 3, 0):
            dic[i] = Aruco(i, size=27, topLeft=Position(0, 0, 0))
        for k, (x, y) in enumerate(((0, 0), (110, 0), (0, -95), (110, -95), (0, -190), (110, -190))):
            dic[10 + k] = Aruco(10 + k, size=80, topLeft=Position(x, y, 0))
    elif pap_v == 2:
        for k in range(25):  # 5 x 5 grid, ids 20..44, pitch 35
            i, j = divmod(k, 5)
            dic[20 + k] = Aruco(20 + k, size=30, topLeft=Position(j * 35, i * -35, 0))
    elif pap_v == 4:
        id = 32
        for x_start, y_start in ((0, 0), (150.14, 0), (0, -220.14), (150.14, -220.14)):  # one 2 x 2 group per corner
            for i in range(2):
                for j in range(2):
                    dic[id] = Aruco(id, size=35, topLeft=Position(x=x_start + j * 40, y=y_start + i * -40, z=0))
                    id += 1
    elif pap_v == 3:
        for k, (x, y) in enumerate(((0, 0), (150.4, 0), (0, -220), (150.4, -220))):
            dic[44 + k] = Aruco(44 + k, size=77, topLeft=Position(x, y, 0))
    return dic


def generate_aruco_marker(marker_id, dictionary_id=DICT_4X4_50, marker_size=200, save_path=None):
    """Marker image as cv.aruco.generateImageMarker draws it (aruco.py:180-210): a (4+2)^2 cell image with
    a one-cell black border, scaled to marker_size with nearest-neighbour sampling; saved as PNG."""
    if dictionary_id != DICT_4X4_50:
        raise ValueError("only DICT_4X4_50 is available")
    if not 0 <= marker_id < len(_dict.CODES):
        raise ValueError("marker_id outside the dictionary")
    if save_path is None:
        save_path = f"./aruco/aruco_{marker_id}.png"
    n = _dict.MARKER_SIZE
    code = int(_dict.CODES[marker_id])
    cells = np.zeros((n + 2, n + 2), np.uint8)
    for k in range(n * n):
        if (code >> (n * n - 1 - k)) & 1:
            cells[1 + k // n, 1 + k % n] = 255
    src = np.minimum((np.arange(marker_size) * (n + 2)) // marker_size, n + 1)
    marker_image = cells[np.ix_(src, src)]
    _imwrite(save_path, marker_image)
    print(f"ArUco marker (ID: {marker_id}) saved as {save_path}")
    return marker_image


def _imwrite(path, img):
    from PIL import Image

    Image.fromarray(img).save(path)


def _as_frame(img) -> np.ndarray:
    f = np.asarray(img)
    if f.dtype != np.uint8 or f.ndim not in (2, 3) or (f.ndim == 3 and f.shape[2] not in (1, 3)):
        raise ValueError("detectAruco expects a uint8 image of shape (H, W, 3) in BGR order or (H, W)")
    return f


def detectAruco(img, dictionary_id=DICT_4X4_50, debug=False):
    """Detect DICT_4X4_50 markers with sub-pixel corners (aruco.py:232-274).
    Returns ({id: corners float32 (1,4,2)}, rejected) where rejected is a tuple of (1,4,2) arrays."""
    if dictionary_id != DICT_4X4_50:
        raise ValueError("only DICT_4X4_50 is available")
    f = _as_frame(img)
    ctx = _hip.get_context(f.shape[0], f.shape[1], 1)
    ids, corners, counts, rej, rej_counts = ctx.detect(f)
    rejected = tuple(rej[0, i].reshape(1, 4, 2).copy() for i in range(int(rej_counts[0])))
    detected = {}
    if counts[0] == 0:
        if debug:
            print("No markers found")
        return detected, rejected
    for i in range(int(counts[0])):
        detected[np.int32(ids[0, i])] = corners[0, i].reshape(1, 4, 2).copy()
    if debug:
        print("Detected aruco markers : ", detected.keys())
    return detected, rejected


def locateAruco(aruco: Aruco, img_positions, camera: Camera, metrics: bool = False):
    """World pose of a moving marker seen by a located camera (aruco.py:277-292)."""
    metrics = {}
    assert len(img_positions[0]) == 4
    rvec, tvec = estimate_pose_single_markers(img_positions, aruco.size, camera.mtx, camera.dist)
    aruco2cam = Transform.fromRodrigues(rvec=rvec, tvec=tvec)
    return aruco2cam.combine(camera.world2cam.invert), metrics


def _anchor_correspondences(fixedArucos, corners_position):
    obj, img = [], []
    for aruco in fixedArucos:
        seen = corners_position.get(aruco.id)
        if seen is None:
            continue
        obj.extend(aruco.getCornersAsList())
        img.extend(seen[0])
    return img, obj


def processAruco(fixedArucos, movingArucos, camera: Camera, img, accept_none=False, directUpdate=True, metrics=False):
    """One frame of one camera (aruco.py:295-349): detect, solve the camera pose from the fixed markers,
    update the camera (low-pass), locate the moving markers.
    Returns (rvec, tvec, {id: aruco2world Transform}, metrics) or False when accept_none and too few anchors."""
    metrics_collected = {}
    corners_position, _ = detectAruco(img, debug=False)
    image_points, obj_points = _anchor_correspondences(fixedArucos, corners_position)
    if len(image_points) < 4 or len(obj_points) < 4:
        if accept_none:
            return False
        raise Exception("Could not find any ArUco markers in the image")
    rvec, tvec, met = PnP(np.array(image_points, dtype=np.float32), np.array(obj_points, dtype=np.float32),
                          camera.mtx, camera.dist, metrics)
    if metrics:
        metrics_collected["PnP"] = met
    camera.updateWorldPosition(rvec, tvec, noLowPassFilter=False)
    arucosPosition = {}
    for a in movingArucos:
        seen = corners_position.get(a.id)
        if seen is None:
            continue
        aruco2world, _ = locateAruco(a, seen, camera)
        if directUpdate:
            a.updatePose(aruco2world)
        arucosPosition[a.id] = aruco2world
    return rvec, tvec, arucosPosition, metrics_collected


def PnP(image_points, object_points, mtx, dist, getMetrics=False):
    """Camera pose from 2D-3D correspondences: solvePnP (iterative) then solvePnPRefineLM (aruco.py:352-404).
    Returns rvec (3,1), tvec (3,1), metrics ({"MAE", "RMSE"} in pixels when getMetrics)."""
    image_points = np.array(image_points, dtype=np.float32)
    obj = np.asarray(object_points, dtype=np.float32)
    try:
        if obj.ndim != 2 or obj.shape[1] != 3 or image_points.reshape(-1, 2).shape[0] != obj.shape[0] or obj.shape[0] < 4:
            raise _hip.CamsyncError(-1, "solvePnP needs at least 4 object points [N,3] and as many image points [N,2]")
        n = obj.shape[0]
        ctx = _hip.get_context(8, 8, 1)
        rvec, tvec, ok, mae, rmse = ctx.pnp(obj[None], image_points.reshape(1, n, 2), [n], mtx, dist)
    except _hip.CamsyncError as e:
        print("An error occured while solving PnP")
        print(f"Object point n°: {len(object_points)}, Image point n°: {len(image_points)}")
        raise e
    if ok[0] == 2:
        raise _hip.CamsyncError(-1, "solvePnP needs at least 6 points when they are not coplanar (cv.error in the reference)")
    if ok[0] != 1:
        raise Exception("Could not solve PnP")
    metrics = {}
    if getMetrics:
        metrics["MAE"] = float(mae[0])
        metrics["RMSE"] = float(rmse[0])
    return rvec.reshape(3, 1), tvec.reshape(3, 1), metrics


def _detect_and_locate_cameras(fixedArucos, cameras, frames, after_batch=None):
    """Detection + anchor PnP for every (camera, frame) pair, one launch sequence per frame size
    (camsync_detect_pnp_frames, a camera matrix per frame).  Returns det: det[i] = (detected dict, rvec, tvec, ok, mae,
    rmse) per camera.  after_batch(ctx, camera indices) is called after each batch, while the context still holds that
    batch's detections and poses on the device (the next batch may run on the same context)."""
    n = len(cameras)
    fs = [_as_frame(f) for f in frames]
    anchor_ids = np.array([a.id for a in fixedArucos], dtype=np.int32)
    anchor_obj = (np.stack([a.getCornersAsList() for a in fixedArucos]).astype(np.float32)
                  if fixedArucos else np.zeros((0, 4, 3), np.float32))
    det = [None] * n
    by_shape = {}
    for i, f in enumerate(fs):
        by_shape.setdefault(f.shape, []).append(i)
    for shape, idx in by_shape.items():
        ks = np.stack([np.asarray(cameras[i].mtx, np.float64).reshape(3, 3) for i in idx])
        nd = max(np.asarray(cameras[i].dist).size for i in idx)
        ds = np.zeros((len(idx), nd))
        for k, i in enumerate(idx):
            d = np.asarray(cameras[i].dist, np.float64).ravel()
            ds[k, :d.size] = d
        ctx = _hip.get_context(shape[0], shape[1], len(idx))
        ids, corners, counts, rvec, tvec, ok, mae, rmse = ctx.detect_pnp_frames([fs[i] for i in idx], anchor_ids, anchor_obj, ks, ds)
        for k, i in enumerate(idx):
            found = {np.int32(ids[k, m]): corners[k, m].reshape(1, 4, 2).copy() for m in range(int(counts[k]))}
            det[i] = (found, rvec[k], tvec[k], int(ok[k]), float(mae[k]), float(rmse[k]))
        if after_batch is not None:
            after_batch(ctx, idx)
    return det


def _update_cameras(fixedArucos, cameras, det):
    """The camera half of processAruco(accept_none=True) for every camera in order (aruco.py:317-336): the errors the
    per-camera loop would raise, then Camera.updateWorldPosition.  Returns the indices of the cameras located."""
    located = []
    for i, cam in enumerate(cameras):
        found, rvec, tvec, ok, _, _ = det[i]
        if 4 * sum(1 for a in fixedArucos if found.get(a.id) is not None) < 4:
            continue
        if ok == 2:
            raise _hip.CamsyncError(-1, "solvePnP needs at least 6 points when they are not coplanar (cv.error in the reference)")
        if ok != 1:
            raise Exception("Could not solve PnP")
        cam.updateWorldPosition(rvec.reshape(3, 1), tvec.reshape(3, 1), noLowPassFilter=False)
        located.append(i)
    return located


def _process_cameras_batched(fixedArucos, movingArucos, cameras, frames, metrics):
    """processAruco(..., accept_none=True, directUpdate=False) for every (camera, frame) pair, with the GPU work of
    all cameras in two launch sequences instead of two per camera: one detection + anchor PnP over all frames
    (camsync_detect_pnp_frames, a camera matrix per frame) and one IPPE solve over all moving markers seen.
    Same kernels and same inputs as the per-camera calls, so the results are identical; the cameras are updated in
    order, and an error is raised where the per-camera loop would have raised it.
    Returns one entry per camera: False, or (rvec, tvec, {id: aruco2world}, metrics_collected)."""
    n = len(cameras)
    fixedArucos = list(fixedArucos)
    movingArucos = list(movingArucos)
    det = _detect_and_locate_cameras(fixedArucos, cameras, frames)
    # the moving markers every located camera sees, in the order processAruco visits them
    pairs = []
    for i in range(n):
        found, _, _, ok, _, _ = det[i]
        if sum(1 for a in fixedArucos if found.get(a.id) is not None) >= 1 and ok == 1:
            pairs.extend((i, a) for a in movingArucos if found.get(a.id) is not None)
    poses = {}
    if pairs:
        ctx = _hip.get_context(8, 8, 1)
        pc = np.stack([det[i][0][a.id].reshape(4, 2) for i, a in pairs]).astype(np.float32)
        pk = np.stack([np.asarray(cameras[i].mtx, np.float64).reshape(3, 3) for i, _ in pairs])
        nd = max(np.asarray(cameras[i].dist).size for i, _ in pairs)
        pd = np.zeros((len(pairs), nd))
        for k, (i, _) in enumerate(pairs):
            d = np.asarray(cameras[i].dist, np.float64).ravel()
            pd[k, :d.size] = d
        rv, tv, _ = ctx.ippe_square(pc, np.array([float(a.size) for _, a in pairs]), pk, pd)
        for k, (i, a) in enumerate(pairs):
            poses[(i, a.id)] = (rv[k].reshape(3, 1), tv[k].reshape(3, 1))
    out = []
    for i, cam in enumerate(cameras):
        found, rvec, tvec, ok, mae, rmse = det[i]
        if 4 * sum(1 for a in fixedArucos if found.get(a.id) is not None) < 4:
            out.append(False)
            continue
        if ok == 2:
            raise _hip.CamsyncError(-1, "solvePnP needs at least 6 points when they are not coplanar (cv.error in the reference)")
        if ok != 1:
            raise Exception("Could not solve PnP")
        rvec, tvec = rvec.reshape(3, 1), tvec.reshape(3, 1)
        collected = {"PnP": {"MAE": mae, "RMSE": rmse}} if metrics else {}
        cam.updateWorldPosition(rvec, tvec, noLowPassFilter=False)
        located = {}
        for a in movingArucos:
            if (i, a.id) not in poses:
                continue
            r, t = poses[(i, a.id)]
            located[a.id] = Transform.fromRodrigues(rvec=r, tvec=t).combine(cam.world2cam.invert)
        out.append((rvec, tvec, located, collected))
    return out


def _fuse_on_device(fixedArucos, movingArucos: dict, cameras, frames):
    """processArucoFromMultipleCameras with the per-marker work on the device as well: after the batched detection +
    anchor PnP, camsync_fuse_moving_markers locates every moving marker in every located camera, takes it to the world
    frame and averages its corners over the cameras; one read-back of [markers,4,3] per frame size.  The host keeps
    what is stateful: the cameras' poses and filtered positions, and the low-pass of Aruco.corners.
    Returns {id: mean world corners [4,3]} for the markers at least one camera located."""
    arucos = list(movingArucos.values())
    ids = [a.id for a in arucos]
    sizes = [float(a.size) for a in arucos]
    parts = []  # per batch: (mean corners [M,4,3], cameras that located the marker [M])

    def fuse(ctx, idx):
        if arucos:
            parts.append(ctx.fuse_moving_markers(len(idx), ids, sizes))

    det = _detect_and_locate_cameras(fixedArucos, cameras, frames, after_batch=fuse)
    _update_cameras(fixedArucos, cameras, det)
    out = {}
    for k, a in enumerate(arucos):
        n = sum(int(seen[k]) for _, seen in parts)
        if n == 0:
            continue
        if len(parts) == 1:  # one frame size: the device's mean is the result as it is
            out[a.id] = parts[0][0][k]
        else:
            out[a.id] = sum(fused[k] * int(seen[k]) for fused, seen in parts) / n
    return out


def processArucoFromMultipleCameras(fixedArucos, movingArucos: dict, cameras, frame, getMetrics=False):
    """All cameras look at the same scene (aruco.py:407-470): every camera that sees enough anchors locates
    itself and the moving markers it sees; each moving marker's world corners are averaged over the cameras
    and low-passed into Aruco.corners.  The per-camera loop of the reference (aruco.py:417) is one batch here, and so
    is the per-marker work (_fuse_on_device); the results are those of the loop (to rounding: 1e-12 relative)."""
    assert len(cameras) == len(frame)
    fixedArucos = list(fixedArucos)
    batchable = (0 < len(fixedArucos) <= 64 and len(cameras) > 0
                 and all(getattr(c, "mtx", None) is not None and getattr(c, "dist", None) is not None for c in cameras))
    if batchable and not getMetrics:
        for key, mean in _fuse_on_device(fixedArucos, movingArucos, cameras, frame).items():
            for i, newPos in enumerate(mean):
                movingArucos[key].corners[i].updatePos(newPos)
        return
    if batchable:  # (the pairwise disagreement metric wants every camera's corners: the IPPE poses come back to the host)
        results = _process_cameras_batched(fixedArucos, movingArucos.values(), cameras, frame, getMetrics)
    else:  # nothing to batch over, or more anchors than the fused kernel gathers: the reference's own loop
        results = [processAruco(fixedArucos, movingArucos.values(), cam, f, accept_none=True, directUpdate=False,
                                metrics=getMetrics) for cam, f in zip(cameras, frame)]
    seen_corners = {}  # id -> one [4][3] list per camera that saw the marker
    for res in results:
        if not res:
            continue
        _, _, located, _ = res
        for key, aruco2world in located.items():
            size = movingArucos[key].size
            local = Aruco.getCornersFromTopLeft(np.array([-size / 2, size / 2, 0]), size)
            seen_corners.setdefault(key, []).append([aruco2world.apply(c) for c in local])
    for a in movingArucos.values():
        if a.id not in seen_corners:
            continue
        if getMetrics:
            _pairwise_rmse(np.array(seen_corners[a.id]))
        for i, newPos in enumerate(np.mean(seen_corners[a.id], axis=0)):
            a.corners[i].updatePos(newPos)


def _pairwise_rmse(arr):
    """Disagreement between cameras about one marker (aruco.py:445-466): RMSE over all unordered camera pairs,
    per axis and averaged.  The reference computes it and drops it; it is returned here for callers that want it."""
    pairs = list(combinations(range(len(arr)), 2))
    if not pairs:
        return np.zeros(3), 0.0
    diffs = np.array([arr[i] - arr[j] for i, j in pairs])
    err = np.sum(np.square(diffs) / len(arr), axis=0)
    per_axis = np.sqrt(np.sum(err, axis=0) / 4)
    return per_axis, float(np.sum(per_axis) / 3)


# ---- additions: batched entry points (not in the reference) -------------------------------------

def detectArucoBatch(frames, dictionary_id=DICT_4X4_50):
    """detectAruco over frames [B,H,W,3] (or [B,H,W]) in one launch sequence.
    Returns a list of (detected, rejected) pairs, one per frame, identical to per-frame detectAruco."""
    if dictionary_id != DICT_4X4_50:
        raise ValueError("only DICT_4X4_50 is available")
    f = np.asarray(frames)
    if f.dtype != np.uint8 or f.ndim not in (3, 4):
        raise ValueError("frames must be uint8 [B,H,W,3] or [B,H,W]")
    if f.ndim == 3:
        f = f[..., None]
    B, H, W = f.shape[:3]
    ctx = _hip.get_context(H, W, B)
    ids, corners, counts, rej, rej_counts = ctx.detect(f)
    out = []
    for b in range(B):
        det = {np.int32(ids[b, i]): corners[b, i].reshape(1, 4, 2).copy() for i in range(int(counts[b]))}
        out.append((det, tuple(rej[b, i].reshape(1, 4, 2).copy() for i in range(int(rej_counts[b])))))
    return out


def processArucoBatch(fixedArucos, frames, mtx, dist, getMetrics=False):
    """Camera pose for every frame of a batch taken with one camera model: detection, anchor gathering and
    PnP run back to back on the GPU without visiting the host in between.
    Returns (rvecs [B,3], tvecs [B,3], ok [B] bool, detections list of {id: corners}, metrics dict or None).
    ok[b] is False where fewer than 4 anchor corners were seen (processAruco would raise / return False)."""
    f = np.asarray(frames)
    if f.ndim == 3:
        f = f[..., None]
    B, H, W = f.shape[:3]
    anchor_ids = np.array([a.id for a in fixedArucos], dtype=np.int32)
    anchor_obj = np.stack([a.getCornersAsList() for a in fixedArucos]).astype(np.float32)
    ctx = _hip.get_context(H, W, B)
    ids, corners, counts, rvec, tvec, ok, mae, rmse = ctx.detect_pnp(f, anchor_ids, anchor_obj, mtx, dist)
    dets = [{np.int32(ids[b, i]): corners[b, i].reshape(1, 4, 2).copy() for i in range(int(counts[b]))} for b in range(B)]
    metrics = {"MAE": mae, "RMSE": rmse} if getMetrics else None
    return rvec, tvec, ok == 1, dets, metrics
