"""camera_sync.camera (reference: camera_sync/camera.py:13-256): per-camera state.

On the hot path a Camera is the holder of the intrinsics (`mtx`, `dist`, loaded from
calibration/<name>_<w>-<h>_<focus>.npz) and of the stateful world pose
(`world2cam`, low-passed `world_position`).  A camera device is opened through V4L2 (`v4l2.V4L2Capture`, the
stand-in for cv.VideoCapture(deviceId): raw YUYV or MJPG frames, converted on the GPU), or the Camera takes a
`captureStream` object (anything with VideoCapture's `read()`, e.g. `ingest.RawYUYVStream` / `ingest.MJPEGStream`
over a file or a pipe); image undistortion and the LM solve behind cv.calibrateCamera run on the GPU.
"""
import os

import numpy as np

from .. import _hip
from .position import Position
from .referential import Transform
from .tools import getBaseFolder


class Camera:
    deviceId: int
    world_position: Position
    world2cam: Transform
    mtx: np.ndarray
    dist: np.ndarray
    _focus: int
    _resolution: tuple

    # -- calibration -----------------------------------------------------------------------
    @staticmethod
    def calibrate(frames, pointsToFind=(7, 7)):
        """camera.py:23-77: find the chessboard's inner corners in every frame, refine them with
        cornerSubPix((11,11), (-1,-1), (EPS + MAX_ITER, 30, 0.001)), run cv.calibrateCamera on the frames where the
        board was found, print the mean reprojection error, return (mtx, dist).
        Frames are BGR uint8 images of one size; all of them go through the GPU together."""
        from . import chessboard as cb

        criteria = (cb.TERM_CRITERIA_EPS + cb.TERM_CRITERIA_MAX_ITER, 30, 0.001)
        objp = Camera.boardPoints(pointsToFind)
        frames = [np.asarray(f) for f in frames]
        print("Starting calibration, this may take some time...")
        found = cb.findChessboardCornersBatch(frames, pointsToFind)
        keep = [i for i, (ret, _) in enumerate(found) if ret]
        if not keep:
            raise _hip.CamsyncError(_hip.E_ARG, "no chessboard found in any frame (cv.calibrateCamera would fail on empty input)")
        h, w = frames[0].shape[:2]
        ctx = _hip.get_context(h, w, len(keep))
        batch = np.stack([frames[i] if frames[i].ndim == 3 else frames[i][..., None] for i in keep])
        start = np.stack([found[i][1].reshape(-1, 2) for i in keep])
        refined = ctx.corner_subpix(batch, start, 11, criteria[1], criteria[2])
        objpoints = [objp for _ in keep]
        imgpoints = [r.reshape(-1, 1, 2) for r in refined]
        rms, mtx, dist, rvecs, tvecs = Camera.calibrateFromPoints(objpoints, imgpoints, (w, h), returnAll=True)
        mean_error = 0.0
        for obj, img, rv, tv in zip(objpoints, imgpoints, rvecs, tvecs):
            proj = Camera.projectPoints(obj, rv, tv, mtx, dist)
            mean_error += float(np.sqrt(((img.reshape(-1, 2) - proj) ** 2).sum())) / len(proj)  # cv.norm(L2) / N
        print("total error: {}".format(mean_error / len(objpoints)))
        return mtx, dist

    @staticmethod
    def projectPoints(objpoints, rvec, tvec, mtx, dist) -> np.ndarray:
        """cv.projectPoints (camera.py:67-69) on the host, for the printed evaluation only: [N,2] float64."""
        from .referential import rodrigues_to_matrix

        X = np.asarray(objpoints, np.float64).reshape(-1, 3) @ rodrigues_to_matrix(rvec).T + np.asarray(tvec, np.float64).reshape(3)
        x, y = X[:, 0] / X[:, 2], X[:, 1] / X[:, 2]
        k = np.zeros(8)
        d = np.asarray(dist, np.float64).ravel()[:8]
        k[:d.size] = d
        r2 = x * x + y * y
        rad = (1 + k[0] * r2 + k[1] * r2 ** 2 + k[4] * r2 ** 3) / (1 + k[5] * r2 + k[6] * r2 ** 2 + k[7] * r2 ** 3)
        xd = x * rad + 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        yd = y * rad + k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        mtx = np.asarray(mtx, np.float64)
        return np.stack([mtx[0, 0] * xd + mtx[0, 2], mtx[1, 1] * yd + mtx[1, 2]], axis=1)

    @staticmethod
    def boardPoints(pointsToFind=(7, 7)) -> np.ndarray:
        """Object points of the chessboard exactly as camera.py:38-41 builds them."""
        objp = np.zeros((pointsToFind[0] * pointsToFind[1], 3), np.float32)
        objp[:, :2] = np.mgrid[0 : pointsToFind[0], 0 : pointsToFind[1]].T.reshape(-1, 2)
        return objp

    @staticmethod
    def calibrateFromPoints(objpoints, imgpoints, image_size, rational=False, returnAll=False):
        """The cv.calibrateCamera call of camera.py:61-63 (flags = 0, or CALIB_RATIONAL_MODEL when
        `rational`): objpoints [V][P,3] f32, imgpoints [V][P,1,2] or [V][P,2] f32, image_size (w, h).
        Returns (mtx (3,3), dist (1,5|8)) like Camera.calibrate, or (rms, mtx, dist, rvecs, tvecs)
        when returnAll.  All views must have the same number of points."""
        obj = np.asarray(objpoints, dtype=np.float32)
        img = np.asarray(imgpoints, dtype=np.float32)
        if obj.ndim != 3 or obj.shape[2] != 3:
            raise ValueError("objpoints must be [views][points,3] with the same point count in every view")
        img = img.reshape(obj.shape[0], obj.shape[1], 2)
        w, h = int(image_size[0]), int(image_size[1])
        ctx = _hip.get_context(max(h, 8), max(w, 8), 1)
        rms, mtx, dist, rvecs, tvecs, _ = ctx.calibrate(obj, img, (w, h), rational=rational)
        dist = dist.reshape(1, -1)
        if returnAll:
            return rms, mtx, dist, rvecs, tvecs
        return mtx, dist

    # -- construction ----------------------------------------------------------------------
    def __init__(self, name, deviceId, calibrationFolder: str = None, focus=0, resolution=(640, 480), captureStream=None):
        if calibrationFolder is None:
            calibrationFolder = os.path.join(getBaseFolder(), "./calibration")
        if not os.path.exists(calibrationFolder):
            raise Exception("Calibration folder not found. Searched at " + calibrationFolder)
        self.calibrationFolder = calibrationFolder
        self.name = name
        if deviceId != -1 and captureStream is None:
            # camera.py:98-118: open the capture device, autofocus off, focus from 0..100 to 0..255 in steps of 5, frame
            # size.  (V4L2 on Linux; a device that cannot be opened yields no frames, as with cv.VideoCapture.)
            from .v4l2 import CAP_PROP_AUTOFOCUS, CAP_PROP_FOCUS, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FRAME_WIDTH, V4L2Capture
            captureStream = V4L2Capture(deviceId)
            captureStream.set(CAP_PROP_AUTOFOCUS, 0)
            captureStream.set(CAP_PROP_FOCUS, round(max(0, min(100, focus)) * 255 / 100 / 5) * 5)
            captureStream.set(CAP_PROP_FRAME_WIDTH, resolution[0])
            captureStream.set(CAP_PROP_FRAME_HEIGHT, resolution[1])
        self.deviceId = deviceId
        self.captureStream = captureStream
        self.world_position = Position(0, 0, 0)
        self._focus = focus
        self._resolution = resolution
        self.calibrated = False
        if os.path.exists(self.calibrationFilePath):
            print(f"{self.name} : Found calibration file at {self.calibrationFilePath}")
            self._loadCalibration()
        else:
            print(f"{self.name} : Did not found any calibration file, searched at : {self.calibrationFilePath}")

    @property
    def calibrationFileName(self) -> str:
        return f"{self.name}_{self._resolution[0]}-{self._resolution[1]}_{self._focus}.npz"

    @property
    def calibrationFilePath(self) -> str:
        return os.path.join(self.calibrationFolder, self.calibrationFileName)

    def _loadCalibration(self):
        with np.load(self.calibrationFilePath) as file:  # plain arrays, allow_pickle stays False
            self.mtx = file["mtx"]
            self.dist = file["dist"]
        self.calibrated = True
        return self.mtx, self.dist

    def saveCalibration(self):
        """Same file layout as camera.py:191-201 (keys mtx, dist)."""
        np.savez(self.calibrationFilePath, mtx=self.mtx, dist=self.dist)
        print(f"Calibration file saved in {self.calibrationFilePath}")

    def takePic(self):
        """camera.py:150-154: the next frame of the capture stream, or False when it yields none."""
        if self.captureStream is None:
            raise NotImplementedError("this Camera has no captureStream (deviceId -1: frames are passed in)")
        ret, frame = self.captureStream.read()
        if ret:
            return frame
        return ret

    def calibrateCamera(self, pics, pointsToFind=(7, 7)):
        """camera.py:203-222."""
        mtx, dist = Camera.calibrate(pics, pointsToFind)
        self.mtx, self.dist, self.calibrated = mtx, dist, True
        return mtx, dist

    def calibrateCameraFromPoints(self, objpoints, imgpoints, image_size=None, rational=False):
        """calibrateCamera for callers that already hold the chessboard corners."""
        mtx, dist = Camera.calibrateFromPoints(objpoints, imgpoints, image_size or self._resolution, rational)
        self.mtx, self.dist, self.calibrated = mtx, dist, True
        return mtx, dist

    # -- pose state ------------------------------------------------------------------------
    def updateWorldPosition(self, rvec, tvec, noLowPassFilter: bool = True):
        """camera.py:224-233: world2cam <- (rvec, tvec); world_position <- lowpass(-R^T t)."""
        self.world2cam = Transform.fromRodrigues(rvec=rvec, tvec=tvec)
        self.world_position.updatePos(self.world2cam.invert.apply([0.0, 0.0, 0.0]), noFilter=noLowPassFilter)
        return self.world_position, self.world2cam.rot_mat

    def undistort(self, img):
        """camera.py:235-256: cv.getOptimalNewCameraMatrix(mtx, dist, (w,h), 1, (w,h)), cv.undistort with that
        matrix, crop to the valid region.  Both OpenCV calls are libcamsync_hip entry points."""
        img = np.asarray(img)
        h, w = img.shape[:2]
        newcameramtx, roi = _hip.optimal_new_camera_matrix(self.mtx, self.dist, (w, h), 1, (w, h))
        ctx = _hip.get_context(h, w, 1)
        dst = ctx.undistort(img, self.mtx, self.dist, newcameramtx)
        x, y, w, h = roi
        return dst[y : y + h, x : x + w]
