"""camera_sync.camera (reference: camera_sync/camera.py:13-256): per-camera state.

On the hot path a Camera is the holder of the intrinsics (`mtx`, `dist`, loaded from
calibration/<name>_<w>-<h>_<focus>.npz) and of the stateful world pose
(`world2cam`, low-passed `world_position`).  Frame capture (cv.VideoCapture) is outside the scope of
this package; image undistortion and the LM solve behind cv.calibrateCamera run on the GPU.
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
        """Reference camera.py:23-77 finds chessboard corners in `frames` with OpenCV, then calls
        cv.calibrateCamera.  The corner finder is not part of this package: find the corners with
        any detector and call Camera.calibrateFromPoints, which is the cv.calibrateCamera half."""
        raise NotImplementedError(
            "chessboard corner detection is outside this package's scope; "
            "use Camera.calibrateFromPoints(objpoints, imgpoints, image_size)"
        )

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
    def __init__(self, name, deviceId, calibrationFolder: str = None, focus=0, resolution=(640, 480)):
        if calibrationFolder is None:
            calibrationFolder = os.path.join(getBaseFolder(), "./calibration")
        if not os.path.exists(calibrationFolder):
            raise Exception("Calibration folder not found. Searched at " + calibrationFolder)
        self.calibrationFolder = calibrationFolder
        self.name = name
        if deviceId != -1:
            # camera.py:98-118 opens a cv.VideoCapture here
            raise NotImplementedError("frame capture is outside this package's scope: use deviceId=-1 and pass frames in")
        self.deviceId = deviceId
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
        raise NotImplementedError("frame capture is outside this package's scope")

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
