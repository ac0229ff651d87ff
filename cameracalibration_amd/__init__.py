"""cameracalibration_amd — MI355X-native implementation of camera_sync's
per-frame vision hot path (6figuress/CameraCalibration).

`cameracalibration_amd.camera_sync` mirrors the reference package's modules
(aruco, camera, position, referential, tools); the arithmetic the reference
delegates to OpenCV runs in libcamsync_hip.so (include/camsync.h).
"""
__all__ = ["camera_sync"]
