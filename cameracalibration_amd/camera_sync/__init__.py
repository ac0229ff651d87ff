"""Drop-in mirror of the reference package `camera_sync` for its vision hot path.

Same module and function names as the reference (aruco, camera, position,
referential, tools); every call that the reference hands to OpenCV goes to
libcamsync_hip.so through `cameracalibration_amd._hip` instead.  There is no
CPU fallback: importing works without a GPU, calling a compute function
without the HIP library or a GPU raises.
"""
from . import aruco, camera, ingest, position, referential, tools  # noqa: F401
