"""camera_sync.tools (reference: camera_sync/tools.py:5-6)."""
import os


def getBaseFolder() -> str:
    """Folder of this package; the calibration/ folder is looked up relative to it."""
    return os.path.dirname(os.path.abspath(__file__))
