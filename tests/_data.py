"""Shared fixtures loading for the tests (golden photos, calibrations)."""
import json
import os

import numpy as np
from PIL import Image

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_bgr(name):
    """Decode a golden PNG the way cv.imread does (8-bit BGR)."""
    rgb = np.array(Image.open(os.path.join(GOLDEN, name)).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def goldens():
    with open(os.path.join(GOLDEN, "aruco_goldens.json")) as f:
        return json.load(f)


def calibration(name):
    d = np.load(os.path.join(GOLDEN, "calibration", name))
    return np.ascontiguousarray(d["mtx"], np.float64), np.ascontiguousarray(d["dist"], np.float64)


def v1_anchor_points(ids=(10, 11, 14, 15)):
    """Object corners of the v1 sheet anchors as Aruco.getCornersAsList gives them."""
    tl = {10: (0, 0), 11: (110, 0), 12: (0, -95), 13: (110, -95), 14: (0, -190), 15: (110, -190)}
    out = []
    for i in ids:
        x, y = tl[i]
        s = 80
        out.append([[x, y, 0], [x + s, y, 0], [x + s, y - s, 0], [x, y - s, 0]])
    return np.array(out, np.float32)


def photo_sets():
    """The reference's other photo sets (copies of /root/reference/pics/...): name -> sorted file names under
    tests/golden.  `poses` and `robot_calib` are 640x480 JPEGs of the v1 marker sheet, `poses_blender` are 1920x1080
    Blender renders of it, `calib_blender` are renders of the 8 x 8 chessboard."""
    out = {}
    for d in ("poses", "poses_blender", "robot_calib", "calib_blender"):
        out[d] = [os.path.join(d, f) for f in sorted(os.listdir(os.path.join(GOLDEN, d)))]
    return out
