"""cv.aruco.DICT_4X4_50 (reference camera_sync/aruco.py:182, :233, :247).

Each entry is the 16 inner bits of rotation 0, row-major, MSB first, as two
bytes (b0 << 8 | b1) — the layout of OpenCV's ``Dictionary.bytesList``.

Provenance: ids 10..15 and 20..47 were decoded from the reference's own marker
sheets (camera_sync/blender/10_to_15.png, inkscape/20_to_47.png; SURVEY.md
Appendix B).  The remaining 16 ids have no asset in the reference: their
values are recalled from OpenCV's predefined_dictionaries.hpp and are
*parity unpinned*.
"""
import numpy as np

MARKER_SIZE = 4
MAX_CORRECTION_BITS = 1

_BYTES = {
    0: (181, 50), 1: (15, 154), 2: (51, 45), 3: (153, 70), 4: (84, 158),
    5: (121, 205), 6: (158, 46), 7: (196, 242), 8: (254, 218), 9: (207, 86),
    10: (249, 145), 11: (17, 167), 12: (14, 183), 13: (42, 15), 14: (36, 177),
    15: (38, 62), 16: (70, 101), 17: (102, 0), 18: (108, 94), 19: (118, 175),
    20: (134, 139), 21: (176, 43), 22: (204, 213), 23: (221, 130), 24: (254, 71),
    25: (148, 113), 26: (172, 228), 27: (165, 84), 28: (33, 35), 29: (52, 111),
    30: (68, 21), 31: (87, 178), 32: (158, 207), 33: (240, 203), 34: (8, 174),
    35: (9, 41), 36: (24, 117), 37: (4, 255), 38: (13, 246), 39: (28, 90),
    40: (23, 24), 41: (42, 40), 42: (50, 140), 43: (56, 178), 44: (36, 232),
    45: (46, 235), 46: (45, 63), 47: (75, 100), 48: (80, 46), 49: (80, 19),
}

VERIFIED_IDS = tuple(list(range(10, 16)) + list(range(20, 48)))

CODES = np.array([(_BYTES[i][0] << 8) | _BYTES[i][1] for i in range(50)], dtype=np.uint16)


def marker_bits(marker_id: int) -> np.ndarray:
    """4x4 array of the marker's inner bits (1 = white)."""
    code = int(CODES[marker_id])
    return np.array([(code >> (15 - i)) & 1 for i in range(16)], dtype=np.uint8).reshape(4, 4)
