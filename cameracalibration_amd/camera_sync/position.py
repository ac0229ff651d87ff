"""camera_sync.position (reference: camera_sync/position.py:3-53): 2-D / 3-D points with the
exponential low-pass filter the tracking loop applies to every pose update."""
import numpy as np

ALPHA = 0.4  # weight of the new sample (position.py:3)


def lowPassFilter(new_position, old_position, alpha: float):
    """alpha * new + (1 - alpha) * old  (position.py:50-53)."""
    return alpha * new_position + (1 - alpha) * old_position


class Point:
    """Image point (position.py:6-37)."""

    coords: np.ndarray

    def __init__(self, x: float, y: float):
        self.coords = np.array([x, y])

    @property
    def x(self) -> float:
        return self.coords[0]

    @property
    def y(self) -> float:
        return self.coords[1]

    def updatePos(self, newPos, noFilter=False) -> None:
        """Replace the coordinates, through the low-pass filter unless noFilter (position.py:20-37)."""
        self.coords = newPos if noFilter else lowPassFilter(newPos, self.coords, ALPHA)


class Position(Point):
    """World point (position.py:40-47)."""

    def __init__(self, x: float, y: float, z: float):
        self.coords = np.array([x, y, z])

    @property
    def z(self) -> float:
        return self.coords[2]
