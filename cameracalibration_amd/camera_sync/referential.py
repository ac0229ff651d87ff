"""camera_sync.referential (reference: camera_sync/referential.py:8-156): rigid transforms.

The reference calls cv.Rodrigues twice per construction (referential.py:70, :83); here the
same conversion is host numpy (a 3x3 problem, no reason to visit the GPU), written to
OpenCV's conventions: matrix -> vector first projects onto SO(3) with an SVD, handles the
angle ~ pi branch through the diagonal, and returns a vector of norm in [0, pi].
"""
import numpy as np
from scipy.spatial.transform import Rotation as R

try:  # Python >= 3.11
    from typing import Self
except ImportError:  # the container runs 3.10
    from typing import TypeVar

    Self = TypeVar("Self", bound="Transform")


def rodrigues_to_matrix(rvec) -> np.ndarray:
    """Rotation vector -> 3x3 matrix, as cv.Rodrigues(rvec)[0]."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = float(np.sqrt(r @ r))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    c, s = np.cos(theta), np.sin(theta)
    kx = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s * kx


def matrix_to_rodrigues(rot_mat) -> np.ndarray:
    """3x3 matrix -> rotation vector, as cv.Rodrigues(R)[0].flatten()."""
    m = np.asarray(rot_mat, dtype=np.float64).reshape(3, 3)
    u, _, vt = np.linalg.svd(m)
    q = u @ vt  # nearest orthogonal matrix
    r = np.array([q[2, 1] - q[1, 2], q[0, 2] - q[2, 0], q[1, 0] - q[0, 1]])
    s = np.sqrt((r @ r) * 0.25)
    c = min(1.0, max(-1.0, (np.trace(q) - 1.0) * 0.5))
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        # angle ~ pi: the axis comes from the diagonal, signs from the first row
        t = np.sqrt(np.maximum((np.diag(q) + 1.0) * 0.5, 0.0))
        rx = t[0]
        ry = t[1] * (-1.0 if q[0, 1] < 0 else 1.0)
        rz = t[2] * (-1.0 if q[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((q[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.sqrt(v @ v))
    return r * (theta / (2.0 * s))


class Transform:
    """4x4 rigid transform with cached rvec / quaternion / inverse (referential.py:8-156)."""

    _transf_mat = None
    _rvec = None
    _inv = None
    _quat = None

    @staticmethod
    def fromRotationMatrix(rot_mat, tvec) -> Self:
        """referential.py:14-33."""
        m = np.eye(4)
        m[:3, :3] = rot_mat
        m[:3, 3] = np.array(tvec).flatten()
        return Transform(transf_mat=m)

    @staticmethod
    def fromQuaternion(quat, tvec, scalar_first=True) -> Self:
        """referential.py:35-56; [w,x,y,z] when scalar_first else [x,y,z,w]."""
        return Transform.fromRotationMatrix(rot_mat=R.from_quat(quat, scalar_first=scalar_first).as_matrix(), tvec=tvec)

    @staticmethod
    def fromRodrigues(rvec, tvec) -> Self:
        """referential.py:58-72."""
        return Transform.fromRotationMatrix(rot_mat=rodrigues_to_matrix(np.array(rvec).flatten()), tvec=tvec)

    def __init__(self, transf_mat=None):
        # The reference derives rvec and the quaternion on every construction (referential.py:82-84, an SVD and a
        # scipy call).  They are derived on first use here: the per-frame path builds several transforms per camera
        # and marker and reads neither, and the values are the same either way.
        self._transf_mat = transf_mat

    def _preCompute(self):
        # referential.py:82-84
        self._rvec = matrix_to_rodrigues(self.rot_mat)
        self._quat = R.from_matrix(self.rot_mat).as_quat(scalar_first=True)

    @property
    def transf_mat(self):
        return self._transf_mat

    @property
    def quat(self):
        if self._quat is None:
            self._quat = R.from_matrix(self.rot_mat).as_quat(scalar_first=True)
        return self._quat

    @property
    def rot_mat(self):
        return self.transf_mat[:3, :3]

    @property
    def rvec(self):
        if self._rvec is None:
            self._rvec = matrix_to_rodrigues(self.rot_mat)
        return self._rvec

    @property
    def tvec(self):
        return self.transf_mat[:3, 3]

    @property
    def kine_pose(self):
        """[x, y, z (m), qx, qy, qz, qw] for the inverse kinematics (referential.py:108-117)."""
        return np.array([*self.tvec / 1000, *np.roll(self.quat, -1)])

    @property
    def invert(self) -> Self:
        """Inverse transform, cached (referential.py:119-131)."""
        if self._inv is None:
            rt = self.rot_mat.T
            self._inv = Transform.fromRotationMatrix(rot_mat=rt, tvec=-rt @ self.tvec)
        return self._inv

    def apply(self, point):
        """R @ point + t (referential.py:133-140)."""
        return self.rot_mat @ point + self.tvec

    def combine(self, t2: Self) -> Self:
        """t2 applied after self, i.e. t2 @ self (referential.py:142-156)."""
        return Transform(transf_mat=t2.transf_mat @ self.transf_mat)
