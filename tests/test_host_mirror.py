"""Host-side logic of the camera_sync mirror, following the reference's own CPU tests
(test_postition.py, test_referential.py, test_aruco.py:12-181).  No GPU needed."""
import numpy as np
import pytest

from cameracalibration_amd.camera_sync.aruco import Aruco, getArucosFromPaper, generate_aruco_marker
from cameracalibration_amd.camera_sync.camera import Camera
from cameracalibration_amd.camera_sync.position import ALPHA, Point, Position, lowPassFilter
from cameracalibration_amd.camera_sync.referential import Transform

RZ90 = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
RX90 = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])
RY90 = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]])


# ---- position (test_postition.py) ----
def test_point_and_position():
    p = Point(0.56, 2)
    assert isinstance(p.x, float) and isinstance(p.y, float) and p.x == 0.56 and p.y == 2.0
    assert np.array_equal(Point(1.5, 2.5).coords, np.array([1.5, 2.5]))
    q = Position(1.0, 2.0, 3.0)
    assert (q.x, q.y, q.z) == (1.0, 2.0, 3.0) and isinstance(q, Point)


def test_low_pass_filter_and_update():
    assert np.array_equal(lowPassFilter(np.array([10.0, 20.0]), np.array([5.0, 15.0]), 0.4), np.array([7.0, 17.0]))
    p = Point(1.0, 2.0)
    p.updatePos(np.array([5.0, 6.0]), noFilter=True)
    assert np.array_equal(p.coords, [5.0, 6.0])
    p = Point(5.0, 15.0)
    p.updatePos(np.array([10.0, 20.0]), noFilter=False)
    assert np.array_equal(p.coords, np.array([5.0 + 5.0 * ALPHA, 15.0 + 5.0 * ALPHA]))


# ---- referential (test_referential.py) ----
def test_transform_construction():
    assert np.array_equal(Transform(transf_mat=np.eye(4)).transf_mat, np.eye(4))
    t = Transform.fromRodrigues(rvec=np.array([0.1, 0.2, 0.3]), tvec=np.array([1.0, 2.0, 3.0]))
    assert np.allclose(t.tvec, [1, 2, 3]) and np.allclose(t.rvec, [0.1, 0.2, 0.3])
    t = Transform.fromRotationMatrix(rot_mat=RZ90, tvec=np.array([1.0, 2.0, 3.0]))
    m = np.eye(4); m[:3, :3] = RZ90; m[:3, 3] = [1, 2, 3]
    assert np.array_equal(t.transf_mat, m) and np.array_equal(t.rot_mat, RZ90)
    r = Transform.fromRotationMatrix(rot_mat=RZ90, tvec=[0, 0, 0]).rvec.flatten()
    assert abs(r[2] - np.pi / 2) < 1e-4 and abs(r[0]) < 1e-4 and abs(r[1]) < 1e-4


def test_transform_invert_apply_combine():
    t = Transform.fromRotationMatrix(rot_mat=RZ90, tvec=np.array([1.0, 2.0, 3.0]))
    inv = t.invert
    assert np.array_equal(inv.rot_mat, RZ90.T) and np.array_equal(inv.tvec, -RZ90.T @ np.array([1.0, 2.0, 3.0]))
    assert np.array_equal(np.linalg.inv(t.transf_mat), inv.transf_mat)
    assert np.allclose(t.apply(np.array([1.0, 0, 0])), [1.0, 3.0, 3.0])
    t2 = Transform.fromRotationMatrix(rot_mat=RY90, tvec=np.array([4, 5, 6]))
    c = t.combine(t2)
    assert np.array_equal(c.transf_mat, t2.transf_mat @ t.transf_mat)
    p = np.array([1.0, 1.0, 1.0])
    assert np.allclose(c.apply(p), t2.apply(t.apply(p)))
    pt = np.array([1, 2, 3])
    assert np.allclose(Transform.fromRotationMatrix(RX90, [0, 0, 0]).apply(pt), [1, -3, 2])
    assert np.allclose(Transform.fromRotationMatrix(RY90, [0, 0, 0]).apply(pt), [3, 2, -1])
    assert np.allclose(Transform.fromRotationMatrix(RZ90, [0, 0, 0]).apply(pt), [-2, 1, 3])


def test_transform_chain_inverse_quaternion():
    t1 = Transform.fromRotationMatrix(RZ90, [5, 0, 0])
    t2 = Transform.fromRotationMatrix(RX90, [0, 3, 0])
    p1 = t1.apply(np.array([1, 1, 1])); p2 = t2.apply(p1)
    assert np.allclose(p1, [4, 1, 1]) and np.allclose(p2, [4, 2, 1])
    assert np.allclose(t1.combine(t2).apply(np.array([1, 1, 1])), p2)
    t = Transform.fromRodrigues(rvec=np.array([0.3, -0.5, 0.7]), tvec=np.array([4.2, -1.3, 2.8]))
    for p in (np.array([1, 0, 0]), np.array([0, 5, -2]), np.array([3.5, -2.7, 1.2])):
        assert np.allclose(p, t.invert.apply(t.apply(p)), atol=1e-10)
    assert np.allclose(Transform.fromRotationMatrix(np.diag([-1, 1, -1]), [0, 0, 0]).quat, [0, 0, 1, 0])
    assert np.allclose(Transform.fromRotationMatrix(RZ90, [0, 0, 0]).quat, [np.sqrt(2) / 2, 0, 0, np.sqrt(2) / 2])
    q = Transform.fromQuaternion([np.sqrt(2) / 2, 0, 0, np.sqrt(2) / 2], [1, 2, 3])
    assert np.allclose(q.rot_mat, RZ90) and np.allclose(q.kine_pose[:3], [0.001, 0.002, 0.003])


# ---- Aruco geometry (test_aruco.py:12-181) ----
def test_aruco_class():
    a = Aruco(1, 50)
    assert a.id == 1 and a.size == 50 and a.corners is None and a.isLocated is False
    a = Aruco(2, 50, Position(10, 20, 30))
    assert a.isLocated is True and len(a.corners) == 4
    assert [(c.x, c.y, c.z) for c in a.corners] == [(10, 20, 30), (60, 20, 30), (60, -30, 30), (10, -30, 30)]
    c = Aruco.getCornersFromTopLeft(np.array([10, 20, 30], dtype=float), 40)
    assert np.array_equal(c, [[10, 20, 30], [50, 20, 30], [50, -20, 30], [10, -20, 30]])
    a = Aruco(1, 40, Position(10, 20, 30))
    ctr = a.getCenter()
    assert (ctr.x, ctr.y, ctr.z) == (30, 0, 30)
    lst = a.getCornersAsList()
    assert isinstance(lst, np.ndarray) and lst.shape == (4, 3) and lst.dtype == np.float32
    assert np.array_equal(lst[0], [10, 20, 30])
    a.updatePose(Transform.fromRotationMatrix(np.eye(3), [1, 2, 3]))
    assert np.allclose(a.corners[0], [-19, -18, 3]) and np.allclose(a.corners[2], [21, 22, 3])


def test_get_arucos_from_paper():
    v1 = getArucosFromPaper(1)
    assert len(v1) == 10 and all(k in v1 for k in (0, 1, 10, 15))
    assert v1[11].id == 11 and v1[11].size == 80
    assert (v1[11].corners[0].x, v1[11].corners[0].y, v1[11].corners[0].z) == (110, 0, 0)
    v2 = getArucosFromPaper(2)
    assert len(v2) == 25 and all(v2[i].size == 30 for i in range(20, 45))
    assert (v2[21].corners[0].x, v2[21].corners[0].y) == (35, 0) and (v2[25].corners[0].x, v2[25].corners[0].y) == (0, -35)
    v3 = getArucosFromPaper(3)
    assert sorted(v3) == [44, 45, 46, 47] and v3[44].size == 77
    assert (v3[47].corners[0].x, v3[47].corners[0].y) == (150.4, -220)
    v4 = getArucosFromPaper(4)
    assert sorted(v4) == list(range(32, 48)) and v4[35].size == 35
    assert np.allclose(v4[35].corners[0].coords, [40, -40, 0]) and np.allclose(v4[47].corners[0].coords, [190.14, -260.14, 0])
    assert {k: v.size for k, v in getArucosFromPaper().items()} == {k: v.size for k, v in v1.items()}


@pytest.mark.parametrize("marker_id", [0, 10, 20])
def test_generate_aruco_marker(marker_id, monkeypatch):
    from cameracalibration_amd.camera_sync import aruco as mod

    seen = {}
    monkeypatch.setattr(mod, "_imwrite", lambda path, img: seen.setdefault("path", path))
    monkeypatch.setattr("builtins.print", lambda *a, **k: None)
    img = generate_aruco_marker(marker_id)
    assert seen["path"] == f"./aruco/aruco_{marker_id}.png"
    assert isinstance(img, np.ndarray) and img.shape == (200, 200) and img.dtype == np.uint8
    assert img[0].max() == 0 and img[:, -1].max() == 0  # one-cell black border


# ---- Camera ----
def test_camera_loads_calibration_and_tracks_pose(capsys):
    cam = Camera("Logitec_A", -1, focus=0, resolution=(1280, 720))
    assert cam.calibrated and cam.mtx.shape == (3, 3) and cam.dist.size == 5
    assert cam.calibrationFileName == "Logitec_A_1280-720_0.npz"
    pos, rot = cam.updateWorldPosition(np.array([0.0, 0.0, np.pi / 2]), np.array([1.0, 2.0, 3.0]))
    assert np.allclose(rot, RZ90) and np.allclose(pos.coords, -RZ90.T @ np.array([1.0, 2.0, 3.0]))
    before = pos.coords.copy()
    cam.updateWorldPosition(np.zeros(3), np.array([0.0, 0.0, 10.0]), noLowPassFilter=False)
    assert np.allclose(cam.world_position.coords, ALPHA * np.array([0, 0, -10.0]) + (1 - ALPHA) * before)
    none = Camera("nobody", -1)
    assert not none.calibrated
    with pytest.raises(Exception, match="Calibration folder not found"):
        Camera("x", -1, calibrationFolder="/nonexistent/folder")
    absent = Camera("Logitec_A", 62)  # no such camera here: like cv.VideoCapture, no exception and no frames
    assert not absent.captureStream.isOpened() and absent.takePic() is False
    assert np.array_equal(Camera.boardPoints((7, 7))[:3], [[0, 0, 0], [1, 0, 0], [2, 0, 0]])
