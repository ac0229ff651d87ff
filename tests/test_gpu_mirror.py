"""The camera_sync mirror API on the GPU, written after the reference's own GPU-relevant tests
(test_aruco.py:203-359): same calls, same assertions, plus the batched additions and the
distributed calibration driver with real GPU workers."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import _data, _oracle

pytestmark = pytest.mark.gpu


def _markers(ids):
    from cameracalibration_amd.camera_sync.aruco import getArucosFromPaper

    a = getArucosFromPaper(1)
    return [a[i] for i in ids]


def test_detect_aruco_like_reference():
    """test_aruco.py:203-326."""
    from cameracalibration_amd.camera_sync.aruco import detectAruco

    gold = _data.goldens()
    for name in ("pic_1", "pic_2"):
        res, rejected = detectAruco(_data.load_bgr(name + ".png"))
        for n in range(10, 16):
            assert n in res.keys()
        for k in res.keys():
            assert res[k].shape == (1, 4, 2) and res[k].dtype == np.float32
            assert np.allclose(res[k], np.array(gold[name][str(int(k))], np.float32).reshape(1, 4, 2), atol=1e-3)
        assert isinstance(rejected, tuple) and all(r.shape == (1, 4, 2) for r in rejected)
    # gray input gives the same result as BGR (aruco.py:246 converts first)
    img = _data.load_bgr("pic_2.png")
    g = _oracle.bgr2gray(img)
    res_g, _ = detectAruco(g)
    res_c, _ = detectAruco(img)
    assert res_g.keys() == res_c.keys() and all(np.array_equal(res_g[k], res_c[k]) for k in res_c)
    none, _ = detectAruco(np.full((120, 160, 3), 128, np.uint8))
    assert none == {}


def test_process_aruco_like_reference():
    """test_aruco.py:329-359."""
    from cameracalibration_amd.camera_sync.aruco import processAruco
    from cameracalibration_amd.camera_sync.camera import Camera

    cam = Camera("Logitec_A", -1, focus=0, resolution=(1280, 720))
    frame = _data.load_bgr("pic_2.png")
    moving = _markers([12, 13])
    rvec, tvec, found, metrics = processAruco(_markers([10, 11, 14, 15]), moving, cam, frame, metrics=True)
    assert metrics["PnP"]["MAE"] < 0.6 and metrics["PnP"]["RMSE"] < 0.8
    assert rvec.shape == (3, 1) and tvec.shape == (3, 1)
    assert np.allclose([0.00686839, -2.92023025, 1.16051479], rvec.flatten(), atol=1e-4)
    assert np.allclose([37.15601032, -37.2392403, 580.35313275], tvec.flatten(), atol=1e-4)
    for n in (12, 13):
        assert n in found
    # the moving markers lie on the sheet: their located corners are close to the printed layout (mm)
    for m, want in zip(moving, ([0, -95, 0], [110, -95, 0])):
        c = np.array(m.corners)
        assert np.allclose(c.mean(axis=0), np.array(want) + [40, -40, 0], atol=6.0)
    # camera state was updated through the low-pass filter from (0,0,0)
    assert np.allclose(cam.world_position.coords, 0.4 * cam.world2cam.invert.apply([0.0, 0.0, 0.0]))


def test_process_aruco_errors():
    from cameracalibration_amd.camera_sync.aruco import PnP, processAruco
    from cameracalibration_amd.camera_sync.camera import Camera
    from cameracalibration_amd._hip import CamsyncError

    cam = Camera("Logitec_A", -1, focus=0, resolution=(1280, 720))
    blank = np.full((720, 1280, 3), 200, np.uint8)
    with pytest.raises(Exception, match="Could not find any ArUco markers in the image"):
        processAruco(_markers([10, 11]), [], cam, blank)
    assert processAruco(_markers([10, 11]), [], cam, blank, accept_none=True) is False
    with pytest.raises(CamsyncError):
        PnP(np.zeros((3, 2), np.float32), np.zeros((3, 3), np.float32), cam.mtx, cam.dist)


def test_multiple_cameras_average():
    """aruco.py:407-470 with the same photo seen by two 'cameras': the averaged corners equal one camera's."""
    from cameracalibration_amd.camera_sync.aruco import processAruco, processArucoFromMultipleCameras
    from cameracalibration_amd.camera_sync.camera import Camera
    from cameracalibration_amd.camera_sync.aruco import getArucosFromPaper

    frame = _data.load_bgr("pic_2.png")
    cams = [Camera("Logitec_A", -1, focus=0, resolution=(1280, 720)) for _ in range(2)]
    sheet = getArucosFromPaper(1)
    moving = {12: sheet[12], 13: sheet[13]}
    before = {k: np.array([c.coords for c in v.corners], float) for k, v in moving.items()}
    processArucoFromMultipleCameras(_markers([10, 11, 14, 15]), moving, cams, [frame, frame])
    ref_cam = Camera("Logitec_A", -1, focus=0, resolution=(1280, 720))
    _, _, located, _ = processAruco(_markers([10, 11, 14, 15]), _markers([12, 13]), ref_cam, frame, directUpdate=False)
    for k, a in moving.items():
        size = a.size
        local = [[-size / 2, size / 2, 0], [size / 2, size / 2, 0], [size / 2, -size / 2, 0], [-size / 2, -size / 2, 0]]
        world = np.array([located[k].apply(np.array(c)) for c in local])
        got = np.array([c.coords for c in a.corners], float)
        assert np.allclose(got, 0.4 * world + 0.6 * before[k], atol=1e-9)
    with pytest.raises(AssertionError):
        processArucoFromMultipleCameras([], moving, cams, [frame])


def _eight_camera_scene():
    """Eight 1080p cameras with distinct intrinsics, one camera that sees nothing and one 720p camera looking at the
    reference's pic_2.png with the reference's Logitec_A calibration (the golden-pose camera, test_aruco.py:329-359)."""
    from cameracalibration_amd import synth
    from cameracalibration_amd.camera_sync.camera import Camera

    frames, cams = [], []
    for i in range(8):
        cam = Camera("Logitec_robot", -1, focus=10, resolution=(1920, 1080))
        cam.mtx = cam.mtx.copy()
        cam.mtx[0, 0] *= 1.0 + 0.01 * i
        cam.mtx[1, 1] *= 1.0 - 0.005 * i
        cam.dist = cam.dist * (1.0 + 0.1 * i)
        frames.append(synth.v1_frame(200 + i, cam.mtx, ss=2, device="cuda")[0].cpu().numpy())
        cams.append(cam)
    cams.append(Camera("Logitec_robot", -1, focus=10, resolution=(1920, 1080)))
    frames.append(np.full((1080, 1920, 3), 90, np.uint8))            # sees nothing: skipped
    cams.append(Camera("Logitec_A", -1, focus=0, resolution=(1280, 720)))
    frames.append(_data.load_bgr("pic_2.png"))                        # another resolution: its own batch
    return frames, cams


def _rt_matrix(rvec, tvec):
    m = np.eye(4)
    m[:3, :3] = _oracle.rodrigues_vec2mat(rvec)
    m[:3, 3] = np.asarray(tvec).ravel()
    return m


def test_multiple_cameras_batched_equals_the_per_camera_loop():
    """processArucoFromMultipleCameras runs all cameras as one batch; the reference loops over the
    cameras (aruco.py:417).  Camera poses, filtered positions and the fused marker corners must (1) equal the
    per-camera loop's exactly and (2) equal aruco.py:407-470 restated here in numpy over the CPU oracle's detection,
    solvePnP and IPPE_SQUARE; the pic_2 / Logitec_A camera must also land on the reference's golden pose."""
    import copy
    from cameracalibration_amd.camera_sync.aruco import (Aruco, getArucosFromPaper, processAruco,
                                                         processArucoFromMultipleCameras)

    frames, cams = _eight_camera_scene()
    sheet = getArucosFromPaper(1)
    fixed = [sheet[i] for i in (10, 11, 14, 15)]

    def run(batched, sets=2, metrics=True):
        cs = copy.deepcopy(cams)
        moving = {k: copy.deepcopy(sheet[k]) for k in (12, 13)}
        for _ in range(sets):  # twice: the low-pass filters carry state from the first set to the second
            if batched:
                processArucoFromMultipleCameras(fixed, moving, cs, frames, getMetrics=metrics)
            else:
                seen = {}
                for cam, f in zip(cs, frames):
                    res = processAruco(fixed, moving.values(), cam, f, accept_none=True, directUpdate=False, metrics=True)
                    if not res:
                        continue
                    for key, a2w in res[2].items():
                        size = moving[key].size
                        local = Aruco.getCornersFromTopLeft(np.array([-size / 2, size / 2, 0]), size)
                        seen.setdefault(key, []).append([a2w.apply(c) for c in local])
                for a in moving.values():
                    if a.id in seen:
                        for i, p in enumerate(np.mean(seen[a.id], axis=0)):
                            a.corners[i].updatePos(p)
        return cs, moving

    cb, mb = run(True)
    cl, ml = run(False)
    located = 0
    for a, b in zip(cb, cl):
        assert np.array_equal(a.world_position.coords, b.world_position.coords)
        assert hasattr(a, "world2cam") == hasattr(b, "world2cam")
        if hasattr(a, "world2cam"):
            located += 1
            assert np.array_equal(a.world2cam.transf_mat, b.world2cam.transf_mat)
    assert located == 9
    for k in mb:
        got = np.array([c.coords for c in mb[k].corners], float)
        want = np.array([c.coords for c in ml[k].corners], float)
        assert np.array_equal(got, want)
        assert not np.array_equal(got, np.array([c.coords for c in sheet[k].corners], float))
    # the per-marker work on the device as well (camsync_fuse_moving_markers; two frame sizes = two batches here):
    # the cameras exactly, the fused corners to rounding (the device sums in another order than numpy's matmul)
    cf, mf = run(True, metrics=False)
    for a, b in zip(cf, cl):
        assert np.array_equal(a.world_position.coords, b.world_position.coords)
        assert hasattr(a, "world2cam") == hasattr(b, "world2cam")
    for k in mf:
        got = np.array([c.coords for c in mf[k].corners], float)
        want = np.array([c.coords for c in ml[k].corners], float)
        assert np.allclose(got, want, rtol=1e-12, atol=1e-10), (k, np.abs(got - want).max())

    # ---- against the oracle and the reference's golden pose: one set of frames from fresh state ----
    c1, m1 = run(True, sets=1, metrics=False)
    G = _data.goldens()["pic_2_pose"]
    seen = {}
    for cam, f, got in zip(cams, frames, c1):
        oids, ocor, _ = _oracle.detect(f)
        found = {int(i): c for i, c in zip(oids, ocor)}
        obj = [p for a in fixed if a.id in found for p in a.getCornersAsList()]
        img = [p for a in fixed if a.id in found for p in found[a.id]]
        if len(img) < 4:
            assert not hasattr(got, "world2cam")
            continue
        ok, orv, otv = _oracle.solve_pnp(np.array(obj, np.float32), np.array(img, np.float32), cam.mtx, cam.dist)
        assert ok
        w2c = _rt_matrix(orv, otv)
        assert np.allclose(got.world2cam.transf_mat[:3, :3], w2c[:3, :3], rtol=0, atol=1e-6)
        assert np.allclose(got.world2cam.transf_mat[:3, 3], w2c[:3, 3], rtol=1e-6, atol=1e-6)
        c2w = np.linalg.inv(w2c)
        assert np.allclose(got.world_position.coords, 0.4 * c2w[:3, 3], rtol=1e-6, atol=1e-6)  # low-pass from (0,0,0)
        if tuple(cam._resolution) == (1280, 720):
            assert np.allclose(got.world2cam.transf_mat[:3, 3], G["tvec"], atol=1e-4)
            # the golden rotation is 3.1424 rad > pi: compared as a matrix (Transform keeps the round-tripped rvec)
            assert np.allclose(got.world2cam.transf_mat[:3, :3], _oracle.rodrigues_vec2mat(G["rvec"]), atol=1e-4)
        for k in (12, 13):
            if k not in found:
                continue
            size = sheet[k].size
            ok, mrv, mtv = _oracle.ippe_square(found[k], size, cam.mtx, cam.dist)
            assert ok
            a2w = c2w @ _rt_matrix(mrv, mtv)
            h = size / 2
            local = np.array([[-h, h, 0, 1], [h, h, 0, 1], [h, -h, 0, 1], [-h, -h, 0, 1.0]])
            seen.setdefault(k, []).append((a2w @ local.T).T[:, :3])
    assert set(seen) == {12, 13} and all(len(v) == 9 for v in seen.values())
    for k, a in m1.items():
        before = np.array([c.coords for c in sheet[k].corners], float)
        want = 0.4 * np.mean(seen[k], axis=0) + 0.6 * before
        got = np.array([c.coords for c in a.corners], float)
        assert np.allclose(got, want, rtol=0, atol=1e-4), (k, np.abs(got - want).max())  # millimetres


def test_batched_additions_equal_per_frame_calls():
    from cameracalibration_amd import synth
    from cameracalibration_amd.camera_sync.aruco import PnP, detectAruco, detectArucoBatch, processArucoBatch

    K, dist_ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    frames = np.stack([synth.v1_frame(i, K, ss=2)[0].numpy() for i in range(3)])
    batch = detectArucoBatch(frames)
    anchors = _markers([10, 11, 14, 15])
    rv, tv, ok, dets, met = processArucoBatch(anchors, frames, K, dist_, getMetrics=True)
    for b in range(3):
        single, _ = detectAruco(frames[b])
        assert single.keys() == batch[b][0].keys() == dets[b].keys()
        assert all(np.array_equal(single[k], batch[b][0][k]) for k in single)
        img, obj = [], []
        for a in anchors:
            if single.get(a.id) is not None:
                obj.extend(a.getCornersAsList()); img.extend(single[a.id][0])
        r1, t1, m1 = PnP(np.array(img, np.float32), np.array(obj, np.float32), K, dist_, True)
        assert ok[b] and np.allclose(rv[b], r1.ravel(), atol=1e-12) and np.allclose(tv[b], t1.ravel(), atol=1e-9)
        assert abs(met["MAE"][b] - m1["MAE"]) < 1e-9


def test_camera_calibrate_from_points_vs_oracle():
    from cameracalibration_amd import synth
    from cameracalibration_amd.camera_sync.camera import Camera

    obj, img, _, _ = synth.calib_views(V=9, seed=11, rational=False)
    mtx, d = Camera.calibrateFromPoints(list(obj), [i.reshape(-1, 1, 2) for i in img], (1920, 1080))
    rms0, K0, d0, _, _, _ = _oracle.calibrate(obj, img, (1920, 1080), False)
    assert mtx.shape == (3, 3) and d.shape == (1, 5)
    assert np.allclose(mtx, K0, atol=1e-5) and np.allclose(d.ravel(), d0, atol=1e-6)
    rms, *_ = Camera.calibrateFromPoints(obj, img, (1920, 1080), returnAll=True)
    assert abs(rms - rms0) < 1e-6  # north-star tolerance, px


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, V, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cameracalibration_amd import synth
        from cameracalibration_amd.calibrate_dist import calibrate_distributed

        obj, img, _, _ = synth.calib_views(V=V, seed=3, rational=True)
        rms, K, d, rv, tv, span = calibrate_distributed(obj, img, (1920, 1080), rational=True, device_index=0)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), rms=rms, K=K, d=d, rv=rv, tv=tv, span=span)
    finally:
        dist.destroy_process_group()


def test_distributed_calibration_two_ranks_on_one_gpu(tmp_path):
    """Two processes share GPU 0 (records all-reduced over gloo): same answer as the one-GPU entry point."""
    from cameracalibration_amd import _hip, synth

    V = 31
    mp.spawn(_rank_main, args=(2, _free_port(), V, str(tmp_path)), nprocs=2, join=True)
    res = [np.load(tmp_path / f"r{r}.npz") for r in range(2)]
    obj, img, _, _ = synth.calib_views(V=V, seed=3, rational=True)
    ctx = _hip.Context(32, 32, 1)
    rms1, K1, d1, rv1, tv1, _ = ctx.calibrate(obj, img, (1920, 1080), rational=True)
    rms0 = _oracle.calibrate(obj, img, (1920, 1080), True)[0]
    assert np.array_equal(res[0]["K"], res[1]["K"]) and np.array_equal(res[0]["d"], res[1]["d"])
    assert abs(float(res[0]["rms"]) - rms1) < 1e-9 and abs(float(res[0]["rms"]) - rms0) < 1e-6
    assert np.allclose(res[0]["K"], K1, atol=1e-5) and np.allclose(res[0]["d"], d1, atol=1e-4)
    rv = np.concatenate([r["rv"] for r in res]); tv = np.concatenate([r["tv"] for r in res])
    assert np.allclose(rv, rv1, atol=1e-7) and np.allclose(tv, tv1, atol=1e-6)
    assert [tuple(r["span"]) for r in res] == [(0, 16), (16, 31)]


def _nccl_rank_main(rank, world, port, V, out_dir):
    import torch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from cameracalibration_amd import synth
        from cameracalibration_amd.calibrate_dist import calibrate_distributed

        obj, img, _, _ = synth.calib_views(V=V, seed=5, rational=True)
        stats = {}
        rms, K, d, rv, tv, span, iters = calibrate_distributed(obj, img, (1920, 1080), rational=True, device_index=0,
                                                               return_iters=True, stats=stats)
        np.savez(os.path.join(out_dir, "nccl.npz"), rms=rms, K=K, d=d, rv=rv, tv=tv, span=span, iters=iters, calls=stats["calls"])
    finally:
        dist.destroy_process_group()


def test_v2_sheet_as_anchors_batched_equals_the_per_camera_loop():
    """The reference's v2 sheet (aruco.py:140-150: 25 markers) as fixedArucos is 23 anchors = 92 points with two of
    them moving: more than one point per lane in the PnP kernels, in the fused batch and in the mirror's PnP."""
    import copy
    from cameracalibration_amd import synth
    from cameracalibration_amd.camera_sync.aruco import (Aruco, getArucosFromPaper, processAruco,
                                                         processArucoFromMultipleCameras)
    from cameracalibration_amd.camera_sync.camera import Camera

    sheet = getArucosFromPaper(2)
    layout = [(a.id, float(a.corners[0].coords[0]), float(a.corners[0].coords[1]), float(a.size)) for a in sheet.values()]
    cx = np.mean([m[1] + m[3] / 2 for m in layout])
    cy = np.mean([m[2] - m[3] / 2 for m in layout])
    frames, cams = [], []
    for i in range(3):
        cam = Camera("Logitec_robot", -1, focus=10, resolution=(1920, 1080))
        rng = np.random.default_rng(900 + i)
        R, t = synth.look_at_pose(rng, (cx, cy), dist_range=(300.0, 420.0), tilt_range=(0.0, 35.0))
        frames.append(synth.render_sheet(layout, cam.mtx, R, t, 1080, 1920, seed=900 + i, ss=2, device="cuda").cpu().numpy())
        cams.append(cam)
    ids = sorted(sheet)
    fixed = [sheet[i] for i in ids[:-2]]
    assert len(fixed) == 23

    def run(batched):
        cs = copy.deepcopy(cams)
        moving = {k: copy.deepcopy(sheet[k]) for k in ids[-2:]}
        if batched:
            processArucoFromMultipleCameras(fixed, moving, cs, frames, getMetrics=True)
        else:
            for cam, f in zip(cs, frames):
                res = processAruco(fixed, moving.values(), cam, f, accept_none=True, directUpdate=False, metrics=True)
                assert res, "the camera must see the sheet"
        return cs

    cb, cl = run(True), run(False)
    for a, b in zip(cb, cl):
        assert hasattr(a, "world2cam") and hasattr(b, "world2cam")
        assert np.allclose(a.world2cam.transf_mat, b.world2cam.transf_mat, rtol=1e-9, atol=1e-9)
        assert np.allclose(a.world_position.coords, b.world_position.coords, rtol=1e-9, atol=1e-9)


def test_distributed_calibration_over_rccl_world_1(tmp_path):
    """The RCCL leg of the distributed calibration (process group backend "nccl"), as far as one GPU can run it:
    a one-rank group, every record of every LM iteration all-reduced by RCCL on the device buffer.  The result must be
    bit-identical to the single-GPU entry point (a one-rank sum changes nothing)."""
    from cameracalibration_amd import _hip, synth

    V = 40
    mp.spawn(_nccl_rank_main, args=(1, _free_port(), V, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "nccl.npz")
    obj, img, _, _ = synth.calib_views(V=V, seed=5, rational=True)
    ctx = _hip.Context(32, 32, 1)
    rms1, K1, d1, rv1, tv1, it1 = ctx.calibrate(obj, img, (1920, 1080), rational=True)
    # one collective for the focal sums, one for the first record, then ONE per trial step (the trial's error travels
    # with the record of the trial parameters; a rejected step costs a second one for the re-damped record)
    assert int(r["iters"]) == it1 and int(r["calls"]) >= 2 + it1, (it1, int(r["calls"]))
    assert float(r["rms"]) == rms1
    assert np.array_equal(r["K"], K1) and np.array_equal(r["d"], d1)
    assert np.array_equal(r["rv"], rv1) and np.array_equal(r["tv"], tv1)
    assert abs(rms1 - _oracle.calibrate(obj, img, (1920, 1080), True)[0]) < 1e-6


def test_fuse_moving_markers_needs_the_pose_call_before_it():
    """camsync_fuse_moving_markers works on what camsync_detect_pnp_batch / _frames left on the device: without such a
    call, after a call of another batch size, or after a call that took several chunks, it refuses."""
    from cameracalibration_amd import _hip

    frame = _data.load_bgr("pic_2.png")
    K, d = _data.calibration("Logitec_A_1280-720_0.npz")
    ctx = _hip.Context(720, 1280, 2)
    with pytest.raises(_hip.CamsyncError):
        ctx.fuse_moving_markers(1, [12, 13], [80.0, 80.0])
    anchors = [10, 11, 14, 15]
    ctx.detect_pnp(np.stack([frame, frame]), anchors, _data.v1_anchor_points(anchors), K, d)
    with pytest.raises(_hip.CamsyncError):
        ctx.fuse_moving_markers(1, [12, 13], [80.0, 80.0])
    fused, seen, rv, tv, loc = ctx.fuse_moving_markers(2, [12, 13, 40], [80.0, 80.0, 30.0], want_poses=True)
    assert seen.tolist() == [2, 2, 0] and loc.tolist() == [[1, 1, 0], [1, 1, 0]] and not fused[2].any()
    # both "cameras" saw the same frame: the mean is either camera's corners, and the markers' poses are IPPE's
    oids, ocor, _ = _oracle.detect(frame)
    for m, i in enumerate((12, 13)):
        ok, orv, otv = _oracle.ippe_square(ocor[oids.tolist().index(i)], 80.0, K, d)
        assert ok and np.allclose(rv[0, m], orv, rtol=1e-6, atol=1e-8) and np.allclose(tv[0, m], otv, rtol=1e-6, atol=1e-6)
        assert np.array_equal(rv[0, m], rv[1, m])
    ctx.detect_pnp(np.stack([frame, frame, frame]), anchors, _data.v1_anchor_points(anchors), K, d)  # two chunks
    with pytest.raises(_hip.CamsyncError):
        ctx.fuse_moving_markers(3, [12], [80.0])
    ctx.close()
