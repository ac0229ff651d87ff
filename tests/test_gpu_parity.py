"""GPU parity tests: libcamsync_hip (through the C ABI) against the CPU oracle
and the reference's golden vectors.  Run on the MI355X box with -m gpu."""
import numpy as np
import pytest

from tests import _data, _oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from cameracalibration_amd import _hip
    c = _hip.Context(1080, 1920, max_batch=4)
    yield c
    c.close()


@pytest.fixture(scope="module")
def synth_frames():
    from cameracalibration_amd import synth
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    return [synth.v1_frame(i, K, ss=2)[0].numpy() for i in range(3)]


def _oracle_bin(gray):
    return [orc.adaptive_threshold(gray, w) for w in (3, 13, 23)]


@pytest.mark.parametrize("name", ["pic_1.png", "pic_2.png"])
def test_gray_threshold_photos(ctx, name):
    bgr = _data.load_bgr(name)
    gray, binp = ctx.gray_threshold(bgr)
    og = orc.bgr2gray(bgr)
    assert np.array_equal(gray[0], og)
    for s, ob in enumerate(_oracle_bin(og)):
        got = ((binp[0] >> s) & 1) * 255
        assert np.array_equal(got, ob), f"scale {s}: {np.count_nonzero(got != ob)} pixels differ"


def test_gray_threshold_batch_and_gray_input(ctx, synth_frames):
    batch = np.stack(synth_frames)
    gray, binp = ctx.gray_threshold(batch)
    for b in range(len(batch)):
        og = orc.bgr2gray(batch[b])
        assert np.array_equal(gray[b], og)
        for s, ob in enumerate(_oracle_bin(og)):
            assert np.array_equal(((binp[b] >> s) & 1) * 255, ob)
    # single-channel input goes through unchanged
    g2, b2 = ctx.gray_threshold(gray[0])
    assert np.array_equal(g2[0], gray[0]) and np.array_equal(b2[0], binp[0])


def test_gray_threshold_odd_sizes(ctx):
    rng = np.random.default_rng(5)
    for (h, w) in [(33, 40), (100, 136), (479, 648), (720, 1280)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        # smooth it a little so the threshold is not trivially all-zero
        img[:, : w // 2] //= 2
        gray, binp = ctx.gray_threshold(img)
        og = orc.bgr2gray(img)
        assert np.array_equal(gray[0], og), (h, w)
        for s, ob in enumerate(_oracle_bin(og)):
            assert np.array_equal(((binp[0] >> s) & 1) * 255, ob), (h, w, s)


def _oracle_contours(gray, h, w):
    minp, maxp = int(0.03 * max(h, w)), int(4.0 * max(h, w))
    out = set()
    for s, ob in enumerate(_oracle_bin(gray)):
        for c in orc.find_contours(ob):
            if minp <= len(c) <= maxp:
                out.add((s, int(c[0][1]) * w + int(c[0][0]), len(c)))
    return out


@pytest.mark.parametrize("name", ["pic_1.png", "pic_2.png"])
def test_contours_match_sequential_border_following(ctx, name):
    bgr = _data.load_bgr(name)
    h, w = bgr.shape[:2]
    ctx.detect(bgr)
    d = ctx.debug_contours(0, cap=65536)
    got = set(zip(d["scale"].tolist(), d["start_key"].tolist(), d["length"].tolist()))
    want = _oracle_contours(orc.bgr2gray(bgr), h, w)
    assert got == want, f"missing {len(want - got)} extra {len(got - want)}"


@pytest.mark.parametrize("name", ["pic_1.png", "pic_2.png"])
def test_contour_points_bit_exact(ctx, name):
    """Every kept contour has exactly OpenCV's point list (order and start included), twice in a row
    (the second call must not see anything left over from the first)."""
    bgr = _data.load_bgr(name)
    h, w = bgr.shape[:2]
    minp, maxp = int(0.03 * max(h, w)), int(4.0 * max(h, w))
    want = {}
    for s, ob in enumerate(_oracle_bin(orc.bgr2gray(bgr))):
        for c in orc.find_contours(ob):
            if minp <= len(c) <= maxp:
                want[(s, int(c[0][1]) * w + int(c[0][0]), len(c))] = c
    for _ in range(2):
        ctx.detect(bgr)
        d = ctx.debug_contours(0, cap=65536)
        pts = ctx.debug_points(0)
        assert len(pts) == int(d["length"].sum())
        off = 0
        bad = []
        for s, k, n in zip(d["scale"].tolist(), d["start_key"].tolist(), d["length"].tolist()):
            if not np.array_equal(pts[off:off + n], want[(s, k, n)]):
                bad.append((s, k, n, int(np.argmax((pts[off:off + n] != want[(s, k, n)]).any(axis=1)))))
            off += n
        assert not bad, bad[:10]


def _reorder(q):
    q = q.copy()
    d1 = q[1] - q[0]
    d2 = q[2] - q[0]
    if float(d1[0]) * float(d2[1]) - float(d1[1]) * float(d2[0]) < 0:
        q[[1, 3]] = q[[3, 1]]
    return q


def _oracle_quads(gray):
    out = set()
    for s, ob in enumerate(_oracle_bin(gray)):
        quads, key, clen = orc.quad_candidates(ob)
        for q, n in zip(quads, clen):
            out.add((s, int(n)) + tuple(int(v) for v in _reorder(q).ravel()))
    return out


@pytest.mark.parametrize("name", ["pic_1.png", "pic_2.png"])
def test_quads_bit_exact(ctx, name):
    bgr = _data.load_bgr(name)
    plain = ctx.detect(bgr)
    ctx.debug_identify_all(True)  # (the pipeline identifies only the candidates the selection can ask for)
    try:
        everything = ctx.detect(bgr)
    finally:
        ctx.debug_identify_all(False)
    for a, b in zip(plain, everything):
        assert np.array_equal(a, b)  # identifying all candidates changes nothing in the outputs
    d = ctx.debug_quads(0)
    got = set((int(s), int(n)) + tuple(int(v) for v in c.ravel())
              for s, n, c in zip(d["scale"], d["contour_len"], d["corners"]))
    want = _oracle_quads(orc.bgr2gray(bgr))
    assert got == want, f"missing {sorted(want - got)[:3]} extra {sorted(got - want)[:3]}"
    # identification of every quad candidate agrees with the oracle
    gray = orc.bgr2gray(bgr)
    for c, i, r in zip(d["corners"], d["id"], d["rot"]):
        ok, oid, orot, _ = orc.identify(gray, c)
        assert (int(i) >= 0) == ok
        if ok:
            assert (int(i), int(r)) == (oid, orot)


@pytest.mark.parametrize("name", ["pic_1", "pic_2"])
def test_detect_reference_goldens(ctx, name):
    """The reference's own known-answer test (camera_sync/test_aruco.py:203-326)."""
    bgr = _data.load_bgr(name + ".png")
    ids, corners, counts, rej, rejc = ctx.detect(bgr)
    n = int(counts[0])
    G = _data.goldens()[name]
    assert sorted(ids[0, :n].tolist()) == list(range(10, 16))
    for i in range(n):
        assert np.allclose(corners[0, i], np.array(G[str(int(ids[0, i]))]), atol=1e-3)
    # and the oracle, same order, tighter
    oids, ocor, orej = orc.detect(bgr)
    assert ids[0, :n].tolist() == oids.tolist()
    assert np.abs(corners[0, :n] - ocor).max() <= 1e-3
    assert int(rejc[0]) == len(orej)
    assert np.array_equal(rej[0, :len(orej)], orej)


def test_detect_synthetic_vs_oracle(ctx, synth_frames):
    batch = np.stack(synth_frames)
    ids, corners, counts, rej, rejc = ctx.detect(batch)
    worst = 0.0
    for b in range(len(batch)):
        oids, ocor, orej = orc.detect(batch[b])
        n = int(counts[b])
        assert ids[b, :n].tolist() == oids.tolist()
        worst = max(worst, float(np.abs(corners[b, :n] - ocor).max()))
        assert int(rejc[b]) == len(orej)
        assert np.array_equal(rej[b, :len(orej)], orej)
    assert worst <= 1e-3, worst


def test_pnp_reference_golden_pose(ctx):
    """Golden corners -> golden pose (camera_sync/test_aruco.py:345-353), atol 1e-4."""
    G = _data.goldens()
    K, dist = _data.calibration(G["pic_2_pose"]["camera"])
    anchors = G["pic_2_pose"]["anchors"]
    obj = _data.v1_anchor_points(anchors).reshape(1, 16, 3)
    img = np.array([G["pic_2"][str(i)] for i in anchors], np.float32).reshape(1, 16, 2)
    rvec, tvec, ok, mae, rmse = ctx.pnp(obj, img, [16], K, dist)
    assert ok[0] == 1
    assert np.allclose(rvec[0], G["pic_2_pose"]["rvec"], atol=1e-4)
    assert np.allclose(tvec[0], G["pic_2_pose"]["tvec"], atol=1e-4)
    assert mae[0] < 0.6 and rmse[0] < 0.8
    o_ok, orv, otv = orc.solve_pnp(obj[0], img[0], K, dist)
    assert np.allclose(rvec[0], orv, rtol=1e-6, atol=1e-9) and np.allclose(tvec[0], otv, rtol=1e-6, atol=1e-9)


def test_detect_pnp_fused_reference_pose(ctx):
    """processAruco on pic_2 (camera_sync/test_aruco.py:329-359) through the fused entry point."""
    G = _data.goldens()
    K, dist = _data.calibration(G["pic_2_pose"]["camera"])
    anchors = G["pic_2_pose"]["anchors"]
    bgr = _data.load_bgr("pic_2.png")
    ids, corners, counts, rvec, tvec, ok, mae, rmse = ctx.detect_pnp(bgr, anchors, _data.v1_anchor_points(anchors), K, dist)
    assert ok[0] == 1
    assert np.allclose(rvec[0], G["pic_2_pose"]["rvec"], atol=1e-4)
    assert np.allclose(tvec[0], G["pic_2_pose"]["tvec"], atol=1e-4)
    assert mae[0] < 0.6 and rmse[0] < 0.8


def test_pnp_batch_random_poses_vs_oracle(ctx):
    from cameracalibration_amd import synth
    K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
    rng = np.random.default_rng(11)
    B = 64
    obj = np.broadcast_to(_data.v1_anchor_points().reshape(1, 16, 3), (B, 16, 3)).copy()
    img = np.zeros((B, 16, 2), np.float32)
    for b in range(B):
        R, t = synth.look_at_pose(rng, (95.0, -135.0))
        rv = orc.rodrigues_mat2vec(R)
        img[b] = (orc.project_points(obj[b], rv, t, K, dist) + rng.normal(0, 0.3, (16, 2))).astype(np.float32)
    npts = np.full(B, 16, np.int32)
    npts[5] = 8  # ragged: only the first two anchors seen
    npts[6] = 0  # nothing seen
    rvec, tvec, ok, mae, rmse = ctx.pnp(obj, img, npts, K, dist)
    assert ok[6] == 0
    for b in range(B):
        if b == 6:
            continue
        o_ok, orv, otv = orc.solve_pnp(obj[b, :npts[b]], img[b, :npts[b]], K, dist)
        assert ok[b] == 1 and o_ok
        assert np.allclose(rvec[b], orv, rtol=1e-6, atol=1e-8), (b, rvec[b], orv)
        assert np.allclose(tvec[b], otv, rtol=1e-6, atol=1e-6), (b, tvec[b], otv)


@pytest.mark.parametrize("n_markers", [25, 50])
def test_pnp_more_than_64_points_vs_oracle(ctx, n_markers):
    """cv.solvePnP has no point limit (aruco.py:368): the reference's v2 sheet as anchors is 25 markers = 100 points
    (two points per lane), the whole dictionary 50 markers = 200 points (four per lane)."""
    from cameracalibration_amd import synth
    from cameracalibration_amd.camera_sync.aruco import getArucosFromPaper
    K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
    rng = np.random.default_rng(23)
    sheet = list(getArucosFromPaper(2).values())
    pts = np.array([a.getCornersAsList() for a in sheet], np.float32).reshape(-1, 3)   # 100 points of the v2 sheet
    if n_markers == 50:
        pts = np.concatenate([pts, pts + np.float32([200.0, 0.0, 0.0])])               # a second sheet beside it
    N = len(pts)
    assert N == 4 * n_markers
    B = 12
    obj = np.broadcast_to(pts.reshape(1, N, 3), (B, N, 3)).copy()
    img = np.zeros((B, N, 2), np.float32)
    centre = tuple(pts[:, :2].mean(axis=0))
    for b in range(B):
        R, t = synth.look_at_pose(rng, centre)
        rv = orc.rodrigues_mat2vec(R)
        img[b] = (orc.project_points(obj[b], rv, t, K, dist) + rng.normal(0, 0.3, (N, 2))).astype(np.float32)
    npts = np.full(B, N, np.int32)
    npts[3] = 68   # ragged: a little more than one point per lane
    npts[4] = 64
    rvec, tvec, ok, mae, rmse = ctx.pnp(obj, img, npts, K, dist)
    for b in range(B):
        o_ok, orv, otv = orc.solve_pnp(obj[b, :npts[b]], img[b, :npts[b]], K, dist)
        assert ok[b] == 1 and o_ok
        assert np.allclose(rvec[b], orv, rtol=1e-6, atol=1e-8), (b, rvec[b], orv)
        assert np.allclose(tvec[b], otv, rtol=1e-6, atol=1e-6), (b, tvec[b], otv)
        pr = orc.project_points(obj[b, :npts[b]], rvec[b], tvec[b], K, dist)
        d = np.sqrt(((pr - img[b, :npts[b]]) ** 2).sum(axis=1))
        assert abs(mae[b] - d.mean()) < 1e-6 and abs(rmse[b] - np.sqrt((d ** 2).mean())) < 1e-6


@pytest.mark.parametrize("N", [6, 24, 100])
def test_pnp_non_coplanar_points_vs_oracle(ctx, N):
    """cv.solvePnP(ITERATIVE) on a non-planar object starts from a DLT of the 3x4 projection instead of a homography
    (OpenCV's cvFindExtrinsicCameraParams2; aruco.py:368 passes whatever the caller's markers span, e.g. markers that
    were located in 3-D).  Fewer than 6 non-coplanar points are refused like OpenCV refuses them."""
    from cameracalibration_amd import synth
    K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
    rng = np.random.default_rng(77 + N)
    B = 8
    obj = rng.uniform(-120.0, 120.0, (B, N, 3)).astype(np.float32)
    obj[:, :, 2] *= 0.5
    img = np.zeros((B, N, 2), np.float32)
    for b in range(B):
        R, t = synth.look_at_pose(rng, (0.0, 0.0), dist_range=(500.0, 900.0))
        rv = orc.rodrigues_mat2vec(R)
        img[b] = (orc.project_points(obj[b], rv, t, K, dist) + rng.normal(0, 0.3, (N, 2))).astype(np.float32)
    npts = np.full(B, N, np.int32)
    rvec, tvec, ok, mae, rmse = ctx.pnp(obj, img, npts, K, dist)
    for b in range(B):
        o_ok, orv, otv = orc.solve_pnp(obj[b], img[b], K, dist)
        assert ok[b] == 1 and o_ok, (b, ok[b], o_ok)
        assert np.allclose(rvec[b], orv, rtol=1e-6, atol=1e-7), (b, rvec[b], orv)
        assert np.allclose(tvec[b], otv, rtol=1e-6, atol=1e-5), (b, tvec[b], otv)
        assert mae[b] < 1.5
    # five non-coplanar points: no initialisation exists
    _, _, ok5, _, _ = ctx.pnp(obj[:1, :5], img[:1, :5], np.array([5], np.int32), K, dist)
    assert ok5[0] == 2 and not orc.solve_pnp(obj[0, :5], img[0, :5], K, dist)[0]


def test_ippe_square_vs_oracle(ctx):
    G = _data.goldens()
    K, dist = _data.calibration("Logitec_A_1280-720_0.npz")
    corners = np.array([G["pic_2"][str(i)] for i in (10, 11, 12, 13, 14, 15)], np.float32)
    rvec, tvec, ok = ctx.ippe_square(corners, 80.0, K, dist)
    for i in range(len(corners)):
        o_ok, orv, otv = orc.ippe_square(corners[i], 80.0, K, dist)
        assert ok[i] == 1 and o_ok
        assert np.allclose(rvec[i], orv, rtol=1e-6, atol=1e-8)
        assert np.allclose(tvec[i], otv, rtol=1e-6, atol=1e-6)
        # the marker must reproject onto its corners
        h = 40.0
        sq = np.array([[-h, h, 0], [h, h, 0], [h, -h, 0], [-h, -h, 0]])
        pr = orc.project_points(sq, rvec[i], tvec[i], K, dist)
        assert np.abs(pr - corners[i]).max() < 1.5


@pytest.mark.parametrize("rational", [False, True])
def test_calibrate_vs_oracle(ctx, rational):
    from cameracalibration_amd import synth
    obj, img, K0, d0 = synth.calib_views(V=24, seed=7, rational=rational)
    rms, K, dist, rv, tv, iters = ctx.calibrate(obj, img, (1920, 1080), rational=rational)
    o_rms, oK, odist, orv, otv, oit = orc.calibrate(obj, img, (1920, 1080), rational=rational)
    assert abs(rms - o_rms) <= 1e-6, (rms, o_rms)
    assert rms < 0.45
    assert np.allclose(K, oK, rtol=1e-5)
    assert abs(K[0, 0] - 1400) < 15 and abs(K[1, 1] - 1400) < 15


@pytest.mark.parametrize("V", [200, 2000])
def test_calibrate_vs_oracle_at_full_size(ctx, V):
    """BASELINE configs[3]'s real size (2000 views, rational model): 500 workgroups, so the record is summed over
    workgroups in two levels (k_calib.hip, kLmGroup) -- the path the small cases above never reach; V = 200 crosses the
    first level only.  Run twice on one context: the tickets of the reduction are reset by the kernel itself."""
    from cameracalibration_amd import synth
    obj, img, K0, d0 = synth.calib_views(V=V, seed=42, rational=True)
    o_rms, oK, odist, orv, otv, oit = orc.calibrate(obj, img, (1920, 1080), rational=True)
    first = None
    for rep in range(2):
        rms, K, dist, rv, tv, iters = ctx.calibrate(obj, img, (1920, 1080), rational=True)
        assert abs(rms - o_rms) <= 1e-6, (rep, rms, o_rms)
        assert abs(iters - oit) <= 2, (iters, oit)  # the loop ends when the step is down at rounding level
        assert np.allclose(K, oK, rtol=1e-5), (K, oK)
        assert np.allclose(dist, odist, rtol=1e-4, atol=1e-6), (dist, odist)
        assert np.allclose(rv, orv, rtol=1e-5, atol=1e-6) and np.allclose(tv, otv, rtol=1e-5, atol=1e-6)
        if first is None:
            first = (rms, K.copy(), dist.copy(), rv.copy(), tv.copy())
        else:  # fixed summation order: a repeat is the same to the bit
            assert rms == first[0] and np.array_equal(K, first[1]) and np.array_equal(dist, first[2])
            assert np.array_equal(rv, first[3]) and np.array_equal(tv, first[4])
    assert abs(K[0, 0] - 1400) < 3 and abs(K[1, 1] - 1400) < 3


@pytest.mark.parametrize("size", ["1080p", "4k"])
@pytest.mark.parametrize("channels", [3, 1])
def test_pipeline_threshold_planes_bit_exact(size, channels):
    """The kernel instances the detection pipeline (and bench.py) actually run -- k_gray_thresh3 with the width
    compiled in, 3-channel (MODE 1) and 1-channel (MODE 2) input, no gray image written -- compared bit for bit with
    the oracle's adaptiveThreshold at the three window sizes on full 1080p and 4K frames.  (camsync_gray_threshold,
    which the other threshold tests use, runs the run-time-parameter instance.)"""
    from cameracalibration_amd import _hip, synth
    if size == "1080p":
        K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
        frames = np.stack([synth.v1_frame(70 + i, K, ss=2, device="cuda")[0].cpu().numpy() for i in range(2)])
    else:
        K4 = np.array([[2800.0, 0, 1920.0], [0, 2800.0, 1080.0], [0, 0, 1.0]])
        frames = np.stack([synth.grid_frame(20 + i, K4, ss=2, device="cuda")[0].cpu().numpy() for i in range(2)])
    if channels == 1:
        frames = np.stack([orc.bgr2gray(f) for f in frames])[..., None]
    B, H, W = frames.shape[:3]
    c = _hip.Context(H, W, B)
    c.detect(frames)
    for b in range(B):
        gray = frames[b, :, :, 0] if channels == 1 else orc.bgr2gray(frames[b])
        for s, win in enumerate((3, 13, 23)):
            want = (orc.adaptive_threshold(gray, win) > 0).astype(np.uint8)
            want[0, :] = want[-1, :] = 0
            want[:, 0] = want[:, -1] = 0
            got = c.debug_planes(b, s, H, W)
            assert np.array_equal(got, want), (b, win, int((got != want).sum()))


def _start_rule(binimg):
    """Contour-start candidates of one threshold image by the rule k_contour_starts implements, restated pixel by
    pixel: an outer candidate is a foreground pixel with background to its west whose horizontal run has no
    foreground 8-adjacent in the row above anywhere along it, and that is not an isolated pixel; a hole candidate is
    a background pixel with foreground to its west and directly above all along its zero run (the start pixel is the
    pixel to its west).  Returns sorted (hole << 39 | y * W + x)."""
    b = (binimg > 0).astype(np.uint8)
    H, W = b.shape
    b[0, :] = b[-1, :] = 0
    b[:, 0] = b[:, -1] = 0
    p = np.pad(b, 1)
    up = p[:-2]                                   # rows y-1, padded columns
    above = up[:, :-2] | up[:, 1:-1] | up[:, 2:]  # NW | N | NE
    n_ = up[:, 1:-1]
    west = p[1:-1, :-2]
    east = p[1:-1, 2:]
    dn = p[2:]
    below = dn[:, :-2] | dn[:, 1:-1] | dn[:, 2:]
    out = []
    for y in range(H - 1):
        row = b[y]
        # runs of foreground
        d = np.diff(np.concatenate([[0], row, [0]]).astype(np.int8))
        s_, e_ = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
        if len(s_):
            ca = np.concatenate([[0], np.cumsum(above[y])])
            ok = (ca[e_] - ca[s_]) == 0
            iso = (e_ - s_ == 1) & (below[y][s_] == 0)
            for x in s_[ok & ~iso]:
                out.append(y * W + int(x))
        # zero runs that begin right of a foreground pixel
        z = 1 - row
        d = np.diff(np.concatenate([[0], z, [0]]).astype(np.int8))
        s_, e_ = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
        keep = s_ > 0
        s_, e_ = s_[keep], e_[keep]
        if len(s_):
            cz = np.concatenate([[0], np.cumsum(1 - n_[y])])
            ok = ((cz[e_] - cz[s_]) == 0)
            for x in s_[ok]:
                out.append((1 << 39) | (y * W + int(x) - 1))
    return np.sort(np.array(out, np.uint64))


@pytest.mark.parametrize("case", ["pic_2", "noise", "blocks_136", "synthetic_1080p", "grid_4k", "wide_4088"])
def test_contour_start_candidates_match_the_rule(case):
    """The candidate list of the start stage (before any border is followed) equals the pixel-by-pixel rule on every
    threshold scale: nothing missing (a lost contour) and nothing extra (wasted border following)."""
    from cameracalibration_amd import _hip, synth
    rng = np.random.default_rng(17)
    if case == "pic_2":
        frames = _data.load_bgr("pic_2.png")[None]
    elif case == "noise":
        frames = rng.integers(0, 256, (2, 96, 128, 3), dtype=np.uint8)
    elif case == "blocks_136":
        small = rng.integers(0, 2, (13, 17), dtype=np.uint8) * 200 + 20
        img = np.kron(small, np.ones((8, 8), np.uint8))[:100, :136]
        frames = np.clip(img.astype(np.int32) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)[None, :, :, None]
    elif case == "synthetic_1080p":
        K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
        frames = np.stack([synth.v1_frame(80 + i, K, ss=2, device="cuda")[0].cpu().numpy() for i in range(3)])
    elif case == "grid_4k":
        K4 = np.array([[2800.0, 0, 1920.0], [0, 2800.0, 1080.0], [0, 0, 1.0]])
        frames = synth.grid_frame(30, K4, ss=2, device="cuda")[0].cpu().numpy()[None]
    else:  # widest supported frame: 128 words per row, three column groups, a partly used last word
        small = rng.integers(0, 2, (8, 511), dtype=np.uint8) * 200 + 20
        img = np.kron(small, np.ones((8, 8), np.uint8))[:64, :4088]
        frames = np.clip(img.astype(np.int32) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)[None, :, :, None]
    B, H, W, C = frames.shape
    c = _hip.Context(H, W, B)
    got = c.debug_starts(frames)
    want = []
    for b in range(B):
        gray = frames[b, :, :, 0] if C == 1 else orc.bgr2gray(frames[b])
        for s, win in enumerate((3, 13, 23)):
            r = _start_rule(orc.adaptive_threshold(gray, win))
            want.append(r | np.uint64((b << 40) | (s << 36)))
    want = np.sort(np.concatenate(want))
    assert len(got) == len(want) and np.array_equal(got, want), (len(got), len(want))
