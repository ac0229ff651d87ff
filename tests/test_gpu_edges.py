"""Edge cases and the full-size configurations of BASELINE.json, through the C ABI (ctypes binding):
empty and degenerate frames, smallest and odd frame sizes, 4K frames with 40 markers (configs[2]),
batches that mix contents, capacity and argument errors."""
import numpy as np
import pytest

from cameracalibration_amd import _hip, synth
from tests import _data, _oracle as orc

pytestmark = pytest.mark.gpu


def _same_as_oracle(ctx, frames):
    ids, corners, counts, rej, rejc = ctx.detect(frames)
    for b in range(len(frames)):
        oi, oc, orj = orc.detect(frames[b])
        n = int(counts[b])
        assert ids[b, :n].tolist() == oi.tolist()
        if n:
            assert np.abs(corners[b, :n] - oc).max() <= 1e-3
        assert int(rejc[b]) == len(orj)
        if len(orj):
            assert np.array_equal(rej[b, :len(orj)], orj)  # rejected quads are integer corners, same order
    return counts


def test_empty_and_flat_frames():
    ctx = _hip.Context(96, 128, 4)
    ramp = np.tile(np.arange(128, dtype=np.uint8)[None, :, None] * 2, (96, 1, 3))
    frames = np.stack([np.zeros((96, 128, 3), np.uint8), np.full((96, 128, 3), 255, np.uint8),
                       np.full((96, 128, 3), 128, np.uint8), ramp])
    counts = _same_as_oracle(ctx, frames)
    assert counts.tolist() == [0, 0, 0, 0]


def test_white_noise_small_frame_matches_oracle():
    """Pure white noise: thousands of contours and hundreds of quad candidates in a 96x128 frame."""
    ctx = _hip.Context(96, 128, 1)
    noise = np.random.default_rng(0).integers(0, 256, (1, 96, 128, 3), dtype=np.uint8)
    _same_as_oracle(ctx, noise)


def _squares_frame(rows=30, cols=40):
    img = np.full((rows * 16, cols * 16), 220, np.uint8)
    for r in range(rows):
        for c in range(cols):
            img[r * 16 + 3:r * 16 + 13, c * 16 + 3:c * 16 + 13] = 30
    return img


def test_a_frame_of_1200_squares_equals_the_oracle():
    """1200 small dark squares give several thousand quad candidates in one frame (each is found at several threshold
    scales, as an outer and an inner border) -- more than the 1024 slots a context starts with, and more than the
    selection kernel holds in LDS.  cv.aruco.detectMarkers has no such limit (reference camera_sync/aruco.py:256-264):
    the per-frame lists grow, the chunk is run again, and ids, corners and the rejected candidates (more than the
    binding's default 1024 columns) are the oracle's, in its order."""
    ctx = _hip.Context(480, 640, 2)
    img = _squares_frame()
    frames = np.stack([img, np.full_like(img, 128)])[..., None]
    ids, corners, counts, rej, rejc = ctx.detect(frames)
    assert ctx.debug_regrown() >= 1
    oi, oc, orj = orc.detect(frames[0])
    assert len(orj) > 1024 and int(rejc[0]) == len(orj) and rej.shape[1] >= len(orj)
    assert np.array_equal(rej[0, :len(orj)], orj)
    assert ids[0, :counts[0]].tolist() == oi.tolist()
    assert int(counts[1]) == 0 and int(rejc[1]) == 0
    grown = ctx.debug_regrown()
    _same_as_oracle(ctx, frames)  # (and again on the grown context: nothing grows a second time)
    assert ctx.debug_regrown() == grown
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    small = synth.v1_frame(1, K, ss=2)[0].numpy()[300:540, 600:920].copy()
    ctx2 = _hip.Context(480, 640, 2)
    _same_as_oracle(ctx, small[None])  # the grown lists serve ordinary frames as before
    want, got = ctx2.detect(small[None]), ctx.detect(small[None])
    for a, b in zip(want[:3], got[:3]):
        assert np.array_equal(a, b)


def test_more_markers_in_a_frame_than_the_outputs_start_with():
    """A sheet of 96 markers in one frame: more than the 64 marker slots per frame a context starts with and more than
    the binding's default 64 output columns.  All of them come back, in the oracle's order."""
    rng = np.random.default_rng(5)
    from cameracalibration_amd._dict4x4_50 import marker_bits  # [4, 4] of 0 / 1, 1 = white
    img = np.full((8 * 56 + 8, 12 * 56 + 8), 235, np.uint8)
    for k in range(96):
        r, c = divmod(k, 12)
        cell = np.zeros((6, 6), np.uint8)
        cell[1:5, 1:5] = marker_bits(k % 50) * 255
        img[8 + r * 56:8 + r * 56 + 48, 8 + c * 56:8 + c * 56 + 48] = np.kron(cell, np.ones((8, 8), np.uint8))
    img = np.clip(img.astype(np.int32) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    ctx = _hip.Context(img.shape[0], img.shape[1], 1)
    counts = _same_as_oracle(ctx, img[None, ..., None])
    assert int(counts[0]) == 96
    assert ctx.debug_regrown() >= 1
    ids, corners, cnt, rv, tv, ok, mae, rmse = ctx.detect_pnp(
        img[None, ..., None], [0], np.array([[[0, 0, 0], [40, 0, 0], [40, 40, 0], [0, 40, 0]]], np.float32),
        np.array([[600.0, 0, 340], [0, 600.0, 228], [0, 0, 1]]), None)
    assert int(cnt[0]) == 96 and ids.shape[1] >= 96 and ok[0] == 1


@pytest.mark.parametrize("limit", [dict(points=2000), dict(points=4000), dict(contours=20), dict(tasks=100),
                                   dict(contours=37, tasks=61)])
def test_contour_list_overflow_is_loud_and_harmless(limit):
    """The contour, point and store-task lists are sized for thousands of frames, so one small frame can only fill
    them when their capacities are lowered (testing aid of the C ABI).  Every overflow must end in
    CAMSYNC_E_CAPACITY -- the passes after the border followers read every reserved entry, so entries that
    were reserved but not written must read as empty -- and leave the context usable."""
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    frame = synth.v1_frame(1, K, ss=2)[0].numpy()[300:540, 600:920].copy()
    ctx = _hip.Context(240, 320, 1)
    want = ctx.detect(frame)
    ctx.debug_limit_capacity(**limit)
    with pytest.raises(_hip.CamsyncError) as e:
        ctx.detect(frame)
    assert e.value.code == _hip.E_CAPACITY
    ctx.debug_restore_capacity()
    got = ctx.detect(frame)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("hw", [(32, 32), (33, 40), (47, 104), (240, 320), (481, 648)])
def test_small_and_odd_sizes(hw):
    """Smallest supported frame, heights that are not multiples of 32, widths that are multiples of 8 only."""
    h, w = hw
    rng = np.random.default_rng(h * 1000 + w)
    # a blocky random image has plenty of contours and quads at every size
    small = rng.integers(0, 2, ((h + 7) // 8, (w + 7) // 8), dtype=np.uint8) * 200 + 20
    img = np.kron(small, np.ones((8, 8), np.uint8))[:h, :w]
    img = np.clip(img.astype(np.int32) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
    ctx = _hip.Context(max(h, 32), max(w, 32), 2)
    _same_as_oracle(ctx, np.stack([img, img[::-1].copy()])[..., None])


def test_mixed_batch_and_context_reuse():
    """One context, calls of different sizes and batch sizes back to back (stale scratch must not leak)."""
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    ctx = _hip.Context(1080, 1920, 3)
    big = np.stack([synth.v1_frame(i, K, ss=2)[0].numpy() for i in range(3)])
    c1 = _same_as_oracle(ctx, big)
    assert (c1 >= 5).all()
    pic = _data.load_bgr("pic_2.png")
    _same_as_oracle(ctx, pic[None])
    _same_as_oracle(ctx, big[:1])
    # a batch larger than max_batch is processed in chunks
    ids5, _, counts5, _, _ = ctx.detect(np.concatenate([big, big[:2]]))
    assert counts5.tolist() == c1.tolist() + c1[:2].tolist()


K4 = np.array([[2800.0, 0, 1920.0], [0, 2800.0, 1080.0], [0, 0, 1.0]])


def _grid_frames(n, first=0):
    """n distinct frames of the 4K scene (rendered on the GPU: 20 s per frame on the CPU)."""
    frames, sheets = [], []
    for i in range(first, first + n):
        img, sheet, _, _ = synth.grid_frame(i, K4, ss=2, device="cuda")
        frames.append(img.cpu().numpy())
        sheets.append(sheet)
    return np.stack(frames), sheets


def test_4k_forty_markers_config():
    """BASELINE configs[2]: 3840x2160, 40 markers per frame; every rendered marker is found, nothing else is,
    and the result equals the oracle's (ids and order exactly, corners to 1e-3 px)."""
    frames, sheets = _grid_frames(1)
    ctx = _hip.Context(2160, 3840, 1)
    counts = _same_as_oracle(ctx, frames)
    ids, _, _, _, _ = ctx.detect(frames)
    assert int(counts[0]) == 40
    assert sorted(ids[0, :40].tolist()) == sorted(int(m[0]) for m in sheets[0])


def test_4k_multi_frame_batch_matches_oracle():
    """configs[2] as a batch: 8 distinct 4K frames in one launch sequence (ids, order and rejected quads exactly,
    corners to 1e-3 px, per frame, against the oracle), then the same batch through the resident entry point with
    fewer output columns than markers (counts are clamped to the columns)."""
    import torch
    frames, sheets = _grid_frames(8, first=1)
    ctx = _hip.Context(2160, 3840, 8)
    counts = _same_as_oracle(ctx, frames)
    assert counts.tolist() == [40] * 8
    ids, corners, _, _, _ = ctx.detect(frames)
    for b in range(8):
        assert sorted(ids[b, :40].tolist()) == sorted(int(m[0]) for m in sheets[b])
    lists = ctx.debug_lists()
    assert all(u <= k for u, k in lists.values()), lists
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev)
    B, MM = 8, 16
    d_ids = torch.full((B, MM), -7, dtype=torch.int32, device=dev)
    d_cor = torch.zeros((B, MM, 4, 2), dtype=torch.float32, device=dev)
    d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_rv = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    d_tv = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    d_ok = torch.zeros((B,), dtype=torch.uint8, device=dev)
    anchors = [int(m[0]) for m in sheets[0][:4]]
    obj = np.array([[[x, y, 0], [x + sz, y, 0], [x + sz, y - sz, 0], [x, y - sz, 0]] for (_, x, y, sz) in sheets[0][:4]],
                   np.float32)
    torch.cuda.synchronize()
    ctx.detect_pnp_resident(d_frames.data_ptr(), B, 2160, 3840, 3, anchors, obj, K4, np.zeros(5), MM, d_ids.data_ptr(),
                            d_cor.data_ptr(), d_cnt.data_ptr(), d_rv.data_ptr(), d_tv.data_ptr(), d_ok.data_ptr())
    torch.cuda.synchronize()
    assert d_cnt.cpu().tolist() == [MM] * B                      # 40 markers found, 16 columns given
    assert np.array_equal(d_ids.cpu().numpy(), ids[:, :MM])
    assert np.array_equal(d_cor.cpu().numpy(), corners[:, :MM])


def test_resident_batch_beyond_4gb_equals_host_calls():
    """`camsync_detect_pnp_resident` (the bench's entry point) on 720 resident 1080p frames = 4.5 GB of input,
    8 distinct frames repeated: every frame's ids, corners and pose must equal what the host-buffer entry point
    gives for that frame alone, wherever the frame lies (byte offsets beyond 2^31 and 2^32 included)."""
    import torch
    K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
    H, W, B, MM = 1080, 1920, 720, 16
    base = np.stack([synth.v1_frame(40 + i, K, ss=2)[0].numpy() for i in range(8)])
    anchors = [10, 11, 14, 15]
    obj = _data.v1_anchor_points(anchors)
    ctx = _hip.Context(H, W, max_batch=B)
    ids0, cor0, cnt0, rv0, tv0, ok0, _, _ = ctx.detect_pnp(base, anchors, obj, K, dist, max_markers=MM)
    assert (cnt0 >= 5).all() and (ok0 == 1).all()
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(base).to(dev).repeat(B // 8, 1, 1, 1).contiguous()
    assert frames.numel() > (1 << 32)
    d_ids = torch.full((B, MM), -1, dtype=torch.int32, device=dev)
    d_cor = torch.zeros((B, MM, 4, 2), dtype=torch.float32, device=dev)
    d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_rv = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    d_tv = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    d_ok = torch.zeros((B,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.detect_pnp_resident(frames.data_ptr(), B, H, W, 3, anchors, obj, K, dist, MM, d_ids.data_ptr(), d_cor.data_ptr(),
                            d_cnt.data_ptr(), d_rv.data_ptr(), d_tv.data_ptr(), d_ok.data_ptr())
    torch.cuda.synchronize()
    rep = B // 8
    assert np.array_equal(d_cnt.cpu().numpy(), np.tile(cnt0, rep))
    used = np.arange(MM)[None, :] < np.tile(cnt0, rep)[:, None]  # entries beyond a frame's count are unspecified
    assert np.array_equal(d_ids.cpu().numpy()[used], np.tile(ids0, (rep, 1))[used])
    assert np.array_equal(d_cor.cpu().numpy()[used], np.tile(cor0, (rep, 1, 1, 1))[used])
    assert np.array_equal(d_ok.cpu().numpy(), np.tile(ok0, rep))
    assert np.array_equal(d_rv.cpu().numpy(), np.tile(rv0, (rep, 1)))
    assert np.array_equal(d_tv.cpu().numpy(), np.tile(tv0, (rep, 1)))


def test_argument_errors_and_limits():
    ctx = _hip.Context(64, 64, 1)
    with pytest.raises(_hip.CamsyncError):
        ctx.detect(np.zeros((128, 128, 3), np.uint8))       # larger than the context
    with pytest.raises(_hip.CamsyncError):
        ctx.detect(np.zeros((64, 60, 3), np.uint8))         # width not a multiple of 8
    with pytest.raises(_hip.CamsyncError):
        _hip.Context(16, 16, 1)                             # below the minimum size
    p = ctx.get_params()
    p.maxMarkerPerimeterRate = 400.0                        # would need contours of 25600 points
    ctx.set_params(p)
    with pytest.raises(_hip.CamsyncError):
        ctx.detect(np.zeros((64, 64, 3), np.uint8))
    p.maxMarkerPerimeterRate = 4.0
    ctx.set_params(p)
    ctx.detect(np.zeros((64, 64, 3), np.uint8))
    # PnP: fewer than 4 points is not solvable; 8 points that are not coplanar are solved (DLT start), 5 are refused (status 2)
    K = np.array([[500.0, 0, 32], [0, 500.0, 32], [0, 0, 1]])
    obj = np.zeros((1, 8, 3), np.float32); img = np.zeros((1, 8, 2), np.float32)
    _, _, ok, _, _ = ctx.pnp(obj, img, [3], K, np.zeros(5))
    assert ok[0] == 0
    obj[0] = np.random.default_rng(1).normal(size=(8, 3)); obj[0, :, 2] += 5
    img[0] = (obj[0, :, :2] / obj[0, :, 2:3]) * 500 + 32
    rv, tv, ok, _, _ = ctx.pnp(obj, img, [8], K, np.zeros(5))
    assert ok[0] == 1 and np.allclose(rv[0], 0, atol=1e-6) and np.allclose(tv[0], 0, atol=1e-5)  # the points were projected by the identity pose
    _, _, ok, _, _ = ctx.pnp(obj, img, [5], K, np.zeros(5))
    assert ok[0] == 2


def test_generic_threshold_windows_match_oracle():
    """Detector parameters other than the reference's take the generic (untuned) threshold kernels."""
    ctx = _hip.Context(240, 320, 1)
    p = ctx.get_params()
    p.adaptiveThreshWinSizeMin, p.adaptiveThreshWinSizeMax, p.adaptiveThreshWinSizeStep = 5, 21, 8
    p.adaptiveThreshConstant = 5.0
    ctx.set_params(p)
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    frame = synth.v1_frame(1, K, ss=2)[0].numpy()[300:540, 600:920].copy()
    op = orc.default_params()
    op.adaptiveThreshWinSizeMin, op.adaptiveThreshWinSizeMax, op.adaptiveThreshWinSizeStep = 5, 21, 8
    op.adaptiveThreshConstant = 5.0
    ids, corners, counts, _, _ = ctx.detect(frame)
    oi, oc, _ = orc.detect(frame, params=op)
    n = int(counts[0])
    assert ids[0, :n].tolist() == oi.tolist()
    if n:
        assert np.abs(corners[0, :n] - oc).max() <= 1e-3


def test_page_locked_frame_buffers_give_the_same_result():
    """camsync_host_alloc (frame ingest): frames handed over in page-locked memory are the same frames."""
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    frames = np.stack([synth.v1_frame(i, K, ss=2)[0].numpy() for i in range(2)])
    ctx = _hip.Context(1080, 1920, 2)
    want = ctx.detect(frames)
    pinned = _hip.PinnedFrames(frames.shape)
    pinned.array[...] = frames
    got = ctx.detect(pinned.array)
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    pinned.close()


def test_host_frames_larger_than_the_upload_buffer_arrive_intact():
    """Frames in ordinary host memory are staged through a 64 MB page-locked buffer by several threads (upload_host): a
    batch of 75 MB goes round that buffer, one of 12 MB takes the small pieces, separate frames arrive as a batch does
    -- all give what the same frames give from page-locked memory, which is sent as it is."""
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    base = [synth.v1_frame(40 + i, K, ss=2)[0].numpy() for i in range(3)]
    rng = np.random.default_rng(5)
    frames = np.stack([np.clip(base[i % 3].astype(np.int16) + rng.integers(-3, 4, base[0].shape), 0, 255).astype(np.uint8)
                       for i in range(12)])
    assert frames.nbytes > 64 << 20
    ctx = _hip.Context(1080, 1920, 12)
    pinned = _hip.PinnedFrames(frames.shape)
    pinned.array[...] = frames
    want = ctx.detect(pinned.array)
    assert int(want[2].min()) >= 4
    for got in (ctx.detect(frames), ctx.detect(frames)):
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
    two = ctx.detect(frames[5:7])
    for a, b in zip(want, two):
        assert np.array_equal(a[5:7], b)
    anchors = [10, 11, 14, 15]
    obj = _data.v1_anchor_points(anchors)
    sep = ctx.detect_pnp_frames([frames[i] for i in range(12)], anchors, obj, K, np.zeros(5), 64)
    assert np.array_equal(sep[0], want[0]) and np.array_equal(sep[1], want[1]) and np.array_equal(sep[2], want[2])
    ref = ctx.detect_pnp(pinned.array, anchors, obj, K, np.zeros(5), 64)
    assert ref[5].any()
    for a, b in zip(ref, sep):
        assert np.array_equal(a, b)
    pinned.close()
    # many small frames handed over one by one: several of them share a page-locked piece on their way
    small = [np.ascontiguousarray(frames[i % 12, 200:200 + 96, 300 + (i % 7) * 8:300 + (i % 7) * 8 + 160]) for i in range(420)]
    ctx2 = _hip.Context(96, 160, 420)
    a = ctx2.detect_pnp_frames(small, anchors, obj, K, np.zeros(5), 16)
    b = ctx2.detect_pnp(np.stack(small), anchors, obj, K, np.zeros(5), 16)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_fall_back_candidate_when_the_representative_is_no_marker():
    """A thin black ring close around a marker: the ring's outer quad is the largest of the group of near-duplicates
    (so it is the one selected) and is no marker; the detector then falls back to the marker's own quad.  The pipeline
    identifies fall-backs only in that case (second identification pass): same result as identifying everything, and
    as the oracle."""
    from cameracalibration_amd._dict4x4_50 import CODES
    img = np.full((480, 640), 235, np.uint8)
    cell, x0, y0, mid = 20, 240, 160, 23
    bits = [(int(CODES[mid]) >> (15 - k)) & 1 for k in range(16)]
    img[y0:y0 + 6 * cell, x0:x0 + 6 * cell] = 20
    for k, bit in enumerate(bits):
        if bit:
            r, c = divmod(k, 4)
            img[y0 + (1 + r) * cell:y0 + (2 + r) * cell, x0 + (1 + c) * cell:x0 + (2 + c) * cell] = 235
    gap, t = 14, 5  # white gap, then a ring t pixels thick
    o = gap + t
    ring = np.zeros_like(img, bool)
    ring[y0 - o:y0 + 6 * cell + o, x0 - o:x0 + 6 * cell + o] = True
    ring[y0 - gap:y0 + 6 * cell + gap, x0 - gap:x0 + 6 * cell + gap] = False
    img[ring] = 20
    bgr = np.repeat(img[:, :, None], 3, axis=2)
    ctx = _hip.Context(480, 640, 1)
    plain = ctx.detect(bgr)
    ctx.debug_identify_all(True)
    try:
        everything = ctx.detect(bgr)
        d = ctx.debug_quads(0)
    finally:
        ctx.debug_identify_all(False)
    for a, b in zip(plain, everything):
        assert np.array_equal(a, b)
    n = int(plain[2][0])
    assert plain[0][0, :n].tolist() == [mid]
    _same_as_oracle(ctx, bgr[None])
    # the scene does what it is meant to: an unidentified quad larger than, and close to, the identified one
    per = lambda c: float(np.linalg.norm(c - np.roll(c, -1, axis=0), axis=1).sum())
    valid = [c for c, i in zip(d["corners"], d["id"]) if int(i) == mid]
    assert valid
    v = max(valid, key=per)
    close_bigger = [c for c, i in zip(d["corners"], d["id"]) if int(i) < 0 and per(c) > per(v)
                    and np.linalg.norm(c - v, axis=1).mean() < 0.125 * per(v)]
    assert close_bigger


@pytest.mark.parametrize("group", ["poses", "robot_calib", "poses_blender"])
def test_other_reference_photos_match_the_oracle(group):
    """The reference's other marker-sheet photos (pics/poses, pics/robot_calib: 640x480 JPEGs; pics/blender/poses:
    1920x1080 renders), each set as one batch: ids and order exact, rejected quads exact, corners <= 1e-3 px against the
    oracle; and every frame yields the markers of the sheet it shows (tests/test_oracle_goldens.py::EXPECTED_IDS)."""
    from tests.test_oracle_goldens import EXPECTED_IDS
    names = _data.photo_sets()[group]
    frames = np.stack([_data.load_bgr(n) for n in names])
    ctx = _hip.Context(frames.shape[1], frames.shape[2], len(frames))
    ids, corners, counts, _, _ = ctx.detect(frames)
    for b, n in enumerate(names):
        assert sorted(ids[b, :int(counts[b])].tolist()) == EXPECTED_IDS[n], n
    _same_as_oracle(ctx, frames)
    # contours point for point on the busiest frame of the set (textured desk scenes: thousands of borders per scale)
    h, w = frames.shape[1:3]
    minp, maxp = int(0.03 * max(h, w)), int(4.0 * max(h, w))
    one = _hip.Context(h, w, 1)
    one.detect(frames[-1])
    d = one.debug_contours(0, cap=262144)
    pts = one.debug_points(0)
    want = {}
    gray = orc.bgr2gray(frames[-1])
    for s, win in enumerate((3, 13, 23)):
        for c in orc.find_contours(orc.adaptive_threshold(gray, win)):
            if minp <= len(c) <= maxp:
                want[(s, int(c[0][1]) * w + int(c[0][0]), len(c))] = c
    assert len(d["length"]) == len(want)
    off = 0
    for s, k, n in zip(d["scale"].tolist(), d["start_key"].tolist(), d["length"].tolist()):
        assert np.array_equal(pts[off:off + n], want[(s, k, n)]), (s, k, n)
        off += n


def test_overflowing_scratch_lists_grow_and_the_result_is_the_oracles():
    """cv.aruco.detectMarkers has no capacity limits (aruco.py:256-264): a chunk that overflows one of the detection's
    scratch lists is run again with that list doubled.  The lists are made tiny here (camsync_debug_shrink_lists), so
    the reference's desk photo needs several doublings of every one of them; the results still equal the oracle's, and
    the grown context keeps working (a second call needs no more doublings)."""
    bgr = _data.load_bgr("pic_2.png")
    K, d = _data.calibration("Logitec_A_1280-720_0.npz")
    ctx = _hip.Context(720, 1280, 2)
    ctx.debug_shrink_lists(starts=2048, contours=64, points=4096, tasks=64, seeds=256)
    assert ctx.debug_regrown() == 0
    _same_as_oracle(ctx, np.stack([bgr, bgr[::-1].copy()]))
    n = ctx.debug_regrown()
    assert n >= 5, n
    _same_as_oracle(ctx, bgr[None])
    assert ctx.debug_regrown() == n
    # the path that leaves its results on the device re-runs its chunk as well
    ctx2 = _hip.Context(720, 1280, 1)
    ctx2.debug_shrink_lists(starts=2048, contours=64, points=4096, tasks=64, seeds=256)
    anchors = [10, 11, 14, 15]
    r = ctx2.detect_pnp(bgr, anchors, _data.v1_anchor_points(anchors), K, d)
    G = _data.goldens()["pic_2_pose"]
    assert r[5][0] == 1 and np.allclose(r[3][0], G["rvec"], atol=1e-4) and np.allclose(r[4][0], G["tvec"], atol=1e-4)
    assert ctx2.debug_regrown() >= 5
