"""Compressed frame ingest (baseline JPEG / MJPEG frames).

CPU part: the oracle (oracle/jpeg.cpp, a restatement of libjpeg-turbo's default decoding path, which is what
cv.VideoCapture.read() on an MJPG camera and cv.imread give the reference, camera.py:98-118, 150-154) against Pillow,
which decodes with libjpeg-turbo too: bit for bit on the reference's own 21 JPEG photos and on re-encodings that cover
the samplings, odd sizes, optimised tables, restart intervals and MJPEG frames without Huffman tables.  The reference
holds the JPEG files but no decoded pixels, so this pins the oracle to libjpeg-turbo, not to a reference-held output.

GPU part: camsync_decode_jpeg_batch against the oracle, bit for bit."""
import io
import os

import numpy as np
import pytest
from PIL import Image

from tests import _data, _oracle as orc


def _pil_bgr(data):
    rgb = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _reference_jpegs():
    out = []
    for d in ("calib", "poses", "robot_calib"):
        folder = os.path.join(_data.GOLDEN, d)
        out += [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if f.endswith(".jpg")]
    return out


def _scene(h, w, seed):
    """Smooth gradients, sharp edges and noise: exercises long zero runs, large DC steps and every coefficient."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 3) % 256], -1).astype(np.int32)
    img[h // 4:h // 2, w // 4:w // 2] = (250, 10, 10)
    img[h // 2:, : w // 3] = rng.integers(0, 256, (h - h // 2, w // 3, 3))
    img += rng.integers(-8, 9, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def _encode(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def strip_dht(data):
    """The frame as an MJPEG camera sends it: without its DHT segments (the decoder then uses the Annex K tables, which
    are what Pillow writes unless asked to optimise)."""
    out = bytearray(data[:2])
    i = 2
    while data[i + 1] != 0xDA:
        L = (data[i + 2] << 8) | data[i + 3]
        if data[i + 1] != 0xC4:
            out += data[i:i + 2 + L]
        i += 2 + L
    return bytes(out + data[i:])


def third_tables(data):
    """The same image with the Cr component coded with Huffman tables 2 (copies of tables 1): a frame with three DC and
    three AC tables, which the decoder's wave-per-frame kernel leaves to the shared-wave one."""
    out = bytearray(data[:2])
    i = 2
    extra = bytearray()
    while data[i + 1] != 0xDA:
        L = (data[i + 2] << 8) | data[i + 3]
        seg = data[i:i + 2 + L]
        out += seg
        if data[i + 1] == 0xC4:
            k = 4
            while k < len(seg):
                n = sum(seg[k + 1:k + 17])
                if seg[k] & 15 == 1:
                    extra += bytes([0xFF, 0xC4, 0, 0]) + bytes([(seg[k] & 0xF0) | 2]) + seg[k + 1:k + 17 + n]
                    extra[-(n + 17 + 2):-(n + 17)] = (n + 19).to_bytes(2, "big")
                k += 17 + n
        i += 2 + L
    L = (data[i + 2] << 8) | data[i + 3]
    sos = bytearray(data[i:i + 2 + L])
    assert sos[4] == 3
    sos[5 + 2 * 2 + 1] = 0x22
    return bytes(out + extra + sos + data[i + 2 + L:])


VARIANTS = [
    dict(quality=90, subsampling=0), dict(quality=75, subsampling=1), dict(quality=60, subsampling=2),
    dict(quality=95, subsampling=2, optimize=True), dict(quality=30, subsampling=1, restart_marker_blocks=3),
    dict(quality=85, subsampling=2, restart_marker_rows=1), dict(quality=100, subsampling=0),
    dict(quality=5, subsampling=2),
]


def test_oracle_equals_libjpeg_turbo_on_the_reference_photos():
    files = _reference_jpegs()
    assert len(files) == 21
    for f in files:
        data = open(f, "rb").read()
        assert np.array_equal(orc.jpeg_decode(data), _pil_bgr(data)), f


@pytest.mark.parametrize("hw", [(48, 64), (50, 70), (33, 47), (480, 640), (8, 8), (17, 250)])
def test_oracle_equals_libjpeg_turbo_on_encodings(hw):
    img = _scene(hw[0], hw[1], hw[0] * 1000 + hw[1])
    for kw in VARIANTS:
        data = _encode(img, **kw)
        assert np.array_equal(orc.jpeg_decode(data), _pil_bgr(data)), kw
    gray = _encode(img[..., 0], quality=80)
    assert np.array_equal(orc.jpeg_decode(gray), _pil_bgr(gray))


def test_mjpeg_frames_without_huffman_tables():
    img = _scene(96, 128, 7)
    data = _encode(img, quality=80, subsampling=1)
    bare = strip_dht(data)
    assert len(bare) < len(data) and b"\xff\xc4" not in bare[:600]
    assert np.array_equal(orc.jpeg_decode(bare), _pil_bgr(data))


def test_frames_with_three_huffman_tables():
    img = _scene(64, 96, 11)
    for kw in (dict(quality=80, subsampling=0), dict(quality=70, subsampling=2, optimize=True)):
        data = _encode(img, **kw)
        three = third_tables(data)
        assert three != data and np.array_equal(_pil_bgr(three), _pil_bgr(data))
        assert np.array_equal(orc.jpeg_decode(three), _pil_bgr(data))


def test_what_the_decoder_does_not_take():
    img = _scene(48, 64, 3)
    with pytest.raises(ValueError):
        orc.jpeg_decode(_encode(img, progressive=True))
    with pytest.raises(ValueError):
        orc.jpeg_decode(b"\x89PNG....")
    with pytest.raises(ValueError):
        orc.jpeg_decode(_encode(img, quality=80)[:200])


# ---- GPU: camsync_decode_jpeg_batch against the oracle -----------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_decodes_the_reference_photos_like_the_oracle():
    from cameracalibration_amd import _hip
    datas = [open(f, "rb").read() for f in _reference_jpegs()]
    assert _hip.jpeg_info(datas[0]) == (480, 640, 3)
    ctx = _hip.Context(480, 640, 32)
    got = ctx.decode_jpeg(datas)
    assert got.shape == (21, 480, 640, 3)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], orc.jpeg_decode(d)), i
    # and what the decoded frames are for: the photos of the marker sheet give the oracle's detections
    again = ctx.decode_jpeg(datas[9:17])
    assert np.array_equal(again, got[9:17])


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(48, 64), (50, 70), (33, 47), (8, 8), (17, 250), (480, 640)])
def test_gpu_decodes_every_supported_kind_like_the_oracle(hw):
    """One batch that mixes samplings, table sets (standard, optimised, none), restart intervals and gray frames."""
    from cameracalibration_amd import _hip
    img = _scene(hw[0], hw[1], hw[0] * 1000 + hw[1])
    datas = [_encode(img, **kw) for kw in VARIANTS]
    datas.append(_encode(img[..., 0], quality=80))
    datas.append(strip_dht(_encode(img, quality=80, subsampling=1)))
    datas += [_encode(_scene(hw[0], hw[1], s), quality=70 + s, subsampling=s % 3) for s in range(70)]
    # frames with three Huffman tables per class (the shared-wave kernel), more than one wave of them
    datas += [third_tables(_encode(_scene(hw[0], hw[1], 100 + s), quality=60 + s, subsampling=s % 3)) for s in range(9)]
    # ... and with restart intervals as well: these go to the kernel that shares a wavefront among frames
    datas += [third_tables(_encode(_scene(hw[0], hw[1], 200 + s), quality=65 + s, subsampling=s % 3, restart_marker_blocks=3 + s))
              for s in range(5)]
    ctx = _hip.Context(max(hw[0], 32), max(hw[1], 32), 8)
    got = ctx.decode_jpeg(datas)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], orc.jpeg_decode(d)), i
    # frames left on the device
    import torch
    dev = torch.empty((len(datas),) + hw + (3,), dtype=torch.uint8, device="cuda:0")
    assert ctx.decode_jpeg(datas, d_bgr_ptr=dev.data_ptr()) == (len(datas),) + hw + (3,)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), got)


@pytest.mark.gpu
def test_gpu_frames_whose_speculative_decoders_do_not_fall_into_step():
    """The parallel entropy decoder relies on decoders that start anywhere in the stream falling into step with the real
    one.  In a flat frame every MCU is the same few bits, a decoder that starts out of phase stays out of phase, and the
    true states only spread one subsequence per round: such frames are handed to the sequential kernels after a
    bounded number of rounds.  Flat and striped frames, alone and mixed with ordinary ones, against the oracle."""
    from cameracalibration_amd import _hip
    h, w = 1080, 1920
    flat = np.full((h, w, 3), 97, np.uint8)
    stripes = np.zeros((h, w, 3), np.uint8)
    stripes[:, ::16] = 255
    half = _scene(h, w, 9)
    half[:, w // 2:] = 40
    datas = [_encode(flat, quality=50, subsampling=1), _encode(flat, quality=90, subsampling=2), _encode(flat, quality=75, subsampling=0),
             _encode(stripes, quality=80, subsampling=1), _encode(half, quality=85, subsampling=1), _encode(flat[..., 0], quality=60),
             strip_dht(_encode(_scene(h, w, 10), quality=85, subsampling=1))]
    ctx = _hip.Context(h, w, 8)
    got = ctx.decode_jpeg(datas)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], orc.jpeg_decode(d)), i
    one = ctx.decode_jpeg(datas[:1])
    assert np.array_equal(one[0], got[0])


@pytest.mark.gpu
def test_gpu_frames_with_restart_intervals_decode_in_parts():
    """Restart intervals start in a known state, so a frame's intervals are decoded by up to 256 wavefronts side by
    side (runs of whole intervals each): intervals of one MCU, of a few, of one and of several MCU rows, more intervals
    than wavefronts, the last interval short; 4:2:0, 4:2:2, 4:4:4 and gray."""
    from cameracalibration_amd import _hip
    h, w = 360, 648
    img = _scene(h, w, 77)
    datas = [_encode(img, quality=80, subsampling=2, restart_marker_blocks=1), _encode(img, quality=70, subsampling=1, restart_marker_blocks=7),
             _encode(img, quality=90, subsampling=0, restart_marker_rows=1), _encode(img, quality=60, subsampling=2, restart_marker_rows=5),
             _encode(img[..., 1], quality=85, restart_marker_blocks=13), _encode(img, quality=75, subsampling=1, restart_marker_rows=100),
             _encode(img, quality=75, subsampling=1)]
    ctx = _hip.Context(h, w, 8)
    got = ctx.decode_jpeg(datas)
    for i, d in enumerate(datas):
        assert np.array_equal(got[i], orc.jpeg_decode(d)), i
    # a damaged interval is reported, the others do not hide it
    bad = bytearray(datas[1])
    at = bad.index(b"\xff\xd3", bad.index(b"\xff\xda"))
    bad[at + 1] = 0xD5  # a restart marker out of sequence
    with pytest.raises(_hip.CamsyncError):
        ctx.decode_jpeg([bytes(bad)])
    assert np.array_equal(ctx.decode_jpeg(datas[:1])[0], got[0])


@pytest.mark.gpu
def test_gpu_refuses_what_the_decoder_does_not_take():
    from cameracalibration_amd import _hip
    img = _scene(48, 64, 3)
    ctx = _hip.Context(64, 64, 4)
    good = _encode(img, quality=80)
    sos = good.index(b"\xff\xda")
    empty_scan = good[:sos + 14] + b"\xff\xd9"  # headers, then the end of the image at once
    for bad in (_encode(img, progressive=True), good[:300], _encode(_scene(40, 64, 1), quality=80)):
        with pytest.raises(_hip.CamsyncError) as e:
            ctx.decode_jpeg([good, bad])
        assert e.value.code == _hip.E_ARG
    # entropy-coded data cut short or damaged: the marker / end of data feeds zeros (as libjpeg does) or the frame is
    # reported as corrupt; either way the call returns and the context stays usable
    for cut in (good[:len(good) // 2], empty_scan):
        try:
            ctx.decode_jpeg([cut])
        except _hip.CamsyncError as e2:
            assert e2.code == _hip.E_ARG
    assert np.array_equal(ctx.decode_jpeg([good])[0], orc.jpeg_decode(good))
    # garbled entropy-coded bytes (no new markers), through both Huffman kernels: codes that do not exist, runs past a
    # block's end, a stream that ends early or late
    rng = np.random.default_rng(11)
    sos = good.index(b"\xff\xda")
    for trial in range(8):
        b = bytearray(good if trial % 2 == 0 else third_tables(good))
        first = (sos if trial % 2 == 0 else bytes(b).index(b"\xff\xda")) + 14
        for at in rng.integers(first, len(b) - 2, 12):
            b[at] = int(rng.integers(0, 255))
        try:
            out = ctx.decode_jpeg([good, bytes(b)])
            assert np.array_equal(out[0], orc.jpeg_decode(good))
        except _hip.CamsyncError as e3:
            assert e3.code == _hip.E_ARG
    assert np.array_equal(ctx.decode_jpeg([good])[0], orc.jpeg_decode(good))


@pytest.mark.gpu
def test_gpu_compressed_frames_to_poses_without_leaving_the_device():
    """Ingest feeding the hot path: the reference's photos of the marker sheet (pics/poses, JPEG) are decoded into a
    device buffer and detection + anchor PnP run on that buffer; ids, corners and poses are those of the same call on
    the host copies of the decoded frames, and the ids are the oracle's on the oracle's decoding."""
    import torch
    from cameracalibration_amd import _hip
    files = _reference_jpegs()[9:17]
    datas = [open(f, "rb").read() for f in files]
    B, H, W = len(datas), 480, 640
    ctx = _hip.Context(H, W, B)
    dev = torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda:0")
    assert ctx.decode_jpeg(datas, d_bgr_ptr=dev.data_ptr()) == (B, H, W, 3)
    K, dist = _data.calibration("Logitec_A_1280-720_0.npz")
    anchors = [10, 11, 14, 15]
    obj = _data.v1_anchor_points(anchors)
    MM = 16
    d_ids = torch.full((B, MM), -1, dtype=torch.int32, device="cuda:0"); d_cor = torch.zeros((B, MM, 4, 2), device="cuda:0")
    d_cnt = torch.zeros((B,), dtype=torch.int32, device="cuda:0"); d_rv = torch.zeros((B, 3), dtype=torch.float64, device="cuda:0")
    d_tv = torch.zeros((B, 3), dtype=torch.float64, device="cuda:0"); d_ok = torch.zeros((B,), dtype=torch.uint8, device="cuda:0")
    ctx.detect_pnp_resident(dev.data_ptr(), B, H, W, 3, anchors, obj, K, dist, MM, d_ids.data_ptr(), d_cor.data_ptr(), d_cnt.data_ptr(),
                            d_rv.data_ptr(), d_tv.data_ptr(), d_ok.data_ptr())
    torch.cuda.synchronize()
    host = ctx.detect_pnp(dev.cpu().numpy(), anchors, obj, K, dist, MM)
    assert np.array_equal(d_ids.cpu().numpy(), host[0]) and np.array_equal(d_cor.cpu().numpy(), host[1])
    assert np.array_equal(d_cnt.cpu().numpy(), host[2]) and np.array_equal(d_ok.cpu().numpy(), host[5])
    assert np.array_equal(d_rv.cpu().numpy(), host[3]) and np.array_equal(d_tv.cpu().numpy(), host[4])
    for b in range(B):
        oi, _, _ = orc.detect(orc.jpeg_decode(datas[b]))
        n = int(host[2][b])
        assert sorted(host[0][b][:n].tolist()) == sorted(oi.tolist())
    assert int(host[2].min()) >= 4


@pytest.mark.gpu
def test_camera_takes_pictures_from_an_mjpeg_stream():
    """The mirror's Camera over a stream of MJPEG frames (what a camera in MJPG mode sends): takePic() yields the frames
    cv.VideoCapture.read() would (reference camera.py:150-154), here the reference's own photos of the marker sheet,
    and they carry the oracle's detections."""
    from cameracalibration_amd.camera_sync.camera import Camera
    from cameracalibration_amd.camera_sync.ingest import MJPEGStream, imread
    from cameracalibration_amd.camera_sync.aruco import detectAruco
    files = _reference_jpegs()[9:17]  # pics/poses
    stream = io.BytesIO(b"".join(b"\x00junk" + open(f, "rb").read() for f in files))  # (bytes between frames are skipped)
    cam = Camera("mjpeg", -1, captureStream=MJPEGStream(stream, chunk=4096))
    for f in files[:3]:
        frame = cam.takePic()
        want = orc.jpeg_decode(open(f, "rb").read())
        assert np.array_equal(frame, want)
        assert np.array_equal(imread(f), want)
    arucos = detectAruco(frame)[0]
    oi, _, _ = orc.detect(want)
    assert sorted(arucos.keys()) == sorted(oi.tolist()) and len(oi) >= 4
    rest = cam.captureStream.read_batch(16)
    assert rest.shape == (5, 480, 640, 3)
    assert np.array_equal(rest[4], orc.jpeg_decode(open(files[7], "rb").read()))
    assert cam.takePic() is False
