"""Frame ingest (reference camera_sync/camera.py:98-118, 150-154: cv.VideoCapture.read()): raw YUYV 4:2:2 -> BGR.

PARITY UNPINNED against OpenCV: the reference holds no raw camera frame and no output of the conversion, and cv2 is
not installed here.  The oracle (oracle/ingest.cpp) restates cv.cvtColor(COLOR_YUV2BGR_YUYV); the CPU tests pin it to
the known answers of ITU-R BT.601 (the 75 % / 100 % colour bars and the grey ramp), the GPU tests compare the HIP
path with it bit for bit and run the detector on ingested frames."""
import io

import numpy as np
import pytest

import os

from tests import _data, _oracle as orc

CALIB = os.path.join(_data.GOLDEN, "calibration")


def bgr_to_yuyv(bgr):
    """Test-side inverse (BT.601 limited range, float, chroma averaged over the pixel pair): makes raw frames whose
    conversion lands close to a given picture.  Not part of any parity claim."""
    f = bgr.astype(np.float64)
    b, g, r = f[..., 0], f[..., 1], f[..., 2]
    y = 16 + (65.481 * r + 128.553 * g + 24.966 * b) / 255
    u = 128 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255
    v = 128 + (112.0 * r - 93.786 * g - 18.214 * b) / 255
    h, w = y.shape
    out = np.empty((h, w, 2), np.uint8)
    out[..., 0] = np.clip(np.rint(y), 0, 255)
    out[:, 0::2, 1] = np.clip(np.rint((u[:, 0::2] + u[:, 1::2]) / 2), 0, 255)
    out[:, 1::2, 1] = np.clip(np.rint((v[:, 0::2] + v[:, 1::2]) / 2), 0, 255)
    return out


# ITU-R BT.601 studio-range colour bars: (Y, Cb, Cr) -> (R, G, B)
BARS_100 = [((235, 128, 128), (255, 255, 255)), ((210, 16, 146), (255, 255, 0)), ((170, 166, 16), (0, 255, 255)),
            ((145, 54, 34), (0, 255, 0)), ((106, 202, 222), (255, 0, 255)), ((81, 90, 240), (255, 0, 0)),
            ((41, 240, 110), (0, 0, 255)), ((16, 128, 128), (0, 0, 0))]
BARS_75 = [((180, 128, 128), (191, 191, 191)), ((162, 44, 142), (191, 191, 0)), ((131, 156, 44), (0, 191, 191)),
           ((112, 72, 58), (0, 191, 0)), ((84, 184, 198), (191, 0, 191)), ((65, 100, 212), (191, 0, 0)),
           ((35, 212, 114), (0, 0, 191))]


def test_oracle_colour_bars_and_grey_ramp():
    for (y, cb, cr), (r, g, b) in BARS_100 + BARS_75:
        px = np.array([[[y, cb], [y, cr]]], np.uint8)  # one pixel pair: Y0 U Y1 V
        out = orc.yuyv_to_bgr(px).astype(int)
        assert np.all(np.abs(out - np.array([b, g, r])) <= 2), ((y, cb, cr), out)
    # neutral chroma: B = G = R = the luma stretched from [16, 235] to [0, 255], saturating outside
    ramp = np.zeros((1, 256, 2), np.uint8)
    ramp[0, :, 0] = np.arange(256)
    ramp[0, :, 1] = 128
    out = orc.yuyv_to_bgr(ramp).astype(int)
    assert np.all(out[..., 0] == out[..., 1]) and np.all(out[..., 1] == out[..., 2])
    want = np.clip(np.rint((np.arange(256) - 16) * 255 / 219), 0, 255)
    assert np.all(np.abs(out[0, :, 0] - want) <= 1)
    assert out[0, :17].max() == 0 and out[0, 235:].min() == 255
    assert np.all(np.diff(out[0, :, 0]) >= 0)


def test_oracle_shares_chroma_over_the_pixel_pair_and_rejects_odd_widths():
    px = np.array([[[60, 140], [200, 120]]], np.uint8)
    a = orc.yuyv_to_bgr(px).astype(int)
    # same chroma, different luma: the two pixels differ by the same amount in every channel (away from saturation)
    d = a[0, 1] - a[0, 0]
    assert np.ptp(d) <= 1 and d[0] > 100
    with pytest.raises(ValueError):
        orc.yuyv_to_bgr(np.zeros((2, 3, 2), np.uint8))


def test_round_trip_through_the_test_side_encoder_is_close():
    rng = np.random.default_rng(5)
    img = np.repeat(rng.integers(0, 256, (24, 16, 3), dtype=np.uint8), 2, axis=1)  # pixel pairs share a colour
    back = orc.yuyv_to_bgr(bgr_to_yuyv(img)).astype(int)
    assert np.abs(back - img).max() <= 3


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 48, 64), (1, 720, 1280), (5, 33, 8)])
def test_gpu_yuyv_equals_oracle(shape):
    from cameracalibration_amd import _hip
    B, H, W = shape
    rng = np.random.default_rng(B * 1000 + H)
    raw = rng.integers(0, 256, (B, H, W, 2), dtype=np.uint8)
    raw[0, 0, :8] = [[0, 0], [0, 255], [255, 0], [255, 255], [16, 128], [235, 128], [0, 128], [255, 128]]
    ctx = _hip.Context(max(H, 32), max(W, 32), max_batch=2)  # B > max_batch: the host path goes in chunks
    got = ctx.ingest_yuyv(raw)
    for b in range(B):
        assert np.array_equal(got[b], orc.yuyv_to_bgr(raw[b])), b
    assert np.array_equal(ctx.ingest_yuyv(raw[0]), got[0])


@pytest.mark.gpu
def test_gpu_yuyv_frames_stay_on_the_device_for_detection():
    """Raw frames -> BGR on the GPU -> detect + PnP on the resident frames: equal to detecting the converted picture."""
    import torch
    from cameracalibration_amd import _hip
    from cameracalibration_amd.camera_sync import aruco
    img = _data.load_bgr("pic_2.png")
    H, W = img.shape[:2]
    raw = np.stack([bgr_to_yuyv(img), bgr_to_yuyv(img[:, ::-1])])
    want = [orc.yuyv_to_bgr(r) for r in raw]
    ctx = _hip.Context(H, W, max_batch=2)
    d_raw = torch.from_numpy(raw).cuda()
    d_bgr = torch.empty((2, H, W, 3), dtype=torch.uint8, device="cuda")
    assert ctx.ingest_yuyv(None, d_yuyv_ptr=d_raw.data_ptr(), d_bgr_ptr=d_bgr.data_ptr(), shape=(2, H, W)) is None
    torch.cuda.synchronize()
    assert np.array_equal(d_bgr.cpu().numpy(), np.stack(want))
    ids, cor, cnt, _, _ = ctx.detect(want[0])
    o_ids, o_cor, _ = orc.detect(want[0])
    n = int(cnt[0])
    assert n >= 4 and list(ids[0, :n]) == list(o_ids) and np.abs(cor[0, :n] - np.asarray(o_cor).reshape(n, 4, 2)).max() <= 1e-3
    # the same through the mirror's detectAruco on the ingested host frame
    detected, _ = aruco.detectAruco(ctx.ingest_yuyv(raw[0]))
    assert sorted(int(i) for i in detected) == sorted(int(i) for i in o_ids)


@pytest.mark.gpu
def test_camera_takes_pictures_from_a_raw_stream():
    """Camera.takePic (camera.py:150-154) with a VideoCapture-like stream of raw frames."""
    from cameracalibration_amd.camera_sync.camera import Camera
    from cameracalibration_amd.camera_sync.ingest import RawYUYVStream
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 256, (3, 48, 64, 2), dtype=np.uint8)
    stream = RawYUYVStream(io.BytesIO(frames.tobytes() + b"\x00" * 100), (64, 48))  # a truncated fourth frame at the end
    cam = Camera("nobody", 0, calibrationFolder=CALIB, resolution=(64, 48), captureStream=stream)
    for k in range(3):
        pic = cam.takePic()
        assert pic.shape == (48, 64, 3) and np.array_equal(pic, orc.yuyv_to_bgr(frames[k]))
    assert cam.takePic() is False
    stream.release()
    assert cam.takePic() is False
    batch = RawYUYVStream(io.BytesIO(frames.tobytes()), (64, 48))
    got = batch.read_batch(8)
    assert got.shape == (3, 48, 64, 3) and np.array_equal(got[2], orc.yuyv_to_bgr(frames[2]))
    assert batch.read_batch(2).shape[0] == 0
    with pytest.raises(NotImplementedError):
        Camera("nobody", -1, calibrationFolder=CALIB).takePic()
