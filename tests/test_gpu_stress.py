"""Border-following stress scenes for the segment pipeline (k_contours.hip): borders far longer than a segment, borders
that wind through a grid cell without meeting a seed, nested holes, combs whose teeth make hundreds of local tops, blobs
touching the image frame.  Every kept contour of every threshold scale must equal the oracle's point list (order and
start included), and the detection outputs must equal the oracle's."""
import numpy as np
import pytest

from cameracalibration_amd import _hip
from tests import _oracle as orc

pytestmark = pytest.mark.gpu


def _oracle_bin(gray):
    return [orc.adaptive_threshold(gray, w, 7) for w in (3, 13, 23)]


def _check(frame):
    h, w = frame.shape[:2]
    ctx = _hip.Context(h, w, 1)
    minp, maxp = int(0.03 * max(h, w)), int(4.0 * max(h, w))
    gray = orc.bgr2gray(frame) if frame.ndim == 3 and frame.shape[2] == 3 else frame.reshape(h, w)
    want = {}
    for s, ob in enumerate(_oracle_bin(gray)):
        for c in orc.find_contours(ob):
            if minp <= len(c) <= maxp:
                want[(s, int(c[0][1]) * w + int(c[0][0]), len(c))] = c
    ids, corners, counts, rej, rejc = ctx.detect(frame)
    d = ctx.debug_contours(0, cap=1 << 20)
    pts = ctx.debug_points(0, cap=1 << 22)
    got = set(zip(d["scale"].tolist(), d["start_key"].tolist(), d["length"].tolist()))
    assert got == set(want), (len(got), len(want), sorted(set(want) - got)[:5], sorted(got - set(want))[:5])
    off = 0
    for s, k, n in zip(d["scale"].tolist(), d["start_key"].tolist(), d["length"].tolist()):
        assert np.array_equal(pts[off:off + n], want[(s, k, n)]), (s, k, n)
        off += n
    oi, oc, orj = orc.detect(frame)
    n = int(counts[0])
    assert ids[0, :n].tolist() == oi.tolist()
    assert int(rejc[0]) == len(orj)
    if len(orj):
        assert np.array_equal(rej[0, :len(orj)], orj)
    return len(want)


def _spiral(h, w, pitch):
    """A rectangular spiral of dark ink, `pitch` pixels from arm to arm: one border that is tens of thousands of points
    long (dropped by the length bound) next to shorter ones, with long stretches between grid crossings."""
    img = np.full((h, w), 220, np.uint8)
    x0, y0, x1, y1 = 8, 8, w - 9, h - 9
    t = max(2, pitch // 3)
    while x1 - x0 > 2 * pitch and y1 - y0 > 2 * pitch:
        img[y0:y0 + t, x0:x1] = 30
        img[y0:y1, x1 - t:x1] = 30
        img[y1 - t:y1, x0 + pitch:x1] = 30
        img[y0 + pitch:y1, x0 + pitch:x0 + pitch + t] = 30
        x0 += pitch; y0 += pitch; x1 -= pitch; y1 -= pitch
    return img


def _comb(h, w, tooth):
    """Combs: a bar with teeth pointing up: every tooth top is a start candidate of the same border."""
    img = np.full((h, w), 210, np.uint8)
    for r, y in enumerate(range(20, h - 60, 90)):
        img[y + 30:y + 50, 20:w - 20] = 25
        for x in range(24 + (r % 3), w - 24 - tooth, 2 * tooth):
            img[y + (x // tooth) % 7:y + 30, x:x + tooth] = 25
    return img


def _blobs(h, w, seed, scale):
    """Smooth random field thresholded into blobs with holes; some touch the frame."""
    rng = np.random.default_rng(seed)
    small = rng.normal(size=(h // scale + 3, w // scale + 3))
    ky = np.kron(small, np.ones((scale, scale)))[:h, :w]
    k = np.ones(2 * scale + 1) / (2 * scale + 1)
    ky = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, ky)
    ky = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, ky)
    return np.where(ky > 0.0, 40, 215).astype(np.uint8)


@pytest.mark.parametrize("pitch", [7, 12, 40])
def test_spirals(pitch):
    _check(_spiral(480, 640, pitch)[..., None])


def test_spiral_1080p():
    _check(_spiral(1080, 1920, 24)[..., None])


@pytest.mark.parametrize("tooth", [3, 5])
def test_combs(tooth):
    assert _check(_comb(600, 800, tooth)[..., None]) > 0


@pytest.mark.parametrize("seed,scale", [(1, 6), (2, 11), (3, 23)])
def test_blobs_with_holes(seed, scale):
    img = _blobs(720, 1280, seed, scale)
    rng = np.random.default_rng(100 + seed)
    img = np.clip(img.astype(np.int32) + rng.integers(-5, 6, img.shape), 0, 255).astype(np.uint8)
    assert _check(img[..., None]) > 0


def test_blobs_bgr_4k_strip():
    """A wide 4K-width strip: borders that cross many seed grid columns, three channels."""
    g = _blobs(512, 3840, 9, 17)
    frame = np.stack([g, np.roll(g, 3, axis=1), np.roll(g, -2, axis=0)], axis=-1)
    assert _check(frame) > 0
