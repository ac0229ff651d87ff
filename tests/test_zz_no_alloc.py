"""camsync_pnp_batch_noalloc (the entry point that allocates no device memory) against camsync_pnp_batch, through
the C ABI: two calls in a row on the shared static scratch array, the second with a smaller input, per-frame
cameras included; then the size limit.  The two entry points run the same kernel on the same values, so the
results must be identical bit for bit (that is within the 1e-6 pose bar of the PnP parity tests)."""
import numpy as np
import pytest

from cameracalibration_amd import _hip, synth
from tests import _data, _oracle as orc

pytestmark = pytest.mark.gpu


def _poses(B, N_anchor, seed, per_frame_cam=False):
    K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
    rng = np.random.default_rng(seed)
    anchors = [10, 11, 14, 15][:N_anchor]
    N = 4 * len(anchors)
    obj = np.broadcast_to(_data.v1_anchor_points(anchors).reshape(1, N, 3), (B, N, 3)).copy()
    img = np.zeros((B, N, 2), np.float32)
    Ks = np.repeat(K[None], B, 0).copy()
    ds = np.repeat(np.asarray(dist, np.float64).reshape(1, -1), B, 0).copy()
    for b in range(B):
        if per_frame_cam:
            Ks[b, 0, 0] *= rng.uniform(0.9, 1.1)
            Ks[b, 1, 1] *= rng.uniform(0.9, 1.1)
            ds[b] *= rng.uniform(0.5, 1.5)
        R, t = synth.look_at_pose(rng, (95.0, -135.0))
        rv = orc.rodrigues_mat2vec(R)
        img[b] = (orc.project_points(obj[b], rv, t, Ks[b], ds[b]) + rng.normal(0, 0.3, (N, 2))).astype(np.float32)
    npts = np.full(B, N, np.int32)
    if B > 6:
        npts[5] = 8
        npts[6] = 0
    return (obj, img, npts, Ks, ds) if per_frame_cam else (obj, img, npts, K, dist)


def _same(a, b):
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_noalloc_equals_pnp_batch_twice_second_call_smaller():
    ctx = _hip.Context(64, 64, 1)
    big = _poses(200, 4, seed=21)                       # 200 frames x 16 points
    small = _poses(7, 2, seed=22, per_frame_cam=True)   # then 7 frames x 8 points, a camera per frame
    want_big = ctx.pnp(*big)
    want_small = ctx.pnp(*small)
    got_big = ctx.pnp(*big, noalloc=True)
    got_small = ctx.pnp(*small, noalloc=True)           # reuses the array the larger call filled
    _same(got_big, want_big)
    _same(got_small, want_small)
    assert want_big[2].sum() == 199 and want_small[2].tolist() == [1, 1, 1, 1, 1, 1, 0]
    # and against the oracle, to the PnP parity bar
    obj, img, npts, Ks, ds = small
    for b in range(6):
        o_ok, orv, otv = orc.solve_pnp(obj[b, :npts[b]], img[b, :npts[b]], Ks[b], ds[b])
        assert o_ok and np.allclose(got_small[0][b], orv, rtol=1e-6, atol=1e-8) and np.allclose(got_small[1][b], otv, rtol=1e-6, atol=1e-6)
    # a second context on the same GPU shares the array
    ctx2 = _hip.Context(32, 32, 1)
    _same(ctx2.pnp(*small, noalloc=True), want_small)


def test_noalloc_rejects_sizes_beyond_its_array():
    ctx = _hip.Context(64, 64, 1)
    obj, img, npts, K, dist = _poses(8, 4, seed=3)
    for B, N in ((257, 16), (300, 64), (4096, 4)):
        o = np.zeros((B, N, 3), np.float32); i = np.zeros((B, N, 2), np.float32)
        with pytest.raises(_hip.CamsyncError) as e:
            ctx.pnp(o, i, np.zeros(B, np.int32), K, dist, noalloc=True)
        assert e.value.code == _hip.E_CAPACITY
    # 256 frames x 64 points is exactly the limit
    o = np.zeros((256, 64, 3), np.float32); i = np.zeros((256, 64, 2), np.float32)
    _, _, ok, _, _ = ctx.pnp(o, i, np.zeros(256, np.int32), K, dist, noalloc=True)
    assert ok.sum() == 0
    _same(ctx.pnp(obj, img, npts, K, dist, noalloc=True), ctx.pnp(obj, img, npts, K, dist))
