#!/usr/bin/env python3
"""Developer aid: damaged JPEG frames (random bytes overwritten anywhere: headers, tables, entropy-coded data) must
end in an error code or a decoded frame, never in a crash:  python3 profiles/tools/jpeg_fuzz.py [trials] [seed]"""
import io, os, sys
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cameracalibration_amd import _hip  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
h, w = 240, 320
yy, xx = np.mgrid[0:h, 0:w]
img = np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx + yy) * 3) % 256], -1).astype(np.uint8)
img[60:120, 80:160] = (250, 10, 10)
img[120:, :100] = rng.integers(0, 256, (h - 120, 100, 3))
goods = []
for kw in (dict(quality=85, subsampling=1), dict(quality=60, subsampling=2, optimize=True), dict(quality=90, subsampling=0, restart_marker_blocks=5),
           dict(quality=75, subsampling=1, restart_marker_rows=1)):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    goods.append(b.getvalue())
ctx = _hip.Context(h, w, 8)
want = [ctx.decode_jpeg([g])[0] for g in goods]
errors = decoded = 0
for t in range(trials):
    batch = []
    for g in goods:
        b = bytearray(g)
        n = int(rng.integers(1, 12))
        lo = 2 if t % 3 else b.index(b"\xff\xda") + 10  # two trials in three anywhere, one in the entropy-coded data only
        for at in rng.integers(lo, len(b), n):
            b[at] = int(rng.integers(0, 256))
        if t % 5 == 0:
            b = b[:int(rng.integers(lo, len(b)))]
        batch.append(bytes(b))
    for data in batch:
        try:
            ctx.decode_jpeg([data])
            decoded += 1
        except (_hip.CamsyncError, ValueError):
            errors += 1
    if t % 20 == 19:
        print("trial", t + 1, "errors", errors, "decoded", decoded, flush=True)
# the context is still good
for g, wv in zip(goods, want):
    assert np.array_equal(ctx.decode_jpeg([g])[0], wv)
print("ok: errors", errors, "decoded", decoded)
