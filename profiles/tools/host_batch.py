#!/usr/bin/env python3
"""Developer aid: detect + PnP on 512 host frames per call, pageable against page-locked:  python3 profiles/tools/host_batch.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cameracalibration_amd import _hip, synth  # noqa: E402
from tests import _data  # noqa: E402

dev = torch.device("cuda", 0)
h, w, n = 1080, 1920, 512
Kh, dh = _data.calibration("Logitec_robot_1920-1080_10.npz")
src = np.stack([synth.v1_frame(300 + i, Kh, H=h, W=w, device=dev, ss=2)[0].cpu().numpy() for i in range(8)])
pageable = np.ascontiguousarray(np.tile(src, (n // 8, 1, 1, 1)))
pinned = _hip.PinnedFrames(pageable.shape)
pinned.array[...] = pageable
ctx = _hip.Context(h, w, max_batch=n, device=0)
anchors = [10, 11, 14, 15]
obj = _data.v1_anchor_points(anchors)
for name, buf in (("pageable", pageable), ("page_locked", pinned.array)):
    ctx.detect_pnp(buf, anchors, obj, Kh, dh)
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.detect_pnp(buf, anchors, obj, Kh, dh)
    dt = (time.perf_counter() - t0) / 3
    print(name, "%.0f frames/s %.1f GB/s" % (n / dt, n * h * w * 3 / dt / 1e9))
