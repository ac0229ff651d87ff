#!/usr/bin/env python3
"""Where the time of a small call goes: wall time and per-stage hipEvent times of detect / detect_pnp for 1 and 8
host frames (1080p), pageable and pinned."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cameracalibration_amd import _hip, synth  # noqa: E402
from tests import _data  # noqa: E402

K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
frames = np.stack([synth.v1_frame(100 + i, K, ss=2, device="cuda")[0].cpu().numpy() for i in range(8)])
anchors = [10, 11, 14, 15]
obj = _data.v1_anchor_points(anchors)
out = {}
for B in (1, 8):
    ctx = _hip.Context(1080, 1920, B)
    f = np.ascontiguousarray(frames[:B])
    for _ in range(3):
        ctx.detect_pnp(f, anchors, obj, K, dist)
    ctx.enable_timers(True)
    ctx.reset_timers()
    lat = []
    for _ in range(50):
        t0 = time.perf_counter()
        ctx.detect_pnp(f, anchors, obj, K, dist)
        lat.append((time.perf_counter() - t0) * 1e3)
    tm = ctx.timers()
    out["B%d" % B] = {"wall_p50_ms": float(np.percentile(lat, 50)), "wall_min_ms": float(np.min(lat)),
                      "stage_ms": {k: round(v[0] / 50, 4) for k, v in tm.items()}}
    ctx.close()
print(json.dumps(out, indent=1))
