#!/usr/bin/env python3
"""Time of one detection call on a frame with thousands of quad candidates (the selection instance whose working set is
in global memory):  python3 profiles/tools/busy_frame.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cameracalibration_amd import _hip  # noqa: E402

img = np.full((480, 640), 220, np.uint8)
for r in range(30):
    for c in range(40):
        img[r * 16 + 3:r * 16 + 13, c * 16 + 3:c * 16 + 13] = 30
ctx = _hip.Context(480, 640, 1)
t0 = time.perf_counter()
out = ctx.detect(img[..., None])
first = time.perf_counter() - t0
ctx.enable_timers(True) if hasattr(ctx, "enable_timers") else None
lat = []
for _ in range(5):
    t0 = time.perf_counter()
    out = ctx.detect(img[..., None], max_markers=64, max_rejected=2048)
    lat.append(time.perf_counter() - t0)
print(json.dumps({"first_call_s_with_growth": first, "later_calls_ms": [round(x * 1e3, 2) for x in lat],
                  "rejected": int(out[4][0]), "quads": ctx.debug_counts(0)["quads"], "regrown": ctx.debug_regrown()}))
