#!/usr/bin/env python3
"""Calibration LM alone (configs[3]: V = 2000 views, rational model), for rocprofv3 kernel stats:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_calib -- python3 profiles/tools/calib_only.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cameracalibration_amd import _hip, synth  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
obj, img, _, _ = synth.calib_views(V=V, seed=42, rational=True)
ctx = _hip.Context(32, 32, 1)
ctx.calibrate(obj[:64], img[:64], (1920, 1080), rational=True, max_iter=2)
res = {}
for mi in (1, 30):
    t0 = time.perf_counter()
    rms, _, _, _, _, iters = ctx.calibrate(obj, img, (1920, 1080), rational=True, max_iter=mi)
    res[mi] = (time.perf_counter() - t0, iters, rms)
print(json.dumps({"views": V, "one_iter_s": res[1][0], "full_s": res[30][0], "iters": res[30][1], "rms": res[30][2],
                  "lm_only_iters_per_s": (res[30][1] - 1) / (res[30][0] - res[1][0])}))
