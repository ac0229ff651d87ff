#!/usr/bin/env python3
"""Latency of decoding one 1080p MJPEG frame at a time, and small batches:  python3 profiles/tools/jpeg_single.py"""
import io
import json
import os
import sys
import time

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cameracalibration_amd import _hip, synth  # noqa: E402

K = np.load(os.path.join(os.path.dirname(_hip.__file__), "camera_sync", "calibration", "Logitec_robot_1920-1080_10.npz"))["mtx"]
img = synth.v1_frame(500, np.asarray(K, np.float64), H=1080, W=1920, device=torch.device("cuda", 0), ss=2)[0].cpu().numpy()
b = io.BytesIO()
Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(b, "JPEG", quality=85, subsampling=1)
data = b.getvalue()
ctx = _hip.Context(1080, 1920, 8)
out = {}
for n in (1, 8, 64, 512):
    ctx.decode_jpeg([data] * n)
    t0 = time.perf_counter()
    reps = 5 if n < 64 else 2
    for _ in range(reps):
        ctx.decode_jpeg([data] * n)
    out["batch_%d_ms" % n] = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
for _ in range(5):
    np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
out["libjpeg_turbo_1_core_ms"] = (time.perf_counter() - t0) / 5 * 1e3
out["bytes"] = len(data)
print(json.dumps(out))
