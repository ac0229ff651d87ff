#!/usr/bin/env python3
"""Developer aid: one 1080p frame with restart intervals (one MCU row each), decoded alone:  python3 profiles/tools/jpeg_restart.py"""
import io, json, os, sys, time
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cameracalibration_amd import _hip, synth  # noqa: E402
K = np.load(os.path.join(os.path.dirname(_hip.__file__), "camera_sync", "calibration", "Logitec_robot_1920-1080_10.npz"))["mtx"]
img = synth.v1_frame(500, np.asarray(K, np.float64), H=1080, W=1920, device=torch.device("cuda", 0), ss=2)[0].cpu().numpy()
ctx = _hip.Context(1080, 1920, 8)
out = {}
for name, kw in (("rows_1", dict(restart_marker_rows=1)), ("blocks_8", dict(restart_marker_blocks=8)), ("none", {})):
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(b, "JPEG", quality=85, subsampling=1, **kw)
    data = b.getvalue()
    ctx.decode_jpeg([data])
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.decode_jpeg([data])
    out[name + "_ms"] = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps(out))
