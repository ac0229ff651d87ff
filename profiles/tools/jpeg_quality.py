#!/usr/bin/env python3
"""Developer aid: one 1080p frame decoded alone at several qualities (how fast the subsequence decoders settle):
python3 profiles/tools/jpeg_quality.py"""
import io, json, os, sys, time
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from cameracalibration_amd import _hip, synth  # noqa: E402
K = np.load(os.path.join(os.path.dirname(_hip.__file__), "camera_sync", "calibration", "Logitec_robot_1920-1080_10.npz"))["mtx"]
img = synth.v1_frame(500, np.asarray(K, np.float64), H=1080, W=1920, device=torch.device("cuda", 0), ss=2)[0].cpu().numpy()
rng = np.random.default_rng(1)
noisy = np.clip(img.astype(np.int16) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)
ctx = _hip.Context(1080, 1920, 8)
out = {}
for q in (50, 75, 85, 95, 100):
    for name, im in (("clean", img), ("noisy", noisy)):
        b = io.BytesIO()
        Image.fromarray(np.ascontiguousarray(im[:, :, ::-1])).save(b, "JPEG", quality=q, subsampling=1)
        data = b.getvalue()
        ctx.decode_jpeg([data])
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.decode_jpeg([data])
        out["q%d_%s" % (q, name)] = {"KB": len(data) // 1024, "ms": round((time.perf_counter() - t0) / 3 * 1e3, 2)}
print(json.dumps(out))
