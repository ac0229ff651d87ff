#!/usr/bin/env python3
"""Developer aid: how full the border stage's lists get, per frame, for the bench scene and the photo scene
(64 frames each):  python3 profiles/tools/census.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from cameracalibration_amd import _hip  # noqa: E402
from tests import _data  # noqa: E402

dev = torch.device("cuda", 0)
n = 64
K = np.array([[1400.0, 0, 960], [0, 1400.0, 540], [0, 0, 1]])
out = {}
frames = bench.make_frames(torch, dev, n, 1080, 1920, K, pool=32).cpu().numpy()
ctx = _hip.Context(1080, 1920, max_batch=n)
ctx.detect(frames)
out["bench"] = {k: round(v[0] / n, 1) for k, v in ctx.debug_lists().items()}
photo = _data.load_bgr("pic_2.png")
rng = np.random.default_rng(3)
pf = np.clip(np.rint(photo[None].astype(np.float32) + rng.normal(0, 1, (n,) + photo.shape)), 0, 255).astype(np.uint8)
ctx2 = _hip.Context(photo.shape[0], photo.shape[1], max_batch=n)
ctx2.detect(pf)
out["photo"] = {k: round(v[0] / n, 1) for k, v in ctx2.debug_lists().items()}
print(json.dumps(out, indent=1))
