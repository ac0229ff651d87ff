#!/usr/bin/env python3
"""Developer aid: where an 8-camera set's time goes -- the C call with host frames, the same call with the frames
already on the device, and the mirror's whole set:  python3 profiles/tools/upload_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cameracalibration_amd import _hip, synth  # noqa: E402
from tests import _data  # noqa: E402

dev = torch.device("cuda", 0)
h, w = 1080, 1920
K = np.array([[1400.0, 0, 960], [0, 1400.0, 540], [0, 0, 1]])
views = [synth.v1_frame(100 + i, K, H=h, W=w, device=dev, ss=2)[0].cpu().numpy() for i in range(8)]
ctx = _hip.Context(h, w, max_batch=8)
anchors = [10, 11, 14, 15]
obj = _data.v1_anchor_points(anchors)
dist = np.zeros(5)


def med(f, n=100):
    f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


print("host frames, C call      %.3f ms" % med(lambda: ctx.detect_pnp_frames(views, anchors, obj, K, dist, 16)))
batch = np.stack(views)
print("host batch (contiguous)  %.3f ms" % med(lambda: ctx.detect_pnp(batch, anchors, obj, K, dist, 16)))
d = torch.from_numpy(batch).to(dev)
MM = 16
d_ids = torch.full((8, MM), -1, dtype=torch.int32, device=dev); d_cor = torch.zeros((8, MM, 4, 2), device=dev)
d_cnt = torch.zeros((8,), dtype=torch.int32, device=dev); d_rv = torch.zeros((8, 3), dtype=torch.float64, device=dev)
d_tv = torch.zeros((8, 3), dtype=torch.float64, device=dev); d_ok = torch.zeros((8,), dtype=torch.uint8, device=dev)
print("resident frames          %.3f ms" % med(lambda: ctx.detect_pnp_resident(d.data_ptr(), 8, h, w, 3, anchors, obj, K, dist, MM,
      d_ids.data_ptr(), d_cor.data_ptr(), d_cnt.data_ptr(), d_rv.data_ptr(), d_tv.data_ptr(), d_ok.data_ptr())))
pin = torch.from_numpy(batch).pin_memory()
t = torch.empty_like(d)
def up():
    t.copy_(pin, non_blocking=True); torch.cuda.synchronize()
print("50 MB pinned -> device   %.3f ms" % med(up))
def up2():
    t.copy_(torch.from_numpy(batch)); torch.cuda.synchronize()
print("50 MB pageable -> device %.3f ms" % med(up2))
