"""Developer aid: print what camsync_ingest_yuyv makes of a few hand-made pixel pairs."""
import numpy as np
from cameracalibration_amd import _hip
raw = np.zeros((1, 8, 8, 2), np.uint8)
raw[0, 0, :8] = [[0, 0], [0, 255], [255, 0], [255, 255], [16, 128], [235, 128], [100, 128], [100, 128]]
raw[0, 1, :2] = [[0, 128], [0, 255]]
raw[0, 2, :2] = [[0, 0], [0, 128]]
raw[0, 3, :2] = [[100, 128], [100, 200]]
ctx = _hip.Context(32, 32, max_batch=1)
out = ctx.ingest_yuyv(raw)
for r in range(4):
    print(r, out[0, r, :8].tolist())
