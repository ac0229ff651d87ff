#!/usr/bin/env python3
"""One 1080p frame through detect_pnp, 20 times: for rocprofv3 --kernel-trace --stats (per-kernel latency at batch 1)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cameracalibration_amd import _hip, synth  # noqa: E402
from tests import _data  # noqa: E402

K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
f = np.stack([synth.v1_frame(100 + i, K, ss=2, device="cuda")[0].cpu().numpy() for i in range(B)])
ctx = _hip.Context(1080, 1920, B)
for _ in range(20):
    ctx.detect_pnp(f, [10, 11, 14, 15], _data.v1_anchor_points(), K, dist)
