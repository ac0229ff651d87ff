#!/usr/bin/env python3
"""Developer aid: the 8-camera latency leg alone (for profiling):  python3 profiles/tools/cams.py [sets] [h w]"""
import contextlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench_extra  # noqa: E402

sets = int(sys.argv[1]) if len(sys.argv) > 1 else 100
h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
with contextlib.redirect_stdout(sys.stderr):
    r = bench_extra.leg_8cams(torch.device("cuda", 0), h, w, "Logitec_robot" if h == 1080 else "Logitec_A", 10 if h == 1080 else 0, sets)
print(json.dumps(r))
