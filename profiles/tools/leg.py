#!/usr/bin/env python3
"""One secondary leg of bench_extra.py alone, for profiling:  python3 profiles/tools/leg.py photo|4k|mjpeg [frames] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench_extra  # noqa: E402

which = sys.argv[1]
dev = torch.device("cuda", 0)
if which == "photo":
    r = bench_extra.leg_photo(torch, dev, int(sys.argv[2]) if len(sys.argv) > 2 else 4096, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
elif which == "mjpeg":
    r = bench_extra.leg_mjpeg(torch, dev, int(sys.argv[2]) if len(sys.argv) > 2 else 2048, steps=int(sys.argv[3]) if len(sys.argv) > 3 else 3,
                              oracle_frames=4)
else:
    r = bench_extra.leg_4k(torch, dev, int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
print(json.dumps(r))
