#!/usr/bin/env python3
"""Experiment: the bench workload as P independent pipelines (contexts on their own streams, one host thread each,
frames split evenly) against one context with all frames.  Says what overlapping the latency-bound kernels of one
chunk with the instruction-bound kernels of another is worth.  usage: two_pipelines.py [frames] [pipelines ...]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from cameracalibration_amd import _hip  # noqa: E402
from tests import _data  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
PS = [int(v) for v in sys.argv[2:]] or [1, 2, 4]
SUB = int(os.environ.get("SUB", "1"))      # chunks per pipeline and step
DELAY = float(os.environ.get("DELAY_MS", "0"))  # start of pipeline p delayed by p * DELAY ms
H, W, MM = 1080, 1920, 16
dev = torch.device("cuda:0")
K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
anchors = [10, 11, 14, 15]
anchor_obj = _data.v1_anchor_points(anchors)
frames = bench.make_frames(torch, dev, N, H, W, K)
out = {}
for P in PS:
    per = N // P
    ctxs = [_hip.Context(H, W, max_batch=per // SUB) for _ in range(P)]
    bufs = []
    for p in range(P):
        bufs.append(dict(ids=torch.full((per, MM), -1, dtype=torch.int32, device=dev), cor=torch.zeros((per, MM, 4, 2), dtype=torch.float32, device=dev),
                         cnt=torch.zeros((per,), dtype=torch.int32, device=dev), rv=torch.zeros((per, 3), dtype=torch.float64, device=dev),
                         tv=torch.zeros((per, 3), dtype=torch.float64, device=dev), ok=torch.zeros((per,), dtype=torch.uint8, device=dev)))
    torch.cuda.synchronize()

    def work(p, steps):
        b = bufs[p]
        fr = frames[p * per:(p + 1) * per]
        if DELAY and p:
            time.sleep(p * DELAY * 1e-3)
        for _ in range(steps):
            ctxs[p].detect_pnp_resident(fr.data_ptr(), per, H, W, 3, anchors, anchor_obj, K, dist, MM, b["ids"].data_ptr(), b["cor"].data_ptr(),
                                        b["cnt"].data_ptr(), b["rv"].data_ptr(), b["tv"].data_ptr(), b["ok"].data_ptr())

    def run(steps):
        th = [threading.Thread(target=work, args=(p, steps)) for p in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()

    run(1)
    t0 = time.perf_counter()
    run(4)
    dt = (time.perf_counter() - t0 - (P - 1) * DELAY * 1e-3) / 4
    ok = sum(int(b["ok"].to(torch.int64).sum().item()) for b in bufs)
    out[P] = {"ms_per_step": dt * 1e3, "frames_per_s": N / dt, "frames_with_pose": ok}
    for c in ctxs:
        c.close()
    del bufs
    torch.cuda.empty_cache()
print(json.dumps(out))
