import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from cameracalibration_amd import _hip
from tests import _data
K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
dev = torch.device("cuda:0")
frames = bench.make_frames(torch, dev, 64, 1080, 1920, K)
ctx = _hip.Context(1080, 1920, 64)
f = frames.cpu().numpy()
ids, cor, cnt, _, _ = ctx.detect(f)
q = [ctx.debug_counts(i)["quads"] for i in range(64)]
print("quads per frame", q)
print("lists", ctx.debug_lists())
