#!/usr/bin/env python3
"""Fill level of the border followers' scratch lists for the 4K scene (BASELINE configs[2]) and the bench scene,
per frame, so that the list budgets of camsync_create can be checked on paper.

    python profiles/tools/lists_4k.py > gpurun_out/lists_4k.json
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cameracalibration_amd import _hip, synth  # noqa: E402
from tests import _data  # noqa: E402


def main():
    dev = "cuda"
    out = {}
    K4 = np.array([[2800.0, 0, 1920.0], [0, 2800.0, 1080.0], [0, 0, 1.0]])
    clean = [synth.grid_frame(i, K4, ss=2, device=dev)[0].cpu().numpy() for i in range(4)]
    for B in (1, 4):
        ctx = _hip.Context(2160, 3840, B)
        ids, cor, cnt, _, _ = ctx.detect(np.stack(clean[:B]))
        out["4k_B%d" % B] = {"lists": ctx.debug_lists(), "counts": ctx.debug_counts(0), "markers": cnt.tolist()}
        ctx.close()
    K, _ = _data.calibration("Logitec_robot_1920-1080_10.npz")
    v1 = np.stack([synth.v1_frame(i, K, ss=2, device=dev)[0].cpu().numpy() for i in range(4)])
    ctx = _hip.Context(1080, 1920, 4)
    ids, cor, cnt, _, _ = ctx.detect(v1)
    out["1080p_v1_B4"] = {"lists": ctx.debug_lists(), "markers": cnt.tolist()}
    pic = _data.load_bgr("pic_2.png")
    ctx2 = _hip.Context(pic.shape[0], pic.shape[1], 1)
    ctx2.detect(pic)
    out["pic_2"] = {"lists": ctx2.debug_lists()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
