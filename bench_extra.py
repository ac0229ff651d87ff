#!/usr/bin/env python3
"""Secondary measurements of SURVEY.md §8(d) that are not the bench line (bench.py prints that one):

  * configs[2] on one GPU: detection of 3840x2160 frames with 40 markers each (frames/s, resident input; only with
    --frames4k N);
  * scene dependence: the reference's photo pic_2.png (textured desk scene) as a resident batch of 1280x720 frames;
  * config 5: wall latency of `processArucoFromMultipleCameras` for 8 cameras looking at one scene
    (p50 / p99 over repeated frame-sets, 1280x720 and 1920x1080), host frames in, Python mirror API,
    and the same eight 1080p frames as one batch (processArucoBatch).

    python bench_extra.py [--frames4k 256] [--sets 200] > profiles/rNN_extra.json

Synthetic frames from this repository's renderer (no dataset, no network).  One JSON object on stdout.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def leg_4k(torch, dev, n=256, steps=3, oracle_frames=0, pool=4):
    """configs[2] on one GPU: detect + PnP on n resident 3840x2160 frames with 40 markers each (pool rendered poses x
    per-frame noise).  oracle_frames > 0: the first frames' results are compared with the CPU oracle (ids and order
    exact, corners <= 1e-3 px)."""
    from cameracalibration_amd import _hip, synth
    H, W = 2160, 3840
    K4 = np.array([[2800.0, 0, 1920.0], [0, 2800.0, 1080.0], [0, 0, 1.0]])
    rendered = [synth.grid_frame(i, K4, device=dev, ss=2, noise_sigma=0.0) for i in range(pool)]
    clean = torch.stack([r[0][:, :, 0].to(torch.float32) for r in rendered])
    frames = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    for b0 in range(0, n, 16):
        nb = min(16, n - b0)
        idx = torch.arange(b0, b0 + nb, device=dev) % pool
        noise = torch.randn((nb, H, W, 3), generator=g, device=dev, dtype=torch.float32) * 4.0
        frames[b0:b0 + nb] = (clean[idx][..., None] + noise).round_().clamp_(0, 255).to(torch.uint8)
    MM = 64
    d_ids = torch.full((n, MM), -1, dtype=torch.int32, device=dev)
    d_cor = torch.zeros((n, MM, 4, 2), dtype=torch.float32, device=dev)
    d_cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
    d_rv = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    d_tv = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    d_ok = torch.zeros((n,), dtype=torch.uint8, device=dev)
    ctx = _hip.Context(H, W, max_batch=n, device=dev.index or 0)
    # anchors: the first four markers of every pose's sheet share ids but not places; the pose leg uses pose 0's
    sheet0 = rendered[0][1]
    anchors = [int(m[0]) for m in sheet0[:4]]
    anchor_obj = np.array([[[x, y, 0], [x + sz, y, 0], [x + sz, y - sz, 0], [x, y - sz, 0]] for (_, x, y, sz) in sheet0[:4]],
                          np.float32)

    def step4k():
        ctx.detect_pnp_resident(frames.data_ptr(), n, H, W, 3, anchors, anchor_obj, K4, np.zeros(5), MM,
                                d_ids.data_ptr(), d_cor.data_ptr(), d_cnt.data_ptr(), d_rv.data_ptr(),
                                d_tv.data_ptr(), d_ok.data_ptr())

    try:
        step4k()
        torch.cuda.synchronize()
        # size-independent check: every frame shows its sheet's 40 markers and nothing else
        cnt = d_cnt.cpu().numpy()
        ids = d_ids.cpu().numpy()
        want = [sorted(int(m[0]) for m in r[1]) for r in rendered]
        frames_right = sum(int(cnt[i]) == 40 and sorted(ids[i, :40].tolist()) == want[i % pool] for i in range(n))
        lists = ctx.debug_lists()
        ctx.enable_timers(True)
        ctx.reset_timers()
        t0 = time.perf_counter()
        for _ in range(steps):
            step4k()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tm = ctx.timers()
        ctx.enable_timers(False)
        res = {
            "frames": n, "frames_per_s": n / dt, "ms_per_frame": dt / n * 1e3, "ms_per_step": dt * 1e3,
            "mean_markers_per_frame": float(cnt.mean()),
            "frames_with_exactly_their_40_markers": frames_right,
            "hbm_frac_k1k3_11B_per_px": n / dt * H * W * 11.0 / 8e12,
            "stage_ms_per_step": {k: round(v[0] / steps, 3) for k, v in tm.items()},
            "lists_used_and_capacity": lists,
            "note": "configs[2] on one GPU: detect + PnP on 4 sheet markers, resident input, %d rendered poses x per-frame "
                    "noise (sigma 4)" % pool}
        if oracle_frames > 0:
            from concurrent.futures import ThreadPoolExecutor
            from tests import _oracle as orc
            k = min(oracle_frames, n)
            host = frames[:k].cpu().numpy()
            cor = d_cor[:k].cpu().numpy()
            with ThreadPoolExecutor(min(k, 16)) as ex:
                ores = list(ex.map(orc.detect, list(host)))
            equal, worst = 0, 0.0
            for i, (oi, oc, _) in enumerate(ores):
                m = int(cnt[i])
                if m == len(oi) and ids[i, :m].tolist() == oi.tolist():
                    e = float(np.abs(cor[i, :m] - oc).max()) if m else 0.0
                    worst = max(worst, e)
                    equal += e <= 1e-3
            res["oracle_frames_compared"] = k
            res["oracle_frames_equal"] = int(equal)
            res["max_corner_err_px"] = worst
    except Exception as e:  # keep the other measurements
        res = {"error": "%s: %s" % (type(e).__name__, e)}
    ctx.close()
    del frames, clean
    torch.cuda.empty_cache()
    return res


def leg_photo(torch, dev, n=4096, steps=3, oracle_frames=0):
    """Scene dependence: the reference's own photo pic_2.png (1280x720, textured desk scene, 6 markers) + noise sigma 1
    per frame as a resident batch, detect + PnP."""
    from cameracalibration_amd import _hip
    from tests import _data
    try:
        photos = [_data.load_bgr("pic_2.png")]
        h, w = photos[0].shape[:2]
        base = torch.stack([torch.from_numpy(np.ascontiguousarray(p)).to(dev) for p in photos])
        g = torch.Generator(device=dev).manual_seed(11)
        frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
        for b0 in range(0, n, 64):
            nb = min(64, n - b0)
            idx = torch.arange(b0, b0 + nb, device=dev) % len(photos)
            noise = torch.randn((nb, h, w, 3), generator=g, device=dev, dtype=torch.float32) * 1.0
            frames[b0:b0 + nb] = (base[idx].to(torch.float32) + noise).round_().clamp_(0, 255).to(torch.uint8)
        Kp, dp = _data.calibration("Logitec_A_1280-720_0.npz")
        MM = 16
        d_ids = torch.full((n, MM), -1, dtype=torch.int32, device=dev)
        d_cor = torch.zeros((n, MM, 4, 2), dtype=torch.float32, device=dev)
        d_cnt = torch.zeros((n,), dtype=torch.int32, device=dev)
        d_rv = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        d_tv = torch.zeros((n, 3), dtype=torch.float64, device=dev)
        d_ok = torch.zeros((n,), dtype=torch.uint8, device=dev)
        ctx = _hip.Context(h, w, max_batch=n, device=dev.index or 0)
        anchors = [10, 11, 14, 15]
        anchor_obj = _data.v1_anchor_points(anchors)

        def step_photo():
            ctx.detect_pnp_resident(frames.data_ptr(), n, h, w, 3, anchors, anchor_obj, Kp, dp, MM, d_ids.data_ptr(),
                                    d_cor.data_ptr(), d_cnt.data_ptr(), d_rv.data_ptr(), d_tv.data_ptr(), d_ok.data_ptr())

        step_photo()
        torch.cuda.synchronize()
        lists = ctx.debug_lists()
        ctx.enable_timers(True)
        ctx.reset_timers()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_photo()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        tm = ctx.timers()
        px_rate = n / dt * h * w
        res = {
            "frames": n, "frames_per_s": n / dt, "ms_per_step": dt * 1e3, "pixels_per_s": px_rate,
            "as_1080p_frames_per_s_by_pixels": px_rate / (1920 * 1080),
            "hbm_frac_k1k3_11B_per_px": px_rate * 11.0 / 8e12,
            "frames_with_pose": int(d_ok.to(torch.int64).sum().item()), "mean_markers_per_frame": float(d_cnt.to(torch.float64).mean().item()),
            "stage_ms_per_step": {k: round(v[0] / steps, 3) for k, v in tm.items()}, "lists_used_and_capacity": lists,
            "note": "the reference's pic_2.png (desk scene, 6 markers) + noise sigma 1 per frame; "
                    "detect + PnP, resident input: how much the bench number depends on its flat synthetic background"}
        if oracle_frames > 0:
            from concurrent.futures import ThreadPoolExecutor
            from tests import _oracle as orc
            k = min(oracle_frames, n)
            host = frames[:k].cpu().numpy()
            ids, cor, cnt = d_ids[:k].cpu().numpy(), d_cor[:k].cpu().numpy(), d_cnt[:k].cpu().numpy()
            with ThreadPoolExecutor(min(k, 16)) as ex:
                ores = list(ex.map(orc.detect, list(host)))
            equal = 0
            for i, (oi, oc, _) in enumerate(ores):
                m = int(cnt[i])
                equal += m == len(oi) and ids[i, :m].tolist() == oi.tolist() and (m == 0 or float(np.abs(cor[i, :m] - oc).max()) <= 1e-3)
            res["oracle_frames_compared"] = k
            res["oracle_frames_equal"] = int(equal)
        ctx.close()
        del frames
        torch.cuda.empty_cache()
        return res
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def leg_8cams(dev, h, w, camname, focus, sets=200):
    """configs[4]: wall latency of processArucoFromMultipleCameras for 8 cameras looking at one scene (host frames in,
    mirror API, fused marker corners out)."""
    from cameracalibration_amd import synth
    from cameracalibration_amd.camera_sync import aruco as A
    from cameracalibration_amd.camera_sync.camera import Camera
    try:
        cams = [Camera(camname, -1, focus=focus, resolution=(w, h)) for _ in range(8)]
        Kc = np.asarray(cams[0].mtx, dtype=np.float64)
        views = [synth.v1_frame(100 + i, Kc, H=h, W=w, device=dev, ss=2)[0].cpu().numpy() for i in range(8)]
        sheet = A.getArucosFromPaper(1)
        fixed = [sheet[i] for i in (10, 11, 14, 15)]
        moving = {12: sheet[12], 13: sheet[13]}
        lat = []
        A.processArucoFromMultipleCameras(fixed, moving, cams, views)
        for _ in range(sets):
            t0 = time.perf_counter()
            A.processArucoFromMultipleCameras(fixed, moving, cams, views)
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.array(lat)
        located = sum(hasattr(c, "world2cam") for c in cams)
        return {"sets": sets, "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                "cameras_located": int(located),
                "note": "configs[4]: 8 host frames (pageable memory) per set through the mirror API: one detection + "
                        "anchor-PnP launch sequence for all cameras, one IPPE + fusion call for all moving markers; "
                        "includes the host-to-device copies"}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def leg_mjpeg(torch, dev, n=4096, h=1080, w=1920, steps=3, distinct=32, quality=85, cpu_frames=24, oracle_frames=0):
    """Compressed ingest: n MJPEG frames (rendered views of the v1 sheet, encoded at 4:2:2 with Pillow = libjpeg-turbo,
    Huffman tables stripped as a camera sends them) decoded to BGR frames that stay on the GPU.  The compressed bytes
    start in host memory (that is where a capture delivers them), so the figure includes the copy of the compressed
    data and the host-side header parse.  Beside it: libjpeg-turbo itself on one host core (Pillow), the decoder
    cv.VideoCapture.read() uses."""
    import io
    from PIL import Image
    from cameracalibration_amd import _hip, synth
    try:
        K = np.load(os.path.join(ROOT, "cameracalibration_amd", "camera_sync", "calibration", "Logitec_robot_1920-1080_10.npz"))["mtx"]
        datas = []
        for i in range(distinct):
            img = synth.v1_frame(500 + i, np.asarray(K, np.float64), H=h, W=w, device=dev, ss=2)[0].cpu().numpy()
            b = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(b, "JPEG", quality=quality, subsampling=1)
            d = b.getvalue()
            # drop the DHT segments (the standard tables: what an MJPEG frame means without them)
            out = bytearray(d[:2])
            k = 2
            while d[k + 1] != 0xDA:
                L = (d[k + 2] << 8) | d[k + 3]
                if d[k + 1] != 0xC4:
                    out += d[k:k + 2 + L]
                k += 2 + L
            datas.append(bytes(out + d[k:]))
        frames = [datas[i % distinct] for i in range(n)]
        ctx = _hip.Context(h, w, 8, 0)
        dst = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
        ctx.decode_jpeg(frames, d_bgr_ptr=dst.data_ptr())  # warm-up: buffers
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.decode_jpeg(frames, d_bgr_ptr=dst.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = {"frames": n, "height": h, "width": w, "frames_per_s": n / dt, "ms_per_batch": dt * 1e3,
               "compressed_bytes_per_frame": int(np.mean([len(d) for d in datas])),
               "compressed_GB_per_s": sum(len(f) for f in frames) / dt / 1e9,
               "note": "MJPEG frames (4:2:2, quality %d, no DHT segments) in host memory -> BGR frames resident on the GPU; "
                       "%d distinct frames repeated" % (quality, distinct)}
        # the reference's decoder on one host core
        t0 = time.perf_counter()
        for i in range(cpu_frames):
            np.asarray(Image.open(io.BytesIO(frames[i])).convert("RGB"))
        res["cpu_libjpeg_turbo_1_core_frames_per_s"] = cpu_frames / (time.perf_counter() - t0)
        if oracle_frames:
            from tests import _oracle as orc
            got = dst[:oracle_frames].cpu().numpy()
            res["oracle_frames_compared"] = oracle_frames
            res["oracle_frames_equal"] = int(sum(np.array_equal(got[i], orc.jpeg_decode(frames[i])) for i in range(oracle_frames)))
        ctx.close()
        return res
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames4k", type=int, default=256, help="frames of the 4K configuration (0 = skip)")
    ap.add_argument("--steps4k", type=int, default=3, help="timed passes over the 4K frames")
    ap.add_argument("--only4k", action="store_true", help="skip the 8-camera latency legs")
    ap.add_argument("--sets", type=int, default=200)
    ap.add_argument("--photo-frames", type=int, default=4096, help="frames of the photo leg (the reference's 1280x720 picture)")
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_extra.py needs a GPU (libcamsync_hip has no CPU fallback)")
    dev = torch.device("cuda", 0)
    from cameracalibration_amd import _hip, synth
    from cameracalibration_amd.camera_sync import aruco as A
    from cameracalibration_amd.camera_sync.camera import Camera
    from tests import _data

    out = {}
    # the mirror prints what the reference prints ("Found calibration file ..."): keep stdout for the JSON line
    import contextlib
    _quiet = contextlib.redirect_stdout(sys.stderr)
    _quiet.__enter__()

    if args.frames4k > 0:
        out["config2_4k_one_gpu"] = leg_4k(torch, dev, args.frames4k, args.steps4k)

    if args.only4k:
        _quiet.__exit__(None, None, None)
        print(json.dumps(out))
        return

    out["photos_1280x720"] = leg_photo(torch, dev, args.photo_frames)

    # ---- host frames in: the PCIe-inclusive rate of detect + PnP, pageable against page-locked frame buffers ----
    try:
        h, w, n = 1080, 1920, 512
        Kh, dh = _data.calibration("Logitec_robot_1920-1080_10.npz")
        src = np.stack([synth.v1_frame(300 + i, Kh, H=h, W=w, device=dev, ss=2)[0].cpu().numpy() for i in range(8)])
        pageable = np.ascontiguousarray(np.tile(src, (n // 8, 1, 1, 1)))
        pinned = _hip.PinnedFrames(pageable.shape)
        pinned.array[...] = pageable
        ctx = _hip.Context(h, w, max_batch=n, device=0)
        sheet = A.getArucosFromPaper(1)
        anchors = [10, 11, 14, 15]
        anchor_obj = _data.v1_anchor_points(anchors)
        res = {}
        for name, buf in (("pageable", pageable), ("page_locked", pinned.array)):
            ctx.detect_pnp(buf, anchors, anchor_obj, Kh, dh)
            t0 = time.perf_counter()
            for _ in range(3):
                r = ctx.detect_pnp(buf, anchors, anchor_obj, Kh, dh)
            dt = (time.perf_counter() - t0) / 3
            res[name] = {"frames_per_s": n / dt, "GB_per_s_of_frames": n * h * w * 3 / dt / 1e9}
        res["frames"] = n
        res["note"] = "camsync_detect_pnp_batch on 512 host frames (1920x1080 BGR, 6.2 MB each) per call, results back on the host"
        out["host_frames_1920x1080"] = res
        pinned.close()
        del ctx
    except Exception as e:
        out["host_frames_1920x1080"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- frame ingest: raw YUYV 4:2:2 -> BGR on the device (camsync_ingest_yuyv) ----
    try:
        n, h, w = 512, 1080, 1920
        ctx = _hip.Context(h, w, max_batch=n)
        g = torch.Generator(device="cpu").manual_seed(3)
        raw = torch.randint(0, 256, (n, h, w, 2), dtype=torch.uint8, generator=g)
        d_raw = raw.to(dev)
        d_bgr = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
        pinned = _hip.PinnedFrames((n, h, w, 2))
        pinned.array[...] = raw.numpy()
        res = {}
        for name, call in (("resident", lambda: ctx.ingest_yuyv(None, d_yuyv_ptr=d_raw.data_ptr(), d_bgr_ptr=d_bgr.data_ptr(), shape=(n, h, w))),
                           ("pinned_host_to_device", lambda: ctx.ingest_yuyv(pinned.array, d_bgr_ptr=d_bgr.data_ptr()))):
            call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
            res[name] = {"frames_per_s": n / dt, "GB_per_s_read_plus_written": n * h * w * 5 / dt / 1e9}
        res["note"] = ("512 raw 1920x1080 frames (2 B/px) to BGR frames on the device (3 B/px); 'resident' is the kernel alone "
                       "(HBM-bound: 5 B/px), 'pinned_host_to_device' includes the copy of the raw frames over the host link")
        out["ingest_yuyv_1920x1080"] = res
        pinned.close()
        del ctx, d_raw, d_bgr
    except Exception as e:
        out["ingest_yuyv_1920x1080"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- config 5: 8 cameras, one scene, processArucoFromMultipleCameras (host frames, mirror API) ----
    K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
    for (h, w, camname, focus) in ((720, 1280, "Logitec_A", 0), (1080, 1920, "Logitec_robot", 10)):
        out["config5_8cams_%dx%d" % (w, h)] = leg_8cams(dev, h, w, camname, focus, args.sets)

    # ---- single frame: detectAruco on one host frame ----
    try:
        one = synth.v1_frame(100, K, H=1080, W=1920, device=dev, ss=2)[0].cpu().numpy()
        A.detectAruco(one)
        lat = []
        for _ in range(args.sets):
            t0 = time.perf_counter()
            A.detectAruco(one)
            lat.append((time.perf_counter() - t0) * 1e3)
        out["single_frame_1920x1080_detectAruco"] = {"sets": args.sets, "p50_ms": float(np.percentile(lat, 50)),
                                                     "p99_ms": float(np.percentile(lat, 99)),
                                                     "note": "one pageable host frame in, dict of corners out"}
    except Exception as e:
        out["single_frame_1920x1080_detectAruco"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- one launch sequence for the same 8 frames (detectArucoBatch + processArucoBatch) ----
    try:
        h, w = 1080, 1920
        views = np.stack([synth.v1_frame(100 + i, K, H=h, W=w, device=dev, ss=2)[0].cpu().numpy() for i in range(8)])
        sheet = A.getArucosFromPaper(1)
        fixed = [sheet[i] for i in (10, 11, 14, 15)]
        A.processArucoBatch(fixed, views, K, dist)
        lat = []
        for _ in range(args.sets):
            t0 = time.perf_counter()
            A.processArucoBatch(fixed, views, K, dist)
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.array(lat)
        out["batch8_1920x1080_processArucoBatch"] = {
            "sets": args.sets, "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
            "note": "the 8 frames as one batch: one copy, one launch sequence, camera poses only"}
    except Exception as e:
        out["batch8_1920x1080_processArucoBatch"] = {"error": "%s: %s" % (type(e).__name__, e)}

    _quiet.__exit__(None, None, None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
