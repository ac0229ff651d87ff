#!/usr/bin/env python3
"""bench.py — marker frames/s (detect + anchor PnP) on synthetic 1080p frames.

Workload (BASELINE.json configs[1]): ArUco DICT_4X4_50 detection + anchor
solvePnP on 4096 synthetic 1920x1080 BGR frames per GPU, inputs resident in HBM
before the timed region.  One "step" = one pass of the whole hot path
(gray + 3 adaptive thresholds + contours + quads + identification + sub-pixel
corners + PnP) over the rank's 4096 frames through the C ABI
(camsync_detect_pnp_resident).  Multi-GPU: frames shard across ranks, no
data-path collective (weak scaling); torch.distributed is only the barrier and
the max-over-ranks of the step time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_PX_K1K2 = 8.0   # gray 3r+1w, threshold 1r+3w  (fused kernel k_gray_thresh3)
ALG_BYTES_PER_PX_K1K3 = 11.0  # + contour extraction 3r
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s


def make_frames(torch, dev, n_frames, H, W, K, pool=256, seed=1234):
    """n_frames synthetic frames [n,H,W,3] u8 on `dev`: `pool` rendered poses of the v1 sheet,
    each frame = pose (i % pool) + its own Gaussian noise (sigma 4, seed per frame block)."""
    from cameracalibration_amd import synth
    clean = []
    rng_np = np.random.default_rng(seed)
    for i in range(pool):
        R, t = synth.look_at_pose(np.random.default_rng(seed + i), (95.0, -135.0))
        img = synth.render_sheet(synth.SHEET_V1, K, R, t, H, W, seed=seed + i, ss=2, noise_sigma=0.0, device=dev)
        clean.append(img[:, :, 0].to(torch.float32))
    clean = torch.stack(clean)  # [pool,H,W]
    frames = torch.empty((n_frames, H, W, 3), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    blk = 64
    for b0 in range(0, n_frames, blk):
        nb = min(blk, n_frames - b0)
        idx = (torch.arange(b0, b0 + nb, device=dev) % pool)
        noise = torch.randn((nb, H, W, 3), generator=g, device=dev, dtype=torch.float32) * 4.0
        frames[b0:b0 + nb] = (clean[idx][..., None] + noise).round_().clamp_(0, 255).to(torch.uint8)
    del rng_np
    return frames


def cpu_baseline(sample_frames, anchors_ids, anchor_obj, K, dist, threads):
    """Oracle (CPU restatement) detect + PnP on a bounded sample, frame-parallel threads.
    Returns (frames/s, [(ids, corners, pose_ok, rvec, tvec) per frame])."""
    from tests import _oracle as orc
    orc.lib()

    def one(img):
        ids, cor, _ = orc.detect(img)
        found = {int(i): c for i, c in zip(ids, cor)}  # duplicates: last wins, as detectAruco's dict (aruco.py:270-271)
        obj, pts = [], []
        for a, o in zip(anchors_ids, anchor_obj):
            if a in found:
                obj.append(o)
                pts.append(found[a])
        ok, rv, tv = False, np.zeros(3), np.zeros(3)
        if obj:
            ok, rv, tv = orc.solve_pnp(np.concatenate(obj), np.concatenate(pts), K, dist)
        return ids, cor, ok, rv, tv

    one(sample_frames[0])  # warm
    t0 = time.perf_counter()
    if threads > 1:
        with ThreadPoolExecutor(threads) as ex:
            res = list(ex.map(one, list(sample_frames)))
    else:
        res = [one(f) for f in sample_frames]
    dt = time.perf_counter() - t0
    return len(sample_frames) / dt, res


def compare_with_oracle(res, ids, cor, cnt, ok, rv, tv):
    """Full-size parity of the timed work: the GPU's results for the sampled frames against the oracle's (ids and
    order exact, corners <= 1e-3 px, pose <= 1e-6 relative)."""
    bad_ids = bad_cor = bad_pose = 0
    worst_c = worst_p = 0.0
    for f, (oi, oc, ook, orv, otv) in enumerate(res):
        n = int(cnt[f])
        if n != len(oi) or ids[f, :n].tolist() != oi.tolist():
            bad_ids += 1
            continue
        if n:
            e = float(np.abs(cor[f, :n] - oc).max())
            worst_c = max(worst_c, e)
            bad_cor += e > 1e-3
        if bool(ok[f]) != bool(ook):
            bad_pose += 1
        elif ook:
            e = max(float(np.abs(rv[f] - orv).max() / max(1.0, np.abs(orv).max())),
                    float(np.abs(tv[f] - otv).max() / max(1.0, np.abs(otv).max())))
            worst_p = max(worst_p, e)
            bad_pose += e > 1e-6
    n = len(res)
    return {"oracle_frames_compared": n, "oracle_frames_equal": n - bad_ids - bad_cor - bad_pose,
            "frames_ids_or_order_differ": int(bad_ids), "frames_corners_above_1e-3px": int(bad_cor),
            "frames_pose_above_1e-6_rel": int(bad_pose), "max_corner_err_px": worst_c, "max_pose_rel_err": worst_p}


def run_configs0():
    """BASELINE configs[0]: Camera.calibrate on the reference's nine 640x480 chessboard photos (tests/golden/calib,
    copies of pics/calib) -- corner candidates, cornerSubPix and the calibrateCamera LM on the GPU, no cv2 -- compared
    with the calibration the reference stores for them (calibration/robot_640-480_500.npz)."""
    try:
        import contextlib
        import io
        from PIL import Image
        from cameracalibration_amd.camera_sync.camera import Camera
        from tests import _data
        frames = []
        for i in range(1, 10):
            rgb = np.array(Image.open(os.path.join(_data.GOLDEN, "calib", "frame%d.jpg" % i)).convert("RGB"))
            frames.append(np.ascontiguousarray(rgb[:, :, ::-1]))
        with contextlib.redirect_stdout(io.StringIO()):  # the reference prints progress; stdout carries the JSON line
            Camera.calibrate(frames, (7, 7))
            t0 = time.perf_counter()
            mtx, dist = Camera.calibrate(frames, (7, 7))
            dt = time.perf_counter() - t0
        ref_K, ref_d = _data.calibration("robot_640-480_500.npz")
        return {"ran": True, "frames": len(frames), "seconds": dt, "fx": float(mtx[0, 0]), "fy": float(mtx[1, 1]),
                "max_abs_diff_mtx_vs_reference_npz": float(np.abs(mtx - ref_K).max()),
                "max_abs_diff_dist_vs_reference_npz": float(np.abs(dist.ravel() - ref_d.ravel()).max())}
    except Exception as e:
        return {"ran": False, "error": "%s: %s" % (type(e).__name__, e)}


def launch_ranks(n, argv):
    """Start n ranks of this script (one process per GPU) and wait for them.  The parent never touches a GPU: the
    children are started before any torch / HIP call, with RANK, LOCAL_RANK, WORLD_SIZE and the rendezvous address in
    their environment (what `python -m torch.distributed.run --nproc-per-node n` would set).  Rank 0 inherits stdout
    (the JSON line).  Returns the worst exit code."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:  # a rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def selftest_rank(args, world, rank):
    """CAMSYNC_BENCH_SELFTEST=1: the rank plumbing of this script without a GPU (tests/test_bench_launch.py):
    rendezvous over gloo, barrier, max-over-ranks of a per-rank time, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist_mod
    dist_mod.init_process_group("gloo", rank=rank, world_size=world)
    dist_mod.barrier()
    tt = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.MAX)
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = int(os.environ["LOCAL_RANK"]) + 1
    dist_mod.all_reduce(seen)
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "max_dt": float(tt.item()), "local_ranks": (seen - 1).tolist(),
                          "steps": args.steps, "warmup": args.warmup}))
    dist_mod.barrier()
    dist_mod.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--chunk", type=int, default=4096, help="frames per kernel launch sequence")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-sample", type=int, default=1536,
                    help="frames of the same workload timed on the CPU oracle (about 10 s on 16 threads)")
    ap.add_argument("--cpu-sample-1", type=int, default=64, help="frames of the sample timed on ONE CPU thread as well")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--calib-views", type=int, default=2000)
    ap.add_argument("--poses", type=int, default=256, help="rendered poses the frames cycle through (each frame adds its own noise)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations (4K, photo scene, 8 cameras)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher (no torch, no HIP here)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if os.environ.get("CAMSYNC_BENCH_SELFTEST"):
        return selftest_rank(args, world, rank)

    import torch
    import torch.distributed as dist_mod

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libcamsync_hip has no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, the node has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))

    from cameracalibration_amd import _hip
    from tests import _data

    H, W = args.height, args.width
    K, dist = _data.calibration("Logitec_robot_1920-1080_10.npz")
    if (H, W) != (1080, 1920):
        K = K.copy()
        K[0] *= W / 1920.0
        K[1] *= H / 1080.0
    anchors = [10, 11, 14, 15]
    anchor_obj = _data.v1_anchor_points(anchors)

    frames = make_frames(torch, dev, args.frames, H, W, K, pool=min(args.poses, args.frames), seed=1234 + 100000 * rank)
    B = args.frames
    MM = 16
    d_ids = torch.full((B, MM), -1, dtype=torch.int32, device=dev)
    d_cor = torch.zeros((B, MM, 4, 2), dtype=torch.float32, device=dev)
    d_cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
    d_rv = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    d_tv = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    d_ok = torch.zeros((B,), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    ctx = _hip.Context(H, W, max_batch=min(args.chunk, B), device=local_rank)

    def step():
        ctx.detect_pnp_resident(frames.data_ptr(), B, H, W, 3, anchors, anchor_obj, K, dist, MM,
                                d_ids.data_ptr(), d_cor.data_ptr(), d_cnt.data_ptr(), d_rv.data_ptr(),
                                d_tv.data_ptr(), d_ok.data_ptr())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist_mod.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist_mod.all_reduce(tt, op=dist_mod.ReduceOp.MAX)
        dt = float(tt.item())

    # sanity of the timed work: every frame must have produced its 6 markers and a pose
    n_ok = int(d_ok.to(torch.int64).sum().item())
    mean_markers = float(d_cnt.to(torch.float64).mean().item())

    # per-kernel time for the roofline: three more passes with the ctx's hipEvent stage timers (events on the
    # ctx's own stream, recorded around every stage), averaged
    ctx.enable_timers(True)
    ctx.reset_timers()
    n_timed = 3
    for _ in range(n_timed):
        step()
    tm = ctx.timers()
    ctx.enable_timers(False)
    tm = {k: (v[0] / n_timed, v[1] // n_timed) for k, v in tm.items()}  # per step
    ms0, n0 = tm["gray_threshold"]
    px_per_launch = min(args.chunk, B) * H * W
    # launches can be short at the tail; use totals: bytes of all launches / total time
    alg_bytes = ALG_BYTES_PER_PX_K1K2 * B * H * W
    achieved = alg_bytes / (ms0 * 1e-3) / 1e9 if ms0 > 0 else 0.0
    stage_ms = {k: round(v[0], 3) for k, v in tm.items()}
    # HBM traffic of that kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE cannot be collected from inside this
    # process): profiles/*_pmc_k_gray_thresh3.json holds the corrected bytes per pixel measured with rocprofv3 on this
    # same command; it is scaled to this run's launch size only when the frame geometry matches.
    traffic = phys_bpp = None
    try:
        import glob
        pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_k_gray_thresh3.json")))[-1]))
        if (pm["config"]["height"], pm["config"]["width"]) == (H, W) and pm["config"]["frames_per_launch"] == min(args.chunk, B):
            phys_bpp = pm["traffic_bytes_per_px"]
            traffic = phys_bpp * px_per_launch
    except Exception:
        traffic = None

    # secondary metric: calibration LM iterations/s (configs[3]); with N > 1 the views are sharded over the ranks and
    # the records are summed by RCCL (torch.distributed "nccl") inside camsync_calibrate_dist
    calib = None
    calib_inputs = None
    try:
        from cameracalibration_amd import synth
        from cameracalibration_amd.calibrate_dist import calibrate_distributed
        obj, img, _, _ = synth.calib_views(V=args.calib_views, seed=42, rational=True)
        if world == 1:
            ctx.calibrate(obj, img, (1920, 1080), rational=True, max_iter=1)  # warm: the buffers of this problem size
            # the LM loop alone: the same problem limited to one iteration, subtracted
            t3 = time.perf_counter()
            ctx.calibrate(obj, img, (1920, 1080), rational=True, max_iter=1)
            t4 = time.perf_counter()
            t1 = time.perf_counter()
            rms, _, _, _, _, iters = ctx.calibrate(obj, img, (1920, 1080), rational=True, max_iter=30)
            t2 = time.perf_counter()
            calib_inputs = (obj, img)
            calib = {"views": args.calib_views, "iters": iters, "iters_per_s": iters / (t2 - t1), "rms_px": rms,
                     "includes": "init (homographies + per-view PnP) + LM",
                     "lm_only_iters_per_s": (iters - 1) / max((t2 - t1) - (t4 - t3), 1e-9) if iters > 1 else None}
        else:
            stats = {}
            calibrate_distributed(obj, img, (1920, 1080), rational=True, max_iter=1, device_index=local_rank)  # warm: buffers, communicator
            barrier()
            t1 = time.perf_counter()
            rms, _, _, _, _, span, iters = calibrate_distributed(obj, img, (1920, 1080), rational=True, max_iter=30,
                                                                  device_index=local_rank, return_iters=True, stats=stats)
            barrier()
            tcal = torch.tensor([time.perf_counter() - t1, stats.get("seconds", 0.0)], dtype=torch.float64, device=dev)
            dist_mod.all_reduce(tcal, op=dist_mod.ReduceOp.MAX)
            calib = {"views": args.calib_views, "views_per_rank": span[1] - span[0], "iters": iters,
                     "iters_per_s": iters / float(tcal[0].item()), "rms_px": rms,
                     "allreduce_calls": stats.get("calls"), "allreduce_host_issue_share": float(tcal[1].item()) / float(tcal[0].item()),
                     "includes": "context + init (homographies + per-view PnP) + LM, records summed by RCCL"}
    except Exception as e:  # secondary; never hide the main line
        calib = {"error": "%s: %s" % (type(e).__name__, e)}

    out = None
    if rank == 0:
        value = world * B * args.steps / dt
        alg_frac = achieved / HBM_PEAK_GBS
        out = {
            "metric": "marker frames/sec (detect+PnP) 1080p",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: DICT_4X4_50 detect + anchor PnP, {B} synthetic {W}x{H} BGR frames per GPU, "
                                   f"6 markers (v1 sheet), 4 anchors, {min(args.poses, B)} rendered poses x per-frame noise "
                                   f"(sigma 4), chunk {min(args.chunk, B)}",
                       "frames_per_gpu": B, "height": H, "width": W, "parallelism": f"frames sharded x{world}"},
            # `achieved` follows the contract: ALGORITHMIC bytes (8 B/px for gray + 3 thresholds) over the
            # kernel's time.  The fused kernel physically moves far fewer bytes (`physical`): it does not sit at 91 % of
            # the memory system, it is bounded by vector-instruction issue (profiles/*_notes.md).
            "roofline": {"bound": "hbm", "kernel": "k_gray_thresh3 (K1+K2 fused)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg_frac, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, profiles/)",
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_PX_K1K2 * px_per_launch,
                         "algorithmic_bytes_per_px": ALG_BYTES_PER_PX_K1K2, "px_per_launch": px_per_launch,
                         "avg_launch_ms": ms0 / max(n0, 1),
                         "physical": None if phys_bpp is None else {
                             "bytes_per_px": phys_bpp, "achieved": achieved * phys_bpp / ALG_BYTES_PER_PX_K1K2,
                             "frac": alg_frac * phys_bpp / ALG_BYTES_PER_PX_K1K2, "limiter": "VALU issue, not HBM"},
                         "pipeline_frac_k1k3": (ALG_BYTES_PER_PX_K1K3 * H * W * value / max(world, 1)) / 1e9 / HBM_PEAK_GBS},
            "stage_ms_per_step": stage_ms,
            "check": {"frames_with_pose": n_ok, "mean_markers_per_frame": mean_markers},
        }
        out["configs0"] = run_configs0()
        if not args.no_cpu and world == 1:
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            ns = min(args.cpu_sample, B)
            sample = frames[:ns].cpu().numpy()
            fps, res = cpu_baseline(sample, anchors, anchor_obj, K, dist, threads)
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": threads, "kind": "port",
                                   "sample": f"first {len(sample)} of the same frames, oracle (CPU restatement, not OpenCV) "
                                             f"detect+PnP, {threads} frame-parallel threads"}
            if args.cpu_sample_1 > 0:
                fps1, _ = cpu_baseline(sample[:args.cpu_sample_1], anchors, anchor_obj, K, dist, 1)
                out["cpu_baseline"]["one_thread"] = {"value": fps1, "cores": 1, "sample": f"first {min(args.cpu_sample_1, ns)} frames"}
            # the sampled frames are the ones the timed steps processed: compare the two sets of results
            out["check"].update(compare_with_oracle(res, d_ids[:ns].cpu().numpy(), d_cor[:ns].cpu().numpy(), d_cnt[:ns].cpu().numpy(),
                                                    d_ok[:ns].cpu().numpy(), d_rv[:ns].cpu().numpy(), d_tv[:ns].cpu().numpy()))
        out["calib_lm"] = calib
        if not args.no_cpu and world == 1 and calib_inputs is not None and "error" not in calib:
            # configs[3] at its real size against the oracle (same views, CPU restatement of cv.calibrateCamera)
            from tests import _oracle as orc
            t1 = time.perf_counter()
            o_rms = orc.calibrate(calib_inputs[0], calib_inputs[1], (1920, 1080), rational=True)[0]
            out["check"]["calib_rms_vs_oracle"] = abs(calib["rms_px"] - o_rms)
            out["check"]["calib_oracle_seconds"] = time.perf_counter() - t1
        if not args.no_extra and world == 1:
            # secondary configurations, driver-timed with the rest of this run (bench_extra.py holds the legs)
            import contextlib
            import bench_extra
            del frames, d_ids, d_cor, d_cnt, d_rv, d_tv, d_ok
            torch.cuda.empty_cache()
            with contextlib.redirect_stdout(sys.stderr):  # the mirror prints what the reference prints
                out["config2_4k"] = bench_extra.leg_4k(torch, dev, 256, 3, oracle_frames=0 if args.no_cpu else 8)
                out["photo_scene"] = bench_extra.leg_photo(torch, dev, 4096, 3, oracle_frames=0 if args.no_cpu else 16)
                out["config5_8cams"] = {"1920x1080": bench_extra.leg_8cams(dev, 1080, 1920, "Logitec_robot", 10, 100),
                                        "1280x720": bench_extra.leg_8cams(dev, 720, 1280, "Logitec_A", 0, 100)}
                out["mjpeg_ingest"] = bench_extra.leg_mjpeg(torch, dev, oracle_frames=0 if args.no_cpu else 4)
            for k in ("config2_4k", "photo_scene"):
                out[k].pop("lists_used_and_capacity", None)
        print(json.dumps(out))
    if world > 1:
        dist_mod.barrier()
        dist_mod.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
