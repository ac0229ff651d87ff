"""Multi-GPU camera calibration: cv.calibrateCamera's Levenberg–Marquardt loop (reference:
camera_sync/camera.py:61-63) with the views sharded over ranks.

One process per GPU.  Rank g holds the contiguous share views[lo:hi] (shard_range) and, per LM iteration,
eliminates its views' extrinsic blocks on the GPU (Schur complement) into the record
[S (NI x NI) | rhs (NI) | err | parameter-change sums] — at most 159 doubles — which is summed over ranks with ONE
all-reduce per trial step (the kernel that evaluates a trial step also forms the trial parameters' record, so the
trial's error and the next iteration's normal equations travel in the same message).  Every rank then solves the same
NI x NI system redundantly and back-substitutes its own views, so no parameters are ever broadcast.  The control flow
is CvLevMarq::updateAlt's.

`calibrate_distributed` runs the loop inside the library (camsync_calibrate_dist: device-resident, one host
read-back per trial step) and supplies the collective as a callback: torch.distributed.all_reduce — RCCL when the
process group's backend is "nccl" — on a slice of the torch tensor the library keeps its records in, issued on the
library's own stream (no host synchronisation around the collective).
`lm_calibrate` is the same loop on the host over the camsync_calib_* building blocks (any worker object with
that interface; tests/test_distributed.py runs it over gloo without a GPU).
"""
import ctypes as C
import math

import numpy as np

from . import _hip

DBL_EPSILON = 2.220446049250313e-16


def shard_range(n: int, rank: int, world: int):
    """Contiguous share [lo, hi) of n items for `rank`; the first n % world ranks get one more."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class HipCalibWorker:
    """A rank's share of the views on its GPU (camsync_calib_* of include/camsync.h)."""

    def __init__(self, ctx: "_hip.Context", obj, img, rational=False):
        self.ctx = ctx
        obj = np.asarray(obj, np.float32)
        self.V = obj.shape[0]
        self.ni = ctx.calib_begin(obj, img, rational)

    def focal_acc(self, W, H):
        return self.ctx.calib_focal_acc(W, H)

    def set_intrinsics(self, intr, init_extrinsics):
        self.ctx.calib_set_intrinsics(intr, init_extrinsics)

    def accumulate(self, lam):
        return self.ctx.calib_accumulate(lam, self.ni)

    def step(self, delta):
        return self.ctx.calib_step(delta)

    def accept(self):
        self.ctx.calib_accept()

    def get(self):
        return self.ctx.calib_get(self.V)


def solve_intrinsic_block(acc, ni):
    """Host solve of S d = rhs (camsync_calib_solve: Cholesky with a ridge retry); needs no GPU."""
    lib = _hip.load_library()
    acc = np.ascontiguousarray(acc, np.float64)
    delta = np.zeros(12)
    rc = lib.camsync_calib_solve(acc.ctypes.data_as(C.c_void_p), ni, delta.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise _hip.CamsyncError(rc, "intrinsic block is singular")
    return delta


def make_allreduce(group=None, device=None):
    """Sum a small float64 vector over the ranks of `group` (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return lambda a: np.asarray(a, np.float64)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")

    def allreduce(a):
        t = torch.as_tensor(np.asarray(a, np.float64), dtype=torch.float64).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    return allreduce


def lm_calibrate(worker, image_size, n_views_total, allreduce, max_iter=30):
    """Run the LM loop over `worker`'s share; collective calls are matched on every rank.
    Returns (K, dist, rvecs_local, tvecs_local, iters, err) with err the global sum of squared residuals."""
    W, H = int(image_size[0]), int(image_size[1])
    ni = worker.ni
    L = ni * ni + ni + 1
    f6 = allreduce(worker.focal_acc(W, H))
    det = f6[0] * f6[2] - f6[1] * f6[1]
    if not abs(det) > 0:
        raise _hip.CamsyncError(-3, "focal length initialisation is singular")
    f0 = (f6[3] * f6[2] - f6[4] * f6[1]) / det
    f1 = (f6[0] * f6[4] - f6[1] * f6[3]) / det
    intr = np.zeros(12)
    intr[:4] = [math.sqrt(abs(1.0 / f0)), math.sqrt(abs(1.0 / f1)), 0.5 if not W else (W - 1) * 0.5, 0.5 if not H else (H - 1) * 0.5]
    worker.set_intrinsics(intr, True)

    lam_lg10, it = -3, 0
    prev_err = err = 0.0
    while True:
        acc = allreduce(worker.accumulate(math.exp(lam_lg10 * math.log(10.0))))
        assert acc.size == L
        if it == 0:
            prev_err = acc[-1]
        while True:
            delta = solve_intrinsic_block(acc, ni)
            err = float(allreduce([worker.step(delta)])[0])
            if err > prev_err:
                lam_lg10 += 1
                if lam_lg10 <= 16:
                    acc = allreduce(worker.accumulate(math.exp(lam_lg10 * math.log(10.0))))
                    continue
            break
        lam_lg10 = max(lam_lg10 - 1, -16)
        cur_i, cur_r, cur_t = worker.get()
        worker.accept()
        new_i, new_r, new_t = worker.get()
        # ||p_new - p_old|| / ||p_old|| over all parameters: intrinsics once, extrinsics summed over ranks
        ext = allreduce([((new_r - cur_r) ** 2).sum() + ((new_t - cur_t) ** 2).sum(), (cur_r ** 2).sum() + (cur_t ** 2).sum()])
        dn = ext[0] + ((new_i - cur_i) ** 2).sum()
        pn = ext[1] + (cur_i ** 2).sum()
        it += 1
        if it >= max_iter or math.sqrt(dn) / math.sqrt(pn) < DBL_EPSILON:
            break
        prev_err = err
    out, rvecs, tvecs = worker.get()
    K = np.array([[out[0], 0, out[2]], [0, out[1], out[3]], [0, 0, 1.0]])
    return K, out[4:ni].copy(), rvecs, tvecs, it, err


def make_device_allreduce(work, group=None, stats=None):
    """The camsync_allreduce_fn for torch.distributed: `work` is the float64 device tensor handed to the library
    as its work buffer; the library asks for slices of it to be summed over the ranks in place, on its own stream.

    The collective is issued in stream order: the library's HIP stream is wrapped as a torch ExternalStream and made
    current for the call, so RCCL's kernel waits (on the device) for the kernel that produced the record and the
    library's next kernel waits for RCCL's -- the host synchronises nowhere in here.  With the gloo backend (CPU
    tests) the record goes through the host, which waits for that one stream only.
    `stats` (a dict) receives the number of calls and the host seconds spent issuing them."""
    import time

    import torch
    import torch.distributed as dist

    base = work.data_ptr()
    dev = work.device
    backend = dist.get_backend(group)
    streams = {}

    def allreduce(d_buf, n, stream):
        t0 = time.perf_counter()
        off = (d_buf - base) // 8
        if off < 0 or off + n > work.numel() or (d_buf - base) % 8:
            raise ValueError("all-reduce request outside the work buffer")
        view = work[off:off + n]
        ext = streams.get(stream)
        if ext is None:
            ext = streams[stream] = torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)
        with torch.cuda.stream(ext):
            if backend == "nccl":        # RCCL on the device buffer itself, ordered on the library's stream
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)
            else:                        # gloo (tests): through the host
                h = view.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                view.copy_(h)
                ext.synchronize()
        if stats is not None:
            stats["calls"] = stats.get("calls", 0) + 1
            stats["seconds"] = stats.get("seconds", 0.0) + time.perf_counter() - t0

    return allreduce


def calibrate_distributed(obj, img, image_size, rational=False, max_iter=30, group=None, device_index=None,
                          return_iters=False, stats=None):
    """cv.calibrateCamera over torch.distributed.  Every rank passes the FULL obj [V,P,3] / img [V,P,2] (or
    anything indexable the same way) and works on its own share.  Returns on every rank
    (rms, K (3,3), dist (5|8), rvecs [V_local,3], tvecs [V_local,3], (lo, hi))[, iters]."""
    import torch
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if distributed else (0, 1)
    V = len(obj)
    lo, hi = shard_range(V, rank, world)
    if device_index is None:
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    ctx = _hip.Context(32, 32, 1, device=device_index)
    try:
        P = np.asarray(obj[0]).shape[0]
        o = np.asarray(obj[lo:hi], np.float32).reshape(hi - lo, P, 3)
        i = np.asarray(img[lo:hi], np.float32).reshape(hi - lo, P, 2)
        allreduce, d_work, work = None, None, None
        if distributed:  # (a one-rank group still goes through the collective: that is how the RCCL path is tested on one GPU)
            work = torch.zeros(_hip.CALIB_WORK_DOUBLES, dtype=torch.float64, device=torch.device("cuda", device_index))
            allreduce, d_work = make_device_allreduce(work, group, stats), work.data_ptr()
        rms, K, d, rv, tv, iters = ctx.calibrate_dist(o, i, V, image_size, rational, max_iter, allreduce, d_work)
    finally:
        ctx.close()
    res = (rms, K, d, rv, tv, (lo, hi))
    return res + (iters,) if return_iters else res
