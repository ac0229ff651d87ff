"""Seeded synthetic inputs for tests and bench.py.

Marker sheets are rendered through a pinhole camera by inverse mapping with
supersampling; there is no cv2 here.  torch is used only as an array library
so the same code can render on the GPU for large batches.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ._dict4x4_50 import CODES, VERIFIED_IDS

# getArucosFromPaper(1) big markers (reference camera_sync/aruco.py:131-136): id -> (x, y) of the top-left, 80 mm
SHEET_V1 = [(10, 0.0, 0.0, 80.0), (11, 110.0, 0.0, 80.0), (12, 0.0, -95.0, 80.0),
            (13, 110.0, -95.0, 80.0), (14, 0.0, -190.0, 80.0), (15, 110.0, -190.0, 80.0)]


def rodrigues(rvec):
    r = np.asarray(rvec, dtype=np.float64).ravel()
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def sheet_grid(rng: np.random.Generator, nx=8, ny=5, pitch=100.0, ids=None):
    """nx*ny markers on a jittered grid, sizes 0.45..0.8 of the pitch."""
    if ids is None:
        pool = list(VERIFIED_IDS)
        ids = [pool[i % len(pool)] for i in range(nx * ny)]
    sheet = []
    k = 0
    for j in range(ny):
        for i in range(nx):
            s = float(rng.uniform(0.4, 0.7) * pitch)
            jx = float(rng.uniform(0.12, 0.88 - s / pitch) * pitch)
            jy = float(rng.uniform(0.12, 0.88 - s / pitch) * pitch)
            sheet.append((ids[k], i * pitch + jx, -(j * pitch + jy), s))
            k += 1
    return sheet


def sheet_bounds(sheet, margin):
    xs0 = min(m[1] for m in sheet) - margin
    xs1 = max(m[1] + m[3] for m in sheet) + margin
    ys1 = max(m[2] for m in sheet) + margin
    ys0 = min(m[2] - m[3] for m in sheet) - margin
    return xs0, xs1, ys0, ys1


def look_at_pose(rng: np.random.Generator, center, dist_range=(400.0, 900.0), tilt_range=(0.0, 50.0),
                 roll_range=(0.0, 2 * math.pi)):
    """Random camera pose looking at `center` (sheet coordinates, z = 0)."""
    d = rng.uniform(*dist_range)
    tilt = math.radians(rng.uniform(*tilt_range))
    az = rng.uniform(0, 2 * math.pi)
    roll = rng.uniform(*roll_range)
    # camera position in world
    c = np.array([center[0] + d * math.sin(tilt) * math.cos(az), center[1] + d * math.sin(tilt) * math.sin(az),
                  d * math.cos(tilt)])
    zc = np.array([center[0], center[1], 0.0]) - c
    zc /= np.linalg.norm(zc)
    up = np.array([math.cos(roll), math.sin(roll), 0.0])
    xc = np.cross(up, zc)
    if np.linalg.norm(xc) < 1e-6:
        xc = np.cross(np.array([0.0, 0.0, 1.0]), zc)
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc])  # world -> camera
    t = -R @ c
    return R, t


def render_sheet(sheet, K, R, t, H, W, seed=0, ss=3, noise_sigma=4.0, device="cpu", paper_margin=15.0,
                 paper=235.0, ink=25.0, background=128.0):
    """Render one H x W x 3 BGR uint8 frame of `sheet` seen by camera (K, R, t)."""
    dev = torch.device(device)
    Hm = torch.tensor(np.asarray(K, np.float64) @ np.column_stack([R[:, 0], R[:, 1], t]), dtype=torch.float64)
    Hi = torch.linalg.inv(Hm).to(dev)
    offs = (torch.arange(ss, dtype=torch.float64, device=dev) + 0.5) / ss - 0.5
    ys = torch.arange(H, dtype=torch.float64, device=dev)
    xs = torch.arange(W, dtype=torch.float64, device=dev)
    acc = torch.zeros((H, W), dtype=torch.float32, device=dev)
    x0, x1, y0, y1 = sheet_bounds(sheet, paper_margin)
    bits = torch.tensor(np.array([[(int(CODES[m[0]]) >> (15 - i)) & 1 for i in range(16)] for m in sheet], np.float32),
                        device=dev)
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    # low-frequency background gradient
    gx, gy = (torch.rand(2, generator=g) * 2 - 1).tolist()
    bg = background + 20.0 * (gx * (xs[None, :] / W - 0.5) + gy * (ys[:, None] / H - 0.5))
    bg = bg.to(torch.float32)
    for oy in offs.tolist():
        for ox in offs.tolist():
            u = (xs + ox)[None, :].expand(H, W)
            v = (ys + oy)[:, None].expand(H, W)
            den = Hi[2, 0] * u + Hi[2, 1] * v + Hi[2, 2]
            X = (Hi[0, 0] * u + Hi[0, 1] * v + Hi[0, 2]) / den
            Y = (Hi[1, 0] * u + Hi[1, 1] * v + Hi[1, 2]) / den
            front = den * float(np.sign(np.linalg.det(np.asarray(K, np.float64)))) > 0
            val = bg.clone()
            on_paper = (X >= x0) & (X <= x1) & (Y >= y0) & (Y <= y1) & front
            val = torch.where(on_paper, torch.full_like(val, paper), val)
            for mi, (mid, mx, my, ms) in enumerate(sheet):
                cell = ms / 6.0
                cx = torch.floor((X - mx) / cell)
                cy = torch.floor((my - Y) / cell)
                inside = (cx >= 0) & (cx < 6) & (cy >= 0) & (cy < 6) & front
                inner = inside & (cx >= 1) & (cx < 5) & (cy >= 1) & (cy < 5)
                idx = ((cy - 1).clamp(0, 3) * 4 + (cx - 1).clamp(0, 3)).to(torch.long)
                b = bits[mi][idx]
                mval = torch.where(inner & (b > 0.5), torch.full_like(val, paper), torch.full_like(val, ink))
                val = torch.where(inside, mval, val)
            acc += val
    img = acc / float(ss * ss)
    noise = torch.randn((H, W, 3), generator=g, dtype=torch.float32) * noise_sigma
    out = img[:, :, None] + noise.to(dev)
    return out.round().clamp(0, 255).to(torch.uint8)


def v1_frame(idx, K, H=1080, W=1920, base_seed=1234, device="cpu", ss=3, noise_sigma=4.0):
    """Config-2 style frame: the v1 sheet (ids 10..15) from a random pose.
    Returns (bgr uint8 tensor [H,W,3], R, t)."""
    rng = np.random.default_rng(base_seed + idx)
    R, t = look_at_pose(rng, (95.0, -135.0))
    return render_sheet(SHEET_V1, K, R, t, H, W, seed=base_seed + idx, device=device, ss=ss, noise_sigma=noise_sigma), R, t


def grid_frame(idx, K, H=2160, W=3840, base_seed=4321, device="cpu", ss=2, noise_sigma=4.0):
    """Config-3 style frame: 40 markers on a jittered 8x5 grid (the 34 sheet-verified ids, six of them twice), mild
    perspective.  The roll keeps the sheet's long side along the image's (a free roll puts a third of the markers
    outside a 16:9 frame), and the distance keeps the whole sheet inside it: all 40 markers are visible."""
    rng = np.random.default_rng(base_seed + idx)
    sheet = sheet_grid(rng)
    R, t = look_at_pose(rng, (400.0, -250.0), dist_range=(800.0, 900.0), tilt_range=(0.0, 15.0),
                        roll_range=(math.pi / 2 - 0.15, math.pi / 2 + 0.15))
    return render_sheet(sheet, K, R, t, H, W, seed=base_seed + idx, device=device, ss=ss, noise_sigma=noise_sigma), sheet, R, t


def calib_views(V=2000, seed=42, rational=True, noise=0.3, grid=(7, 7)):
    """Config-4 synthetic board views.  Returns obj [V,P,3] f32, img [V,P,2] f32, K, dist(8)."""
    rng = np.random.default_rng(seed)
    K = np.array([[1400.0, 0, 960.0], [0, 1400.0, 540.0], [0, 0, 1.0]])
    k1, k2, p1, p2, k3, k4, k5, k6 = 0.05, -0.2, 0.001, -0.001, 0.1, 0.01, -0.05, 0.02
    if not rational:
        k4 = k5 = k6 = 0.0
    dist = np.array([k1, k2, p1, p2, k3, k4, k5, k6])
    gx, gy = grid
    objp = np.zeros((gx * gy, 3), np.float32)
    objp[:, :2] = np.mgrid[0:gx, 0:gy].T.reshape(-1, 2)
    obj = np.broadcast_to(objp, (V, gx * gy, 3)).copy()
    img = np.zeros((V, gx * gy, 2), np.float32)
    for v in range(V):
        rvec = rng.normal(0, 0.3, 3)
        tvec = np.array([rng.uniform(-3, 3) - (gx - 1) / 2, rng.uniform(-3, 3) - (gy - 1) / 2, rng.uniform(8, 20)])
        Rm = rodrigues(rvec)
        Pc = objp.astype(np.float64) @ Rm.T + tvec
        x = Pc[:, 0] / Pc[:, 2]
        y = Pc[:, 1] / Pc[:, 2]
        r2 = x * x + y * y
        rad = (1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3) / (1 + k4 * r2 + k5 * r2 ** 2 + k6 * r2 ** 3)
        xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        uv = np.stack([K[0, 0] * xd + K[0, 2], K[1, 1] * yd + K[1, 2]], axis=1)
        uv += rng.normal(0, noise, uv.shape)
        img[v] = uv.astype(np.float32)
    return obj, img, K, dist
