"""Chessboard corners for Camera.calibrate (reference camera_sync/camera.py:48-59) without OpenCV.

The reference calls cv.findChessboardCorners(gray, (7,7)) and then cv.cornerSubPix(gray, corners, (11,11), (-1,-1),
(EPS + MAX_ITER, 30, 0.001)).  Here the pixel work runs on the GPU -- X-corner candidates from an integer ChESS
response (camsync_chess_candidates) and the sub-pixel refinement (camsync_corner_subpix, the same kernel that refines
marker corners) -- and this module is the host logic in between: it assembles the pattern's lattice from the
candidates and puts it into a canonical order.

cv.findChessboardCorners' own detector cannot be restated (its source is not available here), and it does not have
to be: the sub-pixel refinement converges to the saddle points from any start within a few pixels, and the intrinsics
cv.calibrateCamera returns do not depend on which corner of the symmetric 7 x 7 lattice comes first, only on the
order being a consistent right-handed walk of the lattice.  tests/test_chessboard.py pins the whole chain against the
calibration the reference stores for its own chessboard photos.
"""
import numpy as np

from .. import _hip

# cv.TERM_CRITERIA_*: criteria tuples are (type, max_iter, eps) like OpenCV's
TERM_CRITERIA_MAX_ITER = 1
TERM_CRITERIA_EPS = 2


def grow_lattice(points, pattern):
    """Assign lattice coordinates to candidate points by growing from a seed along two local step vectors.
    points [K,2] (strongest first), pattern (cols, rows).  Returns a [rows, cols, 2] array of image points with a
    fully populated pattern, or None."""
    P = np.asarray(points, dtype=np.float64)
    K = len(P)
    cols, rows = int(pattern[0]), int(pattern[1])
    if K < cols * rows:
        return None
    D = np.linalg.norm(P[:, None] - P[None], axis=2) + np.eye(K) * 1e18
    best = None
    full = cols * rows

    def grow(seed, u, v):
        coords = {seed: (0, 0)}
        taken = {(0, 0): seed}
        step = {seed: (u, v)}
        queue = [seed]
        while queue:
            i = queue.pop(0)
            ci = coords[i]
            ui, vi = step[i]
            for di, dj, dvec in ((1, 0, ui), (-1, 0, -ui), (0, 1, vi), (0, -1, -vi)):
                cj = (ci[0] + di, ci[1] + dj)
                if cj in taken:
                    continue
                pred = P[i] + dvec
                dist = np.linalg.norm(P - pred, axis=1)
                j = int(np.argmin(dist))
                if j in coords or dist[j] > 0.3 * np.linalg.norm(dvec):
                    continue
                coords[j] = cj
                taken[cj] = j
                # perspective and lens distortion change the step slowly: carry the step just taken forward
                step[j] = ((P[j] - P[i]) * di, vi) if di else (ui, (P[j] - P[i]) * dj)
                queue.append(j)
        return taken

    def rectangle(taken):
        """The cols x rows block of lattice coordinates that is fully populated, as image points, or None."""
        ii = [c[0] for c in taken]
        jj = [c[1] for c in taken]
        for a, b in ((cols, rows), (rows, cols)):
            for i0 in range(min(ii), max(ii) - a + 2):
                for j0 in range(min(jj), max(jj) - b + 2):
                    if all((i0 + i, j0 + j) in taken for i in range(a) for j in range(b)):
                        G = np.array([[P[taken[(i0 + i, j0 + j)]] for i in range(a)] for j in range(b)])
                        return G if (a, b) == (cols, rows) else np.transpose(G, (1, 0, 2))
        return None

    for seed in range(min(K, 16)):
        # Candidate step pairs from the seed's nearest neighbours, nearest first.  On a board seen at a grazing angle a
        # diagonal neighbour is nearer than a lattice neighbour; a pair with a diagonal in it reaches every other corner
        # or numbers the corners in a sheared frame, neither of which fills the pattern's rectangle, so the pairs are
        # tried until one does.
        near = np.argsort(D[seed])[:8]
        pairs = []
        for a in range(len(near)):
            u = P[near[a]] - P[seed]
            lu = np.linalg.norm(u)
            for b in range(a + 1, len(near)):
                w = P[near[b]] - P[seed]
                lw = np.linalg.norm(w)
                if abs(np.dot(u, w)) < 0.9 * lu * lw and 0.4 < lw / lu < 2.5:
                    pairs.append((abs(np.dot(u, w)) >= 0.5 * lu * lw or not 0.5 < lw / lu < 2.0, a + b, a, u, w))
        pairs.sort(key=lambda t: t[:3])  # roughly perpendicular steps of similar length first
        for _, _, _, u, v in pairs[:16]:
            taken = grow(seed, u, v)
            if len(taken) >= full:
                G = rectangle(taken)
                if G is not None:
                    return G
    return None


def canonical_order(G):
    """G [rows, cols, 2]: a right-handed walk of the lattice (row direction x column direction > 0 in image
    coordinates, x to the right and y down) whose first row runs most nearly along +x."""
    G = np.asarray(G, dtype=np.float64)
    du, dv = G[0, -1] - G[0, 0], G[-1, 0] - G[0, 0]
    if du[0] * dv[1] - du[1] * dv[0] < 0:
        G = G[:, ::-1]
    rows, cols = G.shape[:2]
    best = None
    for k in range(4):
        Gk = np.rot90(G, k, axes=(0, 1))
        if Gk.shape[:2] != (rows, cols):
            continue
        d = Gk[0, -1] - Gk[0, 0]
        score = d[0] / np.linalg.norm(d)
        if best is None or score > best[0]:
            best = (score, Gk)
    return best[1]


def _gray_batch(images):
    fs = [np.asarray(im) for im in images]
    for f in fs:
        if f.dtype != np.uint8 or f.ndim not in (2, 3) or (f.ndim == 3 and f.shape[2] not in (1, 3)):
            raise ValueError("expected uint8 images of shape (H, W) or (H, W, 3) in BGR order")
    return fs


def findChessboardCornersBatch(images, patternSize, scales=(1, 2, 3), max_candidates=2048):
    """cv.findChessboardCorners for a list of same-sized images in one launch sequence per scale.
    Returns a list of (found, corners float32 (N,1,2) or None), unrefined (pixel positions)."""
    fs = _gray_batch(images)
    if not fs:
        return []
    H, W = fs[0].shape[:2]
    batch = np.stack([f if f.ndim == 3 else f[..., None] for f in fs])
    n = int(patternSize[0]) * int(patternSize[1])
    out = [None] * len(fs)
    ctx = _hip.get_context(H, W, len(fs))
    for s in scales:
        todo = [i for i, o in enumerate(out) if o is None]
        if not todo or min(H, W) <= 10 * s:
            break
        cands = ctx.chess_candidates(batch[todo], scale=s, max_cand=max_candidates)
        for i, (xy, _) in zip(todo, cands):
            # the pattern's corners are the strongest responses; allow a few stronger impostors
            for k in sorted({n, n + n // 8, n + n // 3, 2 * n}):
                if len(xy) < n:
                    break
                G = grow_lattice(xy[:k], patternSize)
                if G is not None:
                    out[i] = canonical_order(G).reshape(-1, 1, 2).astype(np.float32)
                    break
    return [(o is not None, o) for o in out]


def findChessboardCorners(image, patternSize, flags=None):
    """cv.findChessboardCorners(image, patternSize) -> (found, corners (N,1,2) float32) (camera.py:52)."""
    return findChessboardCornersBatch([image], patternSize)[0]


def cornerSubPix(image, corners, winSize, zeroZone, criteria):
    """cv.cornerSubPix (camera.py:58): refines `corners` (N,1,2) on `image` and returns them.
    zeroZone must be (-1,-1) (no dead zone), winSize square, criteria (type, max_iter, eps)."""
    if tuple(zeroZone) != (-1, -1) or winSize[0] != winSize[1]:
        raise ValueError("only square windows without a zero zone are supported")
    typ, max_iter, eps = criteria
    if not typ & TERM_CRITERIA_MAX_ITER:
        max_iter = 100
    if not typ & TERM_CRITERIA_EPS:
        eps = 0.0
    img = np.asarray(image)
    pts = np.asarray(corners, dtype=np.float32)
    ctx = _hip.get_context(img.shape[0], img.shape[1], 1)
    return ctx.corner_subpix(img, pts.reshape(1, -1, 2), winSize[0], max_iter, eps).reshape(pts.shape)
