// k_quads.hip — K4: the per-contour part of _findMarkerContours
// (ArucoDetector.detectMarkers step 1c; reference camera_sync/aruco.py:256-264):
// approxPolyDP(eps = 0.03 * n, closed), 4 vertices, convex, minimum side,
// distance to the image border, clockwise reorder.
//
// One wavefront per contour.  The contour points sit in the pool as
// x | y<<16.  Douglas–Peucker's farthest-point searches are wave-wide arg-max
// reductions with OpenCV's tie rule (first maximum in traversal order); all
// distances are integers, so the reductions are exact.  The recursion is
// driven from a 16-entry LDS stack: a polygon that needs more than 8 vertices
// before the clean-up pass can never end with 4, so it is dropped early.
#include "internal.h"
#include "wave_reduce.h"

#include <algorithm>

namespace camsync {

namespace {

// The largest distance of the wave and the first position (smallest j) that has it.  Every lane brings its own best
// (d, j) with j = ~0u when it has none.  Frames are at most 4095 pixels a side (check_frame_args), so squared distances
// and cross products stay below 2^26: 32-bit integers, and two 32-bit reductions instead of one on 64-bit keys.
struct Far { uint32_t d, j; };
__device__ __forceinline__ Far wave_farthest(uint32_t d, uint32_t j) {
    const uint32_t m = wave_allreduce(d, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
    const uint32_t jj = wave_allreduce(d == m ? j : 0xFFFFFFFFu, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
    return Far{m, jj};
}

struct IPt { int x, y; };
__device__ __forceinline__ IPt unpack_pt(uint32_t p) { return IPt{(int)(p & 0xFFFFu), (int)(p >> 16)}; }

__global__ __launch_bounds__(64) void k_quad_fit(const ContourRec* __restrict__ contours, unsigned int contourCap,
                                                 const uint32_t* __restrict__ pool, unsigned int poolCap, int B,
                                                 QuadRec* __restrict__ quads, int* __restrict__ qcount,
                                                 Counters* __restrict__ ctr, DetCfg cfg) {
    __shared__ int s_stack[32][2];
    __shared__ IPt s_dst[16];
    // the points of the contour, when they fit: every Douglas-Peucker pass depends on the result of the one
    // before, so with the points in HBM/L2 a contour costs a chain of about ten memory latencies
    constexpr int kCacheN = 1024;
    __shared__ uint32_t s_pts[kCacheN];
    const int lane = threadIdx.x;
    const unsigned int nC = min(ctr->nContours, contourCap);
    for (unsigned int ci = blockIdx.x; ci < nC; ci += gridDim.x) {
        __syncthreads();
        const ContourRec rec = contours[ci];
        const int n = (int)rec.len;
        if (n <= 0 || n > cfg.maxPerim) continue;  // a slot a border-follower wave reserved and did not use (kept contours never exceed maxPerim)
        // a record is only trusted as far as it can be checked: its frame indexes qcount/quads, its points must lie
        // inside the pool (a record that fails this is a bug upstream; the call then fails with CAMSYNC_E_CAPACITY)
        if (rec.frame >= (unsigned int)B || rec.off >= poolCap || rec.len > poolCap - rec.off) {
            if (lane == 0) atomicOr(&ctr->overflow, 128u);
            continue;
        }
        const uint32_t* __restrict__ G = pool + rec.off;
        const bool cached = n <= kCacheN;
        if (cached)
            for (int i = lane; i < n; i += 64) s_pts[i] = G[i];
        __syncthreads();
        auto P = [&](int i) { return cached ? s_pts[i] : G[i]; };
        double eps = (double)n * cfg.accuracyRate;
        eps *= eps;

        // 1. approximately the two farthest points (3 rounds)
        int pos = 0, rs = 0;
        bool le_eps = false;
        for (int it = 0; it < 3; it++) {
            pos += rs;  // (both below n: one conditional subtraction instead of a division)
            if (pos >= n) pos -= n;
            const IPt sp = unpack_pt(P(pos));
            uint32_t bd = 0, bj = 0xFFFFFFFFu;
            for (int j = 1 + lane; j < n; j += 64) {
                int idx = pos + j;
                if (idx >= n) idx -= n;
                const IPt p = unpack_pt(P(idx));
                const int dx = p.x - sp.x, dy = p.y - sp.y;
                const uint32_t d = (uint32_t)(__mul24(dx, dx) + __mul24(dy, dy));
                if (d > bd) { bd = d; bj = (uint32_t)j; }  // (strict: the first of equal distances stays)
            }
            const Far far = wave_farthest(bd, bj);
            const uint32_t maxd = far.d;
            if (maxd > 0) rs = (int)far.j;
            le_eps = (double)maxd <= eps;
        }

        int top = 0, cnt = 0;
        bool reject = false;
        if (!le_eps) {
            const int a = pos;  // (0 <= pos, rs < n)
            int b = rs + a;
            if (b >= n) b -= n;
            if (lane == 0) {
                s_stack[0][0] = b; s_stack[0][1] = a;  // right slice
                s_stack[1][0] = a; s_stack[1][1] = b;  // slice
            }
            top = 2;
        } else {
            if (lane == 0) s_dst[0] = unpack_pt(P(pos));
            cnt = 1;
        }
        __syncthreads();

        // 3. recursive split
        while (top > 0 && !reject) {
            top--;
            const int a = s_stack[top][0], b = s_stack[top][1];
            __syncthreads();
            const IPt sp = unpack_pt(P(a));
            const IPt ep = unpack_pt(P(b));
            int first = a + 1;
            if (first >= n) first = 0;
            bool le;
            int split = 0;
            if (first != b) {
                int L = b - a - 1;
                if (L < 0) L += n;
                const int dx = ep.x - sp.x, dy = ep.y - sp.y;
                uint32_t bd = 0, bj = 0xFFFFFFFFu;
                for (int t = lane; t < L; t += 64) {
                    int idx = first + t;
                    if (idx >= n) idx -= n;
                    const IPt p = unpack_pt(P(idx));
                    const int c = __mul24(p.y - sp.y, dx) - __mul24(p.x - sp.x, dy);
                    const uint32_t d = (uint32_t)(c < 0 ? -c : c);
                    if (d > bd) { bd = d; bj = (uint32_t)t; }
                }
                const Far far = wave_farthest(bd, bj);
                const double maxd = (double)far.d;
                if (far.d > 0) {
                    split = first + (int)far.j;
                    if (split >= n) split -= n;
                }
                le = maxd * maxd <= eps * (double)(__mul24(dx, dx) + __mul24(dy, dy));
            } else {
                le = true;
            }
            if (le) {
                if (cnt >= 8) { reject = true; break; }
                if (lane == 0) s_dst[cnt] = sp;
                cnt++;
            } else {
                if (top + 2 > 16) { reject = true; break; }
                if (lane == 0) {
                    s_stack[top][0] = split; s_stack[top][1] = b;
                    s_stack[top + 1][0] = a; s_stack[top + 1][1] = split;
                }
                top += 2;
            }
            __syncthreads();
        }
        __syncthreads();
        if (reject || cnt < 4) continue;

        // 4. clean-up pass on <= 8 vertices, then the quad tests; scalar work on lane 0
        if (lane == 0) {
            const int count = cnt;
            int new_count = cnt;
            auto rd = [&](int& p) { IPt v = s_dst[p]; if (++p >= count) p = 0; return v; };
            int p = count - 1;
            IPt start_pt = rd(p);
            int wpos = p;
            IPt pt = rd(p);
            for (int i = 0; i < count && new_count > 2; i++) {
                IPt end_pt = rd(p);
                const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
                const double dist = fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
                const double sip = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
                if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
                    new_count--;
                    s_dst[wpos] = start_pt = end_pt;
                    if (++wpos >= count) wpos = 0;
                    pt = rd(p);
                    i++;
                    continue;
                }
                s_dst[wpos] = start_pt = pt;
                if (++wpos >= count) wpos = 0;
                pt = end_pt;
            }
            bool ok = (new_count == 4);
            IPt q[4];
            if (ok) {
#pragma unroll
                for (int i = 0; i < 4; i++) q[i] = s_dst[i];
                // isContourConvex
                IPt prev = q[2], cur = q[3];
                int dx0 = cur.x - prev.x, dy0 = cur.y - prev.y, orientation = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    prev = cur;
                    cur = q[i];
                    const int dx = cur.x - prev.x, dy = cur.y - prev.y;
                    const int dxdy0 = dx * dy0, dydx0 = dy * dx0;
                    orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
                    dx0 = dx;
                    dy0 = dy;
                }
                if (orientation == 3) ok = false;
            }
            if (ok) {
                const int maxdim = max(cfg.W, cfg.H);
                double minDistSq = (double)maxdim * maxdim;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const double ddx = (double)(q[j].x - q[(j + 1) & 3].x), ddy = (double)(q[j].y - q[(j + 1) & 3].y);
                    minDistSq = fmin(minDistSq, ddx * ddx + ddy * ddy);
                }
                const double mcd = (double)n * cfg.minCornerDistanceRate;
                if (minDistSq < mcd * mcd) ok = false;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (q[j].x < cfg.minDistToBorder || q[j].y < cfg.minDistToBorder ||
                        q[j].x > cfg.W - 1 - cfg.minDistToBorder || q[j].y > cfg.H - 1 - cfg.minDistToBorder)
                        ok = false;
            }
            if (ok) {
                // _reorderCandidatesCorners
                const double dx1 = q[1].x - q[0].x, dy1 = q[1].y - q[0].y, dx2 = q[2].x - q[0].x, dy2 = q[2].y - q[0].y;
                if (dx1 * dy2 - dy1 * dx2 < 0.0) { IPt t = q[1]; q[1] = q[3]; q[3] = t; }
                const int slot = atomicAdd(&qcount[rec.frame], 1);
                if (slot >= kQuadCap) {
                    atomicOr(&ctr->overflow, 8u);
                } else {
                    QuadRec r;
#pragma unroll
                    for (int j = 0; j < 4; j++) { r.c[2 * j] = (int16_t)q[j].x; r.c[2 * j + 1] = (int16_t)q[j].y; }
                    r.scanKey = rec.scanKey;
                    r.scale = rec.scale;
                    r.valid = 0;
                    r.clen = n;
                    r.id = -1;
                    r.rot = 0;
                    quads[(size_t)rec.frame * kQuadCap + slot] = r;
                }
            }
        }
    }
}

}  // namespace

void launch_quad_fit(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    // one wave per contour, a few hundred contours per frame; contours differ a lot in length, so the fewer of them a
    // workgroup takes in its fixed stride the better the hardware's scheduler evens the CUs out (4096 1080p frames, ms:
    // 8192 workgroups 2.16, 32768: 1.76, 131072: 1.64, 262144: 1.55, 524288: 1.58)
    const unsigned blocks = (unsigned)std::min(std::max(B * 128, 256), 256 * 1024);
    hipLaunchKernelGGL(k_quad_fit, dim3(blocks), dim3(64), 0, st, b.contours, b.contourCap, b.pool, b.poolCap, B, b.quads,
                       b.qcount, b.counters, c);
}

}  // namespace camsync
