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

// A point is x | y << 16 with 0 <= x, y <= 4095: as a pair of 16-bit integers, the difference of two points is one
// packed subtraction and a squared distance or a cross product with a fixed vector one v_dot2_i32_i16.
typedef short pt16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pt16 pt_diff(uint32_t p, uint32_t q) { return __builtin_bit_cast(pt16, p) - __builtin_bit_cast(pt16, q); }
__device__ __forceinline__ uint32_t sq_norm(pt16 a) { return (uint32_t)__builtin_amdgcn_sdot2(a, a, 0, false); }
// |a x (dx, dy)| = |a.y dx - a.x dy|
__device__ __forceinline__ pt16 cross_coef(int dx, int dy) { return __builtin_bit_cast(pt16, ((uint32_t)(uint16_t)(-dy)) | ((uint32_t)(uint16_t)dx << 16)); }
__device__ __forceinline__ uint32_t abs_cross(pt16 a, pt16 coef) {
    const int c = __builtin_amdgcn_sdot2(a, coef, 0, false);
    return (uint32_t)max(c, -c);
}
// i < 2 n  ->  i mod n
__device__ __forceinline__ int wrap_once(int i, int n) { return (int)min((uint32_t)i, (uint32_t)(i - n)); }

struct IPt { int x, y; };
__device__ __forceinline__ IPt unpack_pt(uint32_t p) { return IPt{(int)(p & 0xFFFFu), (int)(p >> 16)}; }

// lanes of one wavefront exchange data through LDS in program order: only the compiler has to keep it
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Douglas-Peucker's clean-up pass on the <= 8 vertices in dst, then the quad tests of _findMarkerContours (4 vertices,
// convex, minimum side, distance to the image border), the clockwise reorder and the append to the frame's list.
// Scalar work: one lane per contour runs it.
__device__ __forceinline__ void finish_polygon(IPt* __restrict__ s_dst, int cnt, double eps, int n, const ContourRec& rec,
                                               QuadRec* __restrict__ quads, int* __restrict__ qcount,
                                               Counters* __restrict__ ctr, const DetCfg& cfg) {
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
    if (new_count != 4) return;
    IPt q[4];
#pragma unroll
    for (int i = 0; i < 4; i++) q[i] = s_dst[i];
    {   // isContourConvex
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
        if (orientation == 3) return;
    }
    const int maxdim = max(cfg.W, cfg.H);
    double minDistSq = (double)maxdim * maxdim;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const double ddx = (double)(q[j].x - q[(j + 1) & 3].x), ddy = (double)(q[j].y - q[(j + 1) & 3].y);
        minDistSq = fmin(minDistSq, ddx * ddx + ddy * ddy);
    }
    const double mcd = (double)n * cfg.minCornerDistanceRate;
    if (minDistSq < mcd * mcd) return;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (q[j].x < cfg.minDistToBorder || q[j].y < cfg.minDistToBorder ||
            q[j].x > cfg.W - 1 - cfg.minDistToBorder || q[j].y > cfg.H - 1 - cfg.minDistToBorder)
            return;
    // _reorderCandidatesCorners
    const double dx1 = q[1].x - q[0].x, dy1 = q[1].y - q[0].y, dx2 = q[2].x - q[0].x, dy2 = q[2].y - q[0].y;
    if (dx1 * dy2 - dy1 * dx2 < 0.0) { IPt t = q[1]; q[1] = q[3]; q[3] = t; }
    const int slot = atomicAdd(&qcount[rec.frame], 1);
    if (slot >= cfg.quadCap) {
        atomicOr(&ctr->overflow, 8u);
        atomicMax(&ctr->needQuads, (unsigned int)slot + 1u);
        return;
    }
    QuadRec r;
#pragma unroll
    for (int j = 0; j < 4; j++) { r.c[2 * j] = (int16_t)q[j].x; r.c[2 * j + 1] = (int16_t)q[j].y; }
    r.scanKey = rec.scanKey;
    r.scale = rec.scale;
    r.valid = 0;
    r.clen = n;
    r.id = -1;
    r.rot = 0;
    quads[(size_t)rec.frame * cfg.quadCap + slot] = r;
}

__global__ __launch_bounds__(64) void k_quad_fit(const ContourRec* __restrict__ contours, const uint32_t* __restrict__ list,
                                                 const unsigned int* __restrict__ count, unsigned int contourCap,
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
    const unsigned int nC = min(*count, contourCap);
    for (unsigned int li = blockIdx.x; li < nC; li += gridDim.x) {
        __syncthreads();
        const ContourRec rec = contours[min(list[li], contourCap - 1u)];
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
            const uint32_t sp = P(pos);
            uint32_t bd = 0, bj = 0xFFFFFFFFu;
            for (int j = 1 + lane; j < n; j += 64) {
                const uint32_t d = sq_norm(pt_diff(P(wrap_once(pos + j, n)), sp));
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
                const pt16 coef = cross_coef(dx, dy);
                const uint32_t spw = P(a);
                uint32_t bd = 0, bj = 0xFFFFFFFFu;
                for (int t = lane; t < L; t += 64) {
                    const uint32_t d = abs_cross(pt_diff(P(wrap_once(first + t, n)), spw), coef);
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
            // every slice still on the stack ends in at least one more vertex: with more than 8 in sight the polygon is
            // lost already (the test above would say so some rounds later)
            if (cnt + top > 8) { reject = true; break; }
            __syncthreads();
        }
        __syncthreads();
        if (reject || cnt < 4) continue;

        // 4. clean-up pass on <= 8 vertices, then the quad tests; scalar work on lane 0
        if (lane == 0) finish_polygon(s_dst, cnt, eps, n, rec, quads, qcount, ctr, cfg);
    }
}


// The same for contours of up to kFitWave = 2048 points (nearly all that reach this kernel: marker outlines are a
// few hundred to a thousand points): always in LDS, the farthest point and its position as one key (distance << 11 |
// 2047 - offset: one wave reduction per round instead of two; a closed chain of 2048 points spans at most 1024 pixels,
// so cross products and squared lengths stay below 2^21), four points per lane in flight per scan step.
constexpr int kFitWaveN = 2048, kFitWaveKS = 11;
__global__ __launch_bounds__(64) void k_quad_wave(const ContourRec* __restrict__ contours, const uint32_t* __restrict__ list,
                                                  const unsigned int* __restrict__ count, unsigned int contourCap,
                                                  const uint32_t* __restrict__ pool, unsigned int poolCap, int B,
                                                  QuadRec* __restrict__ quads, int* __restrict__ qcount,
                                                  Counters* __restrict__ ctr, DetCfg cfg) {
    constexpr int KS = kFitWaveKS;
    constexpr uint32_t KMASK = (1u << KS) - 1u;
    static_assert(kFitWaveN <= (1 << KS) && (uint64_t)kFitWaveN * kFitWaveN / 2 <= (1ull << (32 - KS)), "key layout");
    __shared__ int s_stack[32][2];
    __shared__ IPt s_dst[16];
    __shared__ uint32_t s_pts[kFitWaveN + 256];  // (+ what a scan's last step may read past the end: read, never used)
    const int lane = threadIdx.x;
    const unsigned int nC = min(*count, contourCap);
    const auto wave_max = [](uint32_t v) { return wave_allreduce(v, [](uint32_t x, uint32_t y) { return x > y ? x : y; }); };
    for (unsigned int li = blockIdx.x; li < nC; li += gridDim.x) {
        const ContourRec rec = contours[min(list[li], contourCap - 1u)];
        const int n = (int)rec.len;
        if (n <= 0 || n > kFitWaveN || n > cfg.maxPerim || rec.frame >= (unsigned int)B || rec.off >= poolCap || rec.len > poolCap - rec.off)
            continue;  // (k_quad_bin lists none of these)
        const uint32_t* __restrict__ G = pool + rec.off;
        wave_lds_order();
        // all of the contour's loads in flight at once, eight per lane and step (a loop of load, wait, store pays one
        // memory latency per 64 points)
        for (int i0 = 0; i0 < n; i0 += 512) {
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = (i0 + 64 * q + lane < n) ? G[i0 + 64 * q + lane] : 0u;
#pragma unroll
            for (int q = 0; q < 8; q++)
                if (i0 + 64 * q + lane < n) s_pts[i0 + 64 * q + lane] = v[q];
        }
        wave_lds_order();
        double eps = (double)n * cfg.accuracyRate;
        eps *= eps;

        // 1. approximately the two farthest points (3 rounds)
        int pos = 0, rs = 0;
        bool le_eps = false;
        for (int it = 0; it < 3; it++) {
            pos += rs;
            if (pos >= n) pos -= n;
            const uint32_t sp = s_pts[pos];
            uint32_t best = 0;
            for (int j = 1 + lane; j < n; j += 256) {  // (uniform trip count: j - lane is)
                const uint32_t p0 = s_pts[wrap_once(pos + j, n)], p1 = s_pts[wrap_once(pos + j + 64, n)];
                const uint32_t p2 = s_pts[wrap_once(pos + j + 128, n)], p3 = s_pts[wrap_once(pos + j + 192, n)];
                const uint32_t k0 = j < n ? (sq_norm(pt_diff(p0, sp)) << KS) | (KMASK - (uint32_t)j) : 0u;
                const uint32_t k1 = j + 64 < n ? (sq_norm(pt_diff(p1, sp)) << KS) | (KMASK - (uint32_t)(j + 64)) : 0u;
                const uint32_t k2 = j + 128 < n ? (sq_norm(pt_diff(p2, sp)) << KS) | (KMASK - (uint32_t)(j + 128)) : 0u;
                const uint32_t k3 = j + 192 < n ? (sq_norm(pt_diff(p3, sp)) << KS) | (KMASK - (uint32_t)(j + 192)) : 0u;
                best = max(max(best, k0), max(k1, max(k2, k3)));
            }
            best = wave_max(best);
            const uint32_t maxd = best >> KS;
            if (maxd > 0) rs = (int)(KMASK - (best & KMASK));
            le_eps = (double)maxd <= eps;
        }
        if (le_eps) continue;  // (one vertex: no quad)

        int top = 2, cnt = 0;
        bool reject = false;
        {
            const int a = pos;
            int b = rs + a;
            if (b >= n) b -= n;
            if (lane == 0) {
                s_stack[0][0] = b; s_stack[0][1] = a;  // right slice
                s_stack[1][0] = a; s_stack[1][1] = b;  // slice
            }
        }
        wave_lds_order();

        // 2. recursive split
        while (top > 0) {
            top--;
            const int a = s_stack[top][0], b = s_stack[top][1];
            const uint32_t spw = s_pts[a];
            const IPt sp = unpack_pt(spw), ep = unpack_pt(s_pts[b]);
            int first = a + 1;
            if (first >= n) first = 0;
            bool le = true;
            int split = 0;
            if (first != b) {
                int L = b - a - 1;
                if (L < 0) L += n;
                const int dx = ep.x - sp.x, dy = ep.y - sp.y;
                const pt16 coef = cross_coef(dx, dy);
                uint32_t best = 0;
                for (int t = lane; t < L; t += 256) {
                    const uint32_t p0 = s_pts[wrap_once(first + t, n)], p1 = s_pts[wrap_once(first + t + 64, n)];
                    const uint32_t p2 = s_pts[wrap_once(first + t + 128, n)], p3 = s_pts[wrap_once(first + t + 192, n)];
                    const uint32_t k0 = t < L ? (abs_cross(pt_diff(p0, spw), coef) << KS) | (KMASK - (uint32_t)t) : 0u;
                    const uint32_t k1 = t + 64 < L ? (abs_cross(pt_diff(p1, spw), coef) << KS) | (KMASK - (uint32_t)(t + 64)) : 0u;
                    const uint32_t k2 = t + 128 < L ? (abs_cross(pt_diff(p2, spw), coef) << KS) | (KMASK - (uint32_t)(t + 128)) : 0u;
                    const uint32_t k3 = t + 192 < L ? (abs_cross(pt_diff(p3, spw), coef) << KS) | (KMASK - (uint32_t)(t + 192)) : 0u;
                    best = max(max(best, k0), max(k1, max(k2, k3)));
                }
                best = wave_max(best);
                const double maxd = (double)(best >> KS);
                if ((best >> KS) > 0) split = wrap_once(first + (int)(KMASK - (best & KMASK)), n);
                le = maxd * maxd <= eps * (double)(__mul24(dx, dx) + __mul24(dy, dy));
            }
            wave_lds_order();  // (the stack entry has been read)
            if (le) {
                if (lane == 0) s_dst[cnt] = sp;
                cnt++;
            } else {
                if (lane == 0) {
                    s_stack[top][0] = split; s_stack[top][1] = b;
                    s_stack[top + 1][0] = a; s_stack[top + 1][1] = split;
                }
                top += 2;
            }
            // every slice still on the stack ends in at least one more vertex: with more than 8 in sight the polygon is
            // lost (approxPolyDP's clean-up pass only removes vertices, and the quad test wants 4 of at most 8)
            if (cnt + top > 8) { reject = true; break; }
            wave_lds_order();
        }
        if (reject || cnt < 4) continue;
        wave_lds_order();
        // 3. clean-up pass on <= 8 vertices, then the quad tests; scalar work on lane 0
        if (lane == 0) finish_polygon(s_dst, cnt, eps, n, rec, quads, qcount, ctr, cfg);
    }
}


// ---- short contours: four per wavefront -----------------------------------------------------------------
// A textured scene has hundreds of contours per frame that are a hundred-odd points long (the reference's desk photo:
// 500 per frame, 165 points on average).  With a wavefront per contour such a contour is two or three points per lane,
// and the cost is the fixed part of every Douglas-Peucker round (two wave reductions, the stack, the barriers): ~1800
// instructions per contour.  Here a contour gets a row of 16 lanes and a wavefront carries four contours through the
// same rounds in lock-step: the reductions stay inside the row (DPP row rotations), the stack and the points are per
// row in LDS, and the scalar tail runs on four lanes at once.  Same arithmetic, same tie rules, same results.
constexpr int kFitShort = 256;  // longest contour the rows take (bins: <= 64 and <= 128 in rows of 8 lanes, <= 256 in rows of 16)
// A row keeps its contour twice over in LDS (point i also at i + n), so that a scan that starts anywhere on the closed
// curve runs over consecutive words without wrapping, four of them per lane and round in flight at once (one LDS
// latency per four points instead of one per point); a tail of four words per lane is what the last round may read
// past a scan's end (read, never used).
constexpr int kQuadBins = 5;
constexpr int kFitWave = 2048;  // longest contour the level-by-level kernel takes with a whole wavefront

// reduction over the 16 lanes of a row, result in every lane of the row (all lanes of the wave must be active)
template <class Op>
__device__ __forceinline__ uint32_t row_allreduce(uint32_t v, Op op) {
    v = wave_detail::step<8>(v, op);
    v = wave_detail::step<4>(v, op);
    v = wave_detail::step<2>(v, op);
    v = wave_detail::step<1>(v, op);
    return v;
}
// the same over the 8 lanes of a half row: lane i with 7 - i (row_half_mirror), then with i ^ 1 and i ^ 2 (quad_perm)
template <class Op>
__device__ __forceinline__ uint32_t half_row_allreduce(uint32_t v, Op op) {
    v = op(v, wave_detail::dpp_mov<0x141>(v));
    v = op(v, wave_detail::dpp_mov<0xB1>(v));
    v = op(v, wave_detail::dpp_mov<0x4E>(v));
    return v;
}
__device__ __forceinline__ Far row_farthest(uint32_t d, uint32_t j) {
    const uint32_t m = row_allreduce(d, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
    const uint32_t jj = row_allreduce(d == m ? j : 0xFFFFFFFFu, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
    return Far{m, jj};
}

// Douglas-Peucker level by level instead of slice by slice.  The recursion splits every slice independently of the
// others and emits the start points of the final slices in the order of the curve, so the slices can be processed a
// generation at a time: one pass over the contour's points finds, for every slice at once, the point farthest from its
// chord (a lane walks its points in curve order, keeps the best of the slice it is in and hands it to the slice's LDS
// cell when it crosses a slice boundary), then one lane per slice decides leaf or split, and a ballot gives the new
// boundaries their places.  A polygon that has more than 8 slices at any time is lost (approxPolyDP's clean-up pass
// only removes vertices, and a polygon of more than 8 cannot come down to 4... it would be rejected by the depth-first
// formulation when the ninth vertex is emitted): at most three passes for a contour that is no quad, two and a
// confirming one for a quad, where the depth-first formulation needed eight to fifteen rounds with their fixed costs.
// The farthest point and its position travel as one key, distance << 9 | (511 - offset): the larger distance wins, then
// the smaller offset (OpenCV keeps the first of equal distances).  A closed chain of n <= 512 points spans at most 256
// pixels, so distances (cross products, squared lengths) stay below 2^17 and the key below 2^26.
// LPC lanes per contour (8 or 16), contours of up to NMAX points; the contours to take are list[0 .. *count).
template <int LPC, int NMAX, int KS>
__global__ __launch_bounds__(64) void k_quad_rows(const ContourRec* __restrict__ contours, const uint32_t* __restrict__ list,
                                                  const unsigned int* __restrict__ count, unsigned int contourCap,
                                                  const uint32_t* __restrict__ pool, unsigned int poolCap, int B,
                                                  QuadRec* __restrict__ quads, int* __restrict__ qcount,
                                                  Counters* __restrict__ ctr, DetCfg cfg) {
    static_assert(NMAX % LPC == 0 && (LPC == 8 || LPC == 16 || LPC == 64), "rows");
    // key = distance << KS | (KMASK - offset): offsets need KS bits, distances (a closed chain of n points spans at most
    // n / 2 pixels: cross products and squared lengths below n^2 / 2) the rest of the 32
    static_assert(NMAX <= (1 << KS) && (uint64_t)NMAX * NMAX / 2 <= (1ull << (32 - KS)), "key layout");
    constexpr uint32_t KMASK = (1u << KS) - 1u;
    constexpr int G = 64 / LPC;                  // contours per wavefront
    constexpr int kRow = 2 * NMAX + 4 * LPC;     // a contour twice over + what a scan's last round may read past its end
    __shared__ uint32_t s_ptsAll[G][kRow];
    __shared__ uint32_t s_bestAll[G][8];
    __shared__ int s_bndAll[G][2][18];
    __shared__ IPt s_dstAll[G][16];
    const int lane = threadIdx.x, g = lane / LPC, l = lane % LPC;
    uint32_t* const s_pts = s_ptsAll[g];
    uint32_t* const s_best = s_bestAll[g];
    IPt* const s_dst = s_dstAll[g];
    const unsigned int nC = min(*count, contourCap);
    const auto row_max = [](uint32_t v) {
        const auto mx = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
        return LPC == 64 ? wave_allreduce(v, mx) : LPC == 16 ? row_allreduce(v, mx) : half_row_allreduce(v, mx);
    };
    // The next batch's records and points are fetched while this one is worked on (a record, then the points it names:
    // two dependent trips to memory per batch, which a wavefront that takes its batches one after the other would sit
    // out each time -- that wait was a third of this kernel's time).
    const unsigned int stride = gridDim.x * (unsigned int)G;
    constexpr int kPre = NMAX / LPC;  // points per lane
    const auto load_rec = [&](unsigned int b) {
        ContourRec r;
        r.frame = 0; r.pos = 0; r.len = 0; r.off = 0; r.scale = 0; r.hole = 0; r.scanKey = 0;
        const unsigned int ci = b + (unsigned int)g;
        if (b < nC && ci < nC) r = contours[min(list[ci], contourCap - 1u)];
        return r;
    };
    // (empty slots, and the contours the wavefront-per-contour kernel takes: it also reports records that fail its checks)
    const auto is_live = [&](const ContourRec& r) {
        const int n = (int)r.len;
        return n > 0 && n <= NMAX && n <= cfg.maxPerim && r.frame < (unsigned int)B && r.off < poolCap && r.len <= poolCap - r.off;
    };
    uint32_t pre[kPre];
    const auto load_points = [&](const ContourRec& r, int n) {
        const uint32_t* __restrict__ G = pool + r.off;
#pragma unroll
        for (int q = 0; q < kPre; q++) pre[q] = (l + LPC * q < n) ? G[l + LPC * q] : 0u;
    };
    unsigned int base = blockIdx.x * (unsigned int)G;
    ContourRec recN = load_rec(base);
    bool liveN = is_live(recN);
    int nN = liveN ? (int)recN.len : 0;
    load_points(recN, nN);
    ContourRec recNN = load_rec(base + stride);
    for (; base < nC; base += stride) {
        const ContourRec rec = recN;
        const bool live = liveN;
        const int n = nN;
        wave_lds_order();  // the previous batch's readers are done with the rows' LDS
#pragma unroll
        for (int q = 0; q < kPre; q++)
            if (l + LPC * q < n) {
                s_pts[l + LPC * q] = pre[q];
                s_pts[l + LPC * q + n] = pre[q];
            }
        wave_lds_order();
        recN = recNN;
        liveN = is_live(recN);
        nN = liveN ? (int)recN.len : 0;
        load_points(recN, nN);
        recNN = load_rec(base + 2u * stride);
        if (!__any(live)) continue;
        double eps = (double)n * cfg.accuracyRate;
        eps *= eps;
#if defined(QF_KNOCK) && QF_KNOCK == 4
        if (s_pts[l] != 0x12345) continue;
#endif

        // 1. approximately the two farthest points (3 rounds)
        int pos = 0, rs = 0;
        bool le_eps = false;
        for (int it = 0; it < 3; it++) {
            pos += rs;
            if (pos >= n) pos -= n;
            const uint32_t sp = s_pts[pos];
            uint32_t best = 0;
            for (int j = 1 + l; __any(j < n); j += 4 * LPC) {
                const uint32_t* q = s_pts + pos + j;
                const uint32_t p0 = q[0], p1 = q[LPC], p2 = q[2 * LPC], p3 = q[3 * LPC];
                const uint32_t k0 = j < n ? (sq_norm(pt_diff(p0, sp)) << KS) | (KMASK - (uint32_t)j) : 0u;
                const uint32_t k1 = j + LPC < n ? (sq_norm(pt_diff(p1, sp)) << KS) | (KMASK - (uint32_t)(LPC + j)) : 0u;
                const uint32_t k2 = j + 2 * LPC < n ? (sq_norm(pt_diff(p2, sp)) << KS) | (KMASK - (uint32_t)(2 * LPC + j)) : 0u;
                const uint32_t k3 = j + 3 * LPC < n ? (sq_norm(pt_diff(p3, sp)) << KS) | (KMASK - (uint32_t)(3 * LPC + j)) : 0u;
                best = max(max(best, k0), max(k1, max(k2, k3)));
            }
            best = row_max(best);
            const uint32_t maxd = best >> KS;
            if (maxd > 0) rs = (int)(KMASK - (best & KMASK));
            le_eps = (double)maxd <= eps;
        }

        // 2. the slices of the closed curve, in curve order from a = pos: boundaries 0, rs, n (offsets from a)
        const int a = pos;
        bool more = live && !le_eps;  // (a contour that collapses to a point has one vertex: no quad)
#if defined(QF_KNOCK) && QF_KNOCK == 1
        more = false;
#endif
#if defined(QF_KNOCK) && QF_KNOCK == 3
        continue;
#endif
        int nseg = 2, cur = 0, cnt = 0;
        if (l == 0) { s_bndAll[g][0][0] = 0; s_bndAll[g][0][1] = rs; s_bndAll[g][0][2] = n; }
        while (__any(more)) {
            const int* bnd = s_bndAll[g][cur];
            int* nxt = s_bndAll[g][cur ^ 1];
            if (l < 8) s_best[l] = 0;
            wave_lds_order();
            // ---- one pass over the points: the farthest point of every slice ----
            {
                const int ne = more ? n : 0;
                int k = 0, Bk = 0, Bn = more ? bnd[1] : 0;
                uint32_t spw = s_pts[a], best = 0;
                pt16 coef = cross_coef(0, 0);
                if (more) {
                    const IPt s0 = unpack_pt(spw), e0 = unpack_pt(s_pts[a + Bn]);
                    coef = cross_coef(e0.x - s0.x, e0.y - s0.y);
                }
                for (int u = l; __any(u < ne); u += 4 * LPC) {
                    const uint32_t* q = s_pts + a + u;
                    const uint32_t pp[4] = {q[0], q[LPC], q[2 * LPC], q[3 * LPC]};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int uu = u + LPC * i;
                        if (uu < ne) {
                            while (uu >= Bn) {  // into the next slice (boundaries ascend, the last one is n > uu)
                                if (best) atomicMax(&s_best[k], best);
                                best = 0;
                                k++;
                                Bk = Bn;
                                Bn = bnd[k + 1];
                                spw = s_pts[a + Bk];
                                const IPt s0 = unpack_pt(spw), e0 = unpack_pt(s_pts[a + Bn]);
                                coef = cross_coef(e0.x - s0.x, e0.y - s0.y);
                            }
                            if (uu > Bk) best = max(best, (abs_cross(pt_diff(pp[i], spw), coef) << KS) | (KMASK - (uint32_t)(uu - Bk - 1)));
                        }
                    }
                }
                if (best) atomicMax(&s_best[k], best);
            }
            wave_lds_order();
            // ---- one lane per slice: leaf or split ----
            const bool mine = more && l < nseg;
            int Bj = 0, Bj1 = 0, t = 0;
            bool split = false;
            if (mine) {
                Bj = bnd[l];
                Bj1 = bnd[l + 1];
                if (Bj1 - Bj > 1) {
                    const uint32_t key = s_best[l];
                    const double maxd = (double)(key >> KS);
                    t = (int)(KMASK - (key & KMASK));
                    const IPt sp = unpack_pt(s_pts[a + Bj]), ep = unpack_pt(s_pts[a + Bj1]);
                    const int dx = ep.x - sp.x, dy = ep.y - sp.y;
                    split = !(maxd * maxd <= eps * (double)(__mul24(dx, dx) + __mul24(dy, dy)));
                }
            }
            const uint32_t rowSplits = (uint32_t)(__ballot(split) >> (LPC * g)) & 0xFFu;  // (lanes 0..7 of the row: a polygon in progress has at most 8 slices)
            const int nsplit = __popc(rowSplits), pre = __popc(rowSplits & ((1u << l) - 1u));
            const int newN = nseg + nsplit;
            if (more) {
                if (newN > 8) {
                    more = false;                       // more than 8 vertices in sight: no quad
                } else if (nsplit == 0) {
                    if (mine) s_dst[l] = unpack_pt(s_pts[a + Bj]);  // the polygon: the slices' start points in curve order
                    cnt = nseg;
                    more = false;
                } else {
                    if (mine) {
                        nxt[l + pre] = Bj;
                        if (split) nxt[l + pre + 1] = Bj + 1 + t;
                    }
                    if (l == 0) nxt[newN] = n;
                    nseg = newN;
                }
            }
            cur ^= 1;
            wave_lds_order();
        }
#if defined(QF_KNOCK) && QF_KNOCK == 2
        if (cnt == 1000)
#endif
        // 3. clean-up pass and quad tests: one lane per row
        if (l == 0 && live && cnt >= 4) finish_polygon(s_dst, cnt, eps, n, rec, quads, qcount, ctr, cfg);
    }
}


// Sort the contours into bins by length, so that the wavefronts of k_quad_rows carry contours of similar length through
// their rounds in lock-step (a batch takes as long as its longest contour) and the longest go to k_quad_fit:
// bin 0: up to 64 points, 1: up to 128, 2: up to 256 (kFitShort), 3: up to 2048 (kFitWave), 4: longer.
// lists: [kQuadBins][contourCap] contour indices.  A workgroup sorts kBinPerBlock records and reserves its places in
// the five lists with one atomic each (one per wavefront and bin was 100 k atomics on five addresses: 0.5 ms).
constexpr int kBinPerThread = 8, kBinPerBlock = 256 * kBinPerThread;
__global__ __launch_bounds__(256) void k_quad_bin(const ContourRec* __restrict__ contours, unsigned int contourCap, unsigned int poolCap,
                                                  int B, int maxPerim, uint32_t* __restrict__ lists, Counters* __restrict__ ctr) {
    __shared__ unsigned int s_n[kQuadBins], s_at[kQuadBins];
    const unsigned int nC = min(ctr->nContours, contourCap);
    for (unsigned int base = blockIdx.x * (unsigned int)kBinPerBlock; base < nC; base += gridDim.x * (unsigned int)kBinPerBlock) {
        if (threadIdx.x < kQuadBins) s_n[threadIdx.x] = 0;
        __syncthreads();
        int bin[kBinPerThread];
        unsigned int at[kBinPerThread];
#pragma unroll
        for (int q = 0; q < kBinPerThread; q++) {
            const unsigned int i = base + (unsigned int)q * 256u + threadIdx.x;
            bin[q] = -1;
            at[q] = 0;
            if (i < nC) {
                const ContourRec r = contours[i];
                const int n = (int)r.len;
                if (n > 0 && n <= maxPerim) {  // (else: a slot a border-follower wave reserved and did not use)
                    // a record is only trusted as far as it can be checked: its frame indexes qcount/quads, its points must
                    // lie inside the pool (a record that fails this is a bug upstream; the call then fails with CAMSYNC_E_CAPACITY)
                    if (r.frame >= (unsigned int)B || r.off >= poolCap || r.len > poolCap - r.off) atomicOr(&ctr->overflow, 128u);
                    else bin[q] = n <= 64 ? 0 : n <= 128 ? 1 : n <= kFitShort ? 2 : n <= kFitWave ? 3 : 4;
                }
            }
            if (bin[q] >= 0) at[q] = atomicAdd(&s_n[bin[q]], 1u);
        }
        __syncthreads();
        if (threadIdx.x < kQuadBins && s_n[threadIdx.x]) s_at[threadIdx.x] = atomicAdd(&ctr->nBin[threadIdx.x * 32], s_n[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kBinPerThread; q++)
            if (bin[q] >= 0) lists[(size_t)bin[q] * contourCap + s_at[bin[q]] + at[q]] = base + (unsigned int)q * 256u + threadIdx.x;
        __syncthreads();
    }
}

}  // namespace

void launch_quad_fit(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    // the contours by length (k_quad_bin), then the short ones eight or four to a wavefront (k_quad_rows), the long ones
    // one to a wavefront (k_quad_fit).  Contours differ a lot in cost, so the fewer of them a workgroup takes in its
    // fixed stride the better the hardware's scheduler evens the CUs out (4096 1080p frames, wavefront per contour, ms:
    // 8192 workgroups 2.16, 32768: 1.76, 131072: 1.64, 262144: 1.55, 524288: 1.58)
    const unsigned blocks = (unsigned)std::min(std::max(B * 128, 256), 256 * 1024);
    const unsigned binBlocks = (unsigned)std::min<size_t>(std::max<size_t>((size_t)B * 4096 / kBinPerBlock, 16), 4096);
    hipLaunchKernelGGL(k_quad_bin, dim3(binBlocks), dim3(256), 0, st, b.contours, b.contourCap, b.poolCap, B, c.maxPerim, b.binIdx, b.counters);
    const size_t cap = b.contourCap;
    hipLaunchKernelGGL((k_quad_rows<8, 64, 9>), dim3(blocks), dim3(64), 0, st, b.contours, b.binIdx, &b.counters->nBin[0], b.contourCap,
                       b.pool, b.poolCap, B, b.quads, b.qcount, b.counters, c);
    hipLaunchKernelGGL((k_quad_rows<8, 128, 9>), dim3(blocks), dim3(64), 0, st, b.contours, b.binIdx + cap, &b.counters->nBin[32], b.contourCap,
                       b.pool, b.poolCap, B, b.quads, b.qcount, b.counters, c);
    hipLaunchKernelGGL((k_quad_rows<16, kFitShort, 9>), dim3(blocks), dim3(64), 0, st, b.contours, b.binIdx + 2 * cap, &b.counters->nBin[64],
                       b.contourCap, b.pool, b.poolCap, B, b.quads, b.qcount, b.counters, c);
    // (a whole wavefront per contour going level by level, k_quad_rows<64, 2048, 11>, was slower than k_quad_wave's
    // depth-first rounds on the same contours: 2.1 against 1.2 ms on the photo scene, 1.7 against 1.2 ms on the bench scene)
    hipLaunchKernelGGL(k_quad_wave, dim3(blocks), dim3(64), 0, st, b.contours, b.binIdx + 3 * cap, &b.counters->nBin[96], b.contourCap,
                       b.pool, b.poolCap, B, b.quads, b.qcount, b.counters, c);
    hipLaunchKernelGGL(k_quad_fit, dim3(blocks), dim3(64), 0, st, b.contours, b.binIdx + 4 * cap, &b.counters->nBin[128], b.contourCap,
                       b.pool, b.poolCap, B, b.quads, b.qcount, b.counters, c);
}

}  // namespace camsync
