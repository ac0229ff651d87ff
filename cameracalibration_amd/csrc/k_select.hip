// k_select.hip — K5 and the bookkeeping half of K6: filterTooCloseCandidates
// and identifyCandidates of ArucoDetector.detectMarkers (reference
// camera_sync/aruco.py:256-264), one wavefront per frame.
//
// Input: the frame's identified quad candidates in arbitrary order.  They are
// put in the reference detector's order — scale ascending, contours in reverse
// discovery order, then a stable sort by descending perimeter — which is the
// total order (perimeter desc, scale asc, scan key desc).  Grouping of
// near-duplicates, the choice of the representative, the fall-back "close"
// contours, the containment hierarchy and the depth-ordered acceptance follow
// the detector step by step; pair tests run one pair per lane, the order
// dependent bookkeeping runs on lane 0.
#include "internal.h"

#include <algorithm>

namespace camsync {

namespace {

struct P2 { float x, y; };

__device__ __forceinline__ float l2sqr(float ax, float ay, float bx, float by) {
    const float dx = ax - bx, dy = ay - by;
    return dx * dx + dy * dy;
}

__device__ float avg_distance(const float* m1, const float* m2) {
    float minDistSq = 3.402823466e+38f;
#pragma unroll
    for (int fc = 0; fc < 4; fc++) {
        float distSq = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int modC = (c + fc) & 3;
            distSq += l2sqr(m1[2 * modC], m1[2 * modC + 1], m2[2 * c], m2[2 * c + 1]);
        }
        distSq /= 4.f;
        minDistSq = fminf(minDistSq, distSq);
    }
    return sqrtf(minDistSq);
}

__device__ float perimeter4(const float* c) {
    float p = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) p += sqrtf(l2sqr(c[2 * i], c[2 * i + 1], c[2 * ((i + 1) & 3)], c[2 * ((i + 1) & 3) + 1]));
    return p;
}

// pointPolygonTest(measureDist = false) on a 4-gon with float vertices
__device__ int point_in_quad(const float* cnt, float px, float py) {
    int counter = 0;
    float vx = cnt[6], vy = cnt[7];
    for (int i = 0; i < 4; i++) {
        const float v0x = vx, v0y = vy;
        vx = cnt[2 * i];
        vy = cnt[2 * i + 1];
        if ((v0y <= py && vy <= py) || (v0y > py && vy > py) || (v0x < px && vx < px)) {
            if (py == vy && (px == vx || (py == v0y && ((v0x <= px && px <= vx) || (vx <= px && px <= v0x))))) return 0;
            continue;
        }
        double dist = (double)(py - v0y) * (vx - v0x) - (double)(px - v0x) * (vy - v0y);
        if (dist == 0) return 0;
        if (vy < v0y) dist = -dist;
        counter += dist > 0;
    }
    return (counter & 1) ? 1 : -1;
}

__device__ bool marker_in_marker(const float* m1, const float* m2) {
    // a corner outside m2's bounding box is outside m2 (point_in_quad says so for such a point on every edge), and
    // nearly all pairs of a frame end here: eight comparisons instead of four polygon tests
    const float x0 = fminf(fminf(m2[0], m2[2]), fminf(m2[4], m2[6])), x1 = fmaxf(fmaxf(m2[0], m2[2]), fmaxf(m2[4], m2[6]));
    const float y0 = fminf(fminf(m2[1], m2[3]), fminf(m2[5], m2[7])), y1 = fmaxf(fmaxf(m2[1], m2[3]), fmaxf(m2[5], m2[7]));
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (m1[2 * k] < x0 || m1[2 * k] > x1 || m1[2 * k + 1] < y0 || m1[2 * k + 1] > y1) return false;
    return point_in_quad(m2, m1[0], m1[1]) > 0 && point_in_quad(m2, m1[2], m1[3]) > 0 &&
           point_in_quad(m2, m1[4], m1[5]) > 0 && point_in_quad(m2, m1[6], m1[7]) > 0;
}

// NQ: capacity of the LDS working set.  Two instances run over every frame: the small one takes the frames with
// up to 256 candidates (nearly all of them, at a quarter of the LDS), the large one the rest.
template <int NQ>
__global__ __launch_bounds__(64) void k_select(const QuadRec* __restrict__ quads, const int* __restrict__ qcount,
                                               DetCfg cfg, int32_t* __restrict__ outIds, float* __restrict__ outCorners, float* __restrict__ rawCorners,
                                               int32_t* __restrict__ outCounts, float* __restrict__ outRejected,
                                               int32_t* __restrict__ outRejCounts, MarkerRec* __restrict__ markers,
                                               unsigned int markerCap, Counters* __restrict__ ctr) {
    __shared__ float s_c[NQ][8];      // corners, sorted order
    __shared__ float s_perim[NQ];
    __shared__ int s_rank[NQ];
    __shared__ short s_gid[NQ];
    __shared__ short s_closeOf[NQ];   // representative this candidate is a fall-back of, or -1
    __shared__ short s_id[NQ];
    __shared__ uint8_t s_rot[NQ], s_valid[NQ], s_isSel[NQ];
    __shared__ short s_sel[NQ];       // selected candidates (sorted indices)
    __shared__ short s_parent[NQ], s_depth[NQ];
    __shared__ uint8_t s_acc[NQ], s_was[NQ], s_seen[NQ];
    __shared__ short s_use[NQ];       // candidate whose corners/id are reported for a selected entry
    __shared__ int s_nGroups, s_nSel;

    const int lane = threadIdx.x;
    const int f = blockIdx.x;
    const int n = min(qcount[f], kQuadCap);
    if ((n > 256) != (NQ > 256)) return;  // the other instance's frame
    const QuadRec* __restrict__ Q = quads + (size_t)f * kQuadCap;

    if (n == 0) {
        if (lane == 0) { outCounts[f] = 0; outRejCounts[f] = 0; }
        return;
    }

    // ---- order: rank by (perimeter desc, scale asc, scanKey desc) ----
    // perimeter needs the corners first; stage unsorted corners in s_c rows
    for (int i = lane; i < n; i += 64) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (float)Q[i].c[k];
        s_perim[i] = perimeter4(c);
    }
    // (scale asc, scan key desc) as one ascending key, staged in LDS: scan keys are pixel positions (< 2^24), scales < 8
    for (int i = lane; i < n; i += 64) s_rank[i] = (int)((Q[i].scale << 24) | (0xFFFFFFu - (Q[i].scanKey & 0xFFFFFFu)));
    __syncthreads();
    int myRank[NQ / 64];
#pragma unroll
    for (int t = 0; t < NQ / 64; t++) {
        const int i = lane + 64 * t;
        int r = 0;
        if (i < n) {
            const float pi = s_perim[i];
            const uint32_t ki = (uint32_t)s_rank[i];
            for (int j = 0; j < n; j++) {
                const float pj = s_perim[j];
                const uint32_t kj = (uint32_t)s_rank[j];
                // j comes before i (never true for j == i)
                const bool before = (pj != pi) ? pj > pi : kj < ki;
                r += before ? 1 : 0;
            }
        }
        myRank[t] = r;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NQ / 64; t++)
        if (lane + 64 * t < n) s_rank[lane + 64 * t] = myRank[t];
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const int r = s_rank[i];
#pragma unroll
        for (int k = 0; k < 8; k++) s_c[r][k] = (float)Q[i].c[k];
        s_id[r] = (short)Q[i].id;
        s_rot[r] = (uint8_t)Q[i].rot;
        s_valid[r] = (uint8_t)Q[i].valid;
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        s_perim[i] = perimeter4(s_c[i]);
        s_gid[i] = -1;
        s_closeOf[i] = -1;
        s_isSel[i] = 1;
    }
    if (lane == 0) s_nGroups = 0;
    __syncthreads();

    // ---- group near-duplicates ----
    const float rate = cfg.minMarkerDistanceRate;
    for (int i = 0; i < n; i++) {
        const float cix = 0.25f * (s_c[i][0] + s_c[i][2] + s_c[i][4] + s_c[i][6]), ciy = 0.25f * (s_c[i][1] + s_c[i][3] + s_c[i][5] + s_c[i][7]);
        for (int j0 = i + 1; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            bool close = false;
            if (j < n) {
                // the mean distance of matched corners is at least the distance of the centroids: pairs whose centroids
                // are clearly further apart than the bound (1 % and a pixel of slack for rounding) cannot be close
                const float thr = s_perim[j] * rate;
                const float dx = 0.25f * (s_c[j][0] + s_c[j][2] + s_c[j][4] + s_c[j][6]) - cix;
                const float dy = 0.25f * (s_c[j][1] + s_c[j][3] + s_c[j][5] + s_c[j][7]) - ciy;
                const float lim = thr * 1.01f + 1.f;
                if (dx * dx + dy * dy <= lim * lim) close = avg_distance(s_c[i], s_c[j]) < thr;
            }
            unsigned long long m = __ballot(close);
            if (lane == 0) {
                while (m) {
                    const int jj = j0 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    s_isSel[i] = 0;
                    s_isSel[jj] = 0;
                    const int gi = s_gid[i], gj = s_gid[jj];
                    if (gi < 0 && gj < 0) {
                        s_gid[i] = s_gid[jj] = (short)s_nGroups;
                        s_nGroups = s_nGroups + 1;
                    } else if (gi > -1 && gj == -1) {
                        s_gid[jj] = (short)gi;
                    } else if (gj > -1 && gi == -1) {
                        s_gid[i] = (short)gj;
                    }
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();

    // ---- representative + fall-backs per group (one group per lane) ----
    const int nGroups = s_nGroups;
    for (int g = lane; g < nGroups; g += 64) {
        int first = -1, curr = -1;
        for (int id = 0; id < n; id++) {
            if (s_gid[id] != g) continue;
            if (first < 0) {
                first = curr = id;
                s_isSel[id] = 1;
                continue;
            }
            const float dist = avg_distance(s_c[id], s_c[curr]);
            float ms = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++)
                ms += sqrtf(l2sqr(s_c[id][2 * k], s_c[id][2 * k + 1], s_c[id][2 * ((k + 1) & 3)], s_c[id][2 * ((k + 1) & 3) + 1]));
            ms /= (4.f * (float)(cfg.markerSize + 2 * cfg.borderBits));
            if (dist > cfg.minGroupDistance * ms) {
                curr = id;
                s_closeOf[id] = (short)first;
            }
        }
    }
    __syncthreads();

    // ---- selected list (ascending) ----
    if (lane == 0) {
        int ns = 0;
        for (int i = 0; i < n; i++)
            if (s_isSel[i]) s_sel[ns++] = (short)i;
        s_nSel = ns;
    }
    __syncthreads();
    const int ns = s_nSel;

    // ---- containment hierarchy ----
    for (int k = lane; k < ns; k += 64) {
        int parent = -1;
        for (int j = k - 1; j >= 0; j--)
            if (marker_in_marker(s_c[s_sel[k]], s_c[s_sel[j]])) { parent = j; break; }
        s_parent[k] = (short)parent;
        s_depth[k] = 0;
        s_acc[k] = 0;
        s_was[k] = 0;
        s_seen[k] = 0;
        s_use[k] = s_sel[k];
    }
    __syncthreads();

    // first valid fall-back of every representative (smallest index), found by all lanes: the acceptance loop below runs
    // on one lane and used to scan the candidates after each invalid one (s_rank is free again: reused)
    for (int i = lane; i < n; i += 64) s_rank[i] = 0x7FFFFFFF;
    __syncthreads();
    for (int id = lane; id < n; id += 64) {
        const int c = s_closeOf[id];
        if (c >= 0 && s_valid[id]) atomicMin(&s_rank[c], id);
    }
    __syncthreads();

    if (lane == 0) {
        int maxDepth = 0;
        for (int i = ns - 1; i >= 0; i--) {
            const int p = s_parent[i];
            if (p >= 0) s_depth[p] = max((int)s_depth[p], s_depth[i] + 1);
        }
        for (int i = 0; i < ns; i++) maxDepth = max(maxDepth, (int)s_depth[i]);

        // ---- depth-ordered acceptance (identifyCandidates) ----
        int depth = 0, counter = 0;
        while (counter < ns && depth <= maxDepth) {
            for (int v = 0; v < ns; v++) {
                if (s_depth[v] != depth) continue;
                s_was[v] = 1;
                s_seen[v] = 1;
                const int cand = s_sel[v];
                bool valid = s_valid[cand] != 0;
                if (!valid) {
                    const int id = s_rank[cand];
                    if (id < n) { valid = true; s_use[v] = (short)id; }
                }
                s_acc[v] = valid ? 1 : 0;
            }
            for (int v = 0; v < ns; v++) {
                if (s_depth[v] != depth) continue;
                if (s_acc[v]) {
                    int parent = s_parent[v];
                    while (parent != -1) {
                        if (!s_was[parent]) { s_was[parent] = 1; counter++; }
                        parent = s_parent[parent];
                    }
                }
                counter++;
            }
            depth++;
        }

        // ---- outputs ----
        int na = 0, nr = 0;
        for (int v = 0; v < ns; v++) {
            if (s_seen[v] && s_acc[v]) {
                if (na < kMaxMarkerOut) {
                    const int u = s_use[v];
                    const int rot = s_rot[u];
                    // correctCornerPosition: std::rotate(begin, begin + 4 - rot, end)
                    float* o = outCorners + ((size_t)f * kMaxMarkerOut + na) * 8;
                    float* o2 = rawCorners + ((size_t)f * kMaxMarkerOut + na) * 8;
                    for (int k = 0; k < 4; k++) {
                        const int src = (k + 4 - rot) & 3;
                        o[2 * k] = o2[2 * k] = s_c[u][2 * src];
                        o[2 * k + 1] = o2[2 * k + 1] = s_c[u][2 * src + 1];
                    }
                    outIds[(size_t)f * kMaxMarkerOut + na] = s_id[u];
                    const unsigned int slot = atomicAdd(&ctr->nMarkers, 1u);
                    if (slot < markerCap) {
                        markers[slot].frame = (uint32_t)f;
                        markers[slot].slot = (uint32_t)na;
                    } else {
                        atomicOr(&ctr->overflow, 32u);
                    }
                    na++;
                } else {
                    atomicOr(&ctr->overflow, 32u);
                }
            } else {
                if (nr < kMaxRejectedOut) {
                    float* o = outRejected + ((size_t)f * kMaxRejectedOut + nr) * 8;
                    const int u = s_sel[v];
                    for (int k = 0; k < 8; k++) o[k] = s_c[u][k];
                }
                nr++;
            }
        }
        outCounts[f] = na;
        outRejCounts[f] = min(nr, kMaxRejectedOut);
    }
}

// Results of a chunk in the caller's layout (camsync_detect_pnp_resident): ids [B,maxMarkers], corners
// [B,maxMarkers,8], counts [B] clamped to the columns the caller gave.  Lane = one marker slot.
__global__ __launch_bounds__(256) void k_export_markers(const int32_t* __restrict__ outIds, const float* __restrict__ outCorners,
                                                        const int32_t* __restrict__ outCounts, int B, int maxMarkers,
                                                        int32_t* __restrict__ dIds, float* __restrict__ dCorners,
                                                        int32_t* __restrict__ dCounts) {
    const int mm = min(maxMarkers, kMaxMarkerOut);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * mm) return;
    const int f = i / mm, m = i - f * mm;
    const int n = min(max(outCounts[f], 0), mm);
    if (m == 0) dCounts[f] = n;
    if (m >= n) return;
    dIds[(size_t)f * maxMarkers + m] = outIds[(size_t)f * kMaxMarkerOut + m];
    const float4* src = reinterpret_cast<const float4*>(outCorners + ((size_t)f * kMaxMarkerOut + m) * 8);
    float* dst = dCorners + ((size_t)f * maxMarkers + m) * 8;  // the caller's buffer need not be 16-byte aligned
    const float4 a = src[0], b = src[1];
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
}

}  // namespace

void launch_export_markers(const DetBuffers& b, int B, int maxMarkers, int32_t* dIds, float* dCorners, int32_t* dCounts,
                           hipStream_t st) {
    const int mm = std::min(maxMarkers, kMaxMarkerOut);
    const unsigned blocks = (unsigned)(((size_t)B * mm + 255) / 256);
    hipLaunchKernelGGL(k_export_markers, dim3(blocks), dim3(256), 0, st, b.outIds, b.outCorners, b.outCounts, B, maxMarkers,
                       dIds, dCorners, dCounts);
}

void launch_select(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    hipLaunchKernelGGL(k_select<256>, dim3(B), dim3(64), 0, st, b.quads, b.qcount, c, b.outIds, b.outCorners, b.rawCorners, b.outCounts,
                       b.outRejected, b.outRejCounts, b.markers, b.markerCap, b.counters);
    hipLaunchKernelGGL(k_select<kQuadCap>, dim3(B), dim3(64), 0, st, b.quads, b.qcount, c, b.outIds, b.outCorners, b.rawCorners, b.outCounts,
                       b.outRejected, b.outRejCounts, b.markers, b.markerCap, b.counters);
}

}  // namespace camsync
