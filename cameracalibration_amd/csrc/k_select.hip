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

// NQ: capacity of the LDS working set.  Three instances run over every frame: the small one takes the frames with
// up to 256 candidates (nearly all of them, at a quarter of the LDS), the large one those with up to kQuadLds, and the
// instance NQ = 0 the frames beyond that (a frame full of squares: cv.aruco.detectMarkers has no limit), with the same
// working set in global memory (`work`, sel_work_bytes per frame) -- the same code, L2 latencies instead of LDS ones.
// NT: threads per frame.  The phases are loops over candidates, pairs or groups that any number of wavefronts can
// share; what depends on the order of the sequential algorithm (group bookkeeping, acceptance) runs on thread 0, from
// what the others have laid out in the working set.  One wavefront per frame gives the best throughput when there are
// thousands of frames with a few dozen candidates each; with hundreds of candidates per frame (4K frames, forty markers
// at three scales) or few frames, a frame's chain of LDS round trips is the whole stage's time and four wavefronts share it.
constexpr int kGroupRound = 16;  // candidates whose pair tests are laid out before thread 0 replays them

// Bytes of the working set per candidate: corners 32, perimeter / centroid / bound 16, rank 4, eight index arrays 16,
// six flag arrays 6; per block of 64 candidates kGroupRound masks; kGroupRound x 8 words naming the blocks with any bit.
constexpr size_t kSelWorkPerQuad = 32 + 16 + 4 + 16 + 6;
constexpr size_t sel_work_layout(int cap) {
    return ((kSelWorkPerQuad * (size_t)cap + (size_t)kGroupRound * (cap / 64) * 8 + (size_t)kGroupRound * 8 * 8) + 255) & ~(size_t)255;
}

// MODE 0: everything in one launch (every candidate has been identified: the debug path).  MODE 1: order and grouping
// only; lists what the identification has to look at and keeps the order, the selection flags and the fall-back
// relation in `state` ([B][3][quadCap] shorts).  MODE 2: the rest (containment, acceptance, outputs) from that state
// and the identification's results.
template <int NQ, int NT, int MODE>
__global__ __launch_bounds__(NT) void k_select(const QuadRec* __restrict__ quads, const int* __restrict__ qcount,
                                               DetCfg cfg, int32_t* __restrict__ outIds, float* __restrict__ outCorners, float* __restrict__ rawCorners,
                                               int32_t* __restrict__ outCounts, float* __restrict__ outRejected,
                                               int32_t* __restrict__ outRejCounts, MarkerRec* __restrict__ markers,
                                               unsigned int markerCap, Counters* __restrict__ ctr,
                                               uint32_t* __restrict__ idList, int* __restrict__ idCount, int16_t* __restrict__ state,
                                               uint8_t* __restrict__ work) {
    constexpr bool GROUP_ONLY = MODE == 1;
    constexpr bool IN_LDS = NQ > 0;
    constexpr int LQ = IN_LDS ? NQ : 64;  // (LDS arrays of the instance that does not use them: never touched, dropped)
    constexpr int NW = NT / 64;       // wavefronts
    constexpr int NBS = LQ / 64;      // blocks of 64 candidates (LDS instances)
    constexpr int NBWS = 1;           // 64-bit words naming blocks (LDS instances: at most 16 blocks)
    constexpr int PER = IN_LDS ? (NQ + NT - 1) / NT : 8;  // candidates a thread ranks in one pass over the others
    __shared__ float l_c[LQ][8];
    __shared__ float l_perim[LQ];
    __shared__ float l_cenx[LQ], l_ceny[LQ], l_thr[LQ];
    __shared__ int l_rank[LQ];
    __shared__ short l_gid[LQ];
    __shared__ short l_closeOf[LQ];
    __shared__ short l_id[LQ];
    __shared__ uint8_t l_rot[LQ], l_valid[LQ], l_isSel[LQ];
    __shared__ short l_sel[LQ];
    __shared__ short l_parent[LQ], l_depth[LQ];
    __shared__ uint8_t l_acc[LQ], l_was[LQ], l_seen[LQ];
    __shared__ short l_use[LQ];
    __shared__ short l_orig[LQ];
    __shared__ unsigned long long l_close[kGroupRound * NBS];
    __shared__ unsigned long long l_closeBlocks[kGroupRound * NBWS];
    __shared__ int s_nGroups, s_nSel;
    __shared__ int s_nList[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.x;
    const int quadCap = cfg.quadCap;
    const int n = min(qcount[f], quadCap);
    if ((n <= 256 ? 256 : n <= kQuadLds ? kQuadLds : 0) != NQ) return;  // another instance's frame

    // the working set (the comments name the LDS instances' arrays; the third instance has them at `work`)
    float (*s_c)[8];                  // corners, sorted order
    float *s_perim, *s_cenx, *s_ceny, *s_thr;  // perimeter, centroid, and perimeter x minMarkerDistanceRate, of the sorted candidates
    int* s_rank;
    short *s_gid, *s_closeOf;         // group; representative this candidate is a fall-back of, or -1
    short* s_id;
    uint8_t *s_rot, *s_valid, *s_isSel;
    short* s_sel;                     // selected candidates (sorted indices)
    short *s_parent, *s_depth;
    uint8_t *s_acc, *s_was, *s_seen;
    short* s_use;                     // candidate whose corners/id are reported for a selected entry
    short* s_orig;                    // index of the sorted candidate in the frame's quad list
    unsigned long long* s_close;      // [kGroupRound][NB] per candidate of the round and block: the later candidates close to it
    unsigned long long* s_closeBlocks;  // [kGroupRound][NBW] ... and which blocks have any
    int NB, NBW;
    if constexpr (IN_LDS) {
        s_c = l_c; s_perim = l_perim; s_cenx = l_cenx; s_ceny = l_ceny; s_thr = l_thr; s_rank = l_rank; s_gid = l_gid;
        s_closeOf = l_closeOf; s_id = l_id; s_rot = l_rot; s_valid = l_valid; s_isSel = l_isSel; s_sel = l_sel;
        s_parent = l_parent; s_depth = l_depth; s_acc = l_acc; s_was = l_was; s_seen = l_seen; s_use = l_use; s_orig = l_orig;
        s_close = l_close; s_closeBlocks = l_closeBlocks;
        NB = NBS; NBW = NBWS;
    } else {
        const size_t Q = (size_t)quadCap;
        uint8_t* w = work + (size_t)f * sel_work_layout(quadCap);
        s_c = reinterpret_cast<float (*)[8]>(w); w += 32 * Q;
        s_perim = reinterpret_cast<float*>(w); w += 4 * Q;
        s_cenx = reinterpret_cast<float*>(w); w += 4 * Q;
        s_ceny = reinterpret_cast<float*>(w); w += 4 * Q;
        s_thr = reinterpret_cast<float*>(w); w += 4 * Q;
        s_rank = reinterpret_cast<int*>(w); w += 4 * Q;
        NB = quadCap / 64; NBW = 8;
        s_close = reinterpret_cast<unsigned long long*>(w); w += (size_t)kGroupRound * NB * 8;
        s_closeBlocks = reinterpret_cast<unsigned long long*>(w); w += (size_t)kGroupRound * NBW * 8;
        s_gid = reinterpret_cast<short*>(w); w += 2 * Q;
        s_closeOf = reinterpret_cast<short*>(w); w += 2 * Q;
        s_id = reinterpret_cast<short*>(w); w += 2 * Q;
        s_sel = reinterpret_cast<short*>(w); w += 2 * Q;
        s_parent = reinterpret_cast<short*>(w); w += 2 * Q;
        s_depth = reinterpret_cast<short*>(w); w += 2 * Q;
        s_use = reinterpret_cast<short*>(w); w += 2 * Q;
        s_orig = reinterpret_cast<short*>(w); w += 2 * Q;
        s_rot = w; w += Q;
        s_valid = w; w += Q;
        s_isSel = w; w += Q;
        s_acc = w; w += Q;
        s_was = w; w += Q;
        s_seen = w;
    }

    const QuadRec* __restrict__ Q = quads + (size_t)f * quadCap;

    if (n == 0) {
        if (tid == 0) {
            if (GROUP_ONLY) { idCount[2 * f] = 0; idCount[2 * f + 1] = 0; }
            else { outCounts[f] = 0; outRejCounts[f] = 0; }
        }
        return;
    }

    int16_t* const st = state + (size_t)f * 3 * quadCap;  // [0]: quad index of the sorted candidate, [1]: selected, [2]: fall-back of
    if (MODE == 2) {
        for (int r = tid; r < n; r += NT) {
            const int i = st[r];
#pragma unroll
            for (int k = 0; k < 8; k++) s_c[r][k] = (float)Q[i].c[k];
            s_id[r] = (short)Q[i].id;
            s_rot[r] = (uint8_t)Q[i].rot;
            s_valid[r] = (uint8_t)Q[i].valid;
            s_isSel[r] = (uint8_t)st[quadCap + r];
            s_closeOf[r] = st[2 * quadCap + r];
        }
        __syncthreads();
    } else {
    // ---- order: rank by (perimeter desc, scale asc, scanKey desc) ----
    // perimeter needs the corners first
    for (int i = tid; i < n; i += NT) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = (float)Q[i].c[k];
        s_perim[i] = perimeter4(c);
        // (scale asc, scan key desc) as one ascending key: scan keys are pixel positions (< 2^24), scales < 8
        s_rank[i] = (int)((Q[i].scale << 24) | (0xFFFFFFu - (Q[i].scanKey & 0xFFFFFFu)));
    }
    __syncthreads();
    // one pass over the candidates for (up to PER) candidates of the thread's own: the two reads of a step are shared
    // by them, and successive steps do not depend on one another.  (The LDS instances need one pass.)  The ranks go
    // to s_cenx (as bits) until every thread is done with the keys in s_rank.
    for (int base = 0; base < n; base += NT * PER) {
        int myRank[PER];
        float myPerim[PER];
        uint32_t myKey[PER];
#pragma unroll
        for (int t = 0; t < PER; t++) {
            const int i = min(base + tid + NT * t, n - 1);
            myRank[t] = 0;
            myPerim[t] = s_perim[i];
            myKey[t] = (uint32_t)s_rank[i];
        }
        for (int j = 0; j < n; j++) {
            const float pj = s_perim[j];
            const uint32_t kj = (uint32_t)s_rank[j];
#pragma unroll
            for (int t = 0; t < PER; t++) {
                if (base + NT * t >= n) break;  // (uniform)
                // j comes before i (never true for j == i)
                const bool before = (pj != myPerim[t]) ? pj > myPerim[t] : kj < myKey[t];
                myRank[t] += before ? 1 : 0;
            }
        }
#pragma unroll
        for (int t = 0; t < PER; t++)
            if (base + tid + NT * t < n) s_cenx[base + tid + NT * t] = __int_as_float(myRank[t]);
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) s_rank[i] = __float_as_int(s_cenx[i]);
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const int r = s_rank[i];
#pragma unroll
        for (int k = 0; k < 8; k++) s_c[r][k] = (float)Q[i].c[k];
        s_id[r] = (short)Q[i].id;
        s_rot[r] = (uint8_t)Q[i].rot;
        s_valid[r] = (uint8_t)Q[i].valid;
        s_orig[r] = (short)i;
    }
    if (tid == 0) s_nList[0] = s_nList[1] = 0;
    __syncthreads();
    const float rate = cfg.minMarkerDistanceRate;
    for (int i = tid; i < n; i += NT) {
        s_perim[i] = perimeter4(s_c[i]);
        s_cenx[i] = 0.25f * (s_c[i][0] + s_c[i][2] + s_c[i][4] + s_c[i][6]);
        s_ceny[i] = 0.25f * (s_c[i][1] + s_c[i][3] + s_c[i][5] + s_c[i][7]);
        s_thr[i] = s_perim[i] * rate;
        s_gid[i] = -1;
        s_closeOf[i] = -1;
        s_isSel[i] = 1;
    }
    __syncthreads();

    // ---- group near-duplicates ----
    // Rounds of kGroupRound candidates.  The wavefronts share the round's candidates; for candidate i a wavefront
    // tests the later candidates 64 at a time: first the centroids (the mean distance of matched corners is at least
    // the distance of the centroids, so pairs whose centroids are clearly further apart than the bound -- 1 % and a
    // pixel of slack for rounding -- cannot be close), then the corner distances of the pairs that are left, and
    // leaves a mask of the close ones per block.  Thread 0 then replays the round's pairs in the order of the
    // sequential algorithm (i ascending, j ascending) and keeps the groups' books, with candidate i's group id and
    // the group count in registers.
    int nG = 0;  // (thread 0's count of groups)
    const int nBlocks = (n + 63) >> 6;
    for (int i0 = 0; i0 < n; i0 += kGroupRound) {
        for (int slot = wave; slot < kGroupRound; slot += NW) {
            const int i = i0 + slot;
            if (i >= n) break;  // (uniform per wavefront)
            const float cix = s_cenx[i], ciy = s_ceny[i];
            for (int w = 0; w < NBW; w++) {
                unsigned long long blocks = 0;
                for (int b = max((i + 1) >> 6, 64 * w); b < min(nBlocks, 64 * w + 64); b++) {
                    const int j = 64 * b + lane;
                    bool nr = false;
                    if (j > i && j < n) {
                        const float dx = s_cenx[j] - cix, dy = s_ceny[j] - ciy;
                        const float lim = s_thr[j] * 1.01f + 1.f;
                        nr = dx * dx + dy * dy <= lim * lim;
                    }
                    if (!__any(nr)) continue;
                    bool close = false;
                    if (nr) close = avg_distance(s_c[i], s_c[j]) < s_thr[j];
                    const unsigned long long m = __ballot(close);
                    if (m) {
                        blocks |= 1ull << (b & 63);
                        if (lane == 0) s_close[slot * NB + b] = m;
                    }
                }
                if (lane == 0) s_closeBlocks[slot * NBW + w] = blocks;
            }
        }
        __syncthreads();
        if (tid == 0) {
            for (int slot = 0; slot < kGroupRound && i0 + slot < n; slot++) {
                unsigned long long any = 0;
                for (int w = 0; w < NBW; w++) any |= s_closeBlocks[slot * NBW + w];
                if (!any) continue;
                const int i = i0 + slot;
                int gi = s_gid[i];
                for (int w = 0; w < NBW; w++) {
                    unsigned long long blocks = s_closeBlocks[slot * NBW + w];
                    while (blocks) {
                        const int b = 64 * w + __ffsll((long long)blocks) - 1;
                        blocks &= blocks - 1;
                        unsigned long long m = s_close[slot * NB + b];
                        while (m) {
                            const int jj = 64 * b + __ffsll((long long)m) - 1;
                            m &= m - 1;
                            s_isSel[jj] = 0;
                            const int gj = s_gid[jj];
                            if (gi < 0 && gj < 0) {
                                gi = nG++;
                                s_gid[jj] = (short)gi;
                            } else if (gi > -1 && gj == -1) {
                                s_gid[jj] = (short)gi;
                            } else if (gj > -1 && gi == -1) {
                                gi = gj;
                            }
                        }
                    }
                }
                s_gid[i] = (short)gi;
                s_isSel[i] = 0;
            }
        }
        __syncthreads();
    }
    if (tid == 0) s_nGroups = nG;
    __syncthreads();

    // ---- representative + fall-backs per group (one group per thread) ----
    // The members of every group are listed first (ascending, as the sequential algorithm meets them), so that the
    // threads then walk their groups in step: with each thread scanning all candidates for its own, the expensive part
    // ran once per candidate of the frame instead of once per position in the largest group.  Scratch: s_rank (member
    // counts, free between the sort and the fall-back ranks), s_parent (offsets) and s_use (members), which the
    // containment phase initialises later.
    const int nGroups = s_nGroups;
    for (int g = tid; g < nGroups; g += NT) s_rank[g] = 0;
    __syncthreads();
    for (int id = tid; id < n; id += NT) {
        const int g = s_gid[id];
        if (g >= 0) atomicAdd(&s_rank[g], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int off = 0;
        for (int g = 0; g < nGroups; g++) {
            s_parent[g] = (short)off;
            off += s_rank[g];
        }
    }
    __syncthreads();
    for (int g = tid; g < nGroups; g += NT) {
        int k = s_parent[g];
        for (int id = 0; id < n; id++)
            if (s_gid[id] == g) s_use[k++] = (short)id;
    }
    __syncthreads();
    for (int g = tid; g < nGroups; g += NT) {
        const int off = s_parent[g], cnt = s_rank[g];
        const int first = s_use[off];
        int curr = first;
        s_isSel[first] = 1;
        for (int m = 1; m < cnt; m++) {
            const int id = s_use[off + m];
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

    }
    if (GROUP_ONLY) {
        // what the identification has to look at: the selected candidates (pass 1), and the fall-backs of each with
        // their representative (pass 2, skipped for representatives that pass 1 identifies); any order
        uint32_t* const list = idList + (size_t)f * quadCap;
        for (int i = tid; i < n; i += NT) {
            st[i] = s_orig[i];
            st[quadCap + i] = s_isSel[i];
            st[2 * quadCap + i] = s_closeOf[i];
            if (s_isSel[i]) list[atomicAdd(&s_nList[0], 1)] = (uint32_t)(uint16_t)s_orig[i] | 0xFFFF0000u;
            const int c = s_closeOf[i];
            if (c >= 0) list[quadCap - 1 - atomicAdd(&s_nList[1], 1)] = (uint32_t)(uint16_t)s_orig[i] | ((uint32_t)(uint16_t)s_orig[c] << 16);
        }
        __syncthreads();
        if (tid == 0) { idCount[2 * f] = s_nList[0]; idCount[2 * f + 1] = s_nList[1]; }
        return;
    }

    // ---- selected list (ascending): positions by ballot, 64 candidates at a time, on the first wavefront ----
    if (wave == 0) {
        int ns0 = 0;  // uniform
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const bool sel = i < n && s_isSel[i];
            const unsigned long long bal = __ballot(sel);
            if (sel) s_sel[ns0 + (int)__popcll(bal & ((1ull << lane) - 1ull))] = (short)i;
            ns0 += (int)__popcll(bal);
        }
        if (lane == 0) s_nSel = ns0;
    }
    __syncthreads();
    const int ns = s_nSel;

    // ---- containment hierarchy ----
    for (int k = tid; k < ns; k += NT) {
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
    // first valid fall-back of every representative (smallest index), found by all threads: the acceptance loop below
    // runs on one thread and used to scan the candidates after each invalid one (s_rank is free again: reused)
    for (int i = tid; i < n; i += NT) s_rank[i] = 0x7FFFFFFF;
    __syncthreads();
    for (int id = tid; id < n; id += NT) {
        const int c = s_closeOf[id];
        if (c >= 0 && s_valid[id]) atomicMin(&s_rank[c], id);
    }
    __syncthreads();

    if (tid == 0) {
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
    }
    __syncthreads();
    if (wave != 0) return;

    // ---- outputs: accepted markers and rejected candidates in the order of the selected list, 64 entries at a time
    //      (positions by ballot; every lane of the first wavefront writes its own entry) ----
    const int markerOut = cfg.markerOut;
    int na = 0, nr = 0;  // uniform
    for (int v0 = 0; v0 < ns; v0 += 64) {
        const int v = v0 + lane;
        const bool in = v < ns;
        const bool isAcc = in && s_seen[v] && s_acc[v];
        const unsigned long long balA = __ballot(isAcc), balR = __ballot(in && !isAcc);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int myA = na + (int)__popcll(balA & below), myR = nr + (int)__popcll(balR & below);
        const int cntA = (int)__popcll(balA);
        // marker records: one reservation for the chunk's markers that fit the frame's output
        const int fit = max(0, min(cntA, markerOut - na));
        unsigned int slot0 = 0;
        if (fit > 0) {
            if (lane == 0) slot0 = atomicAdd(&ctr->nMarkers, (unsigned int)fit);
            slot0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)slot0);
        }
        if (isAcc) {
            if (myA < markerOut) {
                const int u = s_use[v];
                const int rot = s_rot[u];
                // correctCornerPosition: std::rotate(begin, begin + 4 - rot, end)
                float* o = outCorners + ((size_t)f * markerOut + myA) * 8;
                float* o2 = rawCorners + ((size_t)f * markerOut + myA) * 8;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int src = (k + 4 - rot) & 3;
                    o[2 * k] = o2[2 * k] = s_c[u][2 * src];
                    o[2 * k + 1] = o2[2 * k + 1] = s_c[u][2 * src + 1];
                }
                outIds[(size_t)f * markerOut + myA] = s_id[u];
                const unsigned int slot = slot0 + (unsigned int)(myA - na);
                if (slot < markerCap) {
                    markers[slot].frame = (uint32_t)f;
                    markers[slot].slot = (uint32_t)myA;
                } else {
                    atomicOr(&ctr->overflow, 32u);
                }
            } else {
                atomicOr(&ctr->overflow, 32u);
            }
        } else if (in && myR < quadCap) {
            float* o = outRejected + ((size_t)f * quadCap + myR) * 8;
            const int u = s_sel[v];
#pragma unroll
            for (int k = 0; k < 8; k++) o[k] = s_c[u][k];
        }
        na += cntA;
        nr += (int)__popcll(balR);
    }
    if (lane == 0) {
        outCounts[f] = min(na, markerOut);
        if (na > markerOut) atomicMax(&ctr->needMarkers, (unsigned int)na);
        outRejCounts[f] = min(nr, quadCap);
    }
}

// Results of a chunk in the caller's layout (camsync_detect_pnp_resident): ids [B,maxMarkers], corners
// [B,maxMarkers,8], counts [B] clamped to the columns the caller gave.  Lane = one marker slot.
__global__ __launch_bounds__(256) void k_export_markers(const int32_t* __restrict__ outIds, const float* __restrict__ outCorners,
                                                        const int32_t* __restrict__ outCounts, int B, int maxMarkers, int markerOut,
                                                        int32_t* __restrict__ dIds, float* __restrict__ dCorners,
                                                        int32_t* __restrict__ dCounts) {
    const int mm = min(maxMarkers, markerOut);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * mm) return;
    const int f = i / mm, m = i - f * mm;
    const int n = min(max(outCounts[f], 0), mm);
    if (m == 0) dCounts[f] = n;
    if (m >= n) return;
    dIds[(size_t)f * maxMarkers + m] = outIds[(size_t)f * markerOut + m];
    const float4* src = reinterpret_cast<const float4*>(outCorners + ((size_t)f * markerOut + m) * 8);
    float* dst = dCorners + ((size_t)f * maxMarkers + m) * 8;  // the caller's buffer need not be 16-byte aligned
    const float4 a = src[0], b = src[1];
    dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w;
    dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
}

}  // namespace

void launch_export_markers(const DetBuffers& b, int B, int maxMarkers, int32_t* dIds, float* dCorners, int32_t* dCounts,
                           hipStream_t st) {
    const int mm = std::min(maxMarkers, b.markerOut);
    const unsigned blocks = (unsigned)(((size_t)B * mm + 255) / 256);
    hipLaunchKernelGGL(k_export_markers, dim3(blocks), dim3(256), 0, st, b.outIds, b.outCorners, b.outCounts, B, maxMarkers, b.markerOut,
                       dIds, dCorners, dCounts);
}

namespace {

template <int MODE>
void launch_select_kernels(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    // frames with up to 256 candidates: one wavefront per frame when there are frames enough to fill the chip that way,
    // four otherwise (latency of a small batch); frames with more candidates always get four
    uint8_t* const none = nullptr;
    if (B >= 1024 && MODE != 1)  // (the grouping alone is faster with four even then: 0.23 -> 0.18 ms at 4096 frames)
        hipLaunchKernelGGL((k_select<256, 64, MODE>), dim3(B), dim3(64), 0, st, b.quads, b.qcount, c, b.outIds, b.outCorners, b.rawCorners,
                           b.outCounts, b.outRejected, b.outRejCounts, b.markers, b.markerCap, b.counters, b.idList, b.idCount, b.selState, none);
    else
        hipLaunchKernelGGL((k_select<256, 256, MODE>), dim3(B), dim3(256), 0, st, b.quads, b.qcount, c, b.outIds, b.outCorners, b.rawCorners,
                           b.outCounts, b.outRejected, b.outRejCounts, b.markers, b.markerCap, b.counters, b.idList, b.idCount, b.selState, none);
    hipLaunchKernelGGL((k_select<kQuadLds, 256, MODE>), dim3(B), dim3(256), 0, st, b.quads, b.qcount, c, b.outIds, b.outCorners, b.rawCorners,
                       b.outCounts, b.outRejected, b.outRejCounts, b.markers, b.markerCap, b.counters, b.idList, b.idCount, b.selState, none);
    // (only once a frame has had more candidates than the LDS instances hold: the lists have grown, camsync_api.hip)
    if (c.quadCap > kQuadLds && b.selWork)
        hipLaunchKernelGGL((k_select<0, 256, MODE>), dim3(B), dim3(256), 0, st, b.quads, b.qcount, c, b.outIds, b.outCorners, b.rawCorners,
                           b.outCounts, b.outRejected, b.outRejCounts, b.markers, b.markerCap, b.counters, b.idList, b.idCount, b.selState,
                           b.selWork);
}

}  // namespace

size_t sel_work_bytes(int quadCap) { return sel_work_layout(quadCap); }

void launch_select(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    if (b.identifyAll) launch_select_kernels<0>(c, b, B, st);
    else launch_select_kernels<2>(c, b, B, st);  // (launch_group has run: launch_identify)
}

// The grouping half alone, before the identification: lists the candidates that can be asked for (DetBuffers::idList).
void launch_group(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) { launch_select_kernels<1>(c, b, B, st); }

}  // namespace camsync
