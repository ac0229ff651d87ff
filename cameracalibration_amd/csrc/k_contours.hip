// k_contours.hip — K3: findContours(RETR_LIST, CHAIN_APPROX_NONE) of
// ArucoDetector.detectMarkers step 1b (reference: camera_sync/aruco.py:256-264),
// reformulated for a GPU.
//
// OpenCV's Suzuki–Abe scan is sequential only in how it *finds* border
// starts (it labels pixels as it goes).  The set of borders and their start
// points is a pure function of the binary image:
//   * an outer border starts at the raster-first pixel of its 8-connected
//     component (search for the first neighbour begins west, clockwise);
//   * a hole border starts at the pixel left of the raster-first pixel of the
//     hole (4-connected zero component), search begins east.
// The threshold images arrive as block-linear bit planes (internal.h: one bit
// per pixel, 32x32-pixel blocks of one cache line, image frame zeroed), so:
//   (1) k_contour_starts finds every pixel that can be such a start with a few
//       word-wide bit operations per 32 pixels (outer: p=1, W=NW=N=NE=0;
//       hole: p=0, W=1, N=1), drops isolated pixels, and drops starts whose
//       horizontal run touches the row above further right (the component /
//       hole then begins earlier): that removes the staircase peaks of slanted
//       edges without tracing them.  Starts are staged in LDS and appended to
//       the global list with one atomic per flush.
//   (2) k_trace_short follows each candidate's border exactly like OpenCV's
//       icvFetchContour for a few dozen steps, one lane per candidate, in
//       lock-step with a second tracer walking the same border the other way
//       round (the vertical mirror image of the algorithm).  A candidate is
//       dropped as soon as either tracer meets a crack with a smaller raster
//       key (it is not the canonical start) or the border closes below the
//       minimum perimeter (a border is known all round when the forward tracer is
//       back at its start or when the two tracers meet half-way).  Noise specks
//       and false starts end here.
//   (3) Candidates still alive are long borders.  Border following is a walk on
//       states (pixel, direction to the predecessor) under a bijective successor
//       map, so it can start anywhere: k_contour_starts also lists *seeds*,
//       cracks on a 64-pixel grid that are recognisable from the 3x3
//       neighbourhood, and k_seg_walk2 follows every border from each seed (and
//       from each candidate) to the next seed inside a private LDS window,
//       recording chain codes: the longest dependent chain is one segment, not
//       one border.  k_seg_link turns "ends at the seed at this position" into
//       that seed's id; k_resolve walks each candidate's ring of segment records,
//       applies the start rule (smallest key of the ring) and the perimeter
//       bounds, and cuts accepted borders into copy tasks; k_trace_copy turns
//       chain codes into points by prefix sums.
//   (4) k_trace_store re-follows the (few) borders the short pass itself accepts
//       and writes their points.
// The traces only read the image, so they are independent of each other.
#include "internal.h"

#include <algorithm>

namespace camsync {

namespace {

// fill `seed` along runs of `run` towards lower bit positions.  Bit-reversed, "towards lower" is the direction
// carries travel: adding the seeds to the runs clears each run from its lowest seed up to its end (the carry lands
// on the zero after the run, and later seeds of the same run fall on cleared bits), so the bits the addition
// changed, within the runs, plus the seeds, are the filled runs.  8 instructions instead of the 25 of a Kogge-Stone fill.
__device__ __forceinline__ uint32_t fill_down(uint32_t seed, uint32_t run) {
    const uint32_t r = __brev(run), s = __brev(seed) & r;
    return __brev((((r + s) ^ r) & r) | s);
}

// A wavefront's LDS operations execute in order: lanes of one wavefront exchange data through LDS without s_barrier,
// only the compiler must keep the order.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (3 KB of LDS per wavefront, so that registers, not LDS, bound the waves per CU: 512 / 256 entries 1.72 ms for the
// start pass of the bench scene, 256 / 128 1.68 ms, 128 / 64 1.95 ms -- a flush is an atomic on one shared counter)
constexpr int kStage = 256;  // starts staged per wavefront before a flush
constexpr int kSeedStage = 128;  // border seeds likewise

// word (xw, y) of a block-linear plane, zero outside the plane
__device__ __forceinline__ uint32_t pw_at(const uint32_t* __restrict__ plane, int WS, int H, int xw, int y) {
    if (xw < 0 || xw >= WS || y < 0 || y >= H) return 0u;
    return plane[plane_word_index(WS, xw, y)];
}

// A lane owns one column of plane blocks (32 pixels wide) and walks down it, eight rows (two 16-byte loads) at a
// time; a wavefront is 62 such columns side by side plus one halo column at either end, over a band of rows of one
// plane.  The rows above and below a row are then the lane's own registers from the previous / next step, and only
// the one bit that crosses a word boundary comes from the neighbouring lane (DPP wave shifts): about twenty vector
// instructions per 32 pixels when nothing starts in them.  Two groups of rows are loaded ahead of the one being
// processed.  (A block's 128-byte line is touched in four rounds, microseconds apart, and the chip's wavefronts hold
// more lines in flight than the L2 keeps: FETCH_SIZE shows 2.6 x the planes.  A variant whose lanes pair up to read
// whole lines fetched 1.1 x and was 5 % slower: the pass is bound by instruction issue, not by that traffic.)  Starts are staged in LDS and appended to the global list with one atomic per flush.
// Seed grid pitch G (a power of two, >= 32): the pixels of word bx (32 pixels) that lie on a north-seed column
// (x = 0 mod G, where `north` must hold) or a south-seed column (x = G/2 mod G, where `south` must hold).
constexpr int kSeedGrid = 64;
__device__ __forceinline__ uint32_t seed_columns(int bx, uint32_t north, uint32_t south) {
    if (kSeedGrid == 32) return (north & 1u) | (south & 0x10000u);
    const int wm = bx & (kSeedGrid / 32 - 1);
    return wm == 0 ? (north & 1u) : wm == kSeedGrid / 64 ? (south & 1u) : 0u;
}

// Seed -> id table of one bit plane, indexed by where seeds can be: the rows y = 0 (mod G/2) in full, then the columns
// x = 0 (mod G/2).  Entries are written when a seed is listed and only ever read for positions the followers found to
// be seeds, so the table is never cleared (k_seg_link checks the entry against the seed list: a stale entry of an
// earlier call reads as "not listed").
__host__ __device__ __forceinline__ size_t seed_table_size(int H, int W) {
    return (size_t)(H / (kSeedGrid / 2) + 1) * W + (size_t)(W / (kSeedGrid / 2) + 1) * H;
}
__device__ __forceinline__ size_t seed_table_index(const DetCfg& cfg, uint32_t pos) {
    const int H = cfg.H, W = cfg.W;
    const int y = row_of(cfg, pos), x = (int)(pos - (uint32_t)y * (uint32_t)W);
    constexpr int G2 = kSeedGrid / 2;
    if ((y & (G2 - 1)) == 0) return (size_t)(y / G2) * W + x;
    return (size_t)(H / G2 + 1) * W + (size_t)(x / G2) * H + y;
}

constexpr int kStartCols = 62;  // useful block columns per wavefront (lanes 1..62)

__global__ __launch_bounds__(64) void k_contour_starts(const uint32_t* __restrict__ planes, DetCfg cfg, int B, int bands,
                                                       int bandGroups, uint64_t* __restrict__ cand, unsigned int candCap,
                                                       uint64_t* __restrict__ seeds, unsigned int seedCap,
                                                       uint32_t* __restrict__ seedTable, Counters* __restrict__ ctr) {
    __shared__ uint64_t stage[kStage];
    __shared__ uint64_t seedStage[kSeedStage];
    unsigned int seedStaged = 0;  // uniform
    const int H = cfg.H, WS = cfg.WS, S = cfg.nScales, W = cfg.W, HB = plane_hb(H);
    const int colGroups = (WS + kStartCols - 1) / kStartCols;
    const size_t perPlane = plane_words(H, WS);
    const bool dropIsolated = cfg.minPerim > 1;
    const int lane = threadIdx.x;
    unsigned int staged = 0;  // uniform
    uint32_t* myTable = seedTable;  // this plane's part of the seed table (set below)

    auto flush = [&]() {
        if (staged) {
            unsigned int g = 0;
            if (lane == 0) g = atomicAdd(&ctr->nCand, staged);
            g = __builtin_amdgcn_readfirstlane(g);
            // (a full list fails the call with CAMSYNC_E_CAPACITY; the part that fits is still written, because the
            // next pass reads every entry below min(count, capacity))
            if (g + staged > candCap && lane == 0) atomicOr(&ctr->overflow, 1u);
            const unsigned int fit = g >= candCap ? 0u : min(staged, candCap - g);
            for (unsigned int k = lane; k < fit; k += 64) cand[g + k] = stage[k];
            staged = 0;
        }
    };

    auto flush_seeds = [&]() {
        if (seedStaged) {
            unsigned int g = 0;
            if (lane == 0) g = atomicAdd(&ctr->nSeeds, seedStaged);
            g = __builtin_amdgcn_readfirstlane(g);
            if (g + seedStaged > seedCap && lane == 0) atomicOr(&ctr->overflow, 512u);
            const unsigned int fit = g >= seedCap ? 0u : min(seedStaged, seedCap - g);
            for (unsigned int k = lane; k < fit; k += 64) {
                const uint64_t v = seedStage[k];
                seeds[g + k] = v;
                myTable[seed_table_index(cfg, (uint32_t)(v & 0xFFFFFFFFull))] = g + k;
            }
            seedStaged = 0;
        }
    };

    // which band of which column group of which plane (uniform)
    unsigned int u = blockIdx.x;
    const int band = (int)(u % (unsigned)bands);
    u /= (unsigned)bands;
    const int cg = (int)(u % (unsigned)colGroups);
    const unsigned int fs = u / (unsigned)colGroups;  // frame * S + scale
    if (fs >= (unsigned)B * (unsigned)S) return;
    const int totalGroups = HB * 4;  // groups of eight rows in a plane
    const int g0 = band * bandGroups, g1 = min(totalGroups, g0 + bandGroups);
    if (g0 >= g1) return;
    const int bx = cg * kStartCols - 1 + lane;
    const bool inPlane = bx >= 0 && bx < WS;
    const bool mine = inPlane && lane >= 1 && lane <= kStartCols;  // halo lanes only lend their edge bits
    const uint32_t om = mine ? 0xFFFFFFFFu : 0u;
    // pixels beyond the image width are padding: nothing starts there
    const uint32_t hm = !mine ? 0u : ((bx + 1) * 32 > W ? (1u << (W - bx * 32)) - 1u : 0xFFFFFFFFu);
    const uint32_t* const plane = planes + (size_t)fs * perPlane;
    const uint32_t f = fs / (unsigned int)S, sc = fs % (unsigned int)S;
    myTable = seedTable + (size_t)fs * seed_table_size(H, W);

    // eight consecutive rows of this lane's column, starting at row 8 g (zero outside the plane)
    struct Rows { uint4 a, b; };
    auto load_rows = [&](int g) {
        Rows r;
        r.a = make_uint4(0, 0, 0, 0);
        r.b = make_uint4(0, 0, 0, 0);
        if (inPlane && g < totalGroups) {
            const uint4* src = reinterpret_cast<const uint4*>(plane + ((size_t)(g >> 2) * WS + bx) * 32 + (g & 3) * 8);
            r.a = src[0];
            r.b = src[1];
        }
        return r;
    };
    auto lane_before = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); };
    auto lane_after = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); };
    // a row's pixels shifted by one: bit i = the pixel to the west / east of pixel i
    auto west_of = [&](uint32_t c) { return __funnelshift_l(lane_before(c), c, 1); };   // (c << 1) | (left word >> 31)
    auto east_of = [&](uint32_t c) { return __funnelshift_r(c, lane_after(c), 1); };    // (c >> 1) | (right word << 31)

    // the row above the band's first row (zero above the plane), and its shifted copies
    uint32_t pc = 0;
    if (g0 > 0 && inPlane) pc = plane[plane_word_index(WS, bx, 8 * g0 - 1)];
    uint32_t pw = west_of(pc), pe = east_of(pc);
    Rows cur = load_rows(g0), nx1 = load_rows(g0 + 1), nx2 = load_rows(g0 + 2);
    uint32_t cw = west_of(cur.a.x), ce = east_of(cur.a.x);  // of the row about to be processed

    for (int g = g0; g < g1; g++) {
        const Rows nx3 = load_rows(g + 3);
        const uint32_t w[9] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cur.b.z, cur.b.w, nx1.a.x};
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t c = w[r], sd = w[r + 1];
            const uint32_t sw = west_of(sd), se = east_of(sd);
            const uint32_t above = pc | pw | pe;  // some foreground 8-adjacent in the row above
            uint32_t outer = c & ~(cw | above) & om;
            if (dropIsolated) outer &= (ce | sw | sd | se);
            uint32_t hole = ~c & cw & pc & hm;
            const int y = 8 * g + r;
            if (__any(outer != 0)) {
                // the run that starts here must not touch the row above anywhere
                outer &= ~fill_down(above, c);
                // a run that leaves the word through bit 31 is followed into the next words
                if ((c >> 31) && outer) {
                    const uint32_t top = 1u << (32 - __clz(~c));  // lowest bit of the run ending at bit 31
                    if (outer & top) {
                        bool bad = false;
                        for (int k = bx + 1; k < WS; k++) {
                            const uint32_t ck = pw_at(plane, WS, H, k, y);
                            if (!(ck & 1u)) break;
                            const uint32_t nk = pw_at(plane, WS, H, k, y - 1);
                            const uint32_t nkl = pw_at(plane, WS, H, k - 1, y - 1), nkr = pw_at(plane, WS, H, k + 1, y - 1);
                            const uint32_t ab = nk | (nk << 1) | (nkl >> 31) | (nk >> 1) | (nkr << 31);
                            const int tt = __ffs(~ck) - 1;  // run covers bits [0,tt), tt = -1 means all 32
                            const uint32_t rmask = (tt < 0) ? 0xFFFFFFFFu : ((1u << tt) - 1u);
                            if (ab & rmask) { bad = true; break; }
                            if (tt >= 0) break;
                        }
                        if (bad) outer &= ~top;
                    }
                }
            }
            if (__any(hole != 0)) {
                // the zero run that starts here must have foreground directly above all along
                hole &= ~fill_down(~pc, ~c);
                if (!(c >> 31) && hole) {
                    const uint32_t top = 1u << (32 - __clz(c));  // lowest bit of the zero run ending at bit 31
                    if (hole & top) {
                        bool bad = false;
                        for (int k = bx + 1; k < WS; k++) {
                            const uint32_t zk = ~pw_at(plane, WS, H, k, y);
                            if (!(zk & 1u)) break;
                            const uint32_t nz = ~pw_at(plane, WS, H, k, y - 1);
                            const int tt = __ffs(~zk) - 1;
                            const uint32_t rmask = (tt < 0) ? 0xFFFFFFFFu : ((1u << tt) - 1u);
                            if (nz & rmask) { bad = true; break; }
                            if (tt >= 0) break;
                        }
                        if (bad) hole &= ~top;
                    }
                }
            }
            // ---- append this row's starts: one layer of the wave at a time (a word rarely holds more than one) ----
            if (__any((outer | hole) != 0)) {
                const uint32_t rowPos = (uint32_t)y * W + 32u * bx;
                uint32_t mo = outer, mh = hole;
                for (;;) {
                    const bool has = (mo | mh) != 0;
                    const unsigned long long bal = __ballot(has);
                    if (!bal) break;
                    const unsigned int layer = (unsigned int)__popcll(bal);
                    if (staged + layer > (unsigned int)kStage) flush();
                    if (has) {
                        uint64_t v;
                        if (mo) {
                            v = pack_cand(f, 0, sc, rowPos + (__ffs(mo) - 1));
                            mo &= mo - 1;
                        } else {
                            // the start pixel of a hole border is the pixel left of the hole's first pixel
                            v = pack_cand(f, 1, sc, rowPos + (__ffs(mh) - 1) - 1);
                            mh &= mh - 1;
                        }
                        stage[staged + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull))] = v;
                    }
                    staged += layer;
                }
            }
            // ---- border seeds (internal.h, SegRec): west / east cracks on rows y = 0 / G/2 (mod G), north / south cracks
            //      on columns x = 0 / G/2 (mod G) of the other rows (G = kSeedGrid); a pixel carries at most one seed ----
            {
                const int ym = y & (kSeedGrid - 1);  // uniform
                uint32_t sm;
                if (ym == 0) sm = c & ~cw & (pc | pw | pe) & (sd | sw | se);
                else if (ym == kSeedGrid / 2) sm = c & ~ce & (pc | pw | pe) & (sd | sw | se);
                else sm = c & (pw | cw | sw) & (pe | ce | se) & seed_columns(bx, ~pc, ~sd);
                sm &= om;
                if (__any(sm != 0)) {
                    const uint32_t rowPos = (uint32_t)y * W + 32u * bx;
                    for (;;) {
                        const unsigned long long bal = __ballot(sm != 0);
                        if (!bal) break;
                        const unsigned int layer = (unsigned int)__popcll(bal);
                        if (seedStaged + layer > (unsigned int)kSeedStage) flush_seeds();
                        if (sm) {
                            seedStage[seedStaged + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull))] =
                                pack_cand(f, 0u, sc, rowPos + (__ffs(sm) - 1));
                            sm &= sm - 1;
                        }
                        seedStaged += layer;
                    }
                }
            }
            pc = c; pw = cw; pe = ce;
            cw = sw; ce = se;
        }
        cur = nx1;
        nx1 = nx2;
        nx2 = nx3;
    }
    flush();
    flush_seeds();
}

// 3-bit fields (x-1, x, x+1) of rows y-1, y, y+1 -> neighbour mask, bit d = neighbour in direction d.
// directions: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE.
__device__ __forceinline__ uint32_t pack_mask(uint32_t t, uint32_t m, uint32_t u) {
    return ((m >> 2) & 1u) | (((t >> 2) & 1u) << 1) | (((t >> 1) & 1u) << 2) | ((t & 1u) << 3) | ((m & 1u) << 4) |
           ((u & 1u) << 5) | (((u >> 1) & 1u) << 6) | (((u >> 2) & 1u) << 7);
}

// (x,y) is an interior pixel; the planes have a zero frame.  The three rows are neighbours in one plane block unless
// y is the block's first or last row; offsets are 32-bit words from the plane's base.
__device__ __forceinline__ uint32_t nb_mask(const uint32_t* __restrict__ plane, int WS, int x, int y) {
    const int xm = x - 1, wi = xm >> 5, sh = xm & 31, yr = y & 31;
    const uint32_t blockRow = (uint32_t)WS * 32u;
    const uint32_t i1 = ((uint32_t)(y >> 5) * (uint32_t)WS + (uint32_t)wi) * 32u + (uint32_t)yr;
    const uint32_t i0 = yr == 0 ? i1 - blockRow + 31u : i1 - 1u, i2 = yr == 31 ? i1 + blockRow - 31u : i1 + 1u;
    const uint32_t a = plane[i0], b = plane[i1], c = plane[i2];
    uint32_t a2 = 0, b2 = 0, c2 = 0;
    if (sh > 29) {  // the field straddles two words: same row of the next block
        a2 = plane[i0 + 32];
        b2 = plane[i1 + 32];
        c2 = plane[i2 + 32];
    }
    return pack_mask(__funnelshift_r(a, a2, sh) & 7u, __funnelshift_r(b, b2, sh) & 7u, __funnelshift_r(c, c2, sh) & 7u);
}

// the same mask in the direction numbering of the vertically mirrored image (d -> (8 - d) & 7)
__device__ __forceinline__ uint32_t flip_mask(uint32_t m) {
    // bit d -> bit (8 - d) & 7: the byte reversed (d -> 7 - d) and rotated up by one
    const uint32_t r = __brev(m) >> 24;
    return ((r | (r << 8)) >> 7) & 0xFFu;
}

// direction -> step, two bits per direction holding d + 1
__device__ __forceinline__ int dir_dx(int d) { return (int)((0x901Au >> (2 * d)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int d) { return (int)((0xA901u >> (2 * d)) & 3u) - 1; }

// One border follower with OpenCV's icvFetchContour stepping rule.  FLIP runs the rule on the vertically
// mirrored image, which walks the same border in the opposite sense; it is only used to look for an earlier
// start, never to produce points.
template <bool FLIP>
struct Tracer {
    int x0, y0, x1, y1, cx, cy, s;
    uint32_t m;

    __device__ __forceinline__ uint32_t mask(const uint32_t* plane, int WS, int x, int y) const {
        const uint32_t v = nb_mask(plane, WS, x, y);
        return FLIP ? flip_mask(v) : v;
    }
    __device__ __forceinline__ int dy(int d) const { return FLIP ? -dir_dy(d) : dir_dy(d); }

    // Start at (x, y), whose neighbour mask `raw` (nb_mask, unmirrored) the caller has fetched (both tracers of a
    // candidate start from the same pixel).  The first step goes to the first foreground neighbour clockwise from
    // s_end - 1; returns false for a single-pixel border.
    __device__ __forceinline__ bool init(uint32_t raw, int x, int y, bool isHole) {
        x0 = cx = x;
        y0 = cy = y;
        const int s_end = isHole ? 0 : 4;
        m = FLIP ? flip_mask(raw) : raw;
        // bit j of r = direction s_end + 1 + j: the directions s_end - 1, s_end - 2, ... s_end - 7 are bits 6, 5, ... 0
        const int b = (s_end + 1) & 7;
        const uint32_t r = ((m >> b) | (m << (8 - b))) & 0x7Fu;
        const bool found = r != 0;
        s = found ? (s_end + 1 + (31 - __clz(r))) & 7 : s_end;
        x1 = x0 + dir_dx(s);
        y1 = y0 + dy(s);
        return found;
    }

    // One step.  Reports the pixel being left (px,py) and whether its east crack is on this border.
    // Returns true when the border closed with this step.
    __device__ __forceinline__ bool step(const uint32_t* plane, int WS, int& px, int& py, bool& passedEast) {
        const int s_end = s;
        const uint32_t rot = ((m >> (s_end + 1)) | (m << (7 - s_end))) & 0xFFu;  // bit j <-> direction s_end+1+j
        const int k = __ffs(rot);  // 1..8: m always holds the neighbour we came from
        s = (s_end + k) & 7;
        passedEast = s_end + k >= 9;
        px = cx;
        py = cy;
        const int nx = cx + dir_dx(s), ny = cy + dy(s);
        const bool closed = (nx == x0 && ny == y0 && cx == x1 && cy == y1);
        cx = nx;
        cy = ny;
        s = (s + 4) & 7;
        if (!closed) m = mask(plane, WS, cx, cy);
        return closed;
    }
};

__device__ __forceinline__ uint32_t pack_state(int x, int y, int sdir) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)sdir << 24); }

// ---- long borders -------------------------------------------------------------------------------
// Border following is a walk on states (pixel, direction to the predecessor): the successor T(P, s) is
// found by sweeping counter-clockwise from s + 1 to the first foreground neighbour d, giving (P + d, d + 4);
// the predecessor is found by the same sweep run clockwise.  T is a bijection, a border is one of its cycles,
// and OpenCV's contour is the list of pixels along the cycle from the start state.
//
// Because T is a bijection a border can be entered anywhere: the long borders are followed in segments from seed to
// seed (k_seg_walk2 below), which makes the longest dependent chain one segment instead of one border; the segments'
// records are linked and the start rule applied afterwards (k_seg_link, k_resolve).
//
// The followers are persistent; every lane follows its border inside a private LDS window of its bit plane: 2 x 2
// pieces of 32 columns x kPieceRows rows (a piece = kPieceRows consecutive words of a plane block), 64 columns x
// 2 kPieceRows rows, addressed toroidally (row y mod 2 kPieceRows, word (x >> 5) & 1), so moving the window by one
// piece only loads the pieces that are new: 2 kPieceRows x 8 bytes of LDS per lane.  A step then costs three LDS reads
// instead of dependent trips to L2/HBM.  Windows are moved in the followers' service phases, for all lanes that are
// close to an edge at once; lanes that finish take new work from a queue.
constexpr int kPL = 3;                       // log2(rows per piece): 8-row pieces, 8 KiB of LDS per wavefront
constexpr int kPieceRows = 1 << kPL;
constexpr int kWinRows = 2 * kPieceRows;
constexpr int kTileWords = kWinRows * 2 * 64;  // LDS words per wavefront

struct TileWin {
    int gx, gy;  // window origin in pieces: columns [32 gx, 32 gx + 64), rows [gy << kPL, (gy + 2) << kPL)

    // no window loaded yet: an origin so far out that nothing is covered and every piece of the next move is new
    __device__ __forceinline__ void invalidate() { gx = -(1 << 20); gy = 0; }
    // the 3x3 neighbourhood of (x, y) lies inside the window
    __device__ __forceinline__ bool covers(int x, int y) const {
        return (unsigned)(x - gx * 32 - 1) < 62u && (unsigned)(y - gy * kPieceRows - 1) < (unsigned)(kWinRows - 2);
    }
    // close to leaving the window (never true right after recentre(), which leaves >= 16 columns and
    // kPieceRows / 2 rows of margin)
    __device__ __forceinline__ bool near_edge(int x, int y) const {
        constexpr int my = kPieceRows / 2 - 1;  // row margin that triggers a move along with the others
        return !((unsigned)(x - gx * 32 - 12) < 40u && (unsigned)(y - gy * kPieceRows - my) < (unsigned)(kWinRows - 2 * my));
    }
    // tile word index: (row * 64 + lane) * 2 + column, i.e. one 64-bit read fetches both words of a row.
    // All the loads of a move are issued before the first LDS write: one memory latency per move instead of one
    // per piece.  Pieces beyond the plane are loaded from the nearest piece inside it instead of being zeroed: they
    // are never looked at, because the bit planes carry a zero frame one pixel wide (k_gray_thresh3), so a border
    // pixel has 1 <= x <= W - 2 and 1 <= y <= H - 2 and its neighbourhood lies inside the image.  A move costs the
    // whole wave its instructions whenever one lane needs it, which is at almost every opportunity: it is kept lean.
    __device__ __forceinline__ void recentre(const uint32_t* __restrict__ plane, int WS, int H, int x, int y,
                                                     uint32_t* __restrict__ tile, int lane) {
        const int ngx = (x - 16) >> 5, ngy = (y - kPieceRows / 2) >> kPL;
        const int pieceRows = plane_hb(H) << (5 - kPL);
        const int dx = ngx - gx, dy = ngy - gy;  // piece (ngx + i, ngy + j) is in the old window iff dx + i and dy + j are 0 or 1
        uint4 v[4][kPieceRows / 4];
        bool fresh[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = p & 1, j = p >> 1;
            fresh[p] = (unsigned)(dx + i) > 1u || (unsigned)(dy + j) > 1u;
            if (fresh[p]) {
                const int hx = min(max(ngx + i, 0), WS - 1), hy = min(max(ngy + j, 0), pieceRows - 1);
                const uint4* src = reinterpret_cast<const uint4*>(
                    plane + (unsigned)(((hy >> (5 - kPL)) * WS + hx) * 32 + (hy & ((32 >> kPL) - 1)) * kPieceRows));
#pragma unroll
                for (int q = 0; q < kPieceRows / 4; q++) v[p][q] = src[q];
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (!fresh[p]) continue;
            const int hx = ngx + (p & 1), hy = ngy + (p >> 1);
            uint32_t* dst = tile + ((hy & 1) * kPieceRows) * 128 + lane * 2 + (hx & 1);
#pragma unroll
            for (int q = 0; q < kPieceRows / 4; q++) {
                dst[(4 * q + 0) * 128] = v[p][q].x;
                dst[(4 * q + 1) * 128] = v[p][q].y;
                dst[(4 * q + 2) * 128] = v[p][q].z;
                dst[(4 * q + 3) * 128] = v[p][q].w;
            }
        }
        gx = ngx;
        gy = ngy;
    }
    __device__ __forceinline__ uint32_t mask(const uint32_t* __restrict__ tile, int lane, int x, int y) const {
        const uint2* t2 = reinterpret_cast<const uint2*>(tile) + lane;
        const int xm = x - 1;
        const bool swap = (xm >> 5) & 1;
        const uint2 r0 = t2[((y - 1) & (kWinRows - 1)) * 64], r1 = t2[(y & (kWinRows - 1)) * 64],
                    r2 = t2[((y + 1) & (kWinRows - 1)) * 64];
        // funnel shift of (next word : this word) by xm & 31 brings pixel x-1 to bit 0
        const uint32_t t = __funnelshift_r(swap ? r0.y : r0.x, swap ? r0.x : r0.y, xm) & 7u;
        const uint32_t m = __funnelshift_r(swap ? r1.y : r1.x, swap ? r1.x : r1.y, xm) & 7u;
        const uint32_t u = __funnelshift_r(swap ? r2.y : r2.x, swap ? r2.x : r2.y, xm) & 7u;
        // same bit order as pack_mask: NE,N,NW are the row above reversed
        return (__brev(t) >> 28) | ((m >> 2) & 1u) | ((m & 1u) << 4) | (u << 5);
    }
};


// the initial clockwise search of a fresh start (s_end - 1, s_end - 2, ...): direction of the neighbour the
// border closes through, or -1 for a single pixel.  Rotated so that bit j is direction s_end + j, the first
// hit clockwise is the highest of bits 1..7.
__device__ __forceinline__ int fresh_dir(uint32_t m, bool isHole) {
    const int s_end = isHole ? 0 : 4;
    const uint32_t r = ((m >> s_end) | (m << (8 - s_end))) & 0xFEu;
    return r ? ((s_end + 31 - __clz(r)) & 7) : -1;
}

// the lane's partner in its pair (lane ^ 1)
__device__ __forceinline__ uint32_t from_partner(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}

// A wave takes work from a global queue in batches and hands the items to its idle lanes, and it reserves
// output space (points, contour records, store tasks) in blocks that it sub-allocates itself: the global
// counters are touched once per block instead of once per border (they are single hot addresses).
constexpr unsigned int kSlotBlock = 8, kStoreBatch = 256;

struct WaveQueue {
    unsigned int lo = 0, hi = 0, base = 0;
    bool drained = false;
    // The first batch of every wave is fixed (wave w of nWaves takes batch w), only the later ones come from the shared
    // counter: thousands of waves that all start with an atomic on one address wait in line for it (same-address
    // atomics are served one at a time), which was most of the run time of the passes with little work per item.
    __device__ __forceinline__ void start(unsigned int wave, unsigned int nWaves, unsigned int n, unsigned int batch) {
        base = nWaves * batch;
        const unsigned int b = wave < nWaves ? wave * batch : n;  // (wave >= nWaves: no batch of its own)
        lo = b < n ? b : n;
        hi = b + batch < n ? b + batch : n;
        if (base >= n) drained = true;
    }
    // draw the next batch from the shared counter (uniform call)
    __device__ __forceinline__ void refill(unsigned int* head, unsigned int n, unsigned int batch, int lane) {
        unsigned int b = 0;
        if (lane == 0) b = atomicAdd(head, batch);
        b = __builtin_amdgcn_readfirstlane(b) + base;
        lo = b < n ? b : n;
        hi = b + batch < n ? b + batch : n;
        if (b + batch >= n) drained = true;
    }
    // every lane calls this; returns the item index for lanes with want == true (rank order) or ~0u if none is left
    // for them right now.  `unit` lanes form one consumer (pairs: unit = 2, the even lane's want counts).
    template <int UNIT>
    __device__ __forceinline__ unsigned int pop(bool want, unsigned int* head, unsigned int n, unsigned int batch, int lane) {
        unsigned long long bal = __ballot(want);
        if (UNIT == 2) bal &= 0x5555555555555555ull;
        if (!bal) return ~0u;
        if (lo == hi && !drained) refill(head, n, batch, lane);
        const unsigned int rank = (unsigned int)__popcll(bal & ((1ull << (lane & ~(UNIT - 1))) - 1ull));
        const unsigned int avail = hi - lo, cnt = (unsigned int)__popcll(bal);
        const unsigned int idx = (want && rank < avail) ? lo + rank : ~0u;
        lo += cnt < avail ? cnt : avail;
        return idx;
    }
    __device__ __forceinline__ bool empty() const { return drained && lo == hi; }
};

// Items per batch for n items over nWaves wavefronts: the full batch when there is plenty of work; otherwise just enough
// for every item to be in some wavefront's first (static) batch, so that a small launch spreads over the whole chip
// and touches the shared counter not at all (one 1080p frame has ~1000 seeds: with batches of 128 eight wavefronts
// did all the work while 4088 looked on, and the pass took 375 us of the frame's 930).
__device__ __forceinline__ unsigned int spread_batch(unsigned int n, unsigned int nWaves, unsigned int full) {
    const unsigned int even = (n + nWaves - 1u) / (nWaves ? nWaves : 1u);
    return even < 1u ? 1u : even < full ? even : full;
}

struct WaveAlloc {
    unsigned int cur = 0, end = 0;
    // set when a refill did not fit: the entries between the counter's old value and the capacity belong to
    // nobody, and the caller must make them read as empty (the next pass looks at every entry below the capacity)
    unsigned int orphanLo = 0, orphanHi = 0;
    // uniform call: reserve `need` consecutive entries; refills from the global counter by blocks.
    // Returns ~0u (and raises the overflow bit) when the list is full.  `left` receives the abandoned tail
    // [left.x, left.y) of the previous block when a refill happens (the caller marks those entries empty).
    __device__ __forceinline__ unsigned int take(unsigned int need, unsigned int block, unsigned int* counter, unsigned int cap,
                                                 unsigned int ovfBit, Counters* ctr, int lane, uint2& left) {
        left = make_uint2(0, 0);
        if (cur + need > end) {
            left = make_uint2(cur, end);
            const unsigned int sz = need > block ? need : block;
            unsigned int b = 0;
            if (lane == 0) b = atomicAdd(counter, sz);
            b = __builtin_amdgcn_readfirstlane(b);
            if (b + sz > cap) {
                if (lane == 0) atomicOr(&ctr->overflow, ovfBit);
                cur = end = 0;
                if (b < cap) {
                    orphanLo = b;
                    orphanHi = cap;
                }
                return ~0u;
            }
            cur = b;
            end = b + sz;
        }
        const unsigned int r = cur;
        cur += need;
        return r;
    }
};

__device__ __forceinline__ void mark_empty_contours(ContourRec* __restrict__ contours, unsigned int lo, unsigned int hi, int lane) {
    for (unsigned int i = lo + lane; i < hi; i += 64) {
        ContourRec r;
        r.frame = 0xFFFFFFFFu; r.pos = 0; r.len = 0; r.off = 0; r.scale = 0; r.hole = 0; r.scanKey = 0;
        contours[i] = r;
    }
}
__device__ __forceinline__ void mark_empty_tasks(StoreTask* __restrict__ tasks, unsigned int lo, unsigned int hi, int lane) {
    for (unsigned int i = lo + lane; i < hi; i += 64) {
        StoreTask k;
        k.plane = 0; k.state = 0; k.dst = 0; k.count = 0;
        tasks[i] = k;
    }
}

__device__ __forceinline__ void mark_empty_tasks(CopyTask* __restrict__ tasks, unsigned int lo, unsigned int hi, int lane) {
    for (unsigned int i = lo + lane; i < hi; i += 64) {
        CopyTask k;
        k.chunk = 0; k.start = 0; k.dst = 0; k.n = 0;
        tasks[i] = k;
    }
}
// after a failed reservation: the entries nobody owns (see WaveAlloc) read as empty
template <class Task>
__device__ __forceinline__ void adopt_orphans(WaveAlloc& slotA, WaveAlloc& taskA, ContourRec* __restrict__ contours,
                                              Task* __restrict__ tasks, int lane) {
    if (slotA.orphanHi > slotA.orphanLo) {
        mark_empty_contours(contours, slotA.orphanLo, slotA.orphanHi, lane);
        slotA.orphanLo = slotA.orphanHi = 0;
    }
    if (taskA.orphanHi > taskA.orphanLo) {
        mark_empty_tasks(tasks, taskA.orphanLo, taskA.orphanHi, lane);
        taskA.orphanLo = taskA.orphanHi = 0;
    }
}

// ---- short borders ------------------------------------------------------------------------------
// Most start candidates are noise specks or false starts that are decided within a handful of steps, a few
// per cent need the whole budget.  Lanes therefore work independently: a lane that has decided its candidate
// takes the next one from the wave's batch, instead of the wave waiting for its slowest lane every round
// (which made 94 % of the wave-steps idle).  Each step advances the candidate's forward tracer and, in
// lock-step, a tracer walking the border the other way round; both look for a crack with a smaller key.
// steps a candidate gets here before it goes to the long list (measured on the bench scene, ms of the whole border
// stage: 8: 5.10, 16: 4.34, 20: 4.31, 24: 4.33, 32: 4.43, 48: 4.56, 64: 4.79 -- the 5 % of the candidates that use the
// whole budget cost a third of this pass's iterations at 48, and the segment followers take them over for less).
// The two followers also notice when they meet (SHORT_MEET): a border of up to twice the budget is then known all
// round -- its length and that no crack on it has a smaller key -- and is decided here (a budget of 30 would decide
// every border below 1080p's minimum perimeter of 58 points, but costs more than it saves).  Border stage with it,
// bench / photo scene in ms, budget 16: 3.88 / 7.02,
// 20: 3.90 / 6.99, 24: 3.94 / 7.19, 30: 4.06 / 7.37; without it at 24: 3.95 / 7.42.
#ifndef SHORT_BUDGET
#define SHORT_BUDGET 20
#endif
#ifndef SHORT_MEET
#define SHORT_MEET 1
#endif
constexpr int kShortBudget = SHORT_BUDGET;
constexpr bool MEET = SHORT_MEET != 0;
constexpr unsigned int kWalkBatch = 128, kResolveBatch = 64;
constexpr unsigned int kShortBatch = 256, kShortPoolBlock = 2048, kLongStage = 64;

__global__ __launch_bounds__(256) void k_trace_short(const uint32_t* __restrict__ planes, DetCfg cfg,
                                                     const uint64_t* __restrict__ cand, unsigned int candCap,
                                                     uint64_t* __restrict__ candLong, ContourRec* __restrict__ contours,
                                                     unsigned int contourCap, unsigned int poolCap,
                                                     StoreTask* __restrict__ tasks, unsigned int taskCap, Counters* __restrict__ ctr) {
    __shared__ uint64_t s_long[4][kLongStage];
    __shared__ uint64_t s_batch[4][kShortBatch];
    const int lane = threadIdx.x & 63;
    uint64_t* const stage = s_long[threadIdx.x >> 6];
    uint64_t* const batch = s_batch[threadIdx.x >> 6];
    unsigned int staged = 0;  // uniform
    const unsigned int n = min(ctr->nCand, candCap);
    const int H = cfg.H, W = cfg.W, WS = cfg.WS, S = cfg.nScales;
    const size_t perPlane = plane_words(H, WS);
    (void)H;

    auto flush_long = [&]() {
        if (staged) {
            unsigned int g = 0;
            if (lane == 0) g = atomicAdd(&ctr->nLong, staged);
            g = __builtin_amdgcn_readfirstlane(g);
            if (g + staged > candCap && lane == 0) atomicOr(&ctr->overflow, 1u);
            const unsigned int fit = g >= candCap ? 0u : min(staged, candCap - g);  // see k_contour_starts
            for (unsigned int k = lane; k < fit; k += 64) candLong[g + k] = stage[k];
            staged = 0;
        }
    };

    WaveQueue queue;
    const unsigned int qBatch = spread_batch(n, gridDim.x * (blockDim.x >> 6), kShortBatch);
    queue.start(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), gridDim.x * (blockDim.x >> 6), n, qBatch);
    // The wave's batch of candidates is read into LDS in one coalesced sweep when the batch is drawn: a lane that takes
    // up a candidate then has it after an LDS read instead of a trip to memory that the whole wave waits for.
    unsigned int batchLo = queue.lo;
    auto load_batch = [&]() {
        batchLo = queue.lo;
        for (unsigned int k = lane; k < queue.hi - queue.lo; k += 64) batch[k] = cand[queue.lo + k];
        wave_lds_sync();
    };
    load_batch();
    WaveAlloc poolA, slotA, taskA;
    bool active = false, exhausted = false, isHole = false, bwActive = false, fresh = false;
    uint64_t c = 0;
    int count = 0, startKey = 0;
    int bprev = -1;  // the pixel the follower that walks the other way round has just left
    const uint32_t* plane = planes;
    // the candidate's start pixel; per follower (f forwards, b the other way round, on the vertically mirrored image):
    // the second pixel, the current pixel and the direction to the predecessor
    int x0 = 1, y0 = 1, fx1 = 1, fy1 = 1, fcx = 1, fcy = 1, fs = 0, bx1 = 1, by1 = 1, bcx = 1, bcy = 1, bs = 0;

    // Taking up candidates is the expensive part of an iteration for what it does (a seventh of the lanes finish in
    // any one iteration, and the whole wave pays for their set-up), so it happens every kPick-th iteration only and
    // is kept to decoding the candidate: a fresh lane's followers get their start state in the step phase, from the
    // neighbour mask that every active lane fetches at the top of an iteration anyway.
    constexpr unsigned int kPick = 3;
    bool anyActive = false;  // uniform
    for (unsigned int iter = 0;; iter++) {
        if (iter % kPick == 0 || !anyActive) {
            const bool need = !active && !exhausted;
            if (queue.lo == queue.hi && !queue.drained && __any(need)) {
                queue.refill(&ctr->qShort, n, qBatch, lane);
                load_batch();
            }
            const unsigned int i = queue.pop<1>(need, &ctr->qShort, n, qBatch, lane);
            if (need) {
                if (i != ~0u) {
                    c = batch[i - batchLo];
                    if (c != ~0ull) {  // ~0 would mark a slot that was never filled
                        const uint32_t pos = (uint32_t)(c & 0xFFFFFFFFull);
                        plane = planes + ((size_t)(uint32_t)(c >> 40) * S + (int)((c >> 36) & 7u)) * perPlane;
                        isHole = (c >> 39) & 1u;
                        y0 = row_of(cfg, pos);
                        x0 = (int)pos - y0 * W;
                        fcx = bcx = x0;
                        fcy = bcy = y0;
                        startKey = (int)pos;
                        count = 0;
                        bprev = -1;
                        active = fresh = bwActive = true;
                    }
                } else if (queue.empty()) {
                    exhausted = true;
                }
            }
        }
        anyActive = __any(active);
        if (!anyActive) {
            if (__all(exhausted)) break;
            continue;
        }
        // neighbour masks at the followers' pixels (a fresh lane's two followers stand on the start pixel)
        uint32_t mf = 0, mb = 0;
        if (active) {
            mf = nb_mask(plane, WS, fcx, fcy);
            if (bwActive) mb = flip_mask(nb_mask(plane, WS, bcx, bcy));
        }
        // res: -2 keep going, -1 budget exhausted (long list), 0 drop, > 0 closed with that many points
        int res = -2;
        if (active && fresh) {
            // start states (icvFetchContour): the first step goes to the first foreground neighbour clockwise from
            // s_end - 1; bit j of r = direction s_end + 1 + j, so that neighbour is the highest set bit of bits 0..6
            const int s_end = isHole ? 0 : 4, b = (s_end + 1) & 7;
            const uint32_t rf = ((mf >> b) | (mf << (8 - b))) & 0x7Fu, rb = ((mb >> b) | (mb << (8 - b))) & 0x7Fu;
            const bool found = rf != 0;  // (and then rb != 0: the same neighbours, mirrored)
            fs = found ? (s_end + 1 + (31 - __clz(rf))) & 7 : s_end;
            bs = found ? (s_end + 1 + (31 - __clz(rb))) & 7 : s_end;
            fx1 = x0 + dir_dx(fs);
            fy1 = y0 + dir_dy(fs);
            bx1 = x0 + dir_dx(bs);
            by1 = y0 - dir_dy(bs);
            bwActive = found;
            fresh = false;
            if (!found) res = 1;  // a single pixel
        }
        if (active && res == -2) {
            // one step forwards: sweep counter-clockwise from fs + 1 (mf always holds the neighbour we came from)
            const uint32_t rot = ((mf >> (fs + 1)) | (mf << (7 - fs))) & 0xFFu;
            const int k = __ffs(rot), d = (fs + k) & 7;
            const bool east = fs + k >= 9;  // the east crack of the pixel being left is on this border
            const int key = fcy * W + fcx, nx = fcx + dir_dx(d), ny = fcy + dir_dy(d);
            const bool closed = nx == x0 && ny == y0 && fcx == fx1 && fcy == fy1;
            fcx = nx;
            fcy = ny;
            fs = d ^ 4;
            const int fnew = ny * W + nx;
            const int bkey = bcy * W + bcx;  // where the other follower stands
            if ((!isHole || east) && key < startKey) {
                res = 0;
            } else {
                count++;
                if (closed) res = count;
                // the two followers have just walked the same step from either end: between them they have been all
                // round the border (every crack looked at by one of them), which is 2 count - 2 points long
                else if (MEET && bwActive && fnew == bprev && key == bkey) res = 2 * count - 2;
                else if (count >= kShortBudget) res = -1;
                else if (bwActive) {
                    // ... and one the other way round, on the mirrored image (rows run upwards)
                    const uint32_t brot = ((mb >> (bs + 1)) | (mb << (7 - bs))) & 0xFFu;
                    const int bk = __ffs(brot), bd = (bs + bk) & 7;
                    const bool beast = bs + bk >= 9;
                    const int bnx = bcx + dir_dx(bd), bny = bcy - dir_dy(bd);
                    if (bnx == x0 && bny == y0 && bcx == bx1 && bcy == by1) bwActive = false;
                    bcx = bnx;
                    bcy = bny;
                    bs = bd ^ 4;
                    bprev = bkey;
                    if ((!isHole || beast) && bkey < startKey) res = 0;
                    // (the same, after this follower's step: 2 count - 1 points)
                    else if (MEET && fnew == bkey && key == bny * W + bnx) res = 2 * count - 1;
                }
            }
        }
        // ---- undecided candidates go to the long list through the wave's staging list ----
        {
            const bool und = active && res == -1;
            const unsigned long long bal = __ballot(und);
            if (bal) {
                const unsigned int cnt = (unsigned int)__popcll(bal);
                if (staged + cnt > kLongStage) flush_long();
                if (und) stage[staged + (unsigned int)__popcll(bal & ((1ull << lane) - 1ull))] = c;
                staged += cnt;
            }
        }
        // ---- borders that closed within the perimeter bounds: space from the wave's blocks, one store task ----
        const bool accept = active && res >= cfg.minPerim && res <= cfg.maxPerim;
        unsigned int myOff = ~0u, mySlot = ~0u, myTask = ~0u;
        unsigned long long am = __ballot(accept);
        while (am) {
            const int l = __ffsll((long long)am) - 1;
            am &= am - 1;
            const unsigned int len = (unsigned int)__builtin_amdgcn_readlane(res, l);
            uint2 left;
            const unsigned int off = poolA.take(len, kShortPoolBlock, &ctr->nPool, poolCap, 4u, ctr, lane, left);
            const unsigned int slot = slotA.take(1, kSlotBlock, &ctr->nContours, contourCap, 2u, ctr, lane, left);
            const unsigned int t0 = taskA.take(1, kSlotBlock, &ctr->nTasks, taskCap, 64u, ctr, lane, left);
            if (off == ~0u || slot == ~0u || t0 == ~0u) {
                // a list is full (the call will fail with CAMSYNC_E_CAPACITY): what was obtained must still read
                // as empty, the later passes look at every reserved entry
                if (slot != ~0u) mark_empty_contours(contours, slot, slot + 1, lane);
                if (t0 != ~0u) mark_empty_tasks(tasks, t0, t0 + 1, lane);
                adopt_orphans(slotA, taskA, contours, tasks, lane);
            }
            if (lane == l && off != ~0u && slot != ~0u && t0 != ~0u) {
                myOff = off;
                mySlot = slot;
                myTask = t0;
            }
        }
        if (active && res != -2) {
            active = false;
            if (accept && myOff != ~0u) {
                const uint32_t pos = (uint32_t)(c & 0xFFFFFFFFull);
                ContourRec r;
                r.frame = (uint32_t)(c >> 40);
                r.pos = pos;
                r.len = (uint32_t)res;
                r.off = myOff;
                r.scale = (uint16_t)((c >> 36) & 7u);
                r.hole = isHole ? 1 : 0;
                r.scanKey = pos + (isHole ? 1u : 0u);
                contours[mySlot] = r;
                // a border this short is one store task from its fresh start
                StoreTask k;
                k.plane = r.frame * S + r.scale;
                k.state = pack_state(pos % W, pos / W, 8 + (isHole ? 1 : 0));
                k.dst = myOff;
                k.count = (uint32_t)res;
                tasks[myTask] = k;
            }
        }
    }
    flush_long();
    // entries reserved but never used must read as empty
    mark_empty_contours(contours, slotA.cur, slotA.end, lane);
    mark_empty_tasks(tasks, taskA.cur, taskA.end, lane);
}

// ---- long borders in segments (internal.h, SegRec) ------------------------------------------------------
// Which crack of pixel (x, y) can be a seed: direction of the background neighbour (4 west, 0 east, 2 north, 6 south).
// Rows 0 / G/2 (mod G) carry west / east seeds, the other rows north / south seeds at columns 0 / G/2 (mod G);
// -1: the pixel cannot be a seed.  The start pass applies the same rule word-wide (seed_columns).
__device__ __forceinline__ int seed_dir(int x, int y) {
    // (selects only: as nested conditions this compiled to branches inside the walkers' step)
    constexpr int G2 = kSeedGrid / 2;
    static_assert(kSeedGrid == 64, "the shifts below assume a 64-pixel grid: bit 5 of a coordinate tells the two kinds of line apart");
    const int onRow = (y & (G2 - 1)) == 0, onCol = (x & (G2 - 1)) == 0;
    const int dRow = 4 ^ ((y >> 3) & 4);      // row 0 (mod 64): 4 (west), row 32: 0 (east)
    const int dCol = 2 + ((x >> 3) & 4);      // column 0 (mod 64): 2 (north), column 32: 6 (south)
    return onRow ? dRow : onCol ? dCol : -1;
}
// ... and the border must go on on both sides of the crack (m = neighbour mask, bit d = direction d)
__device__ __forceinline__ bool seed_supported(uint32_t m, int dir) {
    return (dir & 2) ? ((m & 0x38u) && (m & 0x83u))   // north / south crack: something to the left and to the right
                     : ((m & 0x0Eu) && (m & 0xE0u));  // west / east crack: something above and below
}

// ---- the segment walkers --------------------------------------------------------------------------------------
// One lane per work item: items [0, nSeeds) are seeds, [nSeeds, nSeeds + nLong) the start candidates the short pass
// left undecided.  A lane follows its border forwards (successor map T, inside its private LDS window) and records
// the direction of every step, until its sweep passes the crack of a seed: the segment ends there and the record
// keeps that seed's key (k_seg_link turns it into an id afterwards: a table lookup here would stall the whole wave at
// almost every step).  After kSegLen points without a seed the lane closes the segment with a continuation record
// and goes on in it.  A start candidate also stops when it is back at its own start state (a border without seeds)
// and gives up as soon as it meets a crack with a smaller raster key (it is not the canonical start) or exceeds the
// length bound.  Lanes hold their next work item in registers, so that taking it up does not wait for memory.
// CANDS = false: the work items are the seeds; true: the long start candidates, whose step also keeps the start rule's
// books (smaller key met: not the canonical start; closure without a seed; length bound).  Seeds are five items in six
// and their step is shorter, so a wave works on one kind at a time.  `wave` of `nWaves` take a first batch of items
// each without asking the shared counter (wave >= nWaves: this wave only helps to drain the queue).
//
// Everything that touches global memory or is rare -- writing a finished segment's record, taking up the next work
// item, moving a lane's window -- happens in a service phase every kWalk2Steps steps, for all lanes that need it at
// once (with 64 independent lanes whatever one lane needs once in its life the wavefront executes nearly every time:
// a service phase costs about as many instructions as five steps).  Between two phases the steps are written out, LDS
// and registers only; a lane that runs out of window or finishes its segment waits for the next phase.
//
// A step is a table lookup at heart: the 3x3 neighbourhood and the direction to the predecessor decide the next pixel,
// whether the sweep passes the pixel's east crack, and -- for each of the four cracks a seed can sit on -- whether the
// sweep passes it with the border going on on both sides.  The table (4096 entries of 16 bits, built per workgroup at
// the start of the kernel) lives in LDS:
//   index  = row above (x-1, x, x+1) | row << 3 | row below << 6 | direction to the predecessor << 9
//   bits 0-2  q, the direction of the step            bit 3  the sweep passes the east crack
//   bits 4-11 a two-bit field per crack (west, east, north, south: the order rows y = 0 / 32 (mod 64) and columns
//             x = 0 / 32 (mod 64) select them in): bit 0 the sweep passes it and the border goes on on both sides,
//             bit 1 it also passed the east crack before it
//   bits 12-13 dx + 1, bits 14-15 dy + 1 of the step
// (the arithmetic version of the step was ~115 vector instructions, this one ~75).  A step's place in its group's
// chain word is static; the window has its rows y - 1 and y + 1 next to row y at either end as well (18 physical rows:
// row 0 repeats row 16, row 17 repeats row 1), so one address and two immediate offsets fetch the three rows; the two
// words of a row are read by the word the neighbourhood starts in and the other one (address ^ 4), which a funnel shift
// joins.  Steps per service phase, border stage of the bench scene / the photo scene in ms: 4: 4.29 / 8.15, 6: 4.16 /
// 7.84, 8: 4.04 / 7.66, 12: 3.95 / 7.41, 16: 3.98 / 7.41 (a group's chain codes are one 64-bit word; with more steps
// between two phases more lanes stand at the edge of their window waiting for the next one).
constexpr int kWin2Rows = kWinRows + 2;
constexpr int kTile2Words = kWin2Rows * 2 * 64;  // per wavefront
#ifndef WALK2_STEPS
#define WALK2_STEPS 12
#endif
constexpr int kWalk2Steps = WALK2_STEPS;        // steps between two service phases
static_assert(kWalk2Steps >= 2 && kWalk2Steps <= 16, "a group's chain codes are one 64-bit word");
constexpr int kWalk2Waves = 5;                  // wavefronts per workgroup (they share the step table): 53 KB of LDS, three workgroups per CU

struct TileWin2 {
    int gx, gy;  // as TileWin
    __device__ __forceinline__ void invalidate() { gx = -(1 << 20); gy = 0; }
    __device__ __forceinline__ bool near_edge(int x, int y) const {
        // (a move leaves the pixel 16..47 columns and 4..11 rows inside the window)
        // rows 3..12 of 16: one row of hysteresis either side of where a move puts the pixel (a border that runs along a
        // piece boundary would otherwise move its window at every other step)
        return !((unsigned)(x - gx * 32 - 12) < 40u && (unsigned)(y - gy * kPieceRows - 3) < 10u);
    }
    // the 3x3 neighbourhood of (x, y) lies inside the window
    __device__ __forceinline__ bool covers(int x, int y) const {
        return (unsigned)(x - gx * 32 - 1) < 62u && (unsigned)(y - gy * kPieceRows - 1) < (unsigned)(kWinRows - 2);
    }
    // Move the window so that (x, y) is 16..47 columns and 4..11 rows inside it: only the pieces the new window does
    // not share with the old one are loaded.  Word index of (physical row p, lane, column word w): (p * 64 + lane) * 2 + w.
    // (Letting the lanes walk on in the old window while the loads are under way and committing them a phase later
    // keeps 32 more registers alive, 136 in all, and was slower: 5.6 against 4.3 ms for the border stage even when
    // capped at 128.  The wavefronts are not waiting for memory: SQ_WAIT_INST_ANY is a tenth of their cycles.)
    __device__ __forceinline__ void recentre(const uint32_t* __restrict__ plane, int WS, int H, int x, int y,
                                             uint32_t* __restrict__ tile, int lane) {
        const int ngx = (x - 16) >> 5, ngy = (y - kPieceRows / 2) >> kPL;
        const int pieceRows = plane_hb(H) << (5 - kPL);
        const int dx = ngx - gx, dy = ngy - gy;
        uint4 v[4][kPieceRows / 4];
        bool fresh[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int i = p & 1, j = p >> 1;
            fresh[p] = (unsigned)(dx + i) > 1u || (unsigned)(dy + j) > 1u;
            if (fresh[p]) {
                const int hx = min(max(ngx + i, 0), WS - 1), hy = min(max(ngy + j, 0), pieceRows - 1);
                const uint4* src = reinterpret_cast<const uint4*>(
                    plane + (unsigned)(((hy >> (5 - kPL)) * WS + hx) * 32 + (hy & ((32 >> kPL) - 1)) * kPieceRows));
#pragma unroll
                for (int q = 0; q < kPieceRows / 4; q++) v[p][q] = src[q];
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (!fresh[p]) continue;
            const int hx = ngx + (p & 1), hy = ngy + (p >> 1);
            const int upper = hy & 1;
            uint32_t* dst = tile + (upper * kPieceRows + 1) * 128 + lane * 2 + (hx & 1);
#pragma unroll
            for (int q = 0; q < kPieceRows / 4; q++) {
                dst[(4 * q + 0) * 128] = v[p][q].x;
                dst[(4 * q + 1) * 128] = v[p][q].y;
                dst[(4 * q + 2) * 128] = v[p][q].z;
                dst[(4 * q + 3) * 128] = v[p][q].w;
            }
            tile[(upper ? 0 : 17) * 128 + lane * 2 + (hx & 1)] = upper ? v[p][1].w : v[p][0].x;
        }
        gx = ngx;
        gy = ngy;
    }
    // the three 3-bit fields of the neighbourhood of (x, y): rows y - 1, y, y + 1, bit 0 = column x - 1
    __device__ __forceinline__ void fields(const uint32_t* __restrict__ tile, int lane, int x, int y, uint32_t& t, uint32_t& m, uint32_t& u) const {
        const int xm = x - 1;
        const uint32_t a = (uint32_t)(((y & (kWinRows - 1)) + 1) * 128 + lane * 2) + (((uint32_t)xm >> 5) & 1u), b = a ^ 1u;
        t = __funnelshift_r(tile[a - 128], tile[b - 128], xm) & 7u;
        m = __funnelshift_r(tile[a], tile[b], xm) & 7u;
        u = __funnelshift_r(tile[a + 128], tile[b + 128], xm) & 7u;
    }
};

// ring mask (bit d = neighbour in direction d) from the three fields
__device__ __forceinline__ uint32_t ring_of_fields(uint32_t t, uint32_t m, uint32_t u) { return pack_mask(t, m, u); }

__device__ __forceinline__ uint16_t walk_lut_entry(int idx) {
    const uint32_t t = idx & 7, m = (idx >> 3) & 7, u = (idx >> 6) & 7;
    const int s = idx >> 9;
    const uint32_t M = pack_mask(t, m, u);
    const int b0 = (s + 1) & 7;
    const uint32_t rot = ((M >> b0) | (M << (8 - b0))) & 0xFFu;
    const int j = __builtin_ctz(rot | 0x100u);
    const int q = (s + 1 + j) & 7;
    const int tEast = (0 - s) & 7;
    const int east = (unsigned int)(tEast - 1) < (unsigned int)j ? 1 : 0;
    uint32_t e = (uint32_t)q | ((uint32_t)east << 3);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sd = k == 0 ? 4 : k == 1 ? 0 : k == 2 ? 2 : 6;
        const int tSeed = (sd - s) & 7;
        const bool passes = (unsigned int)(tSeed - 1) < (unsigned int)j;
        const bool pass = passes && seed_supported(M, sd);
        const bool eb = pass && east && tEast < tSeed;
        e |= ((pass ? 1u : 0u) | (eb ? 2u : 0u)) << (4 + 2 * k);
    }
    e |= ((0x901Au >> (2 * q)) & 3u) << 12;
    e |= ((0xA901u >> (2 * q)) & 3u) << 14;
    return (uint16_t)e;
}

template <bool CANDS>
__device__ __forceinline__ void seg_walk_lean(const uint32_t* __restrict__ planes, const DetCfg& cfg,
                                              const uint64_t* __restrict__ seeds, unsigned int seedCap,
                                              const uint64_t* __restrict__ candLong, unsigned int candCap,
                                              SegRec* __restrict__ seg, unsigned int segCap,
                                              uint32_t* __restrict__ chain, Counters* __restrict__ ctr,
                                              uint32_t* __restrict__ tile, const uint16_t* __restrict__ lut,
                                              unsigned int wave, unsigned int nWaves) {
    const int lane = threadIdx.x & 63;
    const unsigned int nSeeds = min(ctr->nSeeds, seedCap), nLong = min(ctr->nLong, candCap);
    const unsigned int nItems = min(nSeeds + nLong, segCap);
    const unsigned int base = CANDS ? nSeeds : 0u, nMine = CANDS ? nItems - min(nSeeds, nItems) : min(nSeeds, nItems);
    unsigned int* const head = CANDS ? &ctr->qDecide : &ctr->qWalk;
    const int H = cfg.H, W = cfg.W, WS = cfg.WS, S = cfg.nScales;
    const size_t perPlane = plane_words(H, WS);

    // Lane state, kept in vector registers as plain integers so that the steps compile to straight-line code with selects
    // (per-lane booleans that live across iterations end up as scalar mask pairs, and every `if` on them as a chain of
    // scalar mask instructions)
    //   st: 0 idle, 1 following a border, 2 segment complete, 3 segment complete and the lane goes on in a continuation
    int st = 0, isHole = 0;
    uint64_t c = 0, nextC = 0;
    uint32_t nextI = ~0u;
    bool exhausted = false;
    uint32_t id = 0, count = 0, total = 0, minAll = 0xFFFFFFFFu, minEast = 0xFFFFFFFFu, startPix = 0, state0 = 0;
    uint32_t endNext = kSegNone, endKey = 0;
    // chain codes: grp = the (up to kWalk2Steps) steps since the last service phase, 4 bits each from bit 0; acc = the steps
    // before them that have not filled a word yet, flushed = steps of this segment already written or in acc's words
    uint64_t grp = 0;
    uint32_t count0 = 0;
    uint64_t acc = 0;
    int startKey = 0, cx = 1, cy = 1, s = 0;
    const uint32_t* plane = planes;
    TileWin2 win;
    win.invalidate();
    bool fresh = false;  // a start candidate that has not taken its start state yet
    WaveQueue queue;
    const unsigned int qBatch = spread_batch(nMine, nWaves, kWalkBatch);
    queue.start(wave, nWaves, nMine, qBatch);

    for (;;) {
        // ================= service phase =================
        // ---- the steps of the last group join the segment's chain; a full word goes to memory ----
        {
            const uint32_t e = count - count0;          // steps of the group (0..kWalk2Steps)
            const uint32_t at = count0 & 7u;            // their place in the word being filled
            // (acc holds `at` codes; with the group's up to sixteen that is up to 92 bits: `over` takes what leaves acc)
            const uint64_t over = (kWalk2Steps > 8 && at) ? grp >> (64u - 4u * at) : 0ull;
            acc |= grp << (4u * at);
            const uint32_t full = (at + e) >> 3;        // words that are full now
            if (full >= 1u) {
                chain[(size_t)id * 32 + ((count0 >> 3) & 31u)] = (uint32_t)acc;
                if (kWalk2Steps > 8 && full >= 2u) {
                    chain[(size_t)id * 32 + (((count0 >> 3) + 1u) & 31u)] = (uint32_t)(acc >> 32);
                    acc = over;
                } else {
                    acc = (acc >> 32) | (over << 32);
                }
            }
            grp = 0;
            count0 = count;
        }
        // ---- finished segments: record, continuation ----
        if (st >= 2) {
            uint32_t cont = 0, next = endNext;
            bool goesOn = st == 3;
            if (goesOn) {
                const unsigned int v = atomicAdd(&ctr->nVirt, 1u);
                if (v >= segCap - nItems) {
                    atomicOr(&ctr->overflow, 512u);
                    next = kSegNone;
                    goesOn = false;
                } else {
                    cont = segCap - 1u - v;
                    next = cont;
                }
            }
            if ((count & 7u) != 0u) chain[(size_t)id * 32 + (((count - 1u) >> 3) & 31u)] = (uint32_t)acc;
            SegRec r;
            r.next = next; r.len = count; r.minAll = minAll; r.minEast = minEast; r.chunk = id; r.start = startPix;
            r.pad0 = endKey;
            r.pad1 = (uint32_t)(c >> 36);  // frame << 4 | hole << 3 | scale
            seg[id] = r;
            st = 0;
            if (goesOn) {
                id = cont;
                count = count0 = 0;
                acc = 0;
                minAll = minEast = 0xFFFFFFFFu;
                startPix = (uint32_t)cx | ((uint32_t)cy << 16);
                st = 1;
            }
        }
        // ---- idle lanes take up their next item ----
        if (st == 0 && nextI != ~0u) {
            c = nextC;
            id = nextI;
            nextI = ~0u;
            const uint32_t pos = (uint32_t)(c & 0xFFFFFFFFull);
            plane = planes + ((size_t)(uint32_t)(c >> 40) * S + (int)((c >> 36) & 7u)) * perPlane;
            cy = row_of(cfg, pos);
            cx = (int)pos - cy * W;
            isHole = CANDS ? (int)((c >> 39) & 1u) : 0;
            startKey = cy * W + cx;
            minAll = minEast = 0xFFFFFFFFu;
            if (!CANDS) {
                const int d = seed_dir(cx, cy);
                s = d < 0 ? 4 : d;                   // as if that crack had just been swept
                if (d == 0) minEast = (uint32_t)startKey;  // an east seed: the crack itself is an east crack of the border
            }
            count = count0 = 0;
            total = 0;
            acc = 0;
            startPix = (uint32_t)cx | ((uint32_t)cy << 16);
            win.invalidate();
            st = 1;
            fresh = CANDS;
        }
        // ---- lanes without a next item draw one: after the others have taken theirs up, so that this phase waits for no
        //      load it has issued itself (the item is taken up a phase later at the earliest) ----
        {
            const bool need = nextI == ~0u && !exhausted;
            const unsigned int i = queue.pop<1>(need, head, nMine, qBatch, lane);
            if (need) {
                if (i != ~0u) {
                    nextI = base + i;
                    nextC = CANDS ? candLong[i] : seeds[i];
                } else if (queue.empty()) {
                    exhausted = true;
                }
            }
        }
        if (!__any(st == 1)) {
            if (__all(exhausted && nextI == ~0u)) break;
            continue;
        }
        // ---- windows ----
        if (__any(st == 1 && win.near_edge(cx, cy))) {
            if (st == 1 && win.near_edge(cx, cy)) {
                win.recentre(plane, WS, H, cx, cy, tile, lane);
            }
        }
        // ---- a start candidate takes its start state (P0, s0) (s0 is never -1: single pixels end in the short pass) ----
        if (CANDS && __any(fresh)) {
            if (fresh) {
                fresh = false;
                uint32_t t, m, u;
                win.fields(tile, lane, cx, cy, t, m, u);
                const int s0 = fresh_dir(pack_mask(t, m, u), isHole != 0);
                s = s0 < 0 ? 0 : s0;
                state0 = pack_state(cx, cy, s);
            }
        }

        // ================= kWalk2Steps steps =================
#pragma unroll
        for (int k = 0; k < kWalk2Steps; k++) {
            // (after the service phase the pixel is at least three rows and twelve columns inside the window, so two
            // steps are safe; from the third on a lane may have run out of window, and waits for the next phase)
            const bool act = st == 1 && (k < 2 || win.covers(cx, cy));
            if (act) {
                uint32_t t, m, u;
                win.fields(tile, lane, cx, cy, t, m, u);
                const uint32_t L = lut[t | (m << 3) | (u << 6) | ((uint32_t)s << 9)];
                const uint32_t q = L & 7u, east = (L >> 3) & 1u;
                // the crack a seed can sit on here: rows 0 / 32 (mod 64) west / east, columns 0 / 32 (mod 64) north / south
                const uint32_t fr = (L >> (4u + 2u * (((uint32_t)cy >> 5) & 1u))) & 3u;
                const uint32_t fc = (L >> (8u + 2u * (((uint32_t)cx >> 5) & 1u))) & 3u;
                const uint32_t f = (cy & 31) == 0 ? fr : (cx & 31) == 0 ? fc : 0u;
                const uint32_t atSeed = f & 1u;
                const uint32_t key = (uint32_t)(cy * W + cx);
                uint32_t stop = atSeed;  // no step: the segment ends at this pixel
                if (CANDS) {
                    // a start candidate that meets a crack with a smaller key is not the canonical start
                    const uint32_t abortNow = ((uint32_t)(isHole ^ 1) | east) & ((int)key < startKey ? 1u : 0u) & (atSeed ^ 1u);
                    stop |= abortNow;
                    endNext = abortNow ? kSegAbort : endNext;
                }
                const bool emit = stop == 0u;
                // (the seed's own segment accounts for the seed pixel, but it starts its sweep behind the crack: an east
                // crack passed before the seed's crack in this same sweep is accounted for here)
                const uint32_t eastHere = emit ? east : (f >> 1);
                minEast = min(minEast, eastHere ? key : 0xFFFFFFFFu);
                minAll = min(minAll, emit ? key : 0xFFFFFFFFu);
                grp |= emit ? (uint64_t)q << (4 * k) : 0ull;
                count += emit ? 1u : 0u;
                cx += (int)((L >> 12) & 3u) - 1;   // (a lane that stops does not look at its position again)
                cy += (int)((L >> 14) & 3u) - 1;
                s = (int)(q ^ 4u);
                endNext = atSeed ? kSegPending : endNext;
                endKey = atSeed ? key : endKey;
                int nst = emit ? (count == (uint32_t)kSegLen ? 3 : 1) : 2;
                if (CANDS) {
                    total += emit ? 1u : 0u;
                    const bool closed = emit && pack_state(cx, cy, s) == state0;  // a border without seeds closed
                    const bool tooLong = emit && total > (uint32_t)cfg.maxPerim && !closed;
                    endNext = closed ? kSegSelf : tooLong ? kSegAbort : endNext;
                    nst = (closed || tooLong) ? 2 : nst;
                }
                st = nst;
            }
        }
    }
}

__global__ __launch_bounds__(64 * kWalk2Waves) void k_seg_walk2(const uint32_t* __restrict__ planes, DetCfg cfg,
                                                               const uint64_t* __restrict__ seeds, unsigned int seedCap,
                                                               const uint64_t* __restrict__ candLong, unsigned int candCap,
                                                               SegRec* __restrict__ seg, unsigned int segCap,
                                                               uint32_t* __restrict__ chain, Counters* __restrict__ ctr) {
    __shared__ __align__(8) uint32_t tiles[kWalk2Waves][kTile2Words];
    __shared__ uint16_t lut[4096];
    for (int i = threadIdx.x; i < 4096; i += 64 * kWalk2Waves) lut[i] = walk_lut_entry(i);
    __syncthreads();
    uint32_t* const tile = tiles[threadIdx.x >> 6];
    // Every fourth wave starts on the start candidates (a sixth of the items, but they come without a seed to stop at and
    // their tail is the longer one), all waves then take the seeds, and whoever is done helps with what is left of the
    // candidates.  The seeds' static batches go to the other waves first: with little work (a frame or a few) the waves
    // that start on the candidates then have no seeds of their own, and a frame's two longest chains run side by side.
    const unsigned int w = blockIdx.x * kWalk2Waves + (threadIdx.x >> 6), nW = gridDim.x * kWalk2Waves;
    const unsigned int nCandWaves = (nW + 3u) / 4u;
    const bool candFirst = (w & 3u) == 0u;
    if (candFirst) seg_walk_lean<true>(planes, cfg, seeds, seedCap, candLong, candCap, seg, segCap, chain, ctr, tile, lut, w >> 2, nCandWaves);
    const unsigned int nOther = nW - nCandWaves;
    const unsigned int seedWave = candFirst ? nOther + (w >> 2) : (w >> 2) * 3u + (w & 3u) - 1u;
    seg_walk_lean<false>(planes, cfg, seeds, seedCap, candLong, candCap, seg, segCap, chain, ctr, tile, lut, seedWave, nW);
    if (!candFirst) seg_walk_lean<true>(planes, cfg, seeds, seedCap, candLong, candCap, seg, segCap, chain, ctr, tile, lut, nCandWaves, nCandWaves);
}

// Turn "goes on at the seed at (plane, position)" into that seed's id: one lane per record.
__global__ __launch_bounds__(256) void k_seg_link(SegRec* __restrict__ seg, unsigned int segCap, unsigned int seedCap,
                                                  unsigned int candCap, DetCfg cfg, const uint64_t* __restrict__ seeds,
                                                  const uint32_t* __restrict__ seedTable, Counters* __restrict__ ctr) {
    const int S = cfg.nScales, H = cfg.H, W = cfg.W;
    const unsigned int nSeeds = min(ctr->nSeeds, seedCap), nLong = min(ctr->nLong, candCap);
    const unsigned int nItems = min(nSeeds + nLong, segCap);
    const unsigned int nVirt = min(ctr->nVirt, segCap - nItems);
    const size_t perPlane = seed_table_size(H, W);
    for (unsigned int t = blockIdx.x * 256u + threadIdx.x; t < nItems + nVirt; t += gridDim.x * 256u) {
        const unsigned int id = t < nItems ? t : segCap - 1u - (t - nItems);
        if (seg[id].next != kSegPending) continue;
        const uint32_t pos = seg[id].pad0, fs = seg[id].pad1;  // position; frame << 4 | hole << 3 | scale
        uint32_t v = seedTable[((size_t)(fs >> 4) * S + (fs & 7u)) * perPlane + seed_table_index(cfg, pos)];
        // the entry is this call's only if the seed list agrees (the table is never cleared)
        if (v >= nSeeds || seeds[v] != pack_cand(fs >> 4, 0u, fs & 7u, pos)) {
            v = kSegNone;
            atomicOr(&ctr->overflow, 512u);  // (a seed the start pass could not list)
        }
        seg[id].next = v;
    }
}

// Decide every undecided start candidate from the segment records and cut the accepted borders into copy tasks.
// A candidate's own records lead to the first seed of its border (or close on themselves); from there the seeds'
// records run once round the border.  The border is kept when no crack on it has a smaller raster key than the
// candidate (it is the canonical start) and its length is within the perimeter bounds; its points are then the
// candidate's own stretch followed by the ring from that first seed, cut off at the border's length.
// exclusive prefix sum over the wave's lanes, and the total
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t& total) {
    uint32_t s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)s, d);
        if (lane >= d) s += t;
    }
    total = (uint32_t)__shfl((int)s, 63);
    return s - v;
}

__global__ __launch_bounds__(64) void k_resolve(DetCfg cfg, const uint64_t* __restrict__ candLong, unsigned int candCap,
                                                unsigned int seedCap, const SegRec* __restrict__ seg, unsigned int segCap,
                                                ContourRec* __restrict__ contours, unsigned int contourCap, unsigned int poolCap,
                                                CopyTask* __restrict__ tasks, unsigned int taskCap, uint4* __restrict__ hopLog,
                                                Counters* __restrict__ ctr) {
    const int lane = threadIdx.x;
    // this wave's log of the records pass 1 visits: [hop][lane] = (chunk, start, len), so that pass 2 reads them back in
    // order with loads that do not depend on one another instead of chasing the pointers a second time
    uint4* const myLog = hopLog + (size_t)blockIdx.x * kResolveLogHops * 64 + lane;
    const unsigned int nSeeds = min(ctr->nSeeds, seedCap), nLong = min(ctr->nLong, candCap);
    const unsigned int nItems = min(nSeeds + nLong, segCap);
    const unsigned int nCand = nItems > nSeeds ? nItems - nSeeds : 0u;
    const unsigned int virtLo = segCap - min(ctr->nVirt, segCap - nItems);  // continuation ids are >= virtLo
    const unsigned int hopCap = (unsigned int)cfg.maxPerim + 4u * (unsigned int)kSegLen;  // no ring is longer (in segments) than this
    WaveQueue queue;
    const unsigned int qBatch = spread_batch(nCand, gridDim.x, kResolveBatch);
    queue.start(blockIdx.x, gridDim.x, nCand, qBatch);
    bool exhausted = false;

    auto valid_id = [&](uint32_t v) { return v < nItems || (v >= virtLo && v < segCap); };

    for (;;) {
        const unsigned int i = queue.pop<1>(!exhausted, &ctr->qResolve, nCand, qBatch, lane);
        const bool have = !exhausted && i != ~0u;
        if (!exhausted && i == ~0u && queue.empty()) exhausted = true;
        if (__all(exhausted)) break;

        // ---- pass 1: length of the border, smallest keys, number of records to copy from ----
        uint32_t len = 0, nrec = 0, firstSeed = kSegNone;
        uint64_t c = 0;
        bool ok = false, isHole = false;
        if (have) {
            c = candLong[i];
            isHole = (c >> 39) & 1u;
            const uint32_t startKey = (uint32_t)(c & 0xFFFFFFFFull);
            uint32_t minAll = 0xFFFFFFFFu, minEast = 0xFFFFFFFFu, own = 0;
            uint32_t id = nSeeds + i;
            ok = true;
            bool closed = false;
            // the candidate's own records
            for (unsigned int hop = 0;; hop++) {
                const SegRec r = seg[id];
                if (r.next == kSegAbort || r.next == kSegNone || r.len > (uint32_t)kSegLen || hop > hopCap) { ok = false; break; }
                own += r.len;
                if (nrec < kResolveLogHops) myLog[(size_t)nrec * 64] = make_uint4(r.chunk, r.start, r.len, 0u);
                nrec++;
                minAll = min(minAll, r.minAll);
                minEast = min(minEast, r.minEast);
                if (r.next == kSegSelf) { closed = true; break; }
                if (!valid_id(r.next)) { ok = false; break; }
                if (r.next < nSeeds) { firstSeed = r.next; break; }
                id = r.next;  // a continuation of the candidate's own stretch
            }
            if (ok && closed) {
                len = own;
            } else if (ok) {
                // once round the ring of seed records
                uint32_t ring = 0;
                id = firstSeed;
                for (unsigned int hop = 0;; hop++) {
                    const SegRec r = seg[id];
                    if (r.next == kSegNone || r.next == kSegAbort || r.next == kSegSelf || r.len > (uint32_t)kSegLen || !valid_id(r.next) ||
                        hop > hopCap) { ok = false; break; }
                    ring += r.len;
                    if (nrec < kResolveLogHops) myLog[(size_t)nrec * 64] = make_uint4(r.chunk, r.start, r.len, 0u);
                    nrec++;
                    minAll = min(minAll, r.minAll);
                    minEast = min(minEast, r.minEast);
                    if (ring > (uint32_t)cfg.maxPerim || (isHole ? minEast : minAll) < startKey) { ok = false; break; }
                    id = r.next;
                    if (id == firstSeed) break;
                }
                len = ring;
                if (ok && own > ring) ok = false;  // (cannot happen: the candidate lies on the ring)
            }
            if (ok && (isHole ? minEast : minAll) < startKey) ok = false;  // a crack with a smaller key: not the canonical start
            if (ok && (len < (uint32_t)cfg.minPerim || len > (uint32_t)cfg.maxPerim)) ok = false;
        }
        // ---- space for the batch's accepted borders: their sizes summed over the wave, ONE atomic per list and batch
        //      (the exact amount: nothing is reserved and left unused).  Reserving border by border from per-wave blocks
        //      cost ~80 k atomics per launch on three addresses, and those waited in line at the L2 for most of this
        //      kernel's time. ----
        unsigned int myOff = ~0u, mySlot = ~0u, myTask = ~0u;
        {
            const bool acc = have && ok;
            uint32_t totLen, totRec, totSl;
            const uint32_t oLen = wave_excl_scan(acc ? len : 0u, lane, totLen);
            const uint32_t oRec = wave_excl_scan(acc ? nrec : 0u, lane, totRec);
            const uint32_t oSl = wave_excl_scan(acc ? 1u : 0u, lane, totSl);
            if (totSl) {
                unsigned int bPool = 0, bSlot = 0, bTask = 0;
                if (lane == 0) {
                    bPool = atomicAdd(&ctr->nPool, totLen);
                    bSlot = atomicAdd(&ctr->nContours, totSl);
                    bTask = atomicAdd(&ctr->nCopy, totRec);
                }
                bPool = __builtin_amdgcn_readfirstlane(bPool);
                bSlot = __builtin_amdgcn_readfirstlane(bSlot);
                bTask = __builtin_amdgcn_readfirstlane(bTask);
                const bool poolFits = bPool <= poolCap && totLen <= poolCap - bPool;
                const bool slotFits = bSlot <= contourCap && totSl <= contourCap - bSlot;
                const bool taskFits = bTask <= taskCap && totRec <= taskCap - bTask;
                if (poolFits && slotFits && taskFits) {
                    if (acc) {
                        myOff = bPool + oLen;
                        mySlot = bSlot + oSl;
                        myTask = bTask + oRec;
                    }
                } else {
                    // a full list fails the call with CAMSYNC_E_CAPACITY; what was reserved below the capacities is
                    // nobody's now and must read as empty (the next passes look at every entry below the count)
                    if (lane == 0) atomicOr(&ctr->overflow, (poolFits ? 0u : 4u) | (slotFits ? 0u : 2u) | (taskFits ? 0u : 64u));
                    mark_empty_contours(contours, min(bSlot, contourCap), (unsigned int)min((unsigned long long)bSlot + totSl, (unsigned long long)contourCap), lane);
                    mark_empty_tasks(tasks, min(bTask, taskCap), (unsigned int)min((unsigned long long)bTask + totRec, (unsigned long long)taskCap), lane);
                }
            }
        }
        // ---- pass 2: the record and one copy task per segment, in border order ----
        if (have && ok && myOff != ~0u) {
            const uint32_t pos = (uint32_t)(c & 0xFFFFFFFFull);
            ContourRec r;
            r.frame = (uint32_t)(c >> 40);
            r.pos = pos;
            r.len = len;
            r.off = myOff;
            r.scale = (uint16_t)((c >> 36) & 7u);
            r.hole = isHole ? 1 : 0;
            r.scanKey = pos + (isHole ? 1u : 0u);
            contours[mySlot] = r;
            uint32_t emitted = 0, t = 0;
            if (nrec <= kResolveLogHops) {
                for (unsigned int hop = 0; hop < nrec; hop++) {
                    const uint4 e = myLog[(size_t)hop * 64];
                    const uint32_t n = min(e.z, len - emitted);
                    CopyTask k;
                    k.chunk = e.x;
                    k.start = e.y;
                    k.dst = myOff + emitted;
                    k.n = n;  // (n = 0 reads as an empty task)
                    tasks[myTask + t++] = k;
                    emitted += n;
                }
            } else {  // (more records than the log holds: once more along the pointers)
                uint32_t id = nSeeds + i;
                for (unsigned int hop = 0; hop < nrec; hop++) {
                    const SegRec sr = seg[id];
                    const uint32_t n = min(sr.len, len - emitted);
                    CopyTask k;
                    k.chunk = sr.chunk;
                    k.start = sr.start;
                    k.dst = myOff + emitted;
                    k.n = n;
                    tasks[myTask + t++] = k;
                    emitted += n;
                    id = sr.next;
                    if (id == kSegSelf) break;
                }
            }
            // (tasks reserved beyond the ones written cannot exist: pass 1 counted the same records)
        }
    }
}

// Write the points of the accepted borders: one lane per store task.
__global__ __launch_bounds__(64) void k_trace_store(const uint32_t* __restrict__ planes, DetCfg cfg,
                                                    const StoreTask* __restrict__ tasks, unsigned int taskCap,
                                                    uint32_t* __restrict__ pool, unsigned int poolCap,
                                                    Counters* __restrict__ ctr) {
    __shared__ __align__(8) uint32_t tile[kTileWords];
    const int lane = threadIdx.x;
    const unsigned int n = min(ctr->nTasks, taskCap);
    const int H = cfg.H, WS = cfg.WS;
    const size_t perPlane = plane_words(H, WS);

    bool active = false, exhausted = false;
    int left = 0, cx = 1, cy = 1, s = 0, nq = 0;
    uint32_t oidx = 0, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // pool index of the next point; points not yet written
    const uint32_t* plane = planes;
    TileWin win;
    win.invalidate();
    WaveQueue queue;
    queue.start(blockIdx.x, gridDim.x, n, kStoreBatch);

    for (;;) {
        {
            const bool need = !active && !exhausted;
            const unsigned int i = queue.pop<1>(need, &ctr->qStore, n, kStoreBatch, lane);
            if (need) {
                if (i != ~0u) {
                    const StoreTask k = tasks[i];
                    plane = planes + (size_t)k.plane * perPlane;
                    cx = (int)(k.state & 0xFFFu);
                    cy = (int)((k.state >> 12) & 0xFFFu);
                    s = (int)(k.state >> 24);
                    left = (int)k.count;
                    oidx = k.dst;
                    nq = 0;
                    win.invalidate();
                    active = left > 0 && left <= kCkStride && k.dst < poolCap && k.count <= poolCap - k.dst;  // (empty or not a task: skipped)
                } else if (queue.empty()) {
                    exhausted = true;
                }
            }
        }
        if (!__any(active)) {
            if (__all(exhausted)) break;
            continue;  // e.g. every lane drew an empty task: draw again
        }
        if (__any(active && !win.covers(cx, cy))) {
            if (active && win.near_edge(cx, cy)) win.recentre(plane, WS, H, cx, cy, tile, lane);
        }
        if (active) {
            const uint32_t m = win.mask(tile, lane, cx, cy);
            if (__any(s >= 8)) {  // fresh start: initial search (s - 8 = hole flag); rare, so a uniform branch
                const int s0 = fresh_dir(m, s == 9);
                if (s >= 8) s = s0 < 0 ? 0 : s0;  // a single pixel has count 1 and stops below before s is used
            }
            // the last eight points wait in registers: two 16-byte stores per aligned group of eight instead of
            // eight scattered 4-byte stores (the head and the tail of a task are written point by point)
#pragma unroll
            for (int k = 0; k < 7; k++) q[k] = q[k + 1];
            q[7] = (uint32_t)cx | ((uint32_t)cy << 16);
            nq++;
            --left;
            if ((oidx & 7u) == 7u || left == 0) {
                if (nq >= 8 && (oidx & 7u) == 7u) {
                    uint4* dst = reinterpret_cast<uint4*>(pool + (oidx - 7u));
                    dst[0] = make_uint4(q[0], q[1], q[2], q[3]);
                    dst[1] = make_uint4(q[4], q[5], q[6], q[7]);
                } else {
                    const int m = nq < 8 ? nq : 8;  // only a task's first or last group gets here
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (m > k) pool[oidx - (uint32_t)k] = q[7 - k];
                }
                nq = 0;
            }
            oidx++;
            if (left == 0) {
                active = false;
            } else {
                const uint32_t rot = ((m >> (s + 1)) | (m << (7 - s))) & 0xFFu;
                const int d = (s + __ffs(rot)) & 7;
                cx += dir_dx(d);
                cy += dir_dy(d);
                s = d ^ 4;
            }
        }
    }
}

// Write the points of the kept long borders: a half-wave per copy task.  Lane l of the half takes steps 8 l ..
// 8 l + 7 of the piece's chain chunk; the pixel before step i is the piece's first pixel plus the sum of the
// directions of steps 0 .. i - 1 (a prefix sum over the half-wave), and goes to dst + i or dst - i.
// Inclusive prefix sum over each half of the wave (32 lanes) with DPP row shifts: no trip through LDS.
__device__ __forceinline__ uint32_t half_wave_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);  // rows 1 and 3 take the row before
    return v;
}

// The points of one copy task of at most 128 steps, a row of 32 consecutive steps at a time, one step per lane of the
// half-wave: a prefix sum across the lanes places the row's points, and a store instruction writes 32 consecutive points.
__device__ __forceinline__ void copy_short_task(const uint32_t* __restrict__ chain, const CopyTask& k, unsigned int cnt, bool good,
                                                int l32, uint32_t* __restrict__ pool) {
    uint32_t w[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const unsigned int j = (unsigned int)(32 * m + l32);
        w[m] = (good && j < cnt) ? chain[(size_t)k.chunk * 32 + (j >> 3)] : 0u;
    }
    int cx = (int)(k.start & 0xFFFFu), cy = (int)(k.start >> 16);  // position of the row's first point
#pragma unroll
    for (int m = 0; m < 4; m++) {
        if ((unsigned int)(32 * m) >= cnt) break;  // (the same for all lanes of the half-wave)
        const unsigned int j = (unsigned int)(32 * m + l32);
        const int q = (int)((w[m] >> (4 * (j & 7u))) & 7u);
        const int dx = dir_dx(q), dy = dir_dy(q);
        // both sums in one word: every lane's step is biased by 1 so that the fields never go negative
        const uint32_t inc = half_wave_scan((uint32_t)(dx + 1) | ((uint32_t)(dy + 1) << 16));
        const int sx = (int)(inc & 0xFFFFu) - (l32 + 1), sy = (int)(inc >> 16) - (l32 + 1);  // steps 32 m .. j summed
        if (good && j < cnt) pool[k.dst + j] = (uint32_t)(cx + sx - dx) | ((uint32_t)(cy + sy - dy) << 16);
        cx += __shfl(sx, 31, 32);
        cy += __shfl(sy, 31, 32);
    }
}

__global__ __launch_bounds__(256) void k_trace_copy(const uint32_t* __restrict__ chain, const CopyTask* __restrict__ copies,
                                                    unsigned int copyCap, unsigned int chainCap, uint32_t* __restrict__ pool,
                                                    unsigned int poolCap, Counters* __restrict__ ctr) {
    __shared__ uint32_t s_pts[8][256];  // per half-wave
    const unsigned int n = min(ctr->nCopy, copyCap);
    const int lane = threadIdx.x & 63, l32 = lane & 31;
    const unsigned int wave = blockIdx.x * 4u + (threadIdx.x >> 6), nWaves = gridDim.x * 4u;
    for (unsigned int pair = wave; 2u * pair < n; pair += nWaves) {
        const unsigned int t = 2u * pair + (unsigned int)(lane >> 5);
        CopyTask k;
        k.chunk = 0; k.start = 0; k.dst = 0; k.n = 0;
        if (t < n) k = copies[t];
        const unsigned int cnt = k.n & 0x7FFFFFFFu;
        const bool rev = (k.n >> 31) != 0;
        // (empty entries have n = 0; anything else out of range is not a task)
        const bool good = cnt <= (unsigned int)kCkStride && k.chunk < chainCap && k.dst < poolCap &&
                          (rev ? cnt <= k.dst + 1u : cnt <= poolCap - k.dst);
        // Most segments end at the next seed, a few dozen steps on: one to four rows of 32 steps cover the task, with
        // a fraction of the work of the general case (eight steps per lane, points transposed through LDS).
        if (!rev && cnt <= 128u) {
            copy_short_task(chain, k, cnt, good, l32, pool);
            continue;
        }
        const uint32_t d = (good && (unsigned int)(8 * l32) < cnt) ? chain[(size_t)k.chunk * 32 + l32] : 0u;
        int ox[8], oy[8], x = 0, y = 0;
#pragma unroll
        for (int m = 0; m < 8; m++) {
            ox[m] = x;
            oy[m] = y;
            const int q = (int)((d >> (4 * m)) & 7u);
            x += dir_dx(q);
            y += dir_dy(q);
        }
        // inclusive prefix sums of the lanes' totals within the half-wave (both in one word, biased by 8 per lane)
        const uint32_t inc = half_wave_scan((uint32_t)(x + 8) | ((uint32_t)(y + 8) << 16));
        const int sx = (int)(inc & 0xFFFFu) - 8 * (l32 + 1), sy = (int)(inc >> 16) - 8 * (l32 + 1);
        const int bx = (int)(k.start & 0xFFFFu) + sx - x, by = (int)(k.start >> 16) + sy - y;
        // through LDS, so that a store instruction writes 32 consecutive points instead of every eighth
        uint32_t* const mine = s_pts[threadIdx.x >> 5];
#pragma unroll
        for (int m = 0; m < 8; m++) mine[(8 * l32 + m) ^ l32] = (uint32_t)(bx + ox[m]) | ((uint32_t)(by + oy[m]) << 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const unsigned int i = (unsigned int)(32 * m + l32);
            const uint32_t v = mine[i ^ (i >> 3)];
            if (good && i < cnt) pool[rev ? k.dst - i : k.dst + i] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}


}  // namespace

void launch_contour_starts(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    // one wavefront per (plane, group of 62 block columns, band of rows); a plane is cut into bands (each re-reads one
    // row; whole groups of eight rows) when there are too few planes to fill the chip, and into three for large batches
    const int colGroups = (c.WS + kStartCols - 1) / kStartCols;
    const long long waves = (long long)B * c.nScales * colGroups;
    const int totalGroups = plane_hb(c.H) * 4;
    int bands = 1;
    // (about 32 wavefronts per CU: the chip holds ~26 of them per CU, and one round that does not fill it runs at the
    // latency of its loads -- 256 4K frames: 0.66 ms with 16 per CU, 0.58 ms with 32)
    if (waves < 256 * 32) bands = (int)std::min<long long>((256 * 32 + waves - 1) / waves, (totalGroups + 3) / 4);
    // a wavefront that takes a whole plane runs long, and the chip holds ~26 of them per CU: with one or two rounds of
    // such wavefronts the last ones run on a half-empty chip (12288 planes: 2.08 ms in one band, 1.70 ms in three; more
    // bands than needed cost an atomic per list and wavefront on two shared counters: 2.6 ms with eighteen)
    if (waves >= 256 * 8) bands = std::max(bands, 3);
    const int bandGroups = (totalGroups + bands - 1) / bands;
    bands = (totalGroups + bandGroups - 1) / bandGroups;
    const long long grid = waves * bands;
    hipLaunchKernelGGL(k_contour_starts, dim3((unsigned)grid), dim3(64), 0, st, b.planes, c, B, bands, bandGroups, b.cand, b.candCap,
                       b.seeds, b.seedCap, b.seedTable, b.counters);
}

void launch_trace(const DetCfg& c, const DetBuffers& b, hipStream_t st) {
    hipLaunchKernelGGL(k_trace_short, dim3(256 * 8), dim3(256), 0, st, b.planes, c, b.cand, b.candCap, b.candLong,
                       b.contours, b.contourCap, b.poolCap, b.tasks, b.taskCap, b.counters);
    debug_stage(st, "trace_short");
    // long borders: segments (persistent: single-wave workgroups, 8 KiB of LDS each), links, decision
    hipLaunchKernelGGL(k_seg_walk2, dim3((kTraceWaves + kWalk2Waves - 1) / kWalk2Waves), dim3(64 * kWalk2Waves), 0, st, b.planes, c, b.seeds, b.seedCap,
                       b.candLong, b.candCap, b.seg, b.segCap, b.chain, b.counters);
    debug_stage(st, "seg_walk");
    hipLaunchKernelGGL(k_seg_link, dim3(256 * 8), dim3(256), 0, st, b.seg, b.segCap, b.seedCap, b.candCap, c,
                       b.seeds, b.seedTable, b.counters);
    debug_stage(st, "seg_link");
    hipLaunchKernelGGL(k_resolve, dim3(b.resolveWaves), dim3(64), 0, st, c, b.candLong, b.candCap, b.seedCap, b.seg, b.segCap, b.contours,
                       b.contourCap, b.poolCap, b.copies, b.copyCap, b.resolveLog, b.counters);
    debug_stage(st, "resolve");
    // points of the borders the short pass kept (followed once more from their start), then of the long ones
    hipLaunchKernelGGL(k_trace_store, dim3(256 * 4), dim3(64), 0, st, b.planes, c, b.tasks, b.taskCap, b.pool, b.poolCap, b.counters);
    debug_stage(st, "trace_store");
    hipLaunchKernelGGL(k_trace_copy, dim3(256 * 8), dim3(256), 0, st, b.chain, b.copies, b.copyCap, b.chainCap, b.pool, b.poolCap, b.counters);
}

size_t seed_table_entries(int H, int W) { return seed_table_size(H, W); }

}  // namespace camsync
