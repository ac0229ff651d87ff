// k_threshold.hip — K1+K2: BGR->gray and the multi-window adaptive threshold
// of ArucoDetector.detectMarkers, fused in one pass (reference call sites:
// camera_sync/aruco.py:246 cv.cvtColor, :256-264 detectMarkers step 1a).
//
// Layout: one wavefront (= one workgroup, so no workgroup barriers) walks a
// 512-column strip of one frame top to bottom; 16 columns on each side are
// halo, 480 produce output (1920 = 4 strips).  Each lane owns 8 adjacent
// columns.  Per row the lane reads its 24 BGR bytes once (coalesced), converts
// to luma with two byte dot products, keeps the last 2*rmax+2 gray rows (the
// newest rmax+4 in registers, the older rmax+1 in an LDS ring) and maintains
// the vertical running sums of the three windows in registers as packed 16-bit
// pairs.  The vertical sums are exchanged through LDS and slid horizontally, so
// every input byte is fetched from HBM once (plus the strip halos) and the
// three threshold images leave as bit planes (one bit per pixel and scale).
// The gray image itself is written only when a caller asks for it
// (camsync_gray_threshold, detector parameters the fused kernel does not
// cover): the pipeline's later stages convert the few gray values they need
// from the input frame (internal.h, GraySrc).
//
// What bounds it: vector instruction issue.  While it still wrote the gray image it was bound by memory, and within
// memory by the writes (round 1, profiles/r01_notes.md); without the gray image it moves 3.08 B/px in and 0.38 B/px out
// (counters, profiles/r03_pmc_k_gray_thresh3.json: 3.3 TB/s, a memory floor of ~7 ms for 4096 1080p frames), and the
// ~243 vector instructions per row of 8 pixels a lane (37 gray conversion, 48 vertical sums, 69 horizontal sums,
// 56 thresholds, the rest staging and addressing) keep the vector unit busy in every cycle of its 8.9 ms at three
// waves per SIMD (146 registers, 12 KB of LDS per wave).  The strips of one band are placed on one XCD (the halo
// columns two strips both read meet in one L2) and the plane bits leave in 64-byte pieces.
//
// Integer-exact: luma = (3735 B + 19235 G + 9798 R + 2^14) >> 15;
// out_s = 1  <=>  round(sum/k^2) >= src + idelta  <=>  2 sum >= (2(src+idelta)-1) k^2
//            <=>  sum >= k^2 src + ((2 idelta - 1) k^2 + 1) / 2
// (k^2 is odd, so the rounded mean never ties and the last division is exact).
#include "internal.h"

#include <type_traits>

namespace camsync {

namespace {

constexpr int PX = 8;
constexpr int TPB = 64;
constexpr int ROWPX = PX * TPB;  // 512
constexpr int HALO = 16;         // >= max radius, multiple of PX
constexpr int CHUNKW = ROWPX - 2 * HALO;

// The workgroup is one wavefront and a wavefront's LDS operations execute in order, so exchanging data
// between lanes through LDS needs no s_barrier and, above all, no wait for the global loads in flight
// (which __syncthreads() would add, serialising the row prefetch): only the compiler must keep the order.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t luma(uint32_t b, uint32_t g, uint32_t r) {
    return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15;
}

// luma of the pixel whose B,G,R are bytes 0,1,2 of p, left in byte 2 of the result.  The weights are doubled, so
// that the rounding shift is 16 bits and the gray value a whole byte that a byte permute can pick up:
// (2 (3735 B + 19235 G + 9798 R) + 2^15) >> 16; they split as 2 w = 256 wh + wl (7470 = 29*256+46,
// 38470 = 150*256+70, 19596 = 76*256+140), each half a byte dot product, the second one on top of the first.
__device__ __forceinline__ uint32_t luma_dot_b2(uint32_t p) {
    const uint32_t hi = __builtin_amdgcn_udot4(p, 0x004C961Du, 0u, false);
    return __builtin_amdgcn_udot4(p, 0x008C462Eu, (hi << 8) + 32768u, false);
}

// The 24 (C == 3) or 8 (C == 1) input bytes of columns xbase..xbase+7 of one row, as loaded; converting them
// to gray is a separate step so that several rows of loads can be in flight per wave (a wave with one row in
// flight caps the whole chip at ~2.7 TB/s: 14 waves x 1.5 KB per CU per memory latency).
struct RawRow { uint2 a, b, c; };

__device__ __forceinline__ RawRow issue_row(const uint8_t* __restrict__ in, int C, int W, size_t rowOff, int xbase) {
    // W is a multiple of 8 and so is xbase: a group is inside the image or entirely outside, where
    // BORDER_REPLICATE makes all 8 values the edge pixel of the nearest inside group
    const int xc = min(max(xbase, 0), W - 8);
    RawRow r;
    if (C == 3) {
        const uint2* p = reinterpret_cast<const uint2*>(in + (rowOff + (size_t)xc) * 3);
        r.a = p[0]; r.b = p[1]; r.c = p[2];
    } else {
        r.a = *reinterpret_cast<const uint2*>(in + rowOff + (size_t)xc);
        r.b = r.c = make_uint2(0, 0);
    }
    return r;
}

// issue_row for 3 channels with the lane's byte offset in the row (3 x its clamped first column) worked out once:
// the row is the same for every lane, so the loads take a scalar base and that 32-bit lane offset, and the
// 64-bit address arithmetic per lane and row (a dozen vector instructions in a kernel bound by them) goes away.
// Written so that the compiler keeps that form: the base goes through an empty asm as a pointer to global memory
// (else it re-associates base + offset and hoists the lane part as a 64-bit lane address), and so does the lane
// offset (widened to 64 bits outside the row loop it no longer looks like a 32-bit offset).
__device__ __forceinline__ RawRow issue_row3(const uint8_t* __restrict__ in, size_t rowByte_, uint32_t laneByte) {
    using GBytes = const __attribute__((address_space(1))) uint8_t*;
    using GWords = const __attribute__((address_space(1))) uint64_t*;
    const uint64_t rowByte = rowByte_;  // byte offset of the row in the input (3 x its pixel offset)
    // (the builtin returns int: without the casts the low half would be sign-extended over the high one)
    const uint32_t rowLo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)rowByte);
    const uint32_t rowHi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rowByte >> 32));
    GBytes base = (GBytes)in + (((uint64_t)rowHi << 32) | (uint64_t)rowLo);
    asm("" : "+s"(base));
    uint32_t o = laneByte;
    asm volatile("" : "+v"(o));
    GWords p = (GWords)(base + o);
    const uint64_t pa = p[0], pb = p[1], pc = p[2];
    RawRow r;
    r.a = make_uint2((uint32_t)pa, (uint32_t)(pa >> 32));
    r.b = make_uint2((uint32_t)pb, (uint32_t)(pb >> 32));
    r.c = make_uint2((uint32_t)pc, (uint32_t)(pc >> 32));
    return r;
}

__device__ __forceinline__ uint2 gray_of(const RawRow& q, int C, int W, int xbase) {
    uint2 r;
    if (C == 3) {
        const uint2 a = q.a, b = q.b, c = q.c;
        uint32_t g[8];
        // pixel i starts at byte 3 i of the 24 bytes a.x a.y b.x b.y c.x c.y
        g[0] = luma_dot_b2(a.x);
        g[1] = luma_dot_b2(__builtin_amdgcn_alignbyte(a.y, a.x, 3));
        g[2] = luma_dot_b2(__builtin_amdgcn_alignbyte(b.x, a.y, 2));
        g[3] = luma_dot_b2(__builtin_amdgcn_alignbyte(b.y, b.x, 1));
        g[4] = luma_dot_b2(b.y);
        g[5] = luma_dot_b2(__builtin_amdgcn_alignbyte(c.x, b.y, 3));
        g[6] = luma_dot_b2(__builtin_amdgcn_alignbyte(c.y, c.x, 2));
        g[7] = luma_dot_b2(c.y >> 8);
        // byte 2 of four results -> one word
        r.x = __builtin_amdgcn_perm(g[1], g[0], 0x0C0C0602u) | __builtin_amdgcn_perm(g[3], g[2], 0x06020C0Cu);
        r.y = __builtin_amdgcn_perm(g[5], g[4], 0x0C0C0602u) | __builtin_amdgcn_perm(g[7], g[6], 0x06020C0Cu);
    } else {
        r = q.a;
    }
    if (xbase < 0) r.x = r.y = (r.x & 255u) * 0x01010101u;
    else if (xbase >= W) r.x = r.y = (r.y >> 24) * 0x01010101u;
    return r;
}

// Load the 8 gray values of columns xbase..xbase+7 (clamped to the image) of
// row `row` (already clamped) of frame f, converting from BGR when C == 3.
__device__ __forceinline__ uint2 load_gray8(const uint8_t* __restrict__ in, int C, int W, size_t rowOff,
                                            int xbase, bool aligned) {
    if (aligned) return gray_of(issue_row(in, C, W, rowOff, xbase), C, W, xbase);
    uint32_t g[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int x = min(max(xbase + i, 0), W - 1);
        if (C == 3) {
            const uint8_t* p = in + (rowOff + (size_t)x) * 3;
            g[i] = luma(p[0], p[1], p[2]);
        } else {
            g[i] = in[rowOff + (size_t)x];
        }
    }
    uint2 r;
    r.x = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
    r.y = g[4] | (g[5] << 8) | (g[6] << 16) | (g[7] << 24);
    return r;
}

__device__ __forceinline__ void unpack8(uint2 v, uint32_t o[8]) {
    o[0] = v.x & 255u; o[1] = (v.x >> 8) & 255u; o[2] = (v.x >> 16) & 255u; o[3] = v.x >> 24;
    o[4] = v.y & 255u; o[5] = (v.y >> 8) & 255u; o[6] = (v.y >> 16) & 255u; o[7] = v.y >> 24;
}

// 8 bytes -> 4 words of two 16-bit fields (pixel 2k in the low half of word k)
__device__ __forceinline__ void widen8(uint2 v, uint32_t o[4]) {
    o[0] = __builtin_amdgcn_perm(0u, v.x, 0x0C010C00u);
    o[1] = __builtin_amdgcn_perm(0u, v.x, 0x0C030C02u);
    o[2] = __builtin_amdgcn_perm(0u, v.y, 0x0C010C00u);
    o[3] = __builtin_amdgcn_perm(0u, v.y, 0x0C030C02u);
}

// a + / - one 16-bit half of w: the sub-dword operand selects of the SDWA encoding
// take the field straight out of the packed word, so the sums never have to be unpacked (that unpacking was a
// quarter of this kernel's vector instructions).
#define CAMSYNC_SDWA2(name, op, sel)                                                                              \
    __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t w) {                                           \
        uint32_t d;                                                                                               \
        asm(op " %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" sel                  \
            : "=v"(d) : "v"(a), "v"(w));                                                                          \
        return d;                                                                                                 \
    }
CAMSYNC_SDWA2(sub_lo16, "v_sub_u32_sdwa", "WORD_0")
CAMSYNC_SDWA2(sub_hi16, "v_sub_u32_sdwa", "WORD_1")
#undef CAMSYNC_SDWA2

// a - (sum of values LO - R .. LO + R): the whole words are added as packed pairs first (a field holds at most
// (2R+1) 255, so the pair sums of up to 11 words stay below 2^16), then the two halves of that sum and the
// single values at either end come off a.
template <int LO, int R>
__device__ __forceinline__ uint32_t win_init(uint32_t a, const uint32_t* w) {
    constexpr int F0 = LO - R, F1 = LO + R;                 // first and last value
    constexpr int W0 = (F0 + 1) >> 1, W1 = (F1 - 1) >> 1;  // whole words
    static_assert(F0 >= 0 && W1 >= W0, "window must cover at least one whole word");
    static_assert((W1 - W0 + 1) * (2 * R + 1) * 255 < 65536, "packed partial sums must fit 16 bits");
    uint32_t p = w[W0];
#pragma unroll
    for (int i = W0 + 1; i <= W1; i++) p += w[i];
    uint32_t d;
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
        "v_sub_u32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=&v"(d) : "v"(a), "v"(p));
    if (F0 & 1) d = sub_hi16(d, w[F0 >> 1]);
    if (!(F1 & 1)) d = sub_lo16(d, w[F1 >> 1]);
    return d;
}

// out[1 .. 7] from out[0]: out[I] = out[I - 1] - value(LO + I + R) + value(LO + I - 1 - R).  One asm block: the
// compiler cannot see into asm statements and separates dependent ones by a wait state that these (full-dword
// destinations) do not need; that was 60 s_nop per row.  LO is even, so the halves alternate with I and start
// according to the parity of R.
#define CAMSYNC_SLIDE(o, p, n, q, sn, so)                                                                        \
    "v_sub_u32_sdwa %" #o ", %" #p ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" sn "\n\t" \
    "v_add_u32_sdwa %" #o ", %" #o ", %" #q " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" so "\n\t"
#define CAMSYNC_SLIDE7(sa, sb)                                                                                   \
    CAMSYNC_SLIDE(0, 7, 8, 15, sa, sb) CAMSYNC_SLIDE(1, 0, 9, 16, sb, sa) CAMSYNC_SLIDE(2, 1, 10, 17, sa, sb)      \
    CAMSYNC_SLIDE(3, 2, 11, 18, sb, sa) CAMSYNC_SLIDE(4, 3, 12, 19, sa, sb) CAMSYNC_SLIDE(5, 4, 13, 20, sb, sa)    \
    CAMSYNC_SLIDE(6, 5, 14, 21, sa, sb)
template <int LO, int R>
__device__ __forceinline__ void win_slide7(const uint32_t* w, uint32_t* out) {
    static_assert((LO & 1) == 0, "LO must be even");
#define CAMSYNC_NW(I) "v"(w[(LO + (I) + R) >> 1])
#define CAMSYNC_OW(I) "v"(w[(LO + (I) - 1 - R) >> 1])
#define CAMSYNC_SLIDE_OPS                                                                                         \
    : "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]), "=&v"(out[4]), "=&v"(out[5]), "=&v"(out[6]), "=&v"(out[7])    \
    : "v"(out[0]), CAMSYNC_NW(1), CAMSYNC_NW(2), CAMSYNC_NW(3), CAMSYNC_NW(4), CAMSYNC_NW(5), CAMSYNC_NW(6),      \
      CAMSYNC_NW(7), CAMSYNC_OW(1), CAMSYNC_OW(2), CAMSYNC_OW(3), CAMSYNC_OW(4), CAMSYNC_OW(5), CAMSYNC_OW(6),    \
      CAMSYNC_OW(7)
    if (R & 1)  // I odd: the new value is an even one (low half), the old one odd
        asm(CAMSYNC_SLIDE7("WORD_0", "WORD_1") CAMSYNC_SLIDE_OPS);
    else
        asm(CAMSYNC_SLIDE7("WORD_1", "WORD_0") CAMSYNC_SLIDE_OPS);
#undef CAMSYNC_SLIDE_OPS
#undef CAMSYNC_OW
#undef CAMSYNC_NW
}
#undef CAMSYNC_SLIDE7
#undef CAMSYNC_SLIDE

// `init` MINUS the sliding horizontal sum of radius R over the exchanged vertical sums (the caller folds the
// threshold's constant into init; negated, so that the comparison below is one multiply-add).
// The exchange buffer of one scale is laid out by word: word j (two sums) of every lane is contiguous,
// xb[j * XROW + XPAD + lane], so that whichever word offset a lane reads from its neighbours, the 64 lanes
// of the instruction touch 64 consecutive words (no bank conflicts; a 16-byte lane stride read 4 bytes at a
// time hits every bank four times, and that made this kernel LDS-bound).
constexpr int XPAD = 0;             // lanes of slack on either side of a word row (producing lanes stay 2 lanes inside)
constexpr int XROW = TPB + 2 * XPAD;
template <int R>
__device__ __forceinline__ void hsum8(const uint32_t* xb, int lane, uint32_t init, uint32_t out[8]) {
    // window of pixel i covers values [i-R, i+R] relative to the lane's first; fetch words [-LO/2, 4+LO/2)
    constexpr int LO = (R + 1) & ~1;  // even offset >= R
    constexpr int N = 8 + 2 * LO;     // values fetched
    uint32_t w[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
        const int d = i - LO / 2;  // word offset from the lane's first word: lane + (d >> 2), word d & 3
        w[i] = xb[(d & 3) * XROW + XPAD + lane + (d >> 2)];
    }
    out[0] = win_init<LO, R>(init, w);
    win_slide7<LO, R>(w, out);
}

// bit i of the result = (K c[i] + t[i] < 0), t = (the threshold's constant - 1) - window sum: the >= of the
// definition.  A funnel shift appends the sign bit to the byte under construction: multiply-add and shift per
// pixel, no condition codes (a compare feeding a select costs wait states on this chip).  c = the 8 centre
// values, k = K.
__device__ __forceinline__ uint32_t thresh_bits(const uint32_t t[8], const uint32_t c[8], uint32_t k) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) acc = __builtin_amdgcn_alignbit(acc, __umul24(c[i], k) + t[i], 31);
    return acc;
}

// MODE 1 / 2: the common cases compiled with their parameters fixed (3 / 1 input channels, width a multiple of
// 8, no packed threshold image asked for), so that the row loop is straight-line code whose loads the compiler
// can track across rows; MODE 0 takes them at run time.
template <int R0, int R1, int R2, int MODE>
__global__ __launch_bounds__(TPB) void k_gray_thresh3(const uint8_t* __restrict__ in, int C_, uint8_t* __restrict__ gray,
                                                      uint8_t* __restrict__ bin_, uint32_t* __restrict__ planes, int WS, int H, int W, int bandH, int idelta, int nFrames) {
    const int C = MODE == 1 ? 3 : MODE == 2 ? 1 : C_;
    uint8_t* const bin = MODE ? nullptr : bin_;
    const bool hasGray = gray != nullptr;
    constexpr int RMAX = R2;
    constexpr int RING = RMAX + 1;  // rows y-1 .. y-rmax-1; rows y .. y+rmax sit in the register delay line
    static_assert(R0 <= R1 && R1 <= R2 && R2 <= HALO - 1, "radii must be sorted and fit the halo");
    __shared__ __align__(16) uint8_t ring[RING][ROWPX];
    __shared__ uint32_t xbuf[3][4 * XROW];  // vertical sums of one row, by word (hsum8)
    // STG rows of plane bits of this strip, laid out like the planes (scale, block, row): flushed as 16-byte pieces
    constexpr int NBLK = CHUNKW / 32;
    // rows staged per flush: 16 rows of a block are 64 bytes, the unit memory is written in (8 rows: 32-byte pieces
    // that the L2 has evicted before the block's next piece arrives, +0.9 ms; 32 rows: too much LDS)
    constexpr int STG = 16;
    __shared__ __align__(16) uint32_t stage[3][NBLK][STG];

    const int tid = threadIdx.x;
    // 1-D grid; workgroups 8 apart share an XCD (its L2), and the strips of one band of one frame are made such
    // workgroups: the cache lines of the gray rows that two neighbouring strips both write parts of meet in one L2,
    // and so do the halo columns both read.  Workgroup = (group of 8 units, strip, unit in the group), unit = band
    // of a frame.
    const int nStrips = (W + CHUNKW - 1) / CHUNKW, nBands = (H + bandH - 1) / bandH;
    const int grp = blockIdx.x / (8 * nStrips), rem = blockIdx.x - grp * (8 * nStrips);
    const int unit = grp * 8 + (rem & 7);
    if (unit >= nBands * nFrames) return;
    const int f = unit / nBands;
    const int y0 = (unit - f * nBands) * bandH;
    const int stripIdx = rem >> 3;
    const int y1 = min(H, y0 + bandH);
    const int cx0 = stripIdx * CHUNKW;
    const int xbase = cx0 - HALO + PX * tid;
    const bool aligned = MODE ? true : (W % 8) == 0;
    const size_t frameOff = (size_t)f * H * W;
    const bool inOut = (tid >= HALO / PX) && (tid < TPB - HALO / PX) && xbase < W;  // produces output
    // (MODE 1 / 2: the width is a multiple of 8, so a group that starts inside the image lies inside it: no per-byte path)
    const bool fullOut = MODE ? inOut : inOut && (xbase + 7 < W) && aligned;
    // MODE 1: byte offset of the lane's (clamped) group in a row, see issue_row3
    const uint32_t laneByte = (uint32_t)(min(max(xbase, 0), W - 8) * 3);
    auto issue = [&](size_t rowOff) { return MODE == 1 ? issue_row3(in, rowOff * 3, laneByte) : issue_row(in, C, W, rowOff, aligned ? xbase : 0); };
    constexpr int K0 = (2 * R0 + 1) * (2 * R0 + 1), K1 = (2 * R1 + 1) * (2 * R1 + 1), K2 = (2 * R2 + 1) * (2 * R2 + 1);
    const int c0 = ((2 * idelta - 1) * K0 + 1) >> 1, c1 = ((2 * idelta - 1) * K1 + 1) >> 1, c2 = ((2 * idelta - 1) * K2 + 1) >> 1;

    // vertical running sums, two 16-bit fields per word (at most 23 * 255 per field)
    uint32_t V0[4] = {0, 0, 0, 0}, V1[4] = {0, 0, 0, 0}, V2[4] = {0, 0, 0, 0};

    for (int i = tid; i < 3 * NBLK * STG; i += TPB) (&stage[0][0][0])[i] = 0u;
    const size_t perPlane = plane_words(H, WS);
    uint32_t* const fplanes = planes + (size_t)f * 3 * perPlane;
    const int bx0 = cx0 >> 5;  // first plane block of this strip (CHUNKW is a multiple of 32)
    // write rows [STG g, STG g + STG) of the strip's blocks to the three planes
    auto flush = [&](int g) {
        wave_lds_sync();
        constexpr int Q = STG / 4;  // 16-byte pieces per block and scale
        // (straight-line: a loop with stores in it makes the compiler wait for every load in flight at the top
        // of the row loop)
#pragma unroll
        for (int it = 0; it < (3 * NBLK * Q + TPB - 1) / TPB; it++) {
            const int i = tid + it * TPB;
            const int sc = i / (NBLK * Q), rem = i - sc * (NBLK * Q), blk = rem / Q, q = rem % Q;
            if (i < 3 * NBLK * Q && bx0 + blk < WS) {
                const uint4 v = *reinterpret_cast<const uint4*>(&stage[sc][blk][q * 4]);
                const int row0 = g * STG;  // first row of the group
                uint32_t* dst = fplanes + (size_t)sc * perPlane + ((size_t)(row0 >> 5) * WS + bx0 + blk) * 32 + (row0 & 31) + q * 4;
                *reinterpret_cast<uint4*>(dst) = v;
            }
        }
        wave_lds_sync();
    };
    uint8_t* const myStage = reinterpret_cast<uint8_t*>(&stage[0][0][0]) + (((tid - HALO / PX) >> 2) * STG) * 4 + ((tid - HALO / PX) & 3);

    auto slot = [&](int row) { return (row + 4 * RING) % RING; };
    auto rowOffOf = [&](int row) { return frameOff + (size_t)min(max(row, 0), H - 1) * W; };
    auto accumulate = [&](uint2 g, int d) {  // row y0 + d joins the warm-up sums V_k = sum_{d=-r_k-1}^{r_k-1} g(y0+d)
        uint32_t u[4];
        widen8(g, u);
        if (d >= -R0 - 1 && d <= R0 - 1) {
#pragma unroll
            for (int i = 0; i < 4; i++) V0[i] += u[i];
        }
        if (d >= -R1 - 1 && d <= R1 - 1) {
#pragma unroll
            for (int i = 0; i < 4; i++) V1[i] += u[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) V2[i] += u[i];
    };

    // warm-up: rows y0-RMAX-1 .. y0-1 go to the LDS ring, rows y0 .. y0+RMAX-1 to the delay line
    for (int d = -RMAX - 1; d <= -1; d++) {
        const uint2 g = load_gray8(in, C, W, rowOffOf(y0 + d), xbase, aligned);
        *reinterpret_cast<uint2*>(&ring[slot(y0 + d)][PX * tid]) = g;
        accumulate(g, d);
    }
    // The row loop is unrolled by UNR so that neither the rows in flight nor the delay line are moved between
    // registers every row: moving the destination of a load that is still in flight waits for that load, which
    // took the row prefetch away and left the kernel waiting for memory once per row.
    // A[i] = gray row y + i at the top of the group that starts at row y (A[0 .. RMAX-1] valid, the rest filled
    // by the group's rows); pend[j] = the raw row y + RMAX + j, in flight.
    constexpr int UNR = 4;
    uint2 A[RMAX + UNR];
#pragma unroll
    for (int k = 0; k < RMAX; k++) {
        A[k] = load_gray8(in, C, W, rowOffOf(y0 + k), xbase, aligned);
        accumulate(A[k], k);
    }
#pragma unroll
    for (int k = RMAX; k < RMAX + UNR; k++) A[k] = make_uint2(0, 0);
    RawRow pend[UNR];
#pragma unroll
    for (int k = 0; k < UNR; k++) pend[k] = issue(rowOffOf(y0 + RMAX + k));
    // the next row to fetch and its offset, kept by addition (a 64-bit multiply per row otherwise)
    int rowNext = y0 + RMAX + UNR;
    size_t offNext = rowOffOf(rowNext) * (MODE == 1 ? 3 : 1);  // (MODE 1: in bytes)
    // one row; j = its place in the group (a compile-time constant, so that A and pend are registers)
    // (sy = slot(y): the ring slots of the rows a step reads follow from it by additions -- as remainders of a division
    // by RING they were four scalar multiply-high sequences per row)
    auto wrap = [](int v) { return v >= RING ? v - RING : v; };
    auto do_row = [&](const int y, const int sy, auto jc) {
        constexpr int j = decltype(jc)::value;
        A[RMAX + j] = aligned ? gray_of(pend[j], C, W, xbase) : load_gray8(in, C, W, rowOffOf(y + RMAX), xbase, false);
        // row y + RMAX + UNR, clamped to the last row
        pend[j] = MODE == 1 ? issue_row3(in, offNext, laneByte) : issue(offNext);
        if (rowNext < H - 1) offNext += MODE == 1 ? (size_t)W * 3 : (size_t)W;
        rowNext++;

        uint32_t a[4], b[4];
        // every window takes in the row r_k below the centre (registers) and drops the row r_k + 1 above it (ring)
        widen8(A[R2 + j], a);
        widen8(*reinterpret_cast<const uint2*>(&ring[wrap(sy + (4 * RING - R2 - 1) % RING)][PX * tid]), b);
#pragma unroll
        for (int i = 0; i < 4; i++) V2[i] = V2[i] + a[i] - b[i];
        widen8(A[R1 + j], a);
        widen8(*reinterpret_cast<const uint2*>(&ring[wrap(sy + (4 * RING - R1 - 1) % RING)][PX * tid]), b);
#pragma unroll
        for (int i = 0; i < 4; i++) V1[i] = V1[i] + a[i] - b[i];
        widen8(A[R0 + j], a);
        widen8(*reinterpret_cast<const uint2*>(&ring[wrap(sy + (4 * RING - R0 - 1) % RING)][PX * tid]), b);
#pragma unroll
        for (int i = 0; i < 4; i++) V0[i] = V0[i] + a[i] - b[i];
        const uint2 cen = A[j];  // row y
        // row y replaces row y-RMAX-1 in the ring (read just above)
        *reinterpret_cast<uint2*>(&ring[sy][PX * tid]) = cen;

        // exchange vertical sums (single wavefront: the barrier only orders this wave's LDS traffic)
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            xbuf[0][j * XROW + XPAD + tid] = V0[j];
            xbuf[1][j * XROW + XPAD + tid] = V1[j];
            xbuf[2][j * XROW + XPAD + tid] = V2[j];
        }
        wave_lds_sync();

        if (inOut) {
            uint32_t s0[8], s1[8], s2[8];
            hsum8<R0>(xbuf[0], tid, (uint32_t)(c0 - 1), s0);
            hsum8<R1>(xbuf[1], tid, (uint32_t)(c1 - 1), s1);
            hsum8<R2>(xbuf[2], tid, (uint32_t)(c2 - 1), s2);
            uint32_t cv[8];
            unpack8(cen, cv);
            // one byte per scale, bit i = pixel xbase+i
            const uint32_t pm0 = thresh_bits(s0, cv, K0), pm1 = thresh_bits(s1, cv, K1), pm2 = thresh_bits(s2, cv, K2);
            const size_t off = frameOff + (size_t)y * W + xbase;
            if (MODE != 0 || fullOut) {
                if (bin) {
                    uint32_t ob[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) ob[i] = ((pm0 >> i) & 1u) | (((pm1 >> i) & 1u) << 1) | (((pm2 >> i) & 1u) << 2);
                    uint2 o;
                    o.x = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
                    o.y = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
                    *reinterpret_cast<uint2*>(bin + off) = o;
                }
                if (hasGray) *reinterpret_cast<uint2*>(gray + off) = cen;
            } else {
                for (int i = 0; i < 8; i++)
                    if (xbase + i < W) {
                        if (bin) bin[off + i] = (uint8_t)(((pm0 >> i) & 1u) | (((pm1 >> i) & 1u) << 1) | (((pm2 >> i) & 1u) << 2));
                        if (hasGray) gray[off + i] = (uint8_t)(((i < 4 ? cen.x : cen.y) >> (8 * (i & 3))) & 255u);
                    }
            }
            // bit planes with the 1-pixel image frame zeroed (what findContours works on)
            uint32_t keep = 0xFFu;
            if (y == 0 || y == H - 1) keep = 0;
            if (xbase == 0) keep &= 0xFEu;
            if (xbase + 7 >= W - 1) keep &= ~(1u << ((W - 1) - xbase)) & ((W - xbase) >= 8 ? 0xFFu : ((1u << (W - xbase)) - 1u));
            // stage this row's three bytes; bytes of columns >= W are never written and stay zero
            uint8_t* q = myStage + (y & (STG - 1)) * 4;
            q[0] = (uint8_t)(pm0 & keep);
            q[NBLK * STG * 4] = (uint8_t)(pm1 & keep);
            q[2 * NBLK * STG * 4] = (uint8_t)(pm2 & keep);
        }
        if ((y & (STG - 1)) == STG - 1) flush(y / STG);
    };
    // whole groups without an early exit (an exit between the rows of a group makes the compiler assume that
    // the loads of the first row are the youngest in flight, and wait for all of them at the top of the loop)
    int yg = y0, sg = slot(y0);
    for (; yg + UNR <= y1; yg += UNR) {
        do_row(yg, sg, std::integral_constant<int, 0>());
        do_row(yg + 1, wrap(sg + 1), std::integral_constant<int, 1>());
        do_row(yg + 2, wrap(sg + 2), std::integral_constant<int, 2>());
        do_row(yg + 3, wrap(sg + 3), std::integral_constant<int, 3>());
#pragma unroll
        for (int k = 0; k < RMAX; k++) A[k] = A[k + UNR];
        sg = wrap(sg + UNR);
    }
    static_assert(UNR == 4, "the row groups are written out for four rows");
    if (yg < y1) do_row(yg, sg, std::integral_constant<int, 0>());
    if (yg + 1 < y1) do_row(yg + 1, wrap(sg + 1), std::integral_constant<int, 1>());
    if (yg + 2 < y1) do_row(yg + 2, wrap(sg + 2), std::integral_constant<int, 2>());
    // the band that ends the image also writes the zero rows that pad the last block row
    if (y1 == H) {
        for (int yy = H; yy < plane_hb(H) * 32; yy++) {
            if (inOut) {
                uint8_t* q = myStage + (yy & (STG - 1)) * 4;
                q[0] = 0; q[NBLK * STG * 4] = 0; q[2 * NBLK * STG * 4] = 0;
            }
            if ((yy & (STG - 1)) == STG - 1) flush(yy / STG);
        }
    }
}

// Generic fallback for detector parameters other than the reference's
// (windows 3/13/23): direct sums, one thread per pixel.  Correct, not tuned.
__global__ void k_gray_generic(const uint8_t* __restrict__ in, int C, uint8_t* __restrict__ gray, int H, int W, size_t total) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (C == 3) gray[i] = (uint8_t)luma(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
    else gray[i] = in[i];
}

// one thread per 8 pixels (one plane byte per scale)
__global__ void k_thresh_generic(const uint8_t* __restrict__ gray, uint8_t* __restrict__ bin, uint32_t* __restrict__ planes,
                                 DetCfg cfg, int B) {
    const int H = cfg.H, W = cfg.W, W8 = W / 8;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * H * W8) return;
    const int xb = (int)(i % W8);
    const int y = (int)((i / W8) % H);
    const size_t f = i / ((size_t)W8 * H);
    const uint8_t* g = gray + f * (size_t)W * H;
    const size_t planeBytes = plane_words(H, cfg.WS) * 4;
    uint32_t pm[kMaxScales];
    for (int s = 0; s < cfg.nScales; s++) pm[s] = 0;
    for (int k = 0; k < 8; k++) {
        const int x = xb * 8 + k;
        const int src = g[(size_t)y * W + x];
        uint32_t m = 0;
        for (int s = 0; s < cfg.nScales; s++) {
            const int r = cfg.radius[s];
            int sum = 0;
            for (int dy = -r; dy <= r; dy++) {
                const int yy = min(max(y + dy, 0), H - 1);
                for (int dx = -r; dx <= r; dx++) {
                    const int xx = min(max(x + dx, 0), W - 1);
                    sum += g[(size_t)yy * W + xx];
                }
            }
            const int k2 = (2 * r + 1) * (2 * r + 1);
            const long long t2 = 2 * (src + cfg.idelta) - 1;
            if (2LL * sum >= t2 * k2) {
                m |= 1u << s;
                if (y > 0 && y < H - 1 && x > 0 && x < W - 1) pm[s] |= 1u << k;
            }
        }
        if (bin) bin[f * (size_t)W * H + (size_t)y * W + x] = (uint8_t)m;
    }
    uint8_t* pbase = reinterpret_cast<uint8_t*>(planes) + f * cfg.nScales * planeBytes;
    const int b0 = xb & 3, b1 = (xb == W8 - 1) ? 4 : b0 + 1;  // the row's last byte also clears its word's padding
    for (int s = 0; s < cfg.nScales; s++) {
        for (int b = b0; b < b1; b++)
            pbase[s * planeBytes + plane_word_index(cfg.WS, xb >> 2, y) * 4 + b] = (b == b0) ? (uint8_t)pm[s] : (uint8_t)0;
        if (y == H - 1)
            for (int yy = H; yy < plane_hb(H) * 32; yy++)
                for (int b = b0; b < b1; b++) pbase[s * planeBytes + plane_word_index(cfg.WS, xb >> 2, yy) * 4 + b] = 0;
    }
}

}  // namespace

bool gray_thresh_is_fused(const DetCfg& c) { return c.nScales == 3 && c.radius[0] == 1 && c.radius[1] == 6 && c.radius[2] == 11; }

void launch_gray_thresh(const DetCfg& c, const DetBuffers& b, int B, int C, hipStream_t st) {
    const bool fast = gray_thresh_is_fused(c);
    if (fast) {
        const int chunks = (c.W + CHUNKW - 1) / CHUNKW;
        // Single-wave workgroups of equal length, 12 resident per CU (registers): the launch runs in rounds, and a last
        // round that is mostly empty costs as much as a full one.  Choose the number of bands per strip that
        // minimises rounds x band length; every band re-reads 2 rmax + 2 rows above it to warm its running sums
        // up (about 12 rows' worth of time), and bands are whole plane blocks.
        const long slots = (long)device_cus() * 12;
        const int maxBands = max(1, (c.H + 63) / 64);
        int bands = 1, bandH = (c.H + 31) & ~31;
        long best = -1;
        for (int nb = 1; nb <= maxBands; nb++) {
            const int bh = (((c.H + nb - 1) / nb) + 31) & ~31;
            const int real = (c.H + bh - 1) / bh;
            const long rounds = ((long)B * chunks * real + slots - 1) / slots;
            const long cost = rounds * (bh + 12);
            if (best < 0 || cost < best) {
                best = cost;
                bands = real;
                bandH = bh;
            }
        }
        // 1-D grid, see the kernel: groups of 8 units x strips
        dim3 grid((unsigned)(((long)bands * B + 7) / 8 * 8 * chunks));
        const bool plain = (c.W % 8) == 0 && !b.bin;
        if (plain && C == 3)
            hipLaunchKernelGGL((k_gray_thresh3<1, 6, 11, 1>), grid, dim3(TPB), 0, st, b.in, C, b.gray, b.bin, b.planes, c.WS, c.H, c.W, bandH, c.idelta, B);
        else if (plain && C == 1)
            hipLaunchKernelGGL((k_gray_thresh3<1, 6, 11, 2>), grid, dim3(TPB), 0, st, b.in, C, b.gray, b.bin, b.planes, c.WS, c.H, c.W, bandH, c.idelta, B);
        else
            hipLaunchKernelGGL((k_gray_thresh3<1, 6, 11, 0>), grid, dim3(TPB), 0, st, b.in, C, b.gray, b.bin, b.planes, c.WS, c.H, c.W, bandH, c.idelta, B);
    } else {
        size_t total = (size_t)B * c.H * c.W;
        unsigned blocks = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(k_gray_generic, dim3(blocks), dim3(256), 0, st, b.in, C, b.gray, c.H, c.W, total);
        size_t total8 = total / 8;
        unsigned blocks8 = (unsigned)((total8 + 255) / 256);
        hipLaunchKernelGGL(k_thresh_generic, dim3(blocks8), dim3(256), 0, st, b.gray, b.bin, b.planes, c, B);
    }
}

}  // namespace camsync
