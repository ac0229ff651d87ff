// k_jpeg.hip — compressed frame ingest: baseline JPEG / MJPEG frames -> BGR frames on the device, as
// cv.VideoCapture(...).read() on a camera that delivers MJPG and cv.imread of the reference's stored photos give
// them to the reference (camera_sync/camera.py:98-118, 150-154; both end in libjpeg-turbo with its defaults:
// accurate integer IDCT, triangle-filter chroma upsampling, 16-bit fixed-point YCbCr -> RGB).
//
// Stages per batch:
//   entropy decoding (ITU T.81 Annex F).  A frame's Huffman stream is sequential, but a decoder that enters it anywhere
//                 falls into step with the real one after a while: k_jpeg_unstuff removes the byte stuffing,
//                 k_jpeg_par gives every 1024-bit subsequence of a frame a lane and iterates until each lane's entry
//                 state is the exit state of the lane before it, then decodes once more and stores the coefficients,
//                 k_jpeg_dc sums the DC differences (see "many lanes per frame" below).  Frames with restart intervals
//                 and frames that do not settle take the sequential kernels: k_jpeg_huff_wave gives a frame -- or, with
//                 restart intervals, each run of intervals of a frame -- a wavefront (every lane decodes the symbol that
//                 would start at its bit offset of a 64-bit window, the real symbols are the chain from offset 0),
//                 k_jpeg_huff (more than two Huffman tables of a class) a lane, a few frames of one table set to a
//                 wavefront.  All store the coefficients straight into the frame's coefficient
//                 planes, in coding order; the planes are all zero between calls.
//   k_jpeg_idct   dequantisation + inverse DCT (jidctint.c's two passes, 32-bit) of every block: eight lanes per block,
//                 a row of coefficients each (coalesced 16-byte loads), transposed through LDS for the column pass;
//                 writes the component planes and puts the zeros back into the coefficient planes.
//   k_jpeg_color  chroma upsampling (h2v1 / h2v2 triangle filters, or none) + colour conversion + BGR packing, eight
//                 pixels per lane.
// Bound: 4096 1080p frames (1 GB compressed) take 100 ms: 18 ms the link, ~32 ms entropy decoding (instruction issue:
// five passes over the stream, a lane per subsequence), ~20 ms the inverse DCT and ~12 ms the colour conversion (HBM:
// 128 B + 128 B + 64 B per block; 1.5-3 B + 3 B per pixel).  One frame: 1.3 ms.
//
// Scope (the host parser refuses the rest with CAMSYNC_E_ARG): SOF0 / SOF1, 8-bit, one interleaved scan, 1 or 3
// components (Y / YCbCr), luma sampling 1x1, 2x1 or 2x2 with 1x1 chroma, restart intervals, frames without DHT
// segments (the Annex K tables).  Results equal the CPU restatement under tests/ bit for bit (tests/test_jpeg.py).
#include "internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace camsync {

namespace {

struct JpegDesc {         // one frame
    uint32_t dataOff;     // its entropy-coded bytes in the batch's byte buffer
    uint32_t dataLen;
    uint16_t restart;     // restart interval in MCUs, 0 = none
    uint16_t mcux, mcuy;
    uint8_t nc, hmax, vmax;
    uint8_t tq[3], td[3], ta[3];  // quantisation table; DC / AC table as a slot (0..2) of the workgroup's table set
    uint8_t pad;
};

constexpr int kHuffTableBytes = 1024;  // look[256] u16 | limit[18] u32 | valoff[17] i32 | huffval[256] u8
constexpr int kHuffSetTables = 6;      // DC slots 0..2, AC slots 0..2
constexpr int kOffLimit = 512, kOffValoff = 512 + 72, kOffHuffval = 512 + 72 + 68;

// where the coefficient coded k-th goes in k_jpeg_idct's LDS block (row * 9 + column)
#define ZZ9(n) (uint8_t)(((n) >> 3) * 9 + ((n) & 7))
__constant__ __align__(8) uint8_t c_zigzag9[64] = {
    ZZ9(0),  ZZ9(1),  ZZ9(8),  ZZ9(16), ZZ9(9),  ZZ9(2),  ZZ9(3),  ZZ9(10), ZZ9(17), ZZ9(24), ZZ9(32), ZZ9(25), ZZ9(18),
    ZZ9(11), ZZ9(4),  ZZ9(5),  ZZ9(12), ZZ9(19), ZZ9(26), ZZ9(33), ZZ9(40), ZZ9(48), ZZ9(41), ZZ9(34), ZZ9(27), ZZ9(20),
    ZZ9(13), ZZ9(6),  ZZ9(7),  ZZ9(14), ZZ9(21), ZZ9(28), ZZ9(35), ZZ9(42), ZZ9(49), ZZ9(56), ZZ9(57), ZZ9(50), ZZ9(43),
    ZZ9(36), ZZ9(29), ZZ9(22), ZZ9(15), ZZ9(23), ZZ9(30), ZZ9(37), ZZ9(44), ZZ9(51), ZZ9(58), ZZ9(59), ZZ9(52), ZZ9(45),
    ZZ9(38), ZZ9(31), ZZ9(39), ZZ9(46), ZZ9(53), ZZ9(60), ZZ9(61), ZZ9(54), ZZ9(47), ZZ9(55), ZZ9(62), ZZ9(63)};
#undef ZZ9
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- entropy decoding ------------------------------------------------------------------------------------------------
// groups: [nGroups][1 + kHuffLanes] = table set, then the frames of the workgroup's lanes (-1 = none).
// status[f]: 0 ok, 1 corrupt entropy data.  Coefficients are stored in the order they are coded (zigzag); k_jpeg_idct
// puts them in place.
//
// What the loop is built around: a lane's chain of dependent steps is all there is to hide nothing behind (a few dozen
// waves on the whole chip), so what matters is what a step waits for.  (1) Global loads and stores share one counter on
// this chip and a wait for a load also waits for every store issued before it -- with the coefficient stores in the
// loop, every wait for bitstream bytes became a wait for the previous symbol's store to reach the L2 (1.3 us per
// symbol).  So the loop never waits for memory: each lane's next kilobyte of bitstream sits in an LDS window, refilled
// by the whole wave with one coalesced 16-byte load per lane (a wait for memory once per ~1600 symbols of a lane), and
// the stores are fire-and-forget.  (2) A code longer than the 8-bit look-up finds its length by eight comparisons
// against left-aligned limits instead of a loop of dependent look-ups; with several lanes per wave some lane needs
// that path in most steps.  (3) Few frames per wave: the rare paths (byte stuffing, long codes, restart markers,
// window refills) cost every lane of the wave whenever one lane takes them, and with a wave per SIMD at most there are
// SIMDs to spare.  2048 1080p frames, ms per batch: 64 frames per wave (and loads from memory) 555, 32: 440, 16: 386,
// 8: 333, 4: 288, 2: 288.  What is left is the chain itself: ~0.75 us per symbol on a wave that has its SIMD to itself.
#ifndef HUFF_LANES
#define HUFF_LANES 4
#endif
constexpr int kHuffLanes = HUFF_LANES;  // frames per workgroup (= wavefront)
constexpr int kHuffWindow = 1024; // bytes of bitstream per lane in LDS
constexpr int kHuffSlack = 16;    // a step reads at most this far ahead of its position

__global__ __launch_bounds__(64) void k_jpeg_huff(const uint8_t* __restrict__ bytes, const JpegDesc* __restrict__ desc,
                                                  const int32_t* __restrict__ groups, const uint8_t* __restrict__ tableSets,
                                                  int16_t* __restrict__ coef, uint32_t frameBlocks, int32_t* __restrict__ status) {
    __shared__ __align__(16) uint8_t s_tab[kHuffSetTables * kHuffTableBytes];
    __shared__ __align__(16) uint8_t s_win[kHuffLanes][kHuffWindow];
    const int lane = threadIdx.x;
    const int32_t* grp = groups + (size_t)blockIdx.x * (1 + kHuffLanes);
    {
        const uint4* src = reinterpret_cast<const uint4*>(tableSets + (size_t)grp[0] * kHuffSetTables * kHuffTableBytes);
        uint4* dst = reinterpret_cast<uint4*>(s_tab);
        for (int i = lane; i < kHuffSetTables * kHuffTableBytes / 16; i += 64) dst[i] = src[i];
    }
    __syncthreads();
    const int f = lane < kHuffLanes ? grp[1 + lane] : -1;
    bool active = f >= 0;
    JpegDesc d{};
    if (active) d = desc[f];
    const uint32_t limit = d.dataLen;
    int16_t* const fcoef = coef + (size_t)max(f, 0) * frameBlocks * 64;
    const int nc = d.nc;
    // blocks per row of each component's plane, and the plane's first block
    const uint32_t bw0 = (uint32_t)d.mcux * d.hmax, n0 = bw0 * ((uint32_t)d.mcuy * d.vmax), n1 = (uint32_t)d.mcux * d.mcuy;
    const uint8_t* const win = s_win[lane & (kHuffLanes - 1)];

    uint64_t acc = 0;   // unread bits, left-aligned
    int nb = 0;         // how many of them are valid
    uint32_t ptr = 0;   // next byte of the stream to read
    uint32_t wbase = 0xFFFF0000u;  // stream offset of the window's first byte (none loaded yet)
    bool atMarker = false;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int c = 0, bx = 0, by = 0, k = 0;
    uint32_t mx = 0, my = 0;
    int toRestart = d.restart, nextRst = 0;
    uint32_t block = 0;  // the current block's first coefficient

    for (;;) {
        // ---- windows: every lane's next kHuffSlack bytes are in LDS (refilled by the whole wave, one lane at a time) ----
        unsigned long long need = __ballot(active && (ptr - wbase > (uint32_t)(kHuffWindow - kHuffSlack)));
        if (need) {
            while (need) {
                const int L = __ffsll((long long)need) - 1;
                need &= need - 1;
                const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)(ptr & ~15u), L);
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)d.dataOff, L);
                // (the byte buffer is padded by more than a window, so the load never leaves it)
                const uint4 v = *reinterpret_cast<const uint4*>(bytes + (size_t)off + base + (size_t)lane * 16);
                *reinterpret_cast<uint4*>(&s_win[L][lane * 16]) = v;
                if (lane == L) wbase = base;
            }
            wave_lds_order();
        }
        if (!__any(active)) break;
        if (active) {
            // ---- at least 33 valid bits: a code (<= 16 bits) and its value (<= 16 bits) ----
            if (nb <= 32) {
                if (atMarker) {
                    nb = 64;  // (zeros, as jdhuff.c feeds after a marker)
                } else {
                    const uint32_t rel = ptr - wbase;
                    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(win) + (rel >> 2);
                    const uint32_t w = __builtin_amdgcn_alignbyte(w32[1], w32[0], rel & 3u);  // the four bytes at ptr
                    const uint32_t t = ~w;  // a zero byte of t is an 0xFF byte of w
                    if (ptr + 4 <= limit && ((t - 0x01010101u) & ~t & 0x80808080u) == 0) {
                        acc |= (uint64_t)__builtin_bswap32(w) << (32 - nb);
                        nb += 32;
                        ptr += 4;
                    } else {
                        uint32_t r = rel;
                        while (nb <= 56) {
                            if (ptr >= limit) { atMarker = true; break; }
                            const uint32_t b = win[r];
                            if (b == 0xFF) {
                                const uint32_t nx = ptr + 1 < limit ? win[r + 1] : 0xD9u;
                                if (nx != 0) { atMarker = true; break; }
                                ptr += 2;  // a stuffed zero
                                r += 2;
                            } else {
                                ptr++;
                                r++;
                            }
                            acc |= (uint64_t)b << (56 - nb);
                            nb += 8;
                        }
                        if (atMarker) nb = 64;
                    }
                }
            }
            // ---- one symbol ----
            bool bad = false;
            const int slot = k == 0 ? (c == 0 ? d.td[0] : c == 1 ? d.td[1] : d.td[2]) : 3 + (c == 0 ? d.ta[0] : c == 1 ? d.ta[1] : d.ta[2]);
            const uint8_t* const T = s_tab + slot * kHuffTableBytes;
            const uint32_t e = reinterpret_cast<const uint16_t*>(T)[(uint32_t)(acc >> 56)];
            int l = (int)(e >> 8), sym = (int)(e & 255u);
            if (!e) {
                // longer than 8 bits: the code's length is 9 + the number of lengths whose left-aligned limit it reaches
                const uint32_t* lim = reinterpret_cast<const uint32_t*>(T + kOffLimit);
                const uint32_t code16 = (uint32_t)(acc >> 48);
                l = 9;
#pragma unroll
                for (int j = 9; j <= 15; j++) l += code16 >= lim[j] ? 1 : 0;
                bad = code16 >= lim[16];
                const int32_t* valoff = reinterpret_cast<const int32_t*>(T + kOffValoff);
                sym = T[kOffHuffval + (((int32_t)(code16 >> (16 - l)) + valoff[l]) & 255)];
            }
            acc <<= l;
            nb -= l;
            const int s = k == 0 ? sym : (sym & 15);
            int val = 0;
            if (s) {
                const int sl = min(s, 16);
                const int r = (int)(acc >> (64 - sl));
                acc <<= sl;
                nb -= sl;
                val = r < (1 << (sl - 1)) ? r - (1 << sl) + 1 : r;
            }
            if (k == 0) {
                if (sym > 16) bad = true;
                int pr = c == 0 ? pred0 : c == 1 ? pred1 : pred2;
                pr += val;
                if (c == 0) pred0 = pr; else if (c == 1) pred1 = pr; else pred2 = pr;
                // the block this DC coefficient starts
                const uint32_t bwc = c == 0 ? bw0 : (uint32_t)d.mcux;
                const uint32_t row = c == 0 ? my * d.vmax + (uint32_t)by : my, col = c == 0 ? mx * d.hmax + (uint32_t)bx : mx;
                block = ((c == 0 ? 0u : c == 1 ? n0 : n0 + n1) + row * bwc + col) * 64u;
                if (!bad) fcoef[block] = (int16_t)pr;
                k = 1;
            } else {
                const int r = sym >> 4;
                if (s) {
                    k += r;
                    if (k > 63) bad = true;
                    else fcoef[block + (uint32_t)k] = (int16_t)val;
                    k++;
                } else {
                    k = r == 15 ? k + 16 : 64;
                }
            }
            if (bad) {
                status[f] = 1;
                active = false;
            } else if (k >= 64) {
                // ---- next block, MCU, restart interval ----
                k = 0;
                const int hc = c == 0 ? d.hmax : 1, vc = c == 0 ? d.vmax : 1;
                if (++bx == hc) {
                    bx = 0;
                    if (++by == vc) {
                        by = 0;
                        if (++c == nc) {
                            c = 0;
                            if (++mx == d.mcux) { mx = 0; my++; }
                            if (my == d.mcuy) {
                                active = false;
                            } else if (d.restart && --toRestart == 0) {
                                // the bits left are the padding of the interval's last byte; the marker follows (after
                                // fill bytes).  Rare: read from memory.
                                const uint8_t* const p = bytes + d.dataOff;
                                acc = 0;
                                nb = 0;
                                while (ptr + 1 < limit && p[ptr] == 0xFF && p[ptr + 1] == 0xFF) ptr++;
                                if (ptr + 1 < limit && p[ptr] == 0xFF && p[ptr + 1] == (uint32_t)(0xD0 + nextRst)) {
                                    ptr += 2;
                                    atMarker = false;
                                    nextRst = (nextRst + 1) & 7;
                                    pred0 = pred1 = pred2 = 0;
                                    toRestart = d.restart;
                                } else {
                                    status[f] = 1;
                                    active = false;
                                }
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- entropy decoding, one wavefront per frame ------------------------------------------------------------------------
// A lane per frame leaves a frame to one lane's chain of dependent steps: a quarter of a second for a 1080p frame, which
// is what a caller pays who decodes one frame at a time (Camera.takePic over an MJPG camera).  Here a wavefront takes a
// frame: every lane decodes the symbol that WOULD start at its own bit offset of a 64-bit window (look-up, long codes,
// value bits -- everything that costs instructions -- for each of the tables a symbol there could use), then the
// symbols that really are there are found by following the chain of lengths from offset 0, ~30 scalar instructions per
// symbol (v_readlane), and the lanes on the chain store their coefficients together.  ~10 symbols per window.
// One 1080p frame of 236 KB: 73 ms (a lane per frame: 240 ms); 4096 of them: 282 ms (378 ms).  All four tables are
// worked out for every window: computing a DC table or the second AC table only when the chain gets to a block that
// needs it saved instructions and was slower (79 ms), because the look-ups of the four then wait one after the other.
// The bitstream is unstuffed into an LDS ring first (0xFF 0x00 -> 0xFF; the ring ends at a marker, zeros after it as
// jdhuff.c feeds them), so that bit offsets are plain arithmetic.  Frames with more than two DC or two AC tables go to
// k_jpeg_huff.
constexpr int kRingBytes = 2048;       // unstuffed bytes in LDS (a power of two)
constexpr int kRefillBytes = 256;      // raw bytes per refill: four per lane
constexpr int kRingNeed = 24;          // bytes a window needs ahead of its first bit (63 + 32 bits, whole words)

// symbol that starts at the top of w (32 bits of the stream) under table T: bits consumed | run << 6 | (size == 0) << 10 |
// bad << 11 | value << 16.  lim = the table's limits for lengths 9..16 (uniform).
__device__ __forceinline__ uint32_t huff_symbol(const uint8_t* T, const uint32_t* lim, uint32_t w, bool isDC) {
    const uint32_t e = reinterpret_cast<const uint16_t*>(T)[w >> 24];
    const uint32_t code16 = w >> 16;
    uint32_t ll = 9;
#pragma unroll
    for (int j = 0; j < 7; j++) ll += code16 >= lim[j] ? 1u : 0u;
    const bool bad = !e && code16 >= lim[7];
    const int32_t* valoff = reinterpret_cast<const int32_t*>(T + kOffValoff);
    const uint32_t lsym = T[kOffHuffval + (((int32_t)(code16 >> (16u - ll)) + valoff[ll]) & 255)];
    const uint32_t l = e ? e >> 8 : ll, sym = e ? e & 255u : lsym;
    const uint32_t size = isDC ? min(sym, 16u) : (sym & 15u), run = isDC ? 0u : sym >> 4;
    const uint32_t vb = size ? (w << l) >> (32u - size) : 0u;
    const int val = size && vb < (1u << (size - 1u)) ? (int)vb - (int)(1u << size) + 1 : (int)vb;
    return (l + size) | (run << 6) | ((size == 0u ? 1u : 0u) << 10) | (((bad || (isDC && sym > 16u)) ? 1u : 0u) << 11) | ((uint32_t)val << 16);
}

// list: [n][kWaveEntry] = frame, table set, first byte of the wave's part of the entropy-coded data, its first MCU, its
// number of MCUs, the number (0..7) of the first restart marker it meets.  A frame without restart intervals is one
// part; with them, a part is a run of whole intervals (each starts in a known state: a frame's parts are independent).
constexpr int kWaveEntry = 6;
__global__ __launch_bounds__(64) void k_jpeg_huff_wave(const uint8_t* __restrict__ bytes, const JpegDesc* __restrict__ desc,
                                                       const int32_t* __restrict__ list, const uint8_t* __restrict__ tableSets,
                                                       int16_t* __restrict__ coef, uint32_t frameBlocks, int32_t* __restrict__ status) {
    __shared__ __align__(16) uint8_t s_tab[4 * kHuffTableBytes];  // DC slots 0, 1, AC slots 0, 1
    __shared__ __align__(16) uint32_t s_ring[kRingBytes / 4];
    const int lane = threadIdx.x;
    const int32_t* const entry = list + (size_t)kWaveEntry * blockIdx.x;
    const int f = entry[0];
    {
        const uint8_t* set = tableSets + (size_t)entry[1] * kHuffSetTables * kHuffTableBytes;
        uint4* dst = reinterpret_cast<uint4*>(s_tab);
        for (int i = lane; i < 4 * kHuffTableBytes / 16; i += 64) {
            const int t = i / (kHuffTableBytes / 16), r = i - t * (kHuffTableBytes / 16);
            dst[i] = reinterpret_cast<const uint4*>(set + (size_t)(t < 2 ? t : t + 1) * kHuffTableBytes)[r];  // set: DC 0..2, AC 0..2
        }
    }
    __syncthreads();
    const JpegDesc d = desc[f];
    const uint32_t partStart = (uint32_t)entry[2], firstMcu = (uint32_t)entry[3];
    uint32_t mcusLeft = (uint32_t)entry[4];
    const uint8_t* const p = bytes + d.dataOff + partStart;
    const uint32_t limit = d.dataLen - partStart;
    int16_t* const fcoef = coef + (size_t)f * frameBlocks * 64;
    // the limits of lengths 9..16 of the four tables (uniform)
    uint32_t lim[4][8];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int j = 0; j < 8; j++)
            lim[t][j] = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(s_tab + t * kHuffTableBytes + kOffLimit)[9 + j]);
    const uint32_t bw0 = (uint32_t)d.mcux * d.hmax, n0 = bw0 * ((uint32_t)d.mcuy * d.vmax), n1 = (uint32_t)d.mcux * d.mcuy;
    uint8_t* const ring8 = reinterpret_cast<uint8_t*>(s_ring);

    // all of this is the same in every lane
    uint32_t P = 0;        // bit position in the unstuffed stream (of the current restart interval)
    uint32_t wp = 0;       // unstuffed bytes in the ring so far
    uint32_t rp = 0;       // raw bytes consumed
    uint32_t prevByte = 0; // the raw byte before rp
    bool atMarker = false;
    int pred0 = 0, pred1 = 0, pred2 = 0;
    int c = 0, bx = 0, by = 0, k = 0;
    uint32_t my = firstMcu / d.mcux, mx = firstMcu - my * d.mcux, block = 0;
    int toRestart = d.restart, nextRst = entry[5];
    bool done = mcusLeft == 0u, bad = false, restartNow = false;
    // (a part of zeros decodes one symbol per window at worst: a bound on the windows that no stream reaches)
    uint32_t guard = (mcusLeft * (uint32_t)(d.nc == 1 ? 1 : d.hmax * d.vmax + 2) + 1u) * 80u + 1024u;

    auto refill = [&]() {
        if (atMarker) {  // zeros from here on
#pragma unroll
            for (int j = 0; j < 4; j++) ring8[(wp + 4u * (uint32_t)lane + (uint32_t)j) & (kRingBytes - 1)] = 0;
            wp += kRefillBytes;
            return;
        }
        const uint32_t at = rp + 4u * (uint32_t)lane;
        uint32_t w;
        __builtin_memcpy(&w, p + at, 4);  // (the byte buffer is padded: the load is in bounds wherever the frame ends)
        const uint32_t nextW = (uint32_t)__builtin_amdgcn_update_dpp((int)0, (int)w, 0x130, 0xf, 0xf, false);  // wave_shl:1 = lane + 1's word
        const uint32_t prevW = (uint32_t)__builtin_amdgcn_update_dpp((int)0, (int)w, 0x138, 0xf, 0xf, false);  // wave_shr:1 = lane - 1's word
        const uint32_t peek = lane == 63 ? (uint32_t)p[rp + kRefillBytes] : (nextW & 255u);
        const uint32_t before = lane == 0 ? prevByte : prevW >> 24;
        uint32_t keep = 0, cutHere = 4;  // cutHere: first byte of this lane at which the chunk ends (marker or end of data)
#pragma unroll
        for (int j = 3; j >= 0; j--) {
            const uint32_t b = (w >> (8 * j)) & 255u;
            const uint32_t nx = j < 3 ? (w >> (8 * j + 8)) & 255u : peek;
            const uint32_t pv = j > 0 ? (w >> (8 * j - 8)) & 255u : before;
            const bool ends = at + (uint32_t)j >= limit || (b == 0xFFu && (at + (uint32_t)j + 1u >= limit || nx != 0u));
            if (ends) cutHere = (uint32_t)j;
            if (!(b == 0u && pv == 0xFFu)) keep |= 1u << j;
        }
        const unsigned long long cutMask = __ballot(cutHere < 4u);
        uint32_t cut = kRefillBytes;
        if (cutMask) {
            const int L = __ffsll((long long)cutMask) - 1;
            cut = 4u * (uint32_t)L + (uint32_t)__builtin_amdgcn_readlane((int)cutHere, L);
        }
        // bytes of this lane that go to the ring: kept and before the cut
        const uint32_t mine = cut > 4u * (uint32_t)lane ? min(cut - 4u * (uint32_t)lane, 4u) : 0u;
        keep &= (1u << mine) - 1u;
        const uint32_t cnt = (uint32_t)__popc(keep);
        const unsigned long long b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
        const unsigned long long below = (1ull << lane) - 1ull;
        uint32_t at8 = wp + (uint32_t)__popcll(b0 & below) + 2u * (uint32_t)__popcll(b1 & below) + 4u * (uint32_t)__popcll(b2 & below);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (keep & (1u << j)) ring8[(at8++) & (kRingBytes - 1)] = (uint8_t)(w >> (8 * j));
        wp += (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
        rp += cut;
        if (cut < (uint32_t)kRefillBytes) atMarker = true;
        else prevByte = (uint32_t)__builtin_amdgcn_readlane((int)(w >> 24), 63);
    };

    while (!done && !bad) {
        if (restartNow) {
            // the rest of the interval is padding; the marker follows (after fill bytes)
            while (!atMarker) refill();
            while (rp + 1u < limit && p[rp] == 0xFF && p[rp + 1] == 0xFF) rp++;
            if (rp + 1u < limit && p[rp] == 0xFF && p[rp + 1] == (uint32_t)(0xD0 + nextRst)) {
                rp += 2;
                atMarker = false;
                nextRst = (nextRst + 1) & 7;
                pred0 = pred1 = pred2 = 0;
                toRestart = d.restart;
                P = 0;
                wp = 0;
                prevByte = 0;
            } else {
                bad = true;
                break;
            }
            restartNow = false;
        }
        while (wp - (P >> 3) < (uint32_t)kRingNeed) {
            wave_lds_order();
            refill();
        }
        wave_lds_order();
        // ---- every lane: the symbol that would start at its offset, under each table ----
        const uint32_t o = P + (uint32_t)lane;
        const uint32_t hi = __builtin_bswap32(s_ring[(o >> 5) & (kRingBytes / 4 - 1)]), lo = __builtin_bswap32(s_ring[((o >> 5) + 1u) & (kRingBytes / 4 - 1)]);
        const uint32_t w = (uint32_t)(((((uint64_t)hi << 32) | lo) << (o & 31u)) >> 32);
        const uint32_t pkD0 = huff_symbol(s_tab, lim[0], w, true), pkD1 = huff_symbol(s_tab + kHuffTableBytes, lim[1], w, true);
        const uint32_t pkA0 = huff_symbol(s_tab + 2 * kHuffTableBytes, lim[2], w, false), pkA1 = huff_symbol(s_tab + 3 * kHuffTableBytes, lim[3], w, false);
        // ---- the symbols that are there: the chain from offset 0 (scalar: every value here is the wave's) ----
        int offv = -1, valv = 0;
        uint32_t pos = 0;
        while (pos < 64u) {
            if (k == 0) {
                // a block begins: its DC symbol
                const int td = c == 0 ? d.td[0] : c == 1 ? d.td[1] : d.td[2];
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)(td ? pkD1 : pkD0), (int)pos);
                if (r & (1u << 11)) { bad = true; break; }
                int pr = c == 0 ? pred0 : c == 1 ? pred1 : pred2;
                pr += (int)r >> 16;
                if (c == 0) pred0 = pr; else if (c == 1) pred1 = pr; else pred2 = pr;
                const uint32_t bwc = c == 0 ? bw0 : (uint32_t)d.mcux;
                const uint32_t row = c == 0 ? my * d.vmax + (uint32_t)by : my, col = c == 0 ? mx * d.hmax + (uint32_t)bx : mx;
                block = ((c == 0 ? 0u : c == 1 ? n0 : n0 + n1) + row * bwc + col) * 64u;
                offv = lane == (int)pos ? (int)block : offv;
                valv = lane == (int)pos ? pr : valv;
                k = 1;
                pos += r & 63u;
                if (pos >= 64u) break;
            }
            // the block's AC symbols: the plain ones (a run and a value) in a tight loop
            const int ta = c == 0 ? d.ta[0] : c == 1 ? d.ta[1] : d.ta[2];
            const uint32_t pkA = ta ? pkA1 : pkA0;
            while (k < 64 && pos < 64u) {
                // plain coefficients (a run and a value): one exit test per symbol
                uint32_t r;
                int kn;
                bool plain;
                for (;;) {
                    r = (uint32_t)__builtin_amdgcn_readlane((int)pkA, (int)pos);
                    kn = k + (int)((r >> 6) & 15u);
                    plain = (r & (3u << 10)) == 0u && kn <= 63;
                    if (!plain) break;
                    offv = lane == (int)pos ? (int)(block + (uint32_t)kn) : offv;
                    valv = lane == (int)pos ? (int)r >> 16 : valv;
                    k = kn + 1;
                    pos += r & 63u;
                    if (!(k < 64 && pos < 64u)) break;
                }
                if (plain) break;  // the block or the window is used up
                // end of block, a run of sixteen zeros, a code that does not exist or a run past the block's end
                if ((r & (1u << 11)) || !(r & (1u << 10))) { bad = true; break; }
                k = ((r >> 6) & 15u) == 15u ? k + 16 : 64;
                pos += r & 63u;
            }
            if (bad) break;
            if (k >= 64) {
                k = 0;
                const int hc = c == 0 ? d.hmax : 1, vc = c == 0 ? d.vmax : 1;
                if (++bx == hc) {
                    bx = 0;
                    if (++by == vc) {
                        by = 0;
                        if (++c == d.nc) {
                            c = 0;
                            if (++mx == d.mcux) { mx = 0; my++; }
                            if (--mcusLeft == 0u) { done = true; break; }
                            if (d.restart && --toRestart == 0) { restartNow = true; break; }
                        }
                    }
                }
            }
        }
        if (offv >= 0) fcoef[offv] = (int16_t)valv;
        P += pos;
        if (--guard == 0u) bad = true;
    }
    if (bad && lane == 0) status[f] = 1;
}

// ---- entropy decoding, many lanes per frame ---------------------------------------------------------------------------
// A Huffman stream can be entered anywhere: a decoder that starts at an arbitrary bit with a guessed state (which block
// of the MCU, which coefficient) reads nonsense for a while and then, with high probability, falls into step with the
// real decoder -- from there on it produces what the real one does.  So the frame's stream (byte stuffing removed first,
// k_jpeg_unstuff) is cut into subsequences of kParBits bits with a lane each.  A record per subsequence holds where and in
// which state the decoder leaves it: (bit position of the first symbol beyond it, block of the MCU, coefficient) and the
// number of blocks completed inside it.  k_jpeg_par<false> makes every lane decode its subsequence from the state its
// predecessor's record names (at first a guess: the subsequence's first bit, start of an MCU) and write its own record;
// lanes whose entry state has not changed since they last ran do nothing.  Lane 0 enters with the true state, so after
// r rounds the first r records are true; because of the self-synchronisation nearly all are after three or four, and
// a round in which no record changes proves that every record follows from its predecessor's, i.e. all are true.
// k_jpeg_par<true> then decodes every subsequence once more from its true entry state, now knowing (prefix sum of the
// block counts) which block it is in, and stores the coefficients; DC differences are stored as they are coded and
// summed per component afterwards (k_jpeg_dc).  One 1080p frame: ~1900 lanes instead of one chain of 350 k symbols.
// Where decoders do not fall into step (flat or periodic stretches, quality-100 frames) the true records still spread
// one subsequence per round, and a round that nearly every lane leaves at once takes 10-20 us: a frame of n
// subsequences has settled after n rounds at the latest.  Frames with restart intervals (whose intervals need no
// guessing: k_jpeg_huff_wave decodes them side by side), and frames whose records have not settled after
// kParMaxRounds rounds, take the sequential kernels above.
#ifndef PAR_BITS
#define PAR_BITS 1024
#endif
constexpr uint32_t kParBits = PAR_BITS;  // a multiple of 32
constexpr int kParMaxRounds = 3072;
constexpr unsigned long long kParStateMask = 0xFFFFFFFFFFFFull;  // position | block of the MCU << 32 | coefficient << 40

// One workgroup per frame: the frame's entropy-coded bytes without the stuffed zeros (0xFF 0x00 -> 0xFF), up to the
// first marker, zeros from there to the end of the frame's place in the buffer; its length in bits; its records set to
// the guesses.  list[blockIdx.x] = frame.
__global__ __launch_bounds__(256) void k_jpeg_unstuff(const uint8_t* __restrict__ bytes, const JpegDesc* __restrict__ desc,
                                                      const int32_t* __restrict__ list, uint8_t* __restrict__ ubytes,
                                                      uint32_t* __restrict__ ulenBits, const uint32_t* __restrict__ subBase,
                                                      unsigned long long* __restrict__ exitRec, unsigned long long* __restrict__ lastEntry) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_cut;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = list[blockIdx.x];
    const JpegDesc d = desc[f];
    const uint8_t* const src = bytes + d.dataOff;
    uint8_t* const dst = ubytes + d.dataOff;
    const uint32_t len = d.dataLen, place = (len + 15u) & ~15u;
    uint32_t running = 0;  // bytes written so far (uniform)
    for (uint32_t tile = 0; tile < len; tile += 4096u) {
        if (tid == 0) s_cut = 0xFFFFFFFFu;
        __syncthreads();
        const uint32_t at = tile + 16u * (uint32_t)tid;
        // (the byte buffer is padded beyond its last frame: these loads are in bounds wherever the frame ends)
        const uint4 v = *reinterpret_cast<const uint4*>(src + at);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        const uint32_t before = at > 0 ? src[at - 1] : 0u, after = src[at + 16];
        uint32_t keep = 0, myCut = 16;
#pragma unroll
        for (int j = 15; j >= 0; j--) {
            const uint32_t b = (w[j >> 2] >> (8 * (j & 3))) & 255u;
            const uint32_t nx = j < 15 ? (w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 255u : after;
            const uint32_t pv = j > 0 ? (w[(j - 1) >> 2] >> (8 * ((j - 1) & 3))) & 255u : before;
            if (at + (uint32_t)j >= len || (b == 0xFFu && nx != 0u)) myCut = (uint32_t)j;  // the data ends here
            if (!(b == 0u && pv == 0xFFu)) keep |= 1u << j;
        }
        if (myCut < 16u) atomicMin(&s_cut, at + myCut);
        __syncthreads();
        const uint32_t cut = s_cut;
        const uint32_t mine = cut > at ? min(cut - at, 16u) : 0u;
        keep &= (1u << mine) - 1u;
        const uint32_t cnt = (uint32_t)__popc(keep);
        // exclusive prefix sum over the workgroup
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t base = running;
        for (int k = 0; k < wave; k++) base += s_wave[k];
        const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        uint32_t o = base + inc - cnt;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (keep & (1u << j)) dst[o++] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
        running += total;
        __syncthreads();
        if (cut != 0xFFFFFFFFu) break;
    }
    for (uint32_t i = running + (uint32_t)tid; i < place; i += 256u) dst[i] = 0;
    if (tid == 0) ulenBits[f] = running * 8u;
    // records: the guess for subsequence i + 1 is (its first bit, start of an MCU)
    const uint32_t nSub = (len * 8u + kParBits - 1u) / kParBits;  // (an upper bound of what the unstuffed stream needs)
    for (uint32_t i = (uint32_t)tid; i < nSub; i += 256u) {
        exitRec[subBase[f] + i] = (unsigned long long)((i + 1u) * kParBits);
        lastEntry[subBase[f] + i] = ~0ull;
    }
}

// wg*: per workgroup its frame and its first subsequence.  frameSet[f] = the frame's table set.  changed[0] is set when
// a record changes, changed[1 + f] to `round` when one of frame f does.  skip[f] != 0: the frame takes the sequential
// kernels.
template <bool WRITE>
__global__ __launch_bounds__(256) void k_jpeg_par(const uint8_t* __restrict__ ubytes, const JpegDesc* __restrict__ desc,
                                                  const int32_t* __restrict__ wgFrame, const uint32_t* __restrict__ wgFirst,
                                                  const uint8_t* __restrict__ tableSets, const int32_t* __restrict__ frameSet,
                                                  const uint32_t* __restrict__ ulenBits, const uint32_t* __restrict__ subBase,
                                                  unsigned long long* __restrict__ exitRec, unsigned long long* __restrict__ lastEntry,
                                                  int32_t* __restrict__ changed, int round, const int32_t* __restrict__ skip,
                                                  int16_t* __restrict__ coef, int16_t* __restrict__ dcs, uint32_t frameBlocks,
                                                  int32_t* __restrict__ status) {
    __shared__ __align__(16) uint8_t s_tab[kHuffSetTables * kHuffTableBytes];
    __shared__ uint32_t s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = wgFrame[blockIdx.x];
    if (skip[f]) return;
    {
        const uint4* src = reinterpret_cast<const uint4*>(tableSets + (size_t)frameSet[f] * kHuffSetTables * kHuffTableBytes);
        uint4* dst = reinterpret_cast<uint4*>(s_tab);
        for (int i = tid; i < kHuffSetTables * kHuffTableBytes / 16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const JpegDesc d = desc[f];
    const uint32_t first = wgFirst[blockIdx.x], i = first + (uint32_t)tid;
    const uint32_t nSub = (ulenBits[f] + kParBits - 1u) / kParBits;
    const bool valid = i < nSub;
    unsigned long long* const rec = exitRec + subBase[f];
    const unsigned long long entry = !valid ? 0ull : i == 0u ? 0ull : __hip_atomic_load(&rec[i - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kParStateMask;
    if (!WRITE) {
        if (!valid || entry == lastEntry[subBase[f] + i]) return;
    }
    const int hv = d.hmax * d.vmax, nbMcu = d.nc == 1 ? 1 : hv + 2;
    const uint32_t totalBlocks = (uint32_t)d.mcux * d.mcuy * (uint32_t)nbMcu;
    const uint32_t bw0 = (uint32_t)d.mcux * d.hmax, n0 = bw0 * ((uint32_t)d.mcuy * d.vmax), n1 = (uint32_t)d.mcux * d.mcuy;
    // ---- WRITE: the number of the block this subsequence starts in = blocks completed in all the ones before it ----
    uint32_t q = 0;
    if (WRITE) {
        uint32_t part = 0;
        for (uint32_t t = (uint32_t)tid; t < first; t += 256u) part += (uint32_t)(rec[t] >> 48);
        const uint32_t mineN = valid ? (uint32_t)(rec[i] >> 48) : 0u;
        uint32_t inc = mineN;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
            if (lane >= o) inc += t;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) part += (uint32_t)__shfl_xor((int)part, o);
        if (lane == 63) s_red[wave] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (int k = 0; k < wave; k++) before += s_red[k];
        __syncthreads();
        if (lane == 0) s_red[wave] = part;
        __syncthreads();
        q = s_red[0] + s_red[1] + s_red[2] + s_red[3] + before + inc - mineN;
        if (i == 0u && nSub == 0u) status[f] = 1;  // no entropy-coded data at all
        if (!valid) return;
    }
    // ---- the decoder's state at the entry ----
    const uint32_t p0 = (uint32_t)entry;
    int j = (int)((entry >> 32) & 255u), k = (int)((entry >> 40) & 255u);
    uint32_t nDone = 0;
    bool bad = false;
    if (WRITE && (q >= totalBlocks || (int)(q % (uint32_t)nbMcu) != j)) {
        // beyond the image (padding of the stream's last bytes), or records that do not add up
        if (q < totalBlocks) status[f] = 1;
        return;
    }
    const uint32_t endBit = (i + 1u) * kParBits;
    const uint32_t* const words = reinterpret_cast<const uint32_t*>(ubytes + d.dataOff);
    uint32_t ptr = p0 >> 5;
    uint64_t acc = (uint64_t)__builtin_bswap32(words[ptr]) << (32u + (p0 & 31u));
    int nb = 32 - (int)(p0 & 31u);
    ptr++;
    int16_t* const fcoef = coef + (size_t)f * frameBlocks * 64;
    int16_t* const fdc = dcs + (size_t)f * frameBlocks;
    // where block number q is stored (WRITE), and its place among its component's blocks in coding order (+ the
    // component's first block): the DC differences go to a compact array in that order (k_jpeg_dc)
    uint32_t dcAt = 0;
    auto storage = [&](uint32_t qq, int jj) -> uint32_t {
        const uint32_t m = qq / (uint32_t)nbMcu, my = m / d.mcux, mx = m - my * d.mcux;
        if (jj < hv) {
            const uint32_t by = (uint32_t)jj / d.hmax, bx = (uint32_t)jj - by * d.hmax;
            dcAt = m * (uint32_t)hv + (uint32_t)jj;
            return ((my * d.vmax + by) * bw0 + mx * d.hmax + bx) * 64u;
        }
        dcAt = (jj == hv ? n0 : n0 + n1) + m;
        return ((jj == hv ? n0 : n0 + n1) + m) * 64u;
    };
    uint32_t block = WRITE ? storage(q, j) : 0u;
    while ((uint32_t)(ptr * 32u - (uint32_t)nb) < endBit) {
        if (nb <= 32) {
            acc |= (uint64_t)__builtin_bswap32(words[ptr]) << (32 - nb);
            ptr++;
            nb += 32;
        }
        const int c = j < hv ? 0 : j - hv + 1;
        const int slot = k == 0 ? (c == 0 ? d.td[0] : c == 1 ? d.td[1] : d.td[2]) : 3 + (c == 0 ? d.ta[0] : c == 1 ? d.ta[1] : d.ta[2]);
        const uint8_t* const T = s_tab + slot * kHuffTableBytes;
        const uint32_t e = reinterpret_cast<const uint16_t*>(T)[(uint32_t)(acc >> 56)];
        int l = (int)(e >> 8), sym = (int)(e & 255u);
        if (!e) {
            const uint32_t* lim = reinterpret_cast<const uint32_t*>(T + kOffLimit);
            const uint32_t code16 = (uint32_t)(acc >> 48);
            l = 9;
#pragma unroll
            for (int t = 9; t <= 15; t++) l += code16 >= lim[t] ? 1 : 0;
            if (code16 >= lim[16]) bad = true;
            const int32_t* valoff = reinterpret_cast<const int32_t*>(T + kOffValoff);
            sym = T[kOffHuffval + (((int32_t)(code16 >> (16 - l)) + valoff[l]) & 255)];
        }
        acc <<= l;
        nb -= l;
        if (k == 0 && sym > 16) bad = true;
        const int sz = k == 0 ? min(sym, 16) : (sym & 15);
        int val = 0;
        if (sz) {
            const int r = (int)(acc >> (64 - sz));
            acc <<= sz;
            nb -= sz;
            val = r < (1 << (sz - 1)) ? r - (1 << sz) + 1 : r;
        }
        if (k == 0) {
            if (WRITE) fdc[dcAt] = (int16_t)val;  // the difference to the component's previous block (k_jpeg_dc)
            k = 1;
        } else {
            const int run = sym >> 4;
            if (sz) {
                k += run;
                if (k > 63) {
                    bad = true;
                    k = 64;
                } else {
                    if (WRITE) fcoef[block + (uint32_t)k] = (int16_t)val;
                    k++;
                }
            } else {
                k = run == 15 ? k + 16 : 64;
            }
        }
        if (k >= 64) {
            k = 0;
            nDone++;
            j = j + 1 == nbMcu ? 0 : j + 1;
            if (WRITE) {
                q++;
                if (q >= totalBlocks) break;
                block = storage(q, j);
            }
        }
    }
    if (WRITE) {
        // (codes that do not exist and the like are the speculative decoders' daily bread; on the true path they mean a
        // damaged frame, and so does a stream that ends before the image does)
        if (bad || (i == nSub - 1u && q < totalBlocks)) status[f] = 1;
    } else {
        const unsigned long long out = (unsigned long long)(ptr * 32u - (uint32_t)nb) | ((unsigned long long)(uint32_t)j << 32) |
                                       ((unsigned long long)(uint32_t)k << 40) | ((unsigned long long)min(nDone, 0xFFFFu) << 48);
        lastEntry[subBase[f] + i] = entry;
        if (out != rec[i]) {
            __hip_atomic_store(&rec[i], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            changed[0] = 1;
            changed[1 + f] = round;
        }
    }
}

// DC coefficients from their differences: one workgroup per (component, frame) sums the component's differences in
// the order they are coded (dcs: [frame][frameBlocks], a component's blocks together, component 0 first).  grid (3, frames in the list).
__global__ __launch_bounds__(256) void k_jpeg_dc(int16_t* __restrict__ dcs, const JpegDesc* __restrict__ desc,
                                                 const int32_t* __restrict__ list, const int32_t* __restrict__ skip, uint32_t frameBlocks) {
    __shared__ int32_t s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = list[blockIdx.y], c = (int)blockIdx.x;
    const JpegDesc d = desc[f];
    if (skip[f] || c >= d.nc) return;
    const uint32_t n0 = (uint32_t)d.mcux * d.hmax * ((uint32_t)d.mcuy * d.vmax), n1 = (uint32_t)d.mcux * d.mcuy;
    const uint32_t nblk = c == 0 ? n0 : n1;
    int16_t* const v = dcs + (size_t)f * frameBlocks + (c == 0 ? 0u : c == 1 ? n0 : n0 + n1);
    // tiles of 2048 values, eight consecutive ones a lane (a wavefront reads a kilobyte in one piece)
    int32_t carry = 0;  // uniform: the sum of the tiles before
    for (uint32_t base = 0; base < nblk; base += 2048u) {
        const uint32_t at = base + 8u * (uint32_t)tid;
        int32_t p[8];
        int32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            sum += at + (uint32_t)k < nblk ? (int32_t)v[at + k] : 0;
            p[k] = sum;
        }
        int32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int32_t off = carry + inc - sum;
        for (int k = 0; k < wave; k++) off += s_wave[k];
        carry += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (at + (uint32_t)k < nblk) v[at + k] = (int16_t)(off + p[k]);
        __syncthreads();
    }
}

// ---- dequantisation + inverse DCT -----------------------------------------------------------------------------------
// jidctint.c, jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2; one pass over eight values, arithmetic modulo 2^32
__device__ __forceinline__ int32_t jmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__device__ __forceinline__ int32_t jdescale(int32_t x, int n) { return (int32_t)((uint32_t)x + (1u << (n - 1))) >> n; }
template <int SHIFT>
__device__ __forceinline__ void idct_1d(const int32_t* in, int32_t* out) {
    constexpr int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
                      F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
    uint32_t z2 = (uint32_t)in[2], z3 = (uint32_t)in[6];
    uint32_t z1 = (uint32_t)jmul((int32_t)(z2 + z3), F_0_541);
    uint32_t tmp2 = z1 + (uint32_t)jmul((int32_t)z3, -F_1_847);
    uint32_t tmp3 = z1 + (uint32_t)jmul((int32_t)z2, F_0_765);
    uint32_t tmp0 = ((uint32_t)in[0] + (uint32_t)in[4]) << 13;
    uint32_t tmp1 = ((uint32_t)in[0] - (uint32_t)in[4]) << 13;
    const uint32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = (uint32_t)in[7]; tmp1 = (uint32_t)in[5]; tmp2 = (uint32_t)in[3]; tmp3 = (uint32_t)in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    uint32_t z4 = tmp1 + tmp3;
    const uint32_t z5 = (uint32_t)jmul((int32_t)(z3 + z4), F_1_175);
    tmp0 = (uint32_t)jmul((int32_t)tmp0, F_0_298); tmp1 = (uint32_t)jmul((int32_t)tmp1, F_2_053);
    tmp2 = (uint32_t)jmul((int32_t)tmp2, F_3_072); tmp3 = (uint32_t)jmul((int32_t)tmp3, F_1_501);
    z1 = (uint32_t)jmul((int32_t)z1, -F_0_899); z2 = (uint32_t)jmul((int32_t)z2, -F_2_562);
    z3 = (uint32_t)jmul((int32_t)z3, -F_1_961); z4 = (uint32_t)jmul((int32_t)z4, -F_0_390);
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = jdescale((int32_t)(tmp10 + tmp3), SHIFT); out[7] = jdescale((int32_t)(tmp10 - tmp3), SHIFT);
    out[1] = jdescale((int32_t)(tmp11 + tmp2), SHIFT); out[6] = jdescale((int32_t)(tmp11 - tmp2), SHIFT);
    out[2] = jdescale((int32_t)(tmp12 + tmp1), SHIFT); out[5] = jdescale((int32_t)(tmp12 - tmp1), SHIFT);
    out[3] = jdescale((int32_t)(tmp13 + tmp0), SHIFT); out[4] = jdescale((int32_t)(tmp13 - tmp0), SHIFT);
}

// range_limit[x & RANGE_MASK] of libjpeg's table, samples centred on 128
__device__ __forceinline__ uint32_t range_limit_idct(int32_t x) {
    const int y = x & 1023;
    return (uint32_t)(y < 128 ? y + 128 : y < 512 ? 255 : y < 896 ? 0 : y - 896);
}

constexpr int kIdctBlocksPerWg = 32;
// grid (ceil(frameBlocks / 32), frames).  A frame's planes: component 0 (bw0 * 8 bytes a row), then 1, then 2.
__global__ __launch_bounds__(256) void k_jpeg_idct(int16_t* __restrict__ coef, const JpegDesc* __restrict__ desc,
                                                   const uint16_t* __restrict__ quant, uint8_t* __restrict__ planes,
                                                   uint32_t frameBlocks, const int32_t* __restrict__ frameOf,
                                                   const int16_t* __restrict__ dcs, const int32_t* __restrict__ skip) {
    __shared__ int32_t s_ws[kIdctBlocksPerWg][8 * 9];
    const int f = frameOf ? frameOf[blockIdx.y] : (int)blockIdx.y;
    const JpegDesc d = desc[f];
    const uint32_t bw0 = (uint32_t)d.mcux * d.hmax, n0 = bw0 * ((uint32_t)d.mcuy * d.vmax);
    const uint32_t n1 = d.nc == 3 ? (uint32_t)d.mcux * d.mcuy : 0u;
    const int g = threadIdx.x >> 3, r = threadIdx.x & 7;
    const uint32_t slot = blockIdx.x * kIdctBlocksPerWg + (uint32_t)g;
    if (slot >= n0 + 2u * n1) return;
    const int c = slot < n0 ? 0 : slot < n0 + n1 ? 1 : 2;
    const uint32_t idx = slot - (c == 0 ? 0u : c == 1 ? n0 : n0 + n1);
    const uint32_t bw = c == 0 ? bw0 : (uint32_t)d.mcux;
    const uint32_t brow = idx / bw, bcol = idx - brow * bw;
    uint4* const cp = reinterpret_cast<uint4*>(coef + ((size_t)f * frameBlocks + slot) * 64 + r * 8);
    const uint4 cv = *cp;
    // (the coefficient planes are all zero between calls: the entropy decoders store only what is not; most of a
    // block's high-frequency pieces are zero already and need no store)
    if (cv.x | cv.y | cv.z | cv.w) *cp = make_uint4(0, 0, 0, 0);
    const uint4 qv = *reinterpret_cast<const uint4*>(quant + ((size_t)f * 4 + (c == 0 ? d.tq[0] : c == 1 ? d.tq[1] : d.tq[2])) * 64 + r * 8);
    uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
    const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
    if (r == 0 && dcs && !skip[f]) {
        // the parallel entropy decoder keeps the DC coefficients apart, a component's blocks in the order they are coded
        const uint32_t hc = c == 0 ? d.hmax : 1u, vc = c == 0 ? d.vmax : 1u;
        const uint32_t m = (brow / vc) * d.mcux + bcol / hc, at = m * hc * vc + (brow % vc) * hc + bcol % hc;
        cw[0] = (cw[0] & 0xFFFF0000u) | (uint16_t)dcs[(size_t)f * frameBlocks + (c == 0 ? 0u : c == 1 ? n0 : n0 + n1) + at];
    }
    int32_t* const ws = s_ws[g];
    // this lane holds the coefficients coded 8 r-th .. 8 r + 7-th and their quantisation steps (both in coding order)
    const uint2 zz = *reinterpret_cast<const uint2*>(&c_zigzag9[r * 8]);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t z = i < 2 ? zz.x >> (16 * i) : zz.y >> (16 * (i - 2));
        ws[z & 255u] = jmul((int32_t)(int16_t)(cw[i] & 0xFFFFu), (int32_t)(qw[i] & 0xFFFFu));
        ws[(z >> 8) & 255u] = jmul((int32_t)(int16_t)(cw[i] >> 16), (int32_t)(qw[i] >> 16));
    }
    wave_lds_order();
    int32_t in[8], out[8];
#pragma unroll
    for (int i = 0; i < 8; i++) in[i] = ws[i * 9 + r];  // column r
    idct_1d<13 - 2>(in, out);
    wave_lds_order();
#pragma unroll
    for (int i = 0; i < 8; i++) ws[i * 9 + r] = out[i];
    wave_lds_order();
#pragma unroll
    for (int i = 0; i < 8; i++) in[i] = ws[r * 9 + i];  // row r
    idct_1d<13 + 2 + 3>(in, out);
    uint2 o;
    o.x = range_limit_idct(out[0]) | (range_limit_idct(out[1]) << 8) | (range_limit_idct(out[2]) << 16) | (range_limit_idct(out[3]) << 24);
    o.y = range_limit_idct(out[4]) | (range_limit_idct(out[5]) << 8) | (range_limit_idct(out[6]) << 16) | (range_limit_idct(out[7]) << 24);
    uint8_t* const plane = planes + (size_t)f * frameBlocks * 64 + (size_t)(c == 0 ? 0u : c == 1 ? n0 : n0 + n1) * 64;
    *reinterpret_cast<uint2*>(plane + ((size_t)brow * 8 + r) * (bw * 8) + (size_t)bcol * 8) = o;
}

// ---- upsampling + colour conversion ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t clamp8(int v) { return (uint32_t)min(max(v, 0), 255); }

// jdsample.c h2v1_fancy_upsample / h2v2_fancy_upsample as functions of the output position (the special cases at the
// first and last column, and the context rows of jdmainct.c at the first and last row, are the general formulas with
// the neighbour clamped into the plane); cw x ch = the component's real (downsampled) size
__device__ __forceinline__ int chroma_at(const uint8_t* __restrict__ C, int stride, int cw, int ch, int hmax, int vmax, int x, int y) {
    if (hmax == 1) return C[(size_t)y * stride + x];
    const int i = x >> 1, in = (x & 1) ? min(i + 1, cw - 1) : max(i - 1, 0);
    if (vmax == 1) {
        const uint8_t* row = C + (size_t)y * stride;
        return (3 * row[i] + row[in] + ((x & 1) ? 2 : 1)) >> 2;
    }
    const int j = y >> 1, jn = (y & 1) ? min(j + 1, ch - 1) : max(j - 1, 0);
    const uint8_t* r0 = C + (size_t)j * stride;
    const uint8_t* r1 = C + (size_t)jn * stride;
    const int cur = 3 * r0[i] + r1[i], nbr = 3 * r0[in] + r1[in];
    return (3 * cur + nbr + ((x & 1) ? 7 : 8)) >> 4;
}

// Eight upsampled chroma values for the pixels x0 .. x0 + 7 of row y (x0 a multiple of 8): the same formulas as
// chroma_at, with the row's samples fetched as one aligned word and the two bytes next to it instead of byte by byte.
__device__ __forceinline__ void chroma8(const uint8_t* __restrict__ C, int stride, int cw, int ch, int hmax, int vmax, int x0, int y, int out[8]) {
    if (hmax == 1) {
        const uint2 v = *reinterpret_cast<const uint2*>(C + (size_t)y * stride + x0);  // (the plane is padded to whole blocks)
#pragma unroll
        for (int t = 0; t < 8; t++) out[t] = (int)(((t < 4 ? v.x : v.y) >> (8 * (t & 3))) & 255u);
        return;
    }
    const int i0 = x0 >> 1, last = cw - 1 - i0;  // (>= 0 for a lane that has pixels)
    // s[k + 1] = sample i0 + k, k = -1 .. 4, indices clamped into the row
    auto samples = [&](const uint8_t* row, int s[6]) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(row + i0);
        s[0] = row[max(i0 - 1, 0)];
#pragma unroll
        for (int k = 0; k < 4; k++) s[k + 1] = (int)((w >> (8 * min(k, last))) & 255u);
        s[5] = row[min(i0 + 4, cw - 1)];
    };
    int s[6];
    if (vmax == 1) {
        samples(C + (size_t)y * stride, s);
#pragma unroll
        for (int t = 0; t < 8; t++) out[t] = (3 * s[(t >> 1) + 1] + s[(t >> 1) + ((t & 1) ? 2 : 0)] + ((t & 1) ? 2 : 1)) >> 2;
        return;
    }
    const int j = y >> 1, jn = (y & 1) ? min(j + 1, ch - 1) : max(j - 1, 0);
    int s1[6];
    samples(C + (size_t)j * stride, s);
    samples(C + (size_t)jn * stride, s1);
#pragma unroll
    for (int k = 0; k < 6; k++) s[k] = 3 * s[k] + s1[k];
#pragma unroll
    for (int t = 0; t < 8; t++) out[t] = (3 * s[(t >> 1) + 1] + s[(t >> 1) + ((t & 1) ? 2 : 0)] + ((t & 1) ? 7 : 8)) >> 4;
}

// lane = eight pixels of one row; grid (ceil(H * ceil(W / 8) / 256), frames)
__global__ __launch_bounds__(256) void k_jpeg_color(const uint8_t* __restrict__ planes, const JpegDesc* __restrict__ desc,
                                                    uint32_t frameBlocks, int H, int W, uint8_t* __restrict__ out,
                                                    const int32_t* __restrict__ frameOf) {
    const int f = frameOf ? frameOf[blockIdx.y] : (int)blockIdx.y;
    const int W8 = (W + 7) >> 3;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= (uint32_t)(H * W8)) return;
    const int y = (int)(t / (uint32_t)W8), x0 = (int)(t - (uint32_t)y * W8) * 8;
    const JpegDesc d = desc[f];
    const uint32_t bw0 = (uint32_t)d.mcux * d.hmax, n0 = bw0 * ((uint32_t)d.mcuy * d.vmax), n1 = (uint32_t)d.mcux * d.mcuy;
    const uint8_t* const Y = planes + (size_t)f * frameBlocks * 64;
    const int ys = (int)bw0 * 8;
    uint32_t px[8];
    const uint2 yw = *reinterpret_cast<const uint2*>(Y + (size_t)y * ys + x0);  // (the plane is padded to whole blocks)
    if (d.nc == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) px[i] = (((i < 4 ? yw.x : yw.y) >> (8 * (i & 3))) & 255u) * 0x010101u;
    } else {
        const uint8_t* const Cb = Y + (size_t)n0 * 64;
        const uint8_t* const Cr = Cb + (size_t)n1 * 64;
        const int cs = (int)d.mcux * 8, cw = (W + d.hmax - 1) / d.hmax, ch = (H + d.vmax - 1) / d.vmax;
        int cbv[8], crv[8];
        chroma8(Cb, cs, cw, ch, d.hmax, d.vmax, x0, y, cbv);
        chroma8(Cr, cs, cw, ch, d.hmax, d.vmax, x0, y, crv);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int yy = (int)(((i < 4 ? yw.x : yw.y) >> (8 * (i & 3))) & 255u);
            const int cb = cbv[i] - 128, cr = crv[i] - 128;
            // jdcolor.c: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
            const uint32_t R = clamp8(yy + ((91881 * cr + 32768) >> 16));
            const uint32_t G = clamp8(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
            const uint32_t B = clamp8(yy + ((116130 * cb + 32768) >> 16));
            px[i] = B | (G << 8) | (R << 16);
        }
    }
    const size_t pix = (size_t)f * H * W + (size_t)y * W + x0;
    uint8_t* const o = out + pix * 3;
    if (x0 + 7 < W && (reinterpret_cast<uintptr_t>(o) & 7u) == 0) {  // (24 bytes in three 8-byte stores)
        uint2* o64 = reinterpret_cast<uint2*>(o);
        o64[0] = make_uint2(px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16));
        o64[1] = make_uint2((px[2] >> 16) | (px[3] << 8), px[4] | (px[5] << 24));
        o64[2] = make_uint2((px[5] >> 8) | (px[6] << 16), (px[6] >> 16) | (px[7] << 8));
    } else {
        for (int i = 0; i < 8 && x0 + i < W; i++) {
            o[3 * i] = (uint8_t)px[i];
            o[3 * i + 1] = (uint8_t)(px[i] >> 8);
            o[3 * i + 2] = (uint8_t)(px[i] >> 16);
        }
    }
}

// ---- host: headers, tables, buffers -----------------------------------------------------------------------------------

// ITU T.81 Annex K.3: the tables a frame without DHT segments means (MJPEG)
const uint8_t kStdDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kStdDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kStdDcVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kStdAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kStdAcLumVal[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kStdAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kStdAcChrVal[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct HuffSpec {  // a DHT table as the file gives it
    bool set = false;
    uint8_t bits[16];
    uint8_t vals[256];
    int total = 0;
    void assign(const uint8_t* b, const uint8_t* v, int n) {
        memcpy(bits, b, 16);
        memcpy(vals, v, (size_t)n);
        total = n;
        set = true;
    }
    std::string key() const { return std::string(reinterpret_cast<const char*>(bits), 16) + std::string(reinterpret_cast<const char*>(vals), (size_t)total); }
};

// the device form of a table (kHuffTableBytes); false if the code lengths are not a prefix code
bool build_table(const HuffSpec& h, uint8_t* out) {
    memset(out, 0, kHuffTableBytes);
    uint16_t* look = reinterpret_cast<uint16_t*>(out);
    uint32_t* limit = reinterpret_cast<uint32_t*>(out + kOffLimit);  // [l]: first 16-bit left-aligned code that is longer than l bits
    int32_t* valoff = reinterpret_cast<int32_t*>(out + kOffValoff);
    memcpy(out + kOffHuffval, h.vals, (size_t)h.total);
    int code = 0, p = 0;
    limit[0] = 0;
    for (int l = 1; l <= 16; l++) {
        const int n = h.bits[l - 1];
        if (n) {
            valoff[l] = p - code;
            if (l <= 8)
                for (int i = 0; i < n; i++)
                    for (int j = 0; j < (1 << (8 - l)); j++) look[((code + i) << (8 - l)) + j] = (uint16_t)((l << 8) | h.vals[p + i]);
            p += n;
            code += n;
        }
        if (code > (1 << l)) return false;
        limit[l] = (uint32_t)code << (16 - l);
        code <<= 1;
    }
    return true;
}

struct ParsedFrame {
    JpegDesc d{};
    uint16_t quant[4][64];  // in coding (zigzag) order, as the file has them
    size_t scanOff = 0;  // first entropy-coded byte in the file
    HuffSpec dc[3], ac[3];  // by slot
};

// 0, or an error text
const char* parse_frame(const uint8_t* data, size_t n, int H, int W, ParsedFrame& out) {
    if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return "not a JPEG (no SOI marker)";
    bool qset[4] = {false, false, false, false};
    HuffSpec dcT[4], acT[4];
    bool anyDht = false;
    int nc = 0, compId[3] = {0, 0, 0}, compH[3] = {1, 1, 1}, compV[3] = {1, 1, 1}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    int h = 0, w = 0, restart = 0;
    size_t i = 2, scan = 0;
    memset(out.quant, 0, sizeof out.quant);
    while (!scan) {
        if (i + 4 > n || data[i] != 0xFF) return "truncated or damaged headers";
        const int m = data[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { i += 2; continue; }
        const size_t L = ((size_t)data[i + 2] << 8) | data[i + 3];
        if (L < 2 || i + 2 + L > n) return "truncated or damaged headers";
        const uint8_t* s = data + i + 4;
        const size_t len = L - 2;
        if (m == 0xDB) {
            size_t k = 0;
            while (k < len) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                if (tq > 3 || pq > 1 || k + 1 + (pq ? 128u : 64u) > len) return "bad quantisation table";
                for (int j = 0; j < 64; j++)
                    out.quant[tq][j] = pq ? (uint16_t)((s[k + 1 + 2 * j] << 8) | s[k + 2 + 2 * j]) : s[k + 1 + j];
                qset[tq] = true;
                k += 1 + (pq ? 128 : 64);
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (len < 6 || s[0] != 8) return "only 8-bit samples are supported";
            h = (s[1] << 8) | s[2];
            w = (s[3] << 8) | s[4];
            nc = s[5];
            if ((nc != 1 && nc != 3) || len < 6 + 3 * (size_t)nc) return "only 1 or 3 components (Y / YCbCr) are supported";
            for (int c = 0; c < nc; c++) {
                compId[c] = s[6 + 3 * c];
                compH[c] = s[7 + 3 * c] >> 4;
                compV[c] = s[7 + 3 * c] & 15;
                out.d.tq[c] = s[8 + 3 * c];
                if (out.d.tq[c] > 3) return "bad quantisation table selector";
            }
        } else if (m == 0xC4) {
            size_t k = 0;
            while (k < len) {
                if (k + 17 > len) return "bad Huffman table";
                const int tc = s[k] >> 4, th = s[k] & 15;
                int total = 0;
                for (int l = 0; l < 16; l++) total += s[k + 1 + l];
                if (tc > 1 || th > 3 || total > 256 || k + 17 + (size_t)total > len) return "bad Huffman table";
                (tc ? acT[th] : dcT[th]).assign(s + k + 1, s + k + 17, total);
                anyDht = true;
                k += 17 + (size_t)total;
            }
        } else if (m == 0xDD) {
            if (len < 2) return "bad restart interval";
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!nc || len < 1 || s[0] != nc || len < 1 + 2 * (size_t)nc + 3) return "only one interleaved scan with every component is supported";
            for (int c = 0; c < nc; c++) {
                if (s[1 + 2 * c] != compId[c]) return "only one interleaved scan with every component is supported";
                td[c] = s[2 + 2 * c] >> 4;
                ta[c] = s[2 + 2 * c] & 15;
                if (td[c] > 3 || ta[c] > 3) return "bad Huffman table selector";
            }
            if (s[1 + 2 * nc] != 0 || s[2 + 2 * nc] != 63 || s[3 + 2 * nc] != 0) return "not a sequential scan";
            scan = i + 2 + L;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC8)) {
            return "progressive, lossless and arithmetic-coded JPEG are not supported (baseline / MJPEG frames only)";
        }
        i += 2 + L;
    }
    if (h != H || w != W) return "frame size differs from the batch's";
    if (!anyDht) {
        dcT[0].assign(kStdDcLumBits, kStdDcVal, 12);
        dcT[1].assign(kStdDcChrBits, kStdDcVal, 12);
        acT[0].assign(kStdAcLumBits, kStdAcLumVal, 162);
        acT[1].assign(kStdAcChrBits, kStdAcChrVal, 162);
    }
    if (nc == 1) compH[0] = compV[0] = 1;
    const int hmax = compH[0], vmax = compV[0];
    if (nc == 3) {
        if (compH[1] != 1 || compV[1] != 1 || compH[2] != 1 || compV[2] != 1 ||
            !((hmax == 1 && vmax == 1) || (hmax == 2 && vmax == 1) || (hmax == 2 && vmax == 2)))
            return "chroma sampling other than 4:4:4, 4:2:2 and 4:2:0 is not supported";
    }
    // the tables the components use, as slots of the frame's table set
    int dcSlotOf[4] = {-1, -1, -1, -1}, acSlotOf[4] = {-1, -1, -1, -1}, nd = 0, na = 0;
    for (int c = 0; c < nc; c++) {
        if (!qset[out.d.tq[c]] || !dcT[td[c]].set || !acT[ta[c]].set) return "a table the scan refers to is missing";
        if (dcSlotOf[td[c]] < 0) { dcSlotOf[td[c]] = nd; out.dc[nd++] = dcT[td[c]]; }
        if (acSlotOf[ta[c]] < 0) { acSlotOf[ta[c]] = na; out.ac[na++] = acT[ta[c]]; }
        out.d.td[c] = (uint8_t)dcSlotOf[td[c]];
        out.d.ta[c] = (uint8_t)acSlotOf[ta[c]];
    }
    out.d.nc = (uint8_t)nc;
    out.d.hmax = (uint8_t)hmax;
    out.d.vmax = (uint8_t)vmax;
    out.d.mcux = (uint16_t)((W + 8 * hmax - 1) / (8 * hmax));
    out.d.mcuy = (uint16_t)((H + 8 * vmax - 1) / (8 * vmax));
    out.d.restart = (uint16_t)restart;
    out.scanOff = scan;
    return nullptr;
}

template <class T>
bool regrow_dev(T*& p, size_t& cap, size_t need) {
    if (need <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t n = need + need / 4;
    if (hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    cap = n;
    return true;
}

}  // namespace

struct JpegState {
    uint8_t* d_bytes = nullptr; size_t bytesCap = 0;
    uint8_t* h_bytes = nullptr; size_t hBytesCap = 0;  // page-locked
    JpegDesc* d_desc = nullptr; size_t descCap = 0;
    uint16_t* d_quant = nullptr; size_t quantCap = 0;
    uint8_t* d_tables = nullptr; size_t tablesCap = 0;
    int32_t* d_groups = nullptr; size_t groupsCap = 0;
    int32_t* d_status = nullptr; size_t statusCap = 0;
    int16_t* d_coef = nullptr; size_t coefCap = 0;   // [frames][frameBlocks][64], all zero between calls
    uint8_t* d_planes = nullptr; size_t planesCap = 0;
    uint8_t* d_out = nullptr; size_t outCap = 0;      // BGR frames when the caller's output is on the host
    // the parallel entropy decoder (k_jpeg_unstuff / k_jpeg_par / k_jpeg_dc)
    uint8_t* d_ubytes = nullptr; size_t ubytesCap = 0;            // the frames' streams without byte stuffing
    unsigned long long* d_rec = nullptr; size_t recCap = 0;      // a record per subsequence
    unsigned long long* d_last = nullptr; size_t lastCap = 0;    // the entry state a record was computed from
    int32_t* d_parMeta = nullptr; size_t parMetaCap = 0;         // per frame / per workgroup index arrays, packed
    int32_t* d_changed = nullptr; size_t changedCap = 0;         // [0] any record changed, [1 + f] the last round frame f's did
    int16_t* d_dcs = nullptr; size_t dcsCap = 0;                  // DC coefficients [frames][frameBlocks], in coding order
    hipStream_t copySt = nullptr;                                 // the compressed bytes of the next pass travel on this stream
    hipEvent_t evUp[2] = {};                                      // ... and these say when they have arrived
    bool coefDirty = false;
};

JpegState* jpeg_create() { return new JpegState(); }

void jpeg_destroy(JpegState* s) {
    if (!s) return;
    void* ptrs[] = {s->d_bytes, s->d_desc, s->d_quant, s->d_tables, s->d_groups, s->d_status, s->d_coef, s->d_planes, s->d_out,
                    s->d_ubytes, s->d_rec, s->d_last, s->d_parMeta, s->d_changed, s->d_dcs};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (s->h_bytes) (void)hipHostFree(s->h_bytes);
    for (hipEvent_t e : s->evUp)
        if (e) (void)hipEventDestroy(e);
    if (s->copySt) (void)hipStreamDestroy(s->copySt);
    delete s;
}

int jpeg_info(const uint8_t* data, size_t n, int32_t* height, int32_t* width, int32_t* components) {
    if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return CAMSYNC_E_ARG;
    size_t i = 2;
    while (i + 4 <= n) {
        if (data[i] != 0xFF) return CAMSYNC_E_ARG;
        const int m = data[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { i += 2; continue; }
        const size_t L = ((size_t)data[i + 2] << 8) | data[i + 3];
        if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            if (i + 2 + L > n || L < 8) return CAMSYNC_E_ARG;
            if (height) *height = (data[i + 5] << 8) | data[i + 6];
            if (width) *width = (data[i + 7] << 8) | data[i + 8];
            if (components) *components = data[i + 9];
            return CAMSYNC_OK;
        }
        if (m == 0xDA) return CAMSYNC_E_ARG;
        i += 2 + L;
    }
    return CAMSYNC_E_ARG;
}

// B frames of H x W -> out [B,H,W,3] (device pointer if outOnDevice, else host).  Enqueues on st and waits for it.
int jpeg_decode_batch(JpegState* s, hipStream_t st, const uint8_t* const* jpegs, const size_t* sizes, int B, int H, int W,
                      uint8_t* out, int outOnDevice, std::string& err) {
    auto hipfail = [&](const char* what, hipError_t e) {
        err = std::string(what) + ": " + hipGetErrorString(e);
        return CAMSYNC_E_HIP;
    };
    const uint32_t maxBW = 2u * (uint32_t)((W + 15) / 16), maxBH = 2u * (uint32_t)((H + 15) / 16);
    const size_t frameCoefBytes = (size_t)3u * maxBW * maxBH * 128;  // (4:4:4, the largest of the supported samplings)
    // frames per pass: the coefficient planes of a pass stay under 48 GB; a large batch is cut into four passes so that
    // the compressed bytes of the next pass are packed and sent while the kernels of this one run
    const size_t fit = std::max<size_t>(64, ((size_t)48 << 30) / frameCoefBytes);
    const int chunk = (int)std::min<size_t>(fit, B >= 256 ? (size_t)(B + 3) / 4 : (size_t)B);
    const int nPass = (B + chunk - 1) / chunk;
    std::vector<ParsedFrame> pfAll((size_t)B), pf;
    std::vector<JpegDesc> descs;
    std::vector<uint16_t> quant;
    std::vector<int32_t> groups, status;
    static const bool dbg = getenv("CAMSYNC_DEBUG_JPEG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // ---- headers of all frames; where each frame's entropy-coded bytes lie in its pass's buffer ----
    const auto tParse = now();
    std::vector<size_t> passBytes((size_t)nPass, 0);
    size_t regionBytes = 0;
    for (int k = 0; k < nPass; k++) {
        const int b0 = k * chunk, nb = std::min(chunk, B - b0);
        size_t totalBytes = 0;
        for (int i = 0; i < nb; i++) {
            ParsedFrame& f = pfAll[(size_t)(b0 + i)];
            const char* e = parse_frame(jpegs[b0 + i], sizes[b0 + i], H, W, f);
            if (e) {
                char tmp[256];
                snprintf(tmp, sizeof tmp, "frame %d: %s", b0 + i, e);
                err = tmp;
                return CAMSYNC_E_ARG;
            }
            f.d.dataOff = (uint32_t)totalBytes;
            f.d.dataLen = (uint32_t)(sizes[b0 + i] - f.scanOff);
            totalBytes += (f.d.dataLen + 15u) & ~(size_t)15u;
            if (totalBytes > 0xFFFF0000ull) { err = "more than 4 GiB of compressed data in one pass"; return CAMSYNC_E_ARG; }
        }
        totalBytes += kHuffWindow + 64;  // (a lane's window is loaded whole, wherever its stream ends)
        totalBytes = (totalBytes + 255) & ~(size_t)255;
        passBytes[(size_t)k] = totalBytes;
        regionBytes = std::max(regionBytes, totalBytes);
    }
    // two regions: the pass being decoded and the one being sent
    const size_t nRegions = nPass > 1 ? 2 : 1;
    if (nRegions * regionBytes > s->hBytesCap) {
        if (s->h_bytes) (void)hipHostFree(s->h_bytes);
        s->h_bytes = nullptr;
        s->hBytesCap = 0;
        const size_t n = nRegions * regionBytes + nRegions * regionBytes / 4;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&s->h_bytes), n, hipHostMallocDefault);
        if (e != hipSuccess) return hipfail("hipHostMalloc(compressed frames)", e);
        s->hBytesCap = n;
    }
    if (!regrow_dev(s->d_bytes, s->bytesCap, nRegions * regionBytes)) {
        err = "out of device memory for the JPEG decoder's buffers";
        return CAMSYNC_E_HIP;
    }
    if (!s->copySt) {
        hipError_t e = hipStreamCreateWithFlags(&s->copySt, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->evUp[0], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->evUp[1], hipEventDisableTiming);
        if (e != hipSuccess) return hipfail("stream for the compressed frames", e);
    }
    if (dbg) fprintf(stderr, "jpeg: %d frames in %d passes: headers %.2f ms\n", B, nPass, ms(tParse, now()));
    // The entropy-coded bytes of pass k, packed (16-byte aligned each, zeros in between and after the last) into region
    // k % 2 of the page-locked buffer by several threads; each sends its range as soon as it is packed.  (With hundreds
    // of megabytes to move this is a good part of the call.)
    int devId = 0;
    (void)hipGetDevice(&devId);
    // frames with restart intervals: where each interval starts in the frame's entropy-coded bytes (found while packing)
    std::vector<std::vector<uint32_t>> ivAll((size_t)B);
    auto send_pass = [&](int k) -> int {
        const int b0 = k * chunk, nb = std::min(chunk, B - b0);
        const size_t totalBytes = passBytes[(size_t)k], region = (size_t)(k & 1) * regionBytes;
        std::atomic<int> copyErr{(int)hipSuccess};
        auto pack = [&](int i0, int i1) {
            for (int i = i0; i < i1; i++) {
                const ParsedFrame& f = pfAll[(size_t)(b0 + i)];
                uint8_t* dst = s->h_bytes + region + f.d.dataOff;
                memcpy(dst, jpegs[b0 + i] + f.scanOff, f.d.dataLen);
                const size_t end = i + 1 < nb ? pfAll[(size_t)(b0 + i + 1)].d.dataOff : totalBytes;
                memset(dst + f.d.dataLen, 0, end - f.d.dataOff - f.d.dataLen);
                if (f.d.restart) {
                    std::vector<uint32_t>& iv = ivAll[(size_t)(b0 + i)];
                    iv.assign(1, 0u);
                    const uint8_t *const q = dst, *const qe = dst + f.d.dataLen, *cur = dst;
                    while (cur < qe && (cur = static_cast<const uint8_t*>(memchr(cur, 0xFF, (size_t)(qe - cur)))) != nullptr) {
                        if (cur + 1 >= qe) break;
                        const uint8_t m = cur[1];
                        if (m >= 0xD0 && m <= 0xD7) {
                            if ((uint32_t)(m - 0xD0) != ((uint32_t)(iv.size() - 1) & 7u)) {
                                iv.clear();  // out of sequence: the frame is decoded in one piece, which reports it
                                break;
                            }
                            iv.push_back((uint32_t)(cur + 2 - q));
                            cur += 2;
                        } else if (m == 0x00) {
                            cur += 2;
                        } else if (m == 0xFF) {
                            cur += 1;
                        } else {
                            break;  // another marker: the data ends here
                        }
                    }
                }
            }
            if (i1 > i0) {
                const size_t lo = pfAll[(size_t)(b0 + i0)].d.dataOff, hi = i1 < nb ? pfAll[(size_t)(b0 + i1)].d.dataOff : totalBytes;
                hipError_t ce = hipSetDevice(devId);
                if (ce == hipSuccess) ce = hipMemcpyAsync(s->d_bytes + region + lo, s->h_bytes + region + lo, hi - lo, hipMemcpyHostToDevice, s->copySt);
                if (ce != hipSuccess) copyErr = (int)ce;
            }
        };
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int nt = totalBytes < ((size_t)32 << 20) ? 1 : (int)std::min<unsigned>(std::min<unsigned>(hw, 8u), (unsigned)nb);
        if (nt <= 1) {
            pack(0, nb);
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++) th.emplace_back(pack, (int)((long long)nb * t / nt), (int)((long long)nb * (t + 1) / nt));
            for (auto& x : th) x.join();
        }
        if (copyErr.load() != (int)hipSuccess) return hipfail("hipMemcpyAsync(compressed frames)", (hipError_t)copyErr.load());
        hipError_t e = hipEventRecord(s->evUp[k & 1], s->copySt);
        if (e != hipSuccess) return hipfail("hipEventRecord", e);
        return CAMSYNC_OK;
    };
    {
        const int rc0 = send_pass(0);
        if (rc0) return rc0;
    }
    for (int b0 = 0, pass = 0; b0 < B; b0 += chunk, pass++) {
        const int nb = std::min(chunk, B - b0);
        const auto t0 = now();
        // (region (pass + 1) % 2 is free: the pass that used it has been waited for)
        if (pass + 1 < nPass) {
            const int rcn = send_pass(pass + 1);
            if (rcn) return rcn;
        }
        pf.assign(pfAll.begin() + b0, pfAll.begin() + b0 + nb);
        const size_t totalBytes = passBytes[(size_t)pass];
        const uint8_t* const dBytes = s->d_bytes + (size_t)(pass & 1) * regionBytes;
        // blocks per frame of this pass: the largest of its frames' layouts
        uint32_t frameBlocks = 0;
        for (int i = 0; i < nb; i++) {
            const JpegDesc& d = pf[(size_t)i].d;
            frameBlocks = std::max(frameBlocks, (uint32_t)d.mcux * d.hmax * d.mcuy * d.vmax + (d.nc == 3 ? 2u * d.mcux * d.mcuy : 0u));
        }
        // ---- table sets, groups of 64 frames that share one ----
        std::map<std::string, int> setIndex;
        std::vector<uint8_t> tables;
        std::vector<std::vector<int>> framesOfSet;
        for (int i = 0; i < nb; i++) {
            const ParsedFrame& p = pf[(size_t)i];
            std::string key;
            for (int t = 0; t < 3; t++) key += p.dc[t].set ? p.dc[t].key() : std::string("-");
            for (int t = 0; t < 3; t++) key += p.ac[t].set ? p.ac[t].key() : std::string("-");
            auto it = setIndex.find(key);
            if (it == setIndex.end()) {
                const int id = (int)framesOfSet.size();
                it = setIndex.emplace(key, id).first;
                framesOfSet.emplace_back();
                tables.resize(tables.size() + (size_t)kHuffSetTables * kHuffTableBytes, 0);
                uint8_t* base = tables.data() + (size_t)id * kHuffSetTables * kHuffTableBytes;
                for (int t = 0; t < 3; t++) {
                    if (p.dc[t].set && !build_table(p.dc[t], base + (size_t)t * kHuffTableBytes)) { err = "a Huffman table is not a prefix code"; return CAMSYNC_E_ARG; }
                    if (p.ac[t].set && !build_table(p.ac[t], base + (size_t)(3 + t) * kHuffTableBytes)) { err = "a Huffman table is not a prefix code"; return CAMSYNC_E_ARG; }
                }
            }
            framesOfSet[(size_t)it->second].push_back(i);
        }
        // Frames without restart intervals go to the parallel decoder; the others (and the frames whose records do not
        // settle there) to the sequential kernels: those with at most two DC and two AC tables take a wavefront each
        // (k_jpeg_huff_wave), the rest share wavefronts, kHuffLanes frames of one table set to each (k_jpeg_huff).
        static const bool sequentialOnly = getenv("CAMSYNC_JPEG_SEQUENTIAL") != nullptr;  // (tests: the sequential kernels on every frame)
        std::vector<int32_t> frameSet((size_t)nb, 0), parList, skip((size_t)nb, 1);
        for (size_t set = 0; set < framesOfSet.size(); set++)
            for (int i : framesOfSet[set]) frameSet[(size_t)i] = (int32_t)set;
        for (int i = 0; i < nb; i++)
            if (pf[(size_t)i].d.restart == 0 && !sequentialOnly) {
                parList.push_back(i);
                skip[(size_t)i] = 0;
            }
        int nGroups = 0, nWave = 0;
        size_t waveOff = 0;
        auto build_sequential_lists = [&]() {  // from skip[]
            groups.clear();
            std::vector<int32_t> waveList;
            for (size_t set = 0; set < framesOfSet.size(); set++) {
                std::vector<int> shared;
                for (int i : framesOfSet[set]) {
                    if (!skip[(size_t)i]) continue;
                    if (!pf[(size_t)i].dc[2].set && !pf[(size_t)i].ac[2].set) {
                        // a wavefront per frame, or -- restart intervals start in a known state -- one per run of
                        // intervals, up to 256 wavefronts a frame
                        const JpegDesc& d = pf[(size_t)i].d;
                        const uint32_t totalMcu = (uint32_t)d.mcux * d.mcuy;
                        const std::vector<uint32_t>& iv = ivAll[(size_t)(b0 + i)];
                        const uint32_t nIv = d.restart ? (totalMcu + d.restart - 1u) / d.restart : 1u;
                        if (nIv > 1u && iv.size() == (size_t)nIv) {
                            const uint32_t per = (nIv + 255u) / 256u;
                            for (uint32_t t = 0; t < nIv; t += per) {
                                const uint32_t first = t * d.restart, n = std::min(std::min(per, nIv - t) * (uint32_t)d.restart, totalMcu - first);
                                const int32_t e6[kWaveEntry] = {i, (int32_t)set, (int32_t)iv[t], (int32_t)first, (int32_t)n, (int32_t)(t & 7u)};
                                waveList.insert(waveList.end(), e6, e6 + kWaveEntry);
                            }
                        } else {
                            const int32_t e6[kWaveEntry] = {i, (int32_t)set, 0, 0, (int32_t)totalMcu, 0};
                            waveList.insert(waveList.end(), e6, e6 + kWaveEntry);
                        }
                    } else {
                        shared.push_back(i);
                    }
                }
                for (size_t k = 0; k < shared.size(); k += kHuffLanes) {
                    groups.push_back((int32_t)set);
                    for (size_t j = 0; j < (size_t)kHuffLanes; j++) groups.push_back(k + j < shared.size() ? shared[k + j] : -1);
                }
            }
            nGroups = (int)(groups.size() / (1 + kHuffLanes));
            nWave = (int)(waveList.size() / kWaveEntry);
            waveOff = groups.size();
            groups.insert(groups.end(), waveList.begin(), waveList.end());
            if (groups.empty()) groups.push_back(0);
        };
        // the parallel decoder's index arrays: per frame its table set, its first record and the skip flag; per
        // workgroup of 256 subsequences its frame and its first subsequence
        std::vector<uint32_t> subBase((size_t)nb, 0), wgFirst;
        std::vector<int32_t> wgFrame;
        size_t nRec = 0;
        for (int i : parList) {
            const uint32_t nSub = (uint32_t)(((size_t)pf[(size_t)i].d.dataLen * 8 + kParBits - 1) / kParBits);
            subBase[(size_t)i] = (uint32_t)nRec;
            nRec += nSub;
            for (uint32_t k = 0; k < nSub; k += 256) {
                wgFrame.push_back(i);
                wgFirst.push_back(k);
            }
        }
        if (nRec > 0xFFFFFFF0ull) { err = "too many subsequences in one pass"; return CAMSYNC_E_ARG; }
        const size_t nWg = wgFrame.size();
        // packed: frameSet[nb] | subBase[nb] | skip[nb] | ulenBits[nb] | parList[nb] | wgFrame[nWg] | wgFirst[nWg]
        std::vector<int32_t> parMeta((size_t)5 * nb + 2 * nWg, 0);
        memcpy(&parMeta[0], frameSet.data(), (size_t)nb * 4);
        memcpy(&parMeta[(size_t)nb], subBase.data(), (size_t)nb * 4);
        memcpy(&parMeta[(size_t)2 * nb], skip.data(), (size_t)nb * 4);
        if (!parList.empty()) memcpy(&parMeta[(size_t)4 * nb], parList.data(), parList.size() * 4);
        if (nWg) {
            memcpy(&parMeta[(size_t)5 * nb], wgFrame.data(), nWg * 4);
            memcpy(&parMeta[(size_t)5 * nb + nWg], wgFirst.data(), nWg * 4);
        }
        descs.resize((size_t)nb);
        quant.resize((size_t)nb * 256);
        for (int i = 0; i < nb; i++) {
            descs[(size_t)i] = pf[(size_t)i].d;
            memcpy(&quant[(size_t)i * 256], pf[(size_t)i].quant, 512);
        }
        const auto t1 = now();
        // ---- buffers ----
        const size_t coefNeed = (size_t)nb * frameBlocks * 64;
        const bool freshCoef = coefNeed > s->coefCap;
        if (!regrow_dev(s->d_desc, s->descCap, (size_t)nb) ||
            !regrow_dev(s->d_quant, s->quantCap, (size_t)nb * 256) || !regrow_dev(s->d_tables, s->tablesCap, tables.size()) ||
            !regrow_dev(s->d_status, s->statusCap, (size_t)nb) ||
            (!parList.empty() && (!regrow_dev(s->d_ubytes, s->ubytesCap, totalBytes) || !regrow_dev(s->d_rec, s->recCap, nRec) ||
                                  !regrow_dev(s->d_last, s->lastCap, nRec) || !regrow_dev(s->d_dcs, s->dcsCap, (size_t)nb * frameBlocks))) ||
            !regrow_dev(s->d_parMeta, s->parMetaCap, parMeta.size()) || !regrow_dev(s->d_changed, s->changedCap, (size_t)nb + 1) ||
            !regrow_dev(s->d_coef, s->coefCap, coefNeed) || !regrow_dev(s->d_planes, s->planesCap, (size_t)nb * frameBlocks * 64) ||
            (!outOnDevice && !regrow_dev(s->d_out, s->outCap, (size_t)nb * H * W * 3))) {
            err = "out of device memory for the JPEG decoder's buffers";
            return CAMSYNC_E_HIP;
        }
        hipError_t e;
        if (freshCoef || s->coefDirty) {
            if ((e = hipMemsetAsync(s->d_coef, 0, s->coefCap * sizeof(int16_t), st)) != hipSuccess) return hipfail("hipMemsetAsync", e);
            s->coefDirty = false;
        }
        const auto t2 = now();
        if ((e = hipStreamWaitEvent(st, s->evUp[pass & 1], 0)) != hipSuccess) return hipfail("hipStreamWaitEvent", e);
        const auto t3 = now();
        if ((e = hipMemcpyAsync(s->d_desc, descs.data(), sizeof(JpegDesc) * nb, hipMemcpyHostToDevice, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
        if ((e = hipMemcpyAsync(s->d_quant, quant.data(), quant.size() * 2, hipMemcpyHostToDevice, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
        if ((e = hipMemcpyAsync(s->d_tables, tables.data(), tables.size(), hipMemcpyHostToDevice, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
        if ((e = hipMemcpyAsync(s->d_parMeta, parMeta.data(), parMeta.size() * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
        if ((e = hipMemsetAsync(s->d_status, 0, sizeof(int32_t) * nb, st)) != hipSuccess) return hipfail("hipMemsetAsync", e);
        // ---- entropy decoding ----
        uint8_t* const dst = outOnDevice ? out + (size_t)b0 * H * W * 3 : s->d_out;
        s->coefDirty = true;  // (until the inverse DCT has put the zeros back)
        const int32_t *const d_frameSet = s->d_parMeta, *const d_skip = s->d_parMeta + (size_t)2 * nb, *const d_parList = s->d_parMeta + (size_t)4 * nb,
                      *const d_wgFrame = s->d_parMeta + (size_t)5 * nb;
        const uint32_t *const d_subBase = reinterpret_cast<const uint32_t*>(s->d_parMeta + (size_t)nb),
                       *const d_wgFirst = reinterpret_cast<const uint32_t*>(s->d_parMeta + (size_t)5 * nb + nWg);
        uint32_t* const d_ulen = reinterpret_cast<uint32_t*>(s->d_parMeta + (size_t)3 * nb);
        if (!parList.empty()) {
            if ((e = hipMemsetAsync(s->d_changed, 0, sizeof(int32_t) * ((size_t)nb + 1), st)) != hipSuccess) return hipfail("hipMemsetAsync", e);
            hipLaunchKernelGGL(k_jpeg_unstuff, dim3((unsigned)parList.size()), dim3(256), 0, st, dBytes, s->d_desc, d_parList, s->d_ubytes,
                               d_ulen, d_subBase, s->d_rec, s->d_last);
            // rounds until one changes nothing; the host only looks after some of them (a round in which few lanes have work
            // is a few microseconds, waiting for the stream is more)
            // (the true records spread at least one subsequence a round, so a frame of n subsequences has settled after n
            // rounds at the latest; a round in which nearly every lane finds its entry unchanged takes 10-20 us)
            bool settled = false;
            uint32_t longest = 0;
            for (int i : parList) longest = std::max(longest, (uint32_t)(((size_t)pf[(size_t)i].d.dataLen * 8 + kParBits - 1) / kParBits));
            const int maxRounds = (int)std::min<uint32_t>((uint32_t)kParMaxRounds, longest + 1u);
            int nextLook = std::min(3, maxRounds);
            for (int round = 1; round <= maxRounds && !settled; round++) {
                const bool check = round == nextLook;
                if (check && (e = hipMemsetAsync(s->d_changed, 0, sizeof(int32_t), st)) != hipSuccess) return hipfail("hipMemsetAsync", e);
                hipLaunchKernelGGL((k_jpeg_par<false>), dim3((unsigned)nWg), dim3(256), 0, st, s->d_ubytes, s->d_desc, d_wgFrame, d_wgFirst,
                                   s->d_tables, d_frameSet, d_ulen, d_subBase, s->d_rec, s->d_last, s->d_changed, round, d_skip, s->d_coef,
                                   s->d_dcs, frameBlocks, s->d_status);
                if (!check) continue;
                nextLook = std::min(maxRounds, nextLook + std::max(2, nextLook / 2));
                int32_t flag = 0;
                if ((e = hipMemcpyAsync(&flag, s->d_changed, sizeof flag, hipMemcpyDeviceToHost, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
                if ((e = hipStreamSynchronize(st)) != hipSuccess) return hipfail("JPEG kernels (parallel entropy decoding)", e);
                settled = flag == 0;
                if (settled && getenv("CAMSYNC_DEBUG_JPEG")) fprintf(stderr, "jpeg: %zu frames, %zu subsequences settled within %d rounds\n", parList.size(), nRec, round);
            }
            if (!settled) {
                // the frames whose records still changed in the last round take the sequential kernels
                std::vector<int32_t> last((size_t)nb + 1, 0);
                if ((e = hipMemcpy(last.data(), s->d_changed, sizeof(int32_t) * ((size_t)nb + 1), hipMemcpyDeviceToHost)) != hipSuccess) return hipfail("hipMemcpy", e);
                for (int i : parList)
                    if (last[(size_t)i + 1] == maxRounds) skip[(size_t)i] = 1;
                if ((e = hipMemcpyAsync(s->d_parMeta + (size_t)2 * nb, skip.data(), (size_t)nb * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
            }
            hipLaunchKernelGGL((k_jpeg_par<true>), dim3((unsigned)nWg), dim3(256), 0, st, s->d_ubytes, s->d_desc, d_wgFrame, d_wgFirst,
                               s->d_tables, d_frameSet, d_ulen, d_subBase, s->d_rec, s->d_last, s->d_changed, 0, d_skip, s->d_coef, s->d_dcs,
                               frameBlocks, s->d_status);
            hipLaunchKernelGGL(k_jpeg_dc, dim3(3, (unsigned)parList.size()), dim3(256), 0, st, s->d_dcs, s->d_desc, d_parList, d_skip, frameBlocks);
        }
        build_sequential_lists();
        if (!regrow_dev(s->d_groups, s->groupsCap, groups.size())) {
            err = "out of device memory for the JPEG decoder's buffers";
            return CAMSYNC_E_HIP;
        }
        if ((e = hipMemcpyAsync(s->d_groups, groups.data(), groups.size() * 4, hipMemcpyHostToDevice, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
        if (nGroups)
            hipLaunchKernelGGL(k_jpeg_huff, dim3((unsigned)nGroups), dim3(64), 0, st, dBytes, s->d_desc, s->d_groups, s->d_tables,
                               s->d_coef, frameBlocks, s->d_status);
        if (nWave)
            hipLaunchKernelGGL(k_jpeg_huff_wave, dim3((unsigned)nWave), dim3(64), 0, st, dBytes, s->d_desc, s->d_groups + waveOff,
                               s->d_tables, s->d_coef, frameBlocks, s->d_status);
        hipLaunchKernelGGL(k_jpeg_idct, dim3((frameBlocks + kIdctBlocksPerWg - 1) / kIdctBlocksPerWg, (unsigned)nb), dim3(256), 0, st,
                           s->d_coef, s->d_desc, s->d_quant, s->d_planes, frameBlocks, (const int32_t*)nullptr,
                           parList.empty() ? (const int16_t*)nullptr : s->d_dcs, d_skip);
        const unsigned colorBlocks = (unsigned)(((size_t)H * ((W + 7) / 8) + 255) / 256);
        hipLaunchKernelGGL(k_jpeg_color, dim3(colorBlocks, (unsigned)nb), dim3(256), 0, st, s->d_planes, s->d_desc, frameBlocks, H, W,
                           dst, (const int32_t*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return hipfail("JPEG kernels", e);
        status.assign((size_t)nb, 0);
        if ((e = hipMemcpyAsync(status.data(), s->d_status, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
        if (!outOnDevice)
            if ((e = hipMemcpyAsync(out + (size_t)b0 * H * W * 3, s->d_out, (size_t)nb * H * W * 3, hipMemcpyDeviceToHost, st)) != hipSuccess) return hipfail("hipMemcpyAsync", e);
        if ((e = hipStreamSynchronize(st)) != hipSuccess) return hipfail("hipStreamSynchronize", e);
        if (dbg) fprintf(stderr, "jpeg: pass of %d frames: sending the next pass and tables %.2f ms, buffers %.2f ms, device %.2f ms\n", nb, ms(t0, t1), ms(t1, t2), ms(t3, now()));
        s->coefDirty = false;  // every block the Huffman kernel may have touched has been through k_jpeg_idct
        for (int i = 0; i < nb; i++)
            if (status[(size_t)i]) {
                char tmp[128];
                snprintf(tmp, sizeof tmp, "frame %d: corrupt entropy-coded data", b0 + i);
                err = tmp;
                return CAMSYNC_E_ARG;
            }
    }
    return CAMSYNC_OK;
}

}  // namespace camsync
