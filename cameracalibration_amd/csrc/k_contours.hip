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
// So: (1) a stencil pass lists every pixel that can be such a start
// (outer: p=1, W=NW=N=NE=0; hole: p=0, W=1, N=1), (2) one lane per candidate
// follows the border exactly like OpenCV's icvFetchContour and drops the
// candidate as soon as it meets a crack with a smaller raster key (it is then
// not the canonical start), (3) survivors whose length passes the perimeter
// filter are traced once more to store their points.
// The traces only read the image, so they are independent.
#include "internal.h"

namespace camsync {

namespace {

__device__ __forceinline__ uint32_t ld_row(const uint8_t* __restrict__ img, int H, int W, int y, int xd) {
    // dword xd (pixels 4xd..4xd+3) of row y, with the 1-px image frame zeroed
    if (y <= 0 || y >= H - 1 || xd < 0 || 4 * xd >= W) return 0u;
    uint32_t v = *reinterpret_cast<const uint32_t*>(img + (size_t)y * W + 4 * xd);
    if (xd == 0) v &= 0xFFFFFF00u;
    if (4 * xd + 3 == W - 1) v &= 0x00FFFFFFu;
    return v;
}

// Stencil pass.  One thread per 4 pixels; all scales at once (SWAR on the
// packed scale bits).  W % 4 == 0.
__global__ __launch_bounds__(256) void k_contour_starts(const uint8_t* __restrict__ bin, DetCfg cfg, int B,
                                                        uint64_t* __restrict__ cand, unsigned int candCap,
                                                        Counters* __restrict__ ctr) {
    const int H = cfg.H, W = cfg.W, W4 = W / 4;
    const size_t total = (size_t)B * H * W4;
    const uint32_t scaleMask = ((1u << cfg.nScales) - 1u) * 0x01010101u;
    const bool dropIsolated = cfg.minPerim > 1;
    for (size_t base = (size_t)blockIdx.x * blockDim.x; base < total; base += (size_t)gridDim.x * blockDim.x) {
        size_t i = base + threadIdx.x;
        uint32_t outer = 0, hole = 0;
        int f = 0, y = 0, xd = 0;
        if (i < total) {
            xd = (int)(i % W4);
            y = (int)((i / W4) % H);
            f = (int)(i / ((size_t)W4 * H));
            const uint8_t* img = bin + (size_t)f * H * W;
            uint32_t c = ld_row(img, H, W, y, xd);
            uint32_t cl = ld_row(img, H, W, y, xd - 1), cr = ld_row(img, H, W, y, xd + 1);
            uint32_t n = ld_row(img, H, W, y - 1, xd);
            uint32_t nl = ld_row(img, H, W, y - 1, xd - 1), nr = ld_row(img, H, W, y - 1, xd + 1);
            uint32_t w = (c << 8) | (cl >> 24);
            uint32_t nw = (n << 8) | (nl >> 24);
            uint32_t ne = (n >> 8) | (nr << 24);
            outer = c & ~w & ~nw & ~n & ~ne & scaleMask;
            hole = ~c & w & n & scaleMask;
            if (dropIsolated && outer) {
                uint32_t s = ld_row(img, H, W, y + 1, xd);
                uint32_t sl = ld_row(img, H, W, y + 1, xd - 1), sr = ld_row(img, H, W, y + 1, xd + 1);
                uint32_t e = (c >> 8) | (cr << 24);
                uint32_t sw = (s << 8) | (sl >> 24);
                uint32_t se = (s >> 8) | (sr << 24);
                outer &= (e | sw | s | se);  // w,nw,n,ne are already zero where outer is set
            }
        }
        const int cnt = __popc(outer) + __popc(hole);
        // wave-aggregated append
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d, 64);
            if ((int)(threadIdx.x & 63) >= d) incl += t;
        }
        const int waveTotal = __shfl(incl, 63, 64);
        unsigned int wbase = 0;
        if (waveTotal) {
            if ((threadIdx.x & 63) == 63) wbase = atomicAdd(&ctr->nCand, (unsigned int)waveTotal);
            wbase = __shfl(wbase, 63, 64);
            unsigned int o = wbase + (unsigned int)(incl - cnt);
            if (wbase + (unsigned int)waveTotal > candCap) {
                if ((threadIdx.x & 63) == 63) atomicOr(&ctr->overflow, 1u);
            } else {
                const uint32_t rowPos = (uint32_t)y * W + 4u * xd;
                uint32_t m = outer;
                while (m) {
                    int bit = __ffs(m) - 1;
                    m &= m - 1;
                    cand[o++] = pack_cand(f, 0, bit & 7, rowPos + (bit >> 3));
                }
                m = hole;
                while (m) {
                    int bit = __ffs(m) - 1;
                    m &= m - 1;
                    // start pixel is the one left of the first hole pixel
                    cand[o++] = pack_cand(f, 1, bit & 7, rowPos + (bit >> 3) - 1);
                }
            }
        }
    }
}

__device__ __forceinline__ uint32_t nb_mask(const uint8_t* __restrict__ img, int H, int W, int x, int y, int sc) {
    // bit d = neighbour of (x,y) in direction d is foreground.
    // directions: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE.  (x,y) is interior.
    const uint8_t* r0 = img + (size_t)(y - 1) * W + x;
    const uint8_t* r1 = r0 + W;
    const uint8_t* r2 = r1 + W;
    uint32_t nw = r0[-1], n = r0[0], ne = r0[1], w = r1[-1], e = r1[1], sw = r2[-1], s = r2[0], se = r2[1];
    const bool top = (y - 1) <= 0, bot = (y + 1) >= H - 1, lef = (x - 1) <= 0, rig = (x + 1) >= W - 1;
    uint32_t m = 0;
    m |= (!rig && ((e >> sc) & 1u)) ? 1u : 0u;
    m |= (!rig && !top && ((ne >> sc) & 1u)) ? 2u : 0u;
    m |= (!top && ((n >> sc) & 1u)) ? 4u : 0u;
    m |= (!lef && !top && ((nw >> sc) & 1u)) ? 8u : 0u;
    m |= (!lef && ((w >> sc) & 1u)) ? 16u : 0u;
    m |= (!lef && !bot && ((sw >> sc) & 1u)) ? 32u : 0u;
    m |= (!bot && ((s >> sc) & 1u)) ? 64u : 0u;
    m |= (!rig && !bot && ((se >> sc) & 1u)) ? 128u : 0u;
    return m;
}

__device__ __forceinline__ int dir_dx(int d) { return (d == 0 || d == 1 || d == 7) ? 1 : (d >= 3 && d <= 5) ? -1 : 0; }
__device__ __forceinline__ int dir_dy(int d) { return (d >= 1 && d <= 3) ? -1 : (d >= 5 && d <= 7) ? 1 : 0; }

// Follows one border like OpenCV's icvFetchContour.  STORE = false: count the
// points, abort when the start turns out not to be canonical or the border is
// longer than maxLen.  STORE = true: write the points.  Returns the number of
// points, or 0 when aborted.
template <bool STORE>
__device__ int follow_border(const uint8_t* __restrict__ img, int H, int W, int sc, int x0, int y0, bool isHole,
                             int maxLen, uint32_t* __restrict__ out) {
    const int startKey = y0 * W + x0;
    int s_end = isHole ? 0 : 4;
    int s = s_end;
    uint32_t m = nb_mask(img, H, W, x0, y0, sc);
    bool found = false;
    for (int k = 0; k < 8; k++) {
        s = (s - 1) & 7;
        if (s == s_end) break;
        if ((m >> s) & 1u) { found = true; break; }
    }
    if (!found) {  // single pixel
        if (STORE) out[0] = (uint32_t)x0 | ((uint32_t)y0 << 16);
        return 1;
    }
    const int x1 = x0 + dir_dx(s), y1 = y0 + dir_dy(s);
    int cx = x0, cy = y0, count = 0;
    for (;;) {
        s_end = s;
        // first foreground neighbour counter-clockwise after s_end
        uint32_t rot = ((m >> (s_end + 1)) | (m << (7 - s_end))) & 0xFFu;  // bit j <-> direction s_end+1+j
        int k = __ffs(rot);                                                // 1..8 (m has at least the arrival neighbour)
        if (k == 0) return 0;                                              // cannot happen on a consistent image
        s = (s_end + k) & 7;
        if (!STORE) {
            const int key = cy * W + cx;
            if (isHole) {
                if (s_end + k >= 9 && key < startKey) return 0;  // an east crack of this border lies before the start
            } else {
                if (key < startKey) return 0;
            }
        }
        if (STORE) out[count] = (uint32_t)cx | ((uint32_t)cy << 16);
        count++;
        const int nx = cx + dir_dx(s), ny = cy + dir_dy(s);
        if (nx == x0 && ny == y0 && cx == x1 && cy == y1) break;
        if (count >= maxLen) return STORE ? count : 0;
        cx = nx;
        cy = ny;
        s = (s + 4) & 7;
        m = nb_mask(img, H, W, cx, cy, sc);
    }
    return count;
}

__global__ __launch_bounds__(256) void k_trace_count(const uint8_t* __restrict__ bin, DetCfg cfg,
                                                     const uint64_t* __restrict__ cand, ContourRec* __restrict__ contours,
                                                     unsigned int contourCap, unsigned int poolCap, Counters* __restrict__ ctr) {
    const unsigned int n = min(ctr->nCand, 0xFFFFFFFFu);
    const int H = cfg.H, W = cfg.W;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t c = cand[i];
        const uint32_t pos = (uint32_t)(c & 0xFFFFFFFFull);
        const int sc = (int)((c >> 36) & 7u);
        const bool hole = (c >> 39) & 1u;
        const uint32_t f = (uint32_t)(c >> 40);
        const int x0 = pos % W, y0 = pos / W;
        const uint8_t* img = bin + (size_t)f * H * W;
        // one more point than allowed is enough to know the border is too long
        const int len = follow_border<false>(img, H, W, sc, x0, y0, hole, cfg.maxPerim + 1, nullptr);
        if (len >= cfg.minPerim && len <= cfg.maxPerim && len > 0) {
            const unsigned int off = atomicAdd(&ctr->nPool, (unsigned int)len);
            if (off + (unsigned int)len > poolCap) {
                atomicOr(&ctr->overflow, 4u);
                continue;
            }
            const unsigned int slot = atomicAdd(&ctr->nContours, 1u);
            if (slot >= contourCap) {
                atomicOr(&ctr->overflow, 2u);
                continue;
            }
            ContourRec r;
            r.frame = f;
            r.pos = pos;
            r.len = (uint32_t)len;
            r.off = off;
            r.scale = (uint16_t)sc;
            r.hole = hole ? 1 : 0;
            r.scanKey = pos + (hole ? 1u : 0u);
            contours[slot] = r;
        }
    }
}

__global__ __launch_bounds__(64) void k_trace_store(const uint8_t* __restrict__ bin, DetCfg cfg,
                                                    const ContourRec* __restrict__ contours, unsigned int contourCap,
                                                    uint32_t* __restrict__ pool, const Counters* __restrict__ ctr) {
    const unsigned int n = min(ctr->nContours, contourCap);
    const int H = cfg.H, W = cfg.W;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const ContourRec r = contours[i];
        const uint8_t* img = bin + (size_t)r.frame * H * W;
        follow_border<true>(img, H, W, r.scale, r.pos % W, r.pos / W, r.hole != 0, (int)r.len, pool + r.off);
    }
}

}  // namespace

void launch_contour_starts(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    size_t total = (size_t)B * c.H * (c.W / 4);
    size_t nblk = (total + 255) / 256;
    unsigned blocks = (unsigned)(nblk < (size_t)(256 * 16) ? nblk : (size_t)(256 * 16));
    hipLaunchKernelGGL(k_contour_starts, dim3(blocks), dim3(256), 0, st, b.bin, c, B, b.cand, b.candCap, b.counters);
}

void launch_trace(const DetCfg& c, const DetBuffers& b, hipStream_t st) {
    hipLaunchKernelGGL(k_trace_count, dim3(256 * 8), dim3(256), 0, st, b.bin, c, b.cand, b.contours, b.contourCap,
                       b.poolCap, b.counters);
    hipLaunchKernelGGL(k_trace_store, dim3(256 * 8), dim3(64), 0, st, b.bin, c, b.contours, b.contourCap, b.pool,
                       b.counters);
}

}  // namespace camsync
