// k_subpix.hip — K7: cv::cornerSubPix as ArucoDetector.detectMarkers applies
// it with CORNER_REFINE_SUBPIX (reference camera_sync/aruco.py:251-264):
// win = clamp(round(0.3 * moduleSize), 1, cornerRefinementWinSize), zero zone
// off, criteria (MAX_ITER | EPS, 100, 0.01).
//
// One wavefront per corner.  Every iteration the 64 lanes sample the
// (2 win + 3)^2 bilinear patch into LDS with getRectSubPix's float
// arithmetic, accumulate the five gradient moments in f64 (lane-strided
// partial sums + butterfly) and all lanes take the same 2x2 step.  The
// separable weight table exp(-(i/win)^2) is computed on the host so the
// weights are those of the CPU path bit for bit.
#include "internal.h"
#include "wave_reduce.h"

#include <algorithm>

namespace camsync {

namespace {

constexpr int kMaxWin = 20;
constexpr int kMaxPatch = 2 * kMaxWin + 3;
// The corner moves by a fraction of a pixel per iteration, so the pixels every iteration's patch needs are almost
// always those of the first patch and a few around them: that region is copied to LDS once per corner and the
// patches are sampled from there (same bytes, same arithmetic); a patch that leaves it is sampled from the image.
constexpr int kRegMargin = 4;
constexpr int kRegDim = kMaxPatch + 1 + 2 * kRegMargin;

__device__ __forceinline__ double wave_sum_d(double v) {
    return wave_allreduce(v, [](double a, double b) { return a + b; });
}

// getRectSubPix_8u32f for one output element (i,j) of a pw x pw patch.
struct PatchGeom {
    int ipx, ipy;
    bool fast;
    float a, b, a11, a12, a21, a22, b1, b2;
    double s;
    int rx, rwid;
};

__device__ __forceinline__ PatchGeom patch_geom(float cx, float cy, int pw, int W, int H) {
    PatchGeom g;
    cx -= (pw - 1) * 0.5f;
    cy -= (pw - 1) * 0.5f;
    g.ipx = (int)floorf(cx);
    g.ipy = (int)floorf(cy);
    g.fast = (0 <= g.ipx && g.ipx + pw < W && 0 <= g.ipy && g.ipy + pw < H);
    float a = cx - g.ipx, b = cy - g.ipy;
    if (g.fast) {
        a = fmaxf(a, 0.0001f);
        g.a = a;
        g.b = b;
        g.a12 = a * (1.f - b);
        g.a22 = a * b;
        g.b1 = 1.f - b;
        g.b2 = b;
        g.s = (1. - a) / a;
        g.a11 = g.a21 = 0;
        g.rx = 0;
        g.rwid = pw;
    } else {
        g.a = a;
        g.b = b;
        g.a11 = (1.f - a) * (1.f - b);
        g.a12 = a * (1.f - b);
        g.a21 = (1.f - a) * b;
        g.a22 = a * b;
        g.b1 = 1.f - b;
        g.b2 = b;
        g.s = 0;
        g.rx = g.ipx >= 0 ? 0 : min(-g.ipx, pw);
        g.rwid = (g.ipx < W - pw) ? pw : max(W - g.ipx - 1, 0);
    }
    return g;
}

// the patch lies inside the LDS region: p = first pixel of patch row i, stride = bytes per region row
__device__ __forceinline__ float patch_elem_inside(const uint8_t* __restrict__ p, int stride, const PatchGeom& g, int j) {
    const float t = g.a12 * p[j + 1] + g.a22 * p[j + 1 + stride];
    float prev;
    if (j == 0) {
        prev = (1 - g.a) * (g.b1 * p[0] + g.b2 * p[stride]);
    } else {
        const float tp = g.a12 * p[j] + g.a22 * p[j + stride];
        prev = (float)(tp * g.s);
    }
    return prev + t;
}

// any patch, sampled from the image (rare: patches that touch the image border or have left the LDS region)
__device__ __forceinline__ float patch_elem(const GraySrc& src, int W, int H, const PatchGeom& g, int i, int j) {
    if (g.fast) {
        const size_t p = (size_t)(g.ipy + i) * W + g.ipx;
        const float t = g.a12 * src.at(p + j + 1) + g.a22 * src.at(p + j + 1 + W);
        float prev;
        if (j == 0) {
            prev = (1 - g.a) * (g.b1 * src.at(p) + g.b2 * src.at(p + W));
        } else {
            const float tp = g.a12 * src.at(p + j) + g.a22 * src.at(p + j + W);
            prev = (float)(tp * g.s);
        }
        return prev + t;
    }
    const int y0 = min(max(g.ipy + i, 0), H - 1), y1 = min(max(g.ipy + i + 1, 0), H - 1);
    const size_t r0 = (size_t)y0 * W, r1 = (size_t)y1 * W;
    if (j < g.rx) return src.at(r0) * g.b1 + src.at(r1) * g.b2;
    if (j >= g.rwid) return src.at(r0 + W - 1) * g.b1 + src.at(r1 + W - 1) * g.b2;
    const int x0 = g.ipx + j;
    return src.at(r0 + x0) * g.a11 + src.at(r0 + x0 + 1) * g.a12 + src.at(r1 + x0) * g.a21 + src.at(r1 + x0 + 1) * g.a22;
}

// cv::cornerSubPix for one corner, by one wavefront: window half-size win (<= kMaxWin), criteria (MAX_ITER | EPS,
// max_iters, sqrt(eps2)), zero zone off.  s_buf / s_reg are the wave's LDS scratch (kMaxPatch^2 floats,
// kRegDim^2 bytes).  Every lane returns the refined corner.
__device__ __forceinline__ void refine_corner(const GraySrc& src, int W, int H, float cTx, float cTy, int win, int max_iters,
                                              double eps2, const float* __restrict__ expTab, int lane, float* s_buf,
                                              uint8_t* s_reg, float* s_et, float& outX, float& outY) {
    const int ww = 2 * win + 1, pw = ww + 2;
    // the window's weights (one row of the table) in LDS: the moment loop reads two of them per element
    __syncthreads();
    if (lane < ww) s_et[lane] = expTab[win * (2 * kMaxWin + 1) + lane];
    const float* const et = s_et;
    // (row, column) of this lane's first element and of a step of 64 elements, for the patch and the window
    const int pi0 = lane / pw, pj0 = lane - pi0 * pw, pqi = 64 / pw, pqj = 64 - pqi * pw;
    const int wi0 = lane / ww, wj0 = lane - wi0 * ww, wqi = 64 / ww, wqj = 64 - wqi * ww;
    float cIx = cTx, cIy = cTy;
    int iter = 0;
    double err = 0;
    // the region around the first patch, clipped to the image
    const PatchGeom g0 = patch_geom(cIx, cIy, pw, W, H);
    const int rx0 = max(g0.ipx - kRegMargin, 0), ry0 = max(g0.ipy - kRegMargin, 0);
    const int rx1 = min(g0.ipx + pw + kRegMargin, W - 1), ry1 = min(g0.ipy + pw + kRegMargin, H - 1);
    const int rw = rx1 - rx0 + 1, rh = ry1 - ry0 + 1;
    const bool hasReg = rw > 0 && rh > 0 && rw <= kRegDim && rh <= kRegDim;
    __syncthreads();
    if (hasReg) {
        const int qi = 64 / rw, qj = 64 - qi * rw;
        for (int e = lane, i = lane / rw, j = lane - (lane / rw) * rw; e < rw * rh; e += 64) {
            s_reg[e] = (uint8_t)src.at((size_t)(ry0 + i) * W + rx0 + j);
            i += qi;
            j += qj;
            if (j >= rw) { j -= rw; i++; }
        }
    }
    do {
        const PatchGeom g = patch_geom(cIx, cIy, pw, W, H);
        __syncthreads();
        // element e = lane + 64 k of the patch is (row, column) = (i, j); stepping by 64 elements without dividing
        if (hasReg && g.fast && g.ipx >= rx0 && g.ipx + pw <= rx1 && g.ipy >= ry0 && g.ipy + pw <= ry1) {
            const uint8_t* rp = s_reg + (g.ipy - ry0) * rw + (g.ipx - rx0);
            for (int e = lane, i = pi0, j = pj0; e < pw * pw; e += 64) {
                s_buf[e] = patch_elem_inside(rp + i * rw, rw, g, j);
                i += pqi;
                j += pqj;
                if (j >= pw) { j -= pw; i++; }
            }
        } else {
            for (int e = lane, i = pi0, j = pj0; e < pw * pw; e += 64) {
                s_buf[e] = patch_elem(src, W, H, g, i, j);
                i += pqi;
                j += pqj;
                if (j >= pw) { j -= pw; i++; }
            }
        }
        __syncthreads();
        double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
        for (int e = lane, i = wi0, j = wj0; e < ww * ww; e += 64, i += wqi, j += wqj) {
            if (j >= ww) { j -= ww; i++; }
            const float* sp = s_buf + (i + 1) * pw + (j + 1);
            const double m = (double)(float)(et[i] * et[j]);
            const double tgx = (double)(sp[1] - sp[-1]);
            const double tgy = (double)(sp[pw] - sp[-pw]);
            const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
            const double px = j - win, py = i - win;
            a += gxx;
            b += gxy;
            c += gyy;
            bb1 += gxx * px + gxy * py;
            bb2 += gxy * px + gyy * py;
        }
        a = wave_sum_d(a);
        b = wave_sum_d(b);
        c = wave_sum_d(c);
        bb1 = wave_sum_d(bb1);
        bb2 = wave_sum_d(bb2);
        const double det = a * c - b * b;
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
        const double scale = 1.0 / det;
        const float nx = (float)(cIx + c * scale * bb1 - b * scale * bb2);
        const float ny = (float)(cIy - b * scale * bb1 + a * scale * bb2);
        err = (double)((nx - cIx) * (nx - cIx) + (ny - cIy) * (ny - cIy));
        cIx = nx;
        cIy = ny;
        if (cIx < 0 || cIx >= W || cIy < 0 || cIy >= H) break;
    } while (++iter < max_iters && err > eps2);
    if (fabsf(cIx - cTx) > win || fabsf(cIy - cTy) > win) {
        cIx = cTx;
        cIy = cTy;
    }
    outX = cIx;
    outY = cIy;
}

__global__ __launch_bounds__(64) void k_subpix(const uint8_t* __restrict__ gray, int grayC, const MarkerRec* __restrict__ markers,
                                               unsigned int markerCap, const Counters* __restrict__ ctr,
                                               const float* __restrict__ rawCorners, float* __restrict__ outCorners,
                                               const float* __restrict__ expTab, DetCfg cfg) {
    __shared__ float s_buf[kMaxPatch * kMaxPatch];
    __shared__ uint8_t s_reg[kRegDim * kRegDim];
    __shared__ float s_et[64];
    const int lane = threadIdx.x;
    const unsigned int nM = min(ctr->nMarkers, markerCap);
    const int H = cfg.H, W = cfg.W;
    for (unsigned int job = blockIdx.x; job < nM * 4u; job += gridDim.x) {
        const MarkerRec mr = markers[job >> 2];
        const int corner = job & 3;
        const float* mc = rawCorners + ((size_t)mr.frame * cfg.markerOut + mr.slot) * 8;
        const GraySrc src{gray + (size_t)mr.frame * H * W * grayC, grayC, (uint32_t)(H * W - 1)};
        // window from the marker's average module size (unrefined corners)
        float ms = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float dx = mc[2 * k] - mc[2 * ((k + 1) & 3)], dy = mc[2 * k + 1] - mc[2 * ((k + 1) & 3) + 1];
            ms += sqrtf(dx * dx + dy * dy);
        }
        ms /= (4.f * (float)(cfg.markerSize + 2 * cfg.borderBits));
        int win = max(1, (int)rint((double)(cfg.relWin * ms)));
        win = min(win, cfg.subpixWin);
        win = min(win, kMaxWin);
        const int max_iters = min(max(cfg.subpixMaxIter, 1), 100);
        double eps = fmax(cfg.subpixEps, 0.);
        eps *= eps;
        float cIx, cIy;
        refine_corner(src, W, H, mc[2 * corner], mc[2 * corner + 1], win, max_iters, eps, expTab, lane, s_buf, s_reg, s_et, cIx, cIy);
        if (lane == 0) {
            float* dst = outCorners + ((size_t)mr.frame * cfg.markerOut + mr.slot) * 8;
            dst[2 * corner] = cIx;
            dst[2 * corner + 1] = cIy;
        }
    }
}

// cv.cornerSubPix on caller-supplied points: pts / out f32 [B,N,2], npts [B] (null: N each); one wave per point
__global__ __launch_bounds__(64) void k_subpix_points(const uint8_t* __restrict__ gray, int grayC, int B, int H, int W,
                                                      const float* __restrict__ pts, const int32_t* __restrict__ npts, int N,
                                                      int win, int maxIters, double eps2, const float* __restrict__ expTab,
                                                      float* __restrict__ out) {
    __shared__ float s_buf[kMaxPatch * kMaxPatch];
    __shared__ uint8_t s_reg[kRegDim * kRegDim];
    __shared__ float s_et[64];
    const int lane = threadIdx.x;
    const unsigned int total = (unsigned int)B * (unsigned int)N;
    for (unsigned int job = blockIdx.x; job < total; job += gridDim.x) {
        const unsigned int f = job / (unsigned int)N, k = job - f * (unsigned int)N;
        if (npts && (int)k >= npts[f]) continue;  // (uniform: the whole wave works on one job)
        const GraySrc src{gray + (size_t)f * H * W * grayC, grayC, (uint32_t)(H * W - 1)};
        float cx, cy;
        refine_corner(src, W, H, pts[2 * (size_t)job], pts[2 * (size_t)job + 1], win, maxIters, eps2, expTab, lane, s_buf, s_reg, s_et, cx, cy);
        if (lane == 0) {
            out[2 * (size_t)job] = cx;
            out[2 * (size_t)job + 1] = cy;
        }
    }
}

}  // namespace

void launch_subpix(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    if (!c.subpix) return;
    // one wave per corner (24 per frame in the benchmark scene), a workgroup per corner where the batch allows it: the
    // iteration counts differ, and a fixed stride over few workgroups leaves CUs idle at the end (4096 frames, ms:
    // 8192 workgroups 1.79, 32768: 1.52, 131072: 1.44)
    const unsigned blocks = (unsigned)std::min(std::max(B * 32, 64), 256 * 512);
    hipLaunchKernelGGL(k_subpix, dim3(blocks), dim3(64), 0, st, b.graySrc, b.graySrcC, b.markers, b.markerCap, b.counters,
                       b.rawCorners, b.outCorners, b.expTab, c);
}

void launch_subpix_points(const uint8_t* gray, int grayC, int B, int H, int W, const float* pts, const int32_t* npts, int N,
                          int win, int maxIters, double eps, const float* expTab, float* out, hipStream_t st) {
    const unsigned total = (unsigned)B * (unsigned)N;
    if (!total) return;
    hipLaunchKernelGGL(k_subpix_points, dim3(std::min(total, 256u * 64u)), dim3(64), 0, st, gray, grayC, B, H, W, pts, npts, N,
                       std::min(std::max(win, 1), kMaxWin), std::min(std::max(maxIters, 1), 100), eps * eps, expTab, out);
}

}  // namespace camsync
