// k_undistort.hip — Camera.undistort (reference camera_sync/camera.py:235-256):
// cv.getOptimalNewCameraMatrix (host arithmetic, a 9 x 9 grid of points) and cv.undistort, which OpenCV runs as
// initUndistortRectifyMap(CV_16SC2) + remap(INTER_LINEAR, BORDER_CONSTANT) over stripes of a few rows.
//
// On the GPU the map depends only on (camera, new camera, frame size), so it is computed once per camera and kept:
//   k_undistort_map  one lane per row, walking the row left to right with OpenCV's own running sums
//                    (_x += ir[0] ...), so that every map entry is the value OpenCV's scalar code produces;
//                    entry = source pixel (16 + 16 bits) and the two 5-bit fractions
//   k_remap_bilinear one lane per four destination pixels: gathers the 2 x 2 source pixels of each (through the
//                    cache: neighbouring destinations read neighbouring sources), blends them with the 10-bit
//                    products of the fractions exactly as OpenCV's 15-bit fixed-point weights do, writes 4 x C bytes.
// The remap pass is a pure HBM stream: C bytes read + 8 bytes of map + C bytes written per pixel.
#include "internal.h"

#include <float.h>
#include <math.h>

#include <algorithm>
#include <vector>

namespace camsync {

namespace {

// cv::invert of a 3 x 3 double matrix (closed form)
__host__ __device__ inline bool invert3(const double* S, double* t) {
#define M_(r, c) S[(r) * 3 + (c)]
    double d = M_(0, 0) * (M_(1, 1) * M_(2, 2) - M_(1, 2) * M_(2, 1)) - M_(0, 1) * (M_(1, 0) * M_(2, 2) - M_(1, 2) * M_(2, 0)) +
               M_(0, 2) * (M_(1, 0) * M_(2, 1) - M_(1, 1) * M_(2, 0));
    if (d == 0.) return false;
    d = 1. / d;
    t[0] = (M_(1, 1) * M_(2, 2) - M_(1, 2) * M_(2, 1)) * d;
    t[1] = (M_(0, 2) * M_(2, 1) - M_(0, 1) * M_(2, 2)) * d;
    t[2] = (M_(0, 1) * M_(1, 2) - M_(0, 2) * M_(1, 1)) * d;
    t[3] = (M_(1, 2) * M_(2, 0) - M_(1, 0) * M_(2, 2)) * d;
    t[4] = (M_(0, 0) * M_(2, 2) - M_(0, 2) * M_(2, 0)) * d;
    t[5] = (M_(0, 2) * M_(1, 0) - M_(0, 0) * M_(1, 2)) * d;
    t[6] = (M_(1, 0) * M_(2, 1) - M_(1, 1) * M_(2, 0)) * d;
    t[7] = (M_(0, 1) * M_(2, 0) - M_(0, 0) * M_(2, 1)) * d;
    t[8] = (M_(0, 0) * M_(1, 1) - M_(0, 1) * M_(1, 0)) * d;
#undef M_
    return true;
}

struct UndistortCam {
    double K[9], Ar[9], k[8];
};

// map entry: x = sx (low 16) | sy (high 16) as shorts, y = fraction a (x, 0..31) | b << 5
__global__ __launch_bounds__(64) void k_undistort_map(UndistortCam cam, int H, int W, int stripe0, uint2* __restrict__ map) {
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= H) return;
    const int y0 = (row / stripe0) * stripe0, i = row - y0;
    double Ar[9], ir[9];
#pragma unroll
    for (int q = 0; q < 9; q++) Ar[q] = cam.Ar[q];
    Ar[5] = cam.Ar[5] - y0;
    invert3(Ar, ir);  // (checked on the host for the unshifted matrix; the shift does not change the determinant)
    const double fx = cam.K[0], fy = cam.K[4], u0 = cam.K[2], v0 = cam.K[5];
    const double k1 = cam.k[0], k2 = cam.k[1], p1 = cam.k[2], p2 = cam.k[3], k3 = cam.k[4], k4 = cam.k[5], k5 = cam.k[6], k6 = cam.k[7];
    double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
    for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1. / _w, x = _x * w, y = _y * w;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, _2xy = 2 * x * y;
        const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
        const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
        const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
        const double u = fx * xd + u0, v = fy * yd + v0;
        // saturate_cast<int>: round half to even, clamp
        const double us = fmin(fmax(rint(u * 32), -2147483648.0), 2147483647.0), vs = fmin(fmax(rint(v * 32), -2147483648.0), 2147483647.0);
        const int iu = (int)us, iv = (int)vs;
        const uint32_t sx = (uint32_t)(uint16_t)(int16_t)(iu >> 5), sy = (uint32_t)(uint16_t)(int16_t)(iv >> 5);
        map[(size_t)row * W + j] = make_uint2(sx | (sy << 16), (uint32_t)((iu & 31) | ((iv & 31) << 5)));
    }
}

template <int C>
__global__ __launch_bounds__(256) void k_remap_bilinear(const uint8_t* __restrict__ src, const uint2* __restrict__ map,
                                                        uint8_t* __restrict__ dst, int B, int H, int W) {
    const size_t quads = (size_t)H * (W / 4);
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= quads * B) return;
    const size_t f = t / quads, q = t - f * quads;
    const size_t p0 = q * 4;  // first of four consecutive pixels of one row (W is a multiple of 4)
    const uint8_t* S = src + f * (size_t)H * W * C;
    uint8_t out[4 * C];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const uint2 m = map[p0 + e];
        const int sx = (int)(int16_t)(m.x & 0xFFFFu), sy = (int)(int16_t)(m.x >> 16);
        const int a = (int)(m.y & 31u), b = (int)(m.y >> 5);
        const int w00 = (32 - a) * (32 - b), w01 = a * (32 - b), w10 = (32 - a) * b, w11 = a * b;
        const bool x0in = (unsigned)sx < (unsigned)W, x1in = (unsigned)(sx + 1) < (unsigned)W;
        const bool y0in = (unsigned)sy < (unsigned)H, y1in = (unsigned)(sy + 1) < (unsigned)H;
        const uint8_t* r0 = S + ((size_t)(y0in ? sy : 0) * W) * C;
        const uint8_t* r1 = S + ((size_t)(y1in ? sy + 1 : 0) * W) * C;
        const int xa = x0in ? sx : 0, xb = x1in ? sx + 1 : 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int v00 = (x0in && y0in) ? r0[xa * C + c] : 0, v01 = (x1in && y0in) ? r0[xb * C + c] : 0;
            const int v10 = (x0in && y1in) ? r1[xa * C + c] : 0, v11 = (x1in && y1in) ? r1[xb * C + c] : 0;
            const int v = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
            out[e * C + c] = (uint8_t)((v * 32 + (1 << 14)) >> 15);
        }
    }
    uint8_t* D = dst + (f * (size_t)H * W + p0) * C;
    // 4 C bytes, 4-byte aligned (p0 is a multiple of 4)
    uint32_t* D4 = reinterpret_cast<uint32_t*>(D);
#pragma unroll
    for (int w = 0; w < C; w++)
        D4[w] = (uint32_t)out[4 * w] | ((uint32_t)out[4 * w + 1] << 8) | ((uint32_t)out[4 * w + 2] << 16) | ((uint32_t)out[4 * w + 3] << 24);
}

struct RectD { double x, y, w, h; };

// cvUndistortPointsInternal with criteria (COUNT, 5); P (3 x 3) maps the normalised point to pixels when given
void undistort_pts(std::vector<double>& p, const double K[9], const double k[8], const double* P) {
    const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    for (size_t i = 0; i < p.size() / 2; i++) {
        double x = (p[2 * i] - cx) * ifx, y = (p[2 * i + 1] - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = x0; y = y0; break; }
            const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - dX) * icdist;
            y = (y0 - dY) * icdist;
        }
        if (P) {
            const double xx = P[0] * x + P[1] * y + P[2], yy = P[3] * x + P[4] * y + P[5], ww = 1. / (P[6] * x + P[7] * y + P[8]);
            x = xx * ww;
            y = yy * ww;
        }
        p[2 * i] = x;
        p[2 * i + 1] = y;
    }
}

// icvGetRectangles
void get_rectangles(const double K[9], const double k[8], const double* P, int W, int H, RectD& inner, RectD& outer) {
    const int N = 9;
    std::vector<double> pts(2 * N * N);
    int q = 0;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            pts[q++] = (double)x * (W - 1) / (N - 1);
            pts[q++] = (double)y * (H - 1) / (N - 1);
        }
    undistort_pts(pts, K, k, P);
    double iX0 = -FLT_MAX, iX1 = FLT_MAX, iY0 = -FLT_MAX, iY1 = FLT_MAX;
    double oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    q = 0;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++, q += 2) {
            const double px = pts[q], py = pts[q + 1];
            oX0 = std::min(oX0, px); oX1 = std::max(oX1, px);
            oY0 = std::min(oY0, py); oY1 = std::max(oY1, py);
            if (x == 0) iX0 = std::max(iX0, px);
            if (x == N - 1) iX1 = std::min(iX1, px);
            if (y == 0) iY0 = std::max(iY0, py);
            if (y == N - 1) iY1 = std::min(iY1, py);
        }
    inner = RectD{iX0, iY0, iX1 - iX0, iY1 - iY0};
    outer = RectD{oX0, oY0, oX1 - oX0, oY1 - oY0};
}

}  // namespace

void optimal_new_camera_matrix(const double* K, const double* k8, int W, int H, double alpha, int newW, int newH,
                               double* newK, int32_t* roi) {
    RectD inner, outer;
    get_rectangles(K, k8, nullptr, W, H, inner, outer);
    const double fx0 = (newW - 1) / inner.w, fy0 = (newH - 1) / inner.h, cx0 = -fx0 * inner.x, cy0 = -fy0 * inner.y;
    const double fx1 = (newW - 1) / outer.w, fy1 = (newH - 1) / outer.h, cx1 = -fx1 * outer.x, cy1 = -fy1 * outer.y;
    const double M[9] = {fx0 * (1 - alpha) + fx1 * alpha, 0, cx0 * (1 - alpha) + cx1 * alpha,
                         0, fy0 * (1 - alpha) + fy1 * alpha, cy0 * (1 - alpha) + cy1 * alpha, 0, 0, 1};
    for (int i = 0; i < 9; i++) newK[i] = M[i];
    if (roi) {
        get_rectangles(K, k8, M, W, H, inner, outer);
        // cv::Rect_<double> -> cv::Rect rounds every field; then the intersection with the new image
        int x = (int)lrint(inner.x), y = (int)lrint(inner.y), w = (int)lrint(inner.w), h = (int)lrint(inner.h);
        const int x1 = std::max(x, 0), y1 = std::max(y, 0);
        w = std::min(x + w, newW) - x1;
        h = std::min(y + h, newH) - y1;
        x = x1;
        y = y1;
        if (w <= 0 || h <= 0) x = y = w = h = 0;
        roi[0] = x; roi[1] = y; roi[2] = w; roi[3] = h;
    }
}

bool undistort_camera_ok(const double* newK) {
    double t[9];
    return invert3(newK, t);
}

void launch_undistort_map(const double* K, const double* newK, const double* k8, int H, int W, uint2* map, hipStream_t st) {
    UndistortCam cam;
    for (int i = 0; i < 9; i++) { cam.K[i] = K[i]; cam.Ar[i] = newK[i]; }
    for (int i = 0; i < 8; i++) cam.k[i] = k8[i];
    const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(W, 1)), H);
    hipLaunchKernelGGL(k_undistort_map, dim3((H + 63) / 64), dim3(64), 0, st, cam, H, W, stripe0, map);
}

void launch_remap(const uint8_t* src, const uint2* map, uint8_t* dst, int B, int H, int W, int C, hipStream_t st) {
    const size_t threads = (size_t)B * H * (W / 4);
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    if (C == 3) hipLaunchKernelGGL(k_remap_bilinear<3>, dim3(blocks), dim3(256), 0, st, src, map, dst, B, H, W);
    else hipLaunchKernelGGL(k_remap_bilinear<1>, dim3(blocks), dim3(256), 0, st, src, map, dst, B, H, W);
}

}  // namespace camsync
