/*
 * oracle/undistort.cpp — CPU restatement of Camera.undistort's OpenCV calls
 * (reference camera_sync/camera.py:245-255): cv.getOptimalNewCameraMatrix(mtx, dist, (w,h), 1, (w,h)) and
 * cv.undistort(img, mtx, dist, None, newcameramtx).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restated from OpenCV 4.11's published algorithms (calib3d getOptimalNewCameraMatrix / icvGetRectangles /
 * undistortPoints; imgproc undistort -> initUndistortRectifyMap(CV_16SC2) per stripe -> remap(INTER_LINEAR,
 * BORDER_CONSTANT) with 5 fractional bits and 15-bit weights).  The reference has no test and no stored output for
 * undistort: PARITY UNPINNED against OpenCV; the GPU path is compared with this file bit for bit.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "oracle.h"

namespace {

struct RectD { double x, y, w, h; };

/* cvUndistortPointsInternal, criteria (COUNT, 5): pixel -> normalised, optionally -> pixels of camera P */
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

/* icvGetRectangles: a 9 x 9 grid of image points, undistorted; the rectangle inscribed in / circumscribed about it */
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
        for (int x = 0; x < N; x++) {
            const double px = pts[q], py = pts[q + 1];
            q += 2;
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

int cv_round(double v) { return (int)lrint(v); }

/* cv::invert of a 3 x 3 double matrix (the closed form OpenCV uses for n == 3) */
bool invert3(const double* S, double* t) {
    auto m = [&](int r, int c) { return S[r * 3 + c]; };
    double d = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
               m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
    if (d == 0.) return false;
    d = 1. / d;
    t[0] = (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) * d;
    t[1] = (m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * d;
    t[2] = (m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * d;
    t[3] = (m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) * d;
    t[4] = (m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * d;
    t[5] = (m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * d;
    t[6] = (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)) * d;
    t[7] = (m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * d;
    t[8] = (m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * d;
    return true;
}

int saturate_int(double v) {
    const long r = lrint(v);
    return (int)std::min<long>(std::max<long>(r, INT32_MIN), INT32_MAX);
}

}  // namespace

extern "C" {

/* cv.getOptimalNewCameraMatrix(K, dist, (W,H), alpha, (newW,newH)), centerPrincipalPoint = false
 * (camera.py:246-248).  roi = x, y, w, h of validPixROI. */
int orc_optimal_new_camera_matrix(const double* K, const double* dist, int ndist, int W, int H, double alpha, int newW,
                                  int newH, double* newK, int32_t* roi) {
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist && i < 8; i++) k[i] = dist[i];
    if (newW <= 0 || newH <= 0) { newW = W; newH = H; }
    RectD inner, outer;
    get_rectangles(K, k, nullptr, W, H, inner, outer);
    const double fx0 = (newW - 1) / inner.w, fy0 = (newH - 1) / inner.h, cx0 = -fx0 * inner.x, cy0 = -fy0 * inner.y;
    const double fx1 = (newW - 1) / outer.w, fy1 = (newH - 1) / outer.h, cx1 = -fx1 * outer.x, cy1 = -fy1 * outer.y;
    double M[9] = {fx0 * (1 - alpha) + fx1 * alpha, 0, cx0 * (1 - alpha) + cx1 * alpha,
                   0, fy0 * (1 - alpha) + fy1 * alpha, cy0 * (1 - alpha) + cy1 * alpha, 0, 0, 1};
    memcpy(newK, M, sizeof M);
    if (roi) {
        get_rectangles(K, k, M, W, H, inner, outer);
        // Rect_<double> -> Rect: every field rounded; then intersected with the new image
        int x = cv_round(inner.x), y = cv_round(inner.y), w = cv_round(inner.w), h = cv_round(inner.h);
        const int x1 = std::max(x, 0), y1 = std::max(y, 0);
        w = std::min(x + w, newW) - x1;
        h = std::min(y + h, newH) - y1;
        x = x1;
        y = y1;
        if (w <= 0 || h <= 0) x = y = w = h = 0;
        roi[0] = x; roi[1] = y; roi[2] = w; roi[3] = h;
    }
    return 0;
}

/* cv.undistort(src, K, dist, None, newK) for one u8 image [H,W,C] (camera.py:251). */
int orc_undistort(const uint8_t* src, int H, int W, int C, const double* K, const double* dist, int ndist,
                  const double* newK, uint8_t* dst) {
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist && i < 8; i++) k[i] = dist[i];
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    double Ar[9];
    memcpy(Ar, newK ? newK : K, sizeof Ar);
    const double ar12 = Ar[5];
    const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(W, 1)), H);
    std::vector<int16_t> m1((size_t)stripe0 * W * 2);
    std::vector<uint16_t> m2((size_t)stripe0 * W);
    for (int y = 0; y < H; y += stripe0) {
        const int stripe = std::min(stripe0, H - y);
        Ar[5] = ar12 - y;
        double ir[9];
        if (!invert3(Ar, ir)) return -1;
        /* initUndistortRectifyMap, m1type CV_16SC2 */
        for (int i = 0; i < stripe; i++) {
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            for (int j = 0; j < W; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
                const double w = 1. / _w, x = _x * w, yy = _y * w;
                const double x2 = x * x, y2 = yy * yy;
                const double r2 = x2 + y2, _2xy = 2 * x * yy;
                const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
                const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
                const double yd = (yy * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
                const double u = fx * xd + u0, v = fy * yd + v0;
                const int iu = saturate_int(u * 32), iv = saturate_int(v * 32);
                m1[((size_t)i * W + j) * 2] = (int16_t)(iu >> 5);
                m1[((size_t)i * W + j) * 2 + 1] = (int16_t)(iv >> 5);
                m2[(size_t)i * W + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
            }
        }
        /* remap INTER_LINEAR, BORDER_CONSTANT (0): weights (32 - a)(32 - b) ... * 32 of 2^15, rounded */
        for (int i = 0; i < stripe; i++)
            for (int j = 0; j < W; j++) {
                const int sx = m1[((size_t)i * W + j) * 2], sy = m1[((size_t)i * W + j) * 2 + 1];
                const int fxy = m2[(size_t)i * W + j], a = fxy & 31, b = fxy >> 5;
                const int w00 = (32 - a) * (32 - b), w01 = a * (32 - b), w10 = (32 - a) * b, w11 = a * b;
                uint8_t* D = dst + ((size_t)(y + i) * W + j) * C;
                for (int c = 0; c < C; c++) {
                    auto tap = [&](int xx, int yy) -> int {
                        return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? src[((size_t)yy * W + xx) * C + c] : 0;
                    };
                    const int v = tap(sx, sy) * w00 + tap(sx + 1, sy) * w01 + tap(sx, sy + 1) * w10 + tap(sx + 1, sy + 1) * w11;
                    D[c] = (uint8_t)((v * 32 + (1 << 14)) >> 15);
                }
            }
    }
    return 0;
}

}  // extern "C"
