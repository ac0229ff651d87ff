// homography.h — plane-to-image homography from n <= 64 PPL correspondences, PPL
// points per lane (cv.findHomography(method 0) as used inside solvePnP's planar
// initialisation and calibrateCamera's focal-length initialisation): data
// normalisation as in runKernel, linear solve of the normalised system, then a
// Gauss-Newton polish of the reprojection error.  All lanes return the same H
// (h33 = 1).
#pragma once

#include "pose_math.h"

namespace camsync {

// Sum over a lane's PPL points of f(k), the first term taken as it is (so that one point per lane adds nothing).
template <int PPL, typename F>
__device__ __forceinline__ double lane_sum(F f) {
    double t = f(0);
#pragma unroll
    for (int k = 1; k < PPL; k++) t += f(k);
    return t;
}

// Point i of the set is held by lane i % 64 as its point i / 64.
template <int PPL>
__device__ __forceinline__ bool wave_homography(const double (&Mx_)[PPL], const double (&My_)[PPL], const double (&xn_)[PPL],
                                                const double (&yn_)[PPL], const bool (&active)[PPL], int n, double* H) {
    const double invn = 1.0 / n;
    double Mx[PPL], My[PPL], xn[PPL], yn[PPL], w0[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        Mx[k] = active[k] ? Mx_[k] : 0; My[k] = active[k] ? My_[k] : 0;
        xn[k] = active[k] ? xn_[k] : 0; yn[k] = active[k] ? yn_[k] : 0;
        w0[k] = active[k] ? 1.0 : 0.0;
    }
    // normalisation of both point sets (findHomography's runKernel)
    const double cMx = wsum(lane_sum<PPL>([&](int k) { return Mx[k]; })) * invn, cMy = wsum(lane_sum<PPL>([&](int k) { return My[k]; })) * invn;
    const double cmx = wsum(lane_sum<PPL>([&](int k) { return xn[k]; })) * invn, cmy = wsum(lane_sum<PPL>([&](int k) { return yn[k]; })) * invn;
    double sMx = wsum(lane_sum<PPL>([&](int k) { return active[k] ? fabs(Mx[k] - cMx) : 0; }));
    double sMy = wsum(lane_sum<PPL>([&](int k) { return active[k] ? fabs(My[k] - cMy) : 0; }));
    double smx = wsum(lane_sum<PPL>([&](int k) { return active[k] ? fabs(xn[k] - cmx) : 0; }));
    double smy = wsum(lane_sum<PPL>([&](int k) { return active[k] ? fabs(yn[k] - cmy) : 0; }));
    const double tiny = 2.220446049250313e-16;
    bool hok = !(fabs(sMx) < tiny || fabs(sMy) < tiny || fabs(smx) < tiny || fabs(smy) < tiny);
    if (hok) {
        sMx = n / sMx; sMy = n / sMy; smx = n / smx; smy = n / smy;
        double A[64], b[8];
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            const double xx = (xn[k] - cmx) * smx, yy = (yn[k] - cmy) * smy;
            const double XX = (Mx[k] - cMx) * sMx, YY = (My[k] - cMy) * sMy;
            const double r0[8] = {XX * w0[k], YY * w0[k], w0[k], 0, 0, 0, -xx * XX * w0[k], -xx * YY * w0[k]};
            const double r1[8] = {0, 0, 0, XX * w0[k], YY * w0[k], w0[k], -yy * XX * w0[k], -yy * YY * w0[k]};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const double t = r0[i] * xx + r1[i] * yy;
                b[i] = k == 0 ? t : b[i] + t;
#pragma unroll
                for (int j = i; j < 8; j++) {
                    const double s = r0[i] * r0[j] + r1[i] * r1[j];
                    A[i * 8 + j] = k == 0 ? s : A[i * 8 + j] + s;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            b[i] = wsum(b[i]);
#pragma unroll
            for (int j = i; j < 8; j++) {
                const double s = wsum(A[i * 8 + j]);
                A[i * 8 + j] = s;
                A[j * 8 + i] = s;
            }
        }
        hok = chol_factor<8>(A);
        chol_backsolve<8>(A, b);
        const double Hn[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0};
        const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
        const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
        double T[9];
        mat3_mul(invHnorm, Hn, T);
        mat3_mul(T, Hnorm2, H);
        if (fabs(H[8]) < 1e-300) hok = false;
        const double ih = 1.0 / H[8];
#pragma unroll
        for (int i = 0; i < 9; i++) H[i] *= ih;
    }
    if (hok) {
        // polish on the reprojection error (Gauss-Newton, 8 free entries)
        double cur = -1;
#pragma unroll 1
        for (int it = 0; it < 6; it++) {
            double A[64], b[8], cost = 0;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                double ww = H[6] * Mx[k] + H[7] * My[k] + 1.;
                ww = fabs(ww) > tiny ? 1. / ww : 0;
                const double xi = (H[0] * Mx[k] + H[1] * My[k] + H[2]) * ww, yi = (H[3] * Mx[k] + H[4] * My[k] + H[5]) * ww;
                const double ex = (xi - xn[k]) * w0[k], ey = (yi - yn[k]) * w0[k];
                const double c = ex * ex + ey * ey;
                cost = k == 0 ? c : cost + c;
                const double jx[8] = {Mx[k] * ww * w0[k], My[k] * ww * w0[k], ww * w0[k], 0, 0, 0, -Mx[k] * ww * xi * w0[k], -My[k] * ww * xi * w0[k]};
                const double jy[8] = {0, 0, 0, Mx[k] * ww * w0[k], My[k] * ww * w0[k], ww * w0[k], -Mx[k] * ww * yi * w0[k], -My[k] * ww * yi * w0[k]};
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const double t = jx[i] * ex + jy[i] * ey;
                    b[i] = k == 0 ? t : b[i] + t;
#pragma unroll
                    for (int j = i; j < 8; j++) {
                        const double s = jx[i] * jx[j] + jy[i] * jy[j];
                        A[i * 8 + j] = k == 0 ? s : A[i * 8 + j] + s;
                    }
                }
            }
            cost = wsum(cost);
            if (cur >= 0 && !(cost < cur)) break;
            cur = cost;
            double tr = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                b[i] = wsum(b[i]);
#pragma unroll
                for (int j = i; j < 8; j++) {
                    const double s = wsum(A[i * 8 + j]);
                    A[i * 8 + j] = s;
                    A[j * 8 + i] = s;
                }
                tr += A[i * 8 + i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) A[i * 8 + i] += 1e-14 * tr;
            if (!chol_factor<8>(A)) break;
            chol_backsolve<8>(A, b);
            double Hs[8];
#pragma unroll
            for (int i = 0; i < 8; i++) Hs[i] = H[i] - b[i];
            // accept only if the cost does not grow
            const double c2 = wsum(lane_sum<PPL>([&](int k) {
                double w2 = Hs[6] * Mx[k] + Hs[7] * My[k] + 1.;
                w2 = fabs(w2) > tiny ? 1. / w2 : 0;
                const double x2 = (Hs[0] * Mx[k] + Hs[1] * My[k] + Hs[2]) * w2, y2 = (Hs[3] * Mx[k] + Hs[4] * My[k] + Hs[5]) * w2;
                return ((x2 - xn[k]) * (x2 - xn[k]) + (y2 - yn[k]) * (y2 - yn[k])) * w0[k];
            }));
            if (!(c2 <= cost)) break;
#pragma unroll
            for (int i = 0; i < 8; i++) H[i] = Hs[i];
        }
    }
    return hok;
}

// one point per lane
__device__ __forceinline__ bool wave_homography(double Mx, double My, double xn, double yn, bool active, int n, double* H) {
    const double a[1] = {Mx}, b[1] = {My}, c[1] = {xn}, d[1] = {yn};
    const bool act[1] = {active};
    return wave_homography<1>(a, b, c, d, act, n, H);
}

}  // namespace camsync
