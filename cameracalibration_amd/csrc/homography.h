// homography.h — plane-to-image homography from n <= 64 correspondences, one
// point per lane (cv.findHomography(method 0) as used inside solvePnP's planar
// initialisation and calibrateCamera's focal-length initialisation): data
// normalisation as in runKernel, linear solve of the normalised system, then a
// Gauss-Newton polish of the reprojection error.  All lanes return the same H
// (h33 = 1).
#pragma once

#include "pose_math.h"

namespace camsync {

__device__ __forceinline__ bool wave_homography(double Mx, double My, double xn, double yn, bool active, int n, double* H) {
    const double invn = 1.0 / n;
    if (!active) { Mx = 0; My = 0; xn = 0; yn = 0; }
    // normalisation of both point sets (findHomography's runKernel)
    const double cMx = wsum(Mx) * invn, cMy = wsum(My) * invn, cmx = wsum(xn) * invn, cmy = wsum(yn) * invn;
    double sMx = wsum(active ? fabs(Mx - cMx) : 0), sMy = wsum(active ? fabs(My - cMy) : 0);
    double smx = wsum(active ? fabs(xn - cmx) : 0), smy = wsum(active ? fabs(yn - cmy) : 0);
    const double tiny = 2.220446049250313e-16;
    bool hok = !(fabs(sMx) < tiny || fabs(sMy) < tiny || fabs(smx) < tiny || fabs(smy) < tiny);
    if (hok) {
        sMx = n / sMx; sMy = n / sMy; smx = n / smx; smy = n / smy;
        const double xx = (xn - cmx) * smx, yy = (yn - cmy) * smy;
        const double XX = (Mx - cMx) * sMx, YY = (My - cMy) * sMy;
        const double w0 = active ? 1.0 : 0.0;
        const double r0[8] = {XX * w0, YY * w0, w0, 0, 0, 0, -xx * XX * w0, -xx * YY * w0};
        const double r1[8] = {0, 0, 0, XX * w0, YY * w0, w0, -yy * XX * w0, -yy * YY * w0};
        double A[64], b[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            b[i] = wsum(r0[i] * xx + r1[i] * yy);
#pragma unroll
            for (int j = i; j < 8; j++) {
                const double s = wsum(r0[i] * r0[j] + r1[i] * r1[j]);
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
            double ww = H[6] * Mx + H[7] * My + 1.;
            ww = fabs(ww) > tiny ? 1. / ww : 0;
            const double xi = (H[0] * Mx + H[1] * My + H[2]) * ww, yi = (H[3] * Mx + H[4] * My + H[5]) * ww;
            const double w0 = active ? 1.0 : 0.0;
            const double ex = (xi - xn) * w0, ey = (yi - yn) * w0;
            const double cost = wsum(ex * ex + ey * ey);
            if (cur >= 0 && !(cost < cur)) break;
            cur = cost;
            const double jx[8] = {Mx * ww * w0, My * ww * w0, ww * w0, 0, 0, 0, -Mx * ww * xi * w0, -My * ww * xi * w0};
            const double jy[8] = {0, 0, 0, Mx * ww * w0, My * ww * w0, ww * w0, -Mx * ww * yi * w0, -My * ww * yi * w0};
            double A[64], b[8];
            double tr = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                b[i] = wsum(jx[i] * ex + jy[i] * ey);
#pragma unroll
                for (int j = i; j < 8; j++) {
                    const double s = wsum(jx[i] * jx[j] + jy[i] * jy[j]);
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
            double w2 = Hs[6] * Mx + Hs[7] * My + 1.;
            w2 = fabs(w2) > tiny ? 1. / w2 : 0;
            const double x2 = (Hs[0] * Mx + Hs[1] * My + Hs[2]) * w2, y2 = (Hs[3] * Mx + Hs[4] * My + Hs[5]) * w2;
            const double c2 = wsum(((x2 - xn) * (x2 - xn) + (y2 - yn) * (y2 - yn)) * w0);
            if (!(c2 <= cost)) break;
#pragma unroll
            for (int i = 0; i < 8; i++) H[i] = Hs[i];
        }
    }
    return hok;
}

}  // namespace camsync
