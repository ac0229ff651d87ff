// pose_math.h — f64 device helpers shared by the pose and calibration
// kernels: Rodrigues (cv.Rodrigues, reference referential.py:70,83), the
// pinhole + (k1,k2,p1,p2,k3,k4,k5,k6) projection with analytic Jacobians
// (cv.projectPoints, reference aruco.py:387-393, camera.py:67-69), iterative
// undistortion, fixed-size Cholesky solves.  Everything is written with
// compile-time loop bounds so arrays stay in registers.
#pragma once

#include <hip/hip_runtime.h>

#include "wave_reduce.h"

namespace camsync {

__device__ __forceinline__ double wsum(double v) {
    return wave_allreduce(v, [](double a, double b) { return a + b; });
}

// In-place Cholesky solve of the symmetric positive definite N x N system
// A x = b (A row-major, full storage).  Returns false when a pivot is not
// positive; A is overwritten by its factor.
template <int N>
__device__ __forceinline__ bool chol_factor(double (&A)[N * N]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double s = A[j * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= A[j * N + k] * A[j * N + k];
        if (!(s > 0)) { ok = false; s = 1; }
        const double d = sqrt(s);
        A[j * N + j] = d;
        const double id = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double t = A[i * N + j];
#pragma unroll
            for (int k = 0; k < j; k++) t -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = t * id;
        }
    }
    return ok;
}

template <int N>
__device__ __forceinline__ void chol_backsolve(const double (&L)[N * N], double (&b)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i * N + k] * b[k];
        b[i] = s / L[i * N + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= L[k * N + i] * b[k];
        b[i] = s / L[i * N + i];
    }
}

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
    double t[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
#pragma unroll
    for (int i = 0; i < 9; i++) C[i] = t[i];
}

__device__ __forceinline__ double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// rvec -> R, optional dR/dr as 3 x 9 (cvRodrigues2 layout)
__device__ __forceinline__ void rodrigues_v2m(const double* r_in, double* R, double* J) {
    const double theta = sqrt(r_in[0] * r_in[0] + r_in[1] * r_in[1] + r_in[2] * r_in[2]);
    if (theta < 2.220446049250313e-16) {
#pragma unroll
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        if (J) {
#pragma unroll
            for (int i = 0; i < 27; i++) J[i] = 0;
            J[5] = -1; J[7] = 1; J[11] = 1; J[15] = -1; J[19] = -1; J[21] = 1;
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double r[3] = {r_in[0] * itheta, r_in[1] * itheta, r_in[2] * itheta};
    const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1], r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
    const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx[k];
    if (J) {
        const double drrt[27] = {r[0] + r[0], r[1], r[2], r[1], 0, 0, r[2], 0, 0,
                                 0, r[0], 0, r[0], r[1] + r[1], r[2], 0, r[2], 0,
                                 0, 0, r[0], 0, 0, r[1], r[0], r[1], r[2] + r[2]};
        const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const double ri = r[i];
            const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
#pragma unroll
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1.0 : 0.0) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx[k] + a4 * drx[i * 9 + k];
        }
    }
}

// R -> rvec.  R is first replaced by the nearest rotation (polar factor via
// Newton iteration, equal to the U V^T of its SVD), then converted with
// cvRodrigues2's formulas.
__device__ __forceinline__ void rodrigues_m2v(const double* Rin, double* rv) {
    double R[9];
#pragma unroll
    for (int i = 0; i < 9; i++) R[i] = Rin[i];
    for (int it = 0; it < 12; it++) {
        const double d = det3(R);
        if (fabs(d) < 1e-300) break;
        const double id = 1.0 / d;
        // inverse transpose = cofactor matrix / det
        double C[9];
        C[0] = (R[4] * R[8] - R[5] * R[7]) * id;
        C[1] = (R[5] * R[6] - R[3] * R[8]) * id;
        C[2] = (R[3] * R[7] - R[4] * R[6]) * id;
        C[3] = (R[2] * R[7] - R[1] * R[8]) * id;
        C[4] = (R[0] * R[8] - R[2] * R[6]) * id;
        C[5] = (R[1] * R[6] - R[0] * R[7]) * id;
        C[6] = (R[1] * R[5] - R[2] * R[4]) * id;
        C[7] = (R[2] * R[3] - R[0] * R[5]) * id;
        C[8] = (R[0] * R[4] - R[1] * R[3]) * id;
        double delta = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const double n = 0.5 * (R[i] + C[i]);
            delta += fabs(n - R[i]);
            R[i] = n;
        }
        if (delta < 1e-15) break;
    }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            rv[0] = rv[1] = rv[2] = 0;
        } else {
            double t = (R[0] + 1) * 0.5;
            rx = sqrt(fmax(t, 0.));
            t = (R[4] + 1) * 0.5;
            ry = sqrt(fmax(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(fmax(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rv[0] = rx * theta;
            rv[1] = ry * theta;
            rv[2] = rz * theta;
        }
    } else {
        const double vth = theta / (2 * s);
        rv[0] = rx * vth;
        rv[1] = ry * vth;
        rv[2] = rz * vth;
    }
}

struct Cam {
    double fx, fy, cx, cy;
    double k[8];  // k1 k2 p1 p2 k3 k4 k5 k6
};

__device__ __forceinline__ Cam load_cam(const double* K, const double* dist, int ndist) {
    Cam c;
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
#pragma unroll
    for (int i = 0; i < 8; i++) c.k[i] = (i < ndist) ? dist[i] : 0.0;
    return c;
}

// Projection of one point already in camera coordinates (Xc,Yc,Zc) and the
// 2x3 Jacobian d(u,v)/d(Xc,Yc,Zc).  When WITH_INTR, also d(u,v)/d(fx,fy,cx,cy,k1..k8).
template <bool WITH_JAC, bool WITH_INTR>
__device__ __forceinline__ void project_cam(const Cam& cam, double Xc, double Yc, double Zc, double& u, double& v,
                                            double* duP, double* dvP, double* duI, double* dvI) {
    const double z = Zc ? 1. / Zc : 1;
    const double x = Xc * z, y = Yc * z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double* k = cam.k;
    const double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
    const double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
    const double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
    const double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
    u = xd * cam.fx + cam.cx;
    v = yd * cam.fy + cam.cy;
    if (WITH_JAC) {
        const double dcd = k[0] + 2 * k[1] * r2 + 3 * k[4] * r4;
        const double dic = -icdist2 * icdist2 * (k[5] + 2 * k[6] * r2 + 3 * k[7] * r4);
        const double f = cdist * icdist2, df = dcd * icdist2 + cdist * dic;
        const double dxd_dx = f + 2 * x * x * df + 2 * k[2] * y + 6 * k[3] * x;
        const double dxd_dy = 2 * x * y * df + 2 * k[2] * x + 2 * k[3] * y;
        const double dyd_dx = dxd_dy;
        const double dyd_dy = f + 2 * y * y * df + 6 * k[2] * y + 2 * k[3] * x;
        const double dxP[3] = {z, 0, -x * z}, dyP[3] = {0, z, -y * z};
#pragma unroll
        for (int q = 0; q < 3; q++) {
            duP[q] = cam.fx * (dxd_dx * dxP[q] + dxd_dy * dyP[q]);
            dvP[q] = cam.fy * (dyd_dx * dxP[q] + dyd_dy * dyP[q]);
        }
    }
    if (WITH_INTR) {
        const double fx = cam.fx, fy = cam.fy;
        duI[0] = xd; dvI[0] = 0;
        duI[1] = 0; dvI[1] = yd;
        duI[2] = 1; dvI[2] = 0;
        duI[3] = 0; dvI[3] = 1;
        duI[4] = fx * x * icdist2 * r2; dvI[4] = fy * y * icdist2 * r2;
        duI[5] = fx * x * icdist2 * r4; dvI[5] = fy * y * icdist2 * r4;
        duI[6] = fx * a1; dvI[6] = fy * a3;
        duI[7] = fx * a2; dvI[7] = fy * a1;
        duI[8] = fx * x * icdist2 * r6; dvI[8] = fy * y * icdist2 * r6;
        const double g = -cdist * icdist2 * icdist2;
        duI[9] = fx * x * g * r2; dvI[9] = fy * y * g * r2;
        duI[10] = fx * x * g * r4; dvI[10] = fy * y * g * r4;
        duI[11] = fx * x * g * r6; dvI[11] = fy * y * g * r6;
    }
}

// cvUndistortPoints with the default 5 fixed-point iterations.
__device__ __forceinline__ void undistort_point(const Cam& cam, double u, double v, double& xo, double& yo) {
    const double* k = cam.k;
    double x = (u - cam.cx) / cam.fx, y = (v - cam.cy) / cam.fy;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = x0; y = y0; break; }
        const double dX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double dY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - dX) * icdist;
        y = (y0 - dY) * icdist;
    }
    xo = x;
    yo = y;
}

}  // namespace camsync
