/*
 * oracle/pose.cpp — CPU restatement of the pose / calibration arithmetic the
 * reference reaches through cv.solvePnP + cv.solvePnPRefineLM
 * (camera_sync/aruco.py:368-384), cv.solvePnP(SOLVEPNP_IPPE_SQUARE)
 * (aruco.py:111-117), cv.projectPoints (aruco.py:387-393), cv.Rodrigues
 * (referential.py:70,83) and cv.calibrateCamera (camera.py:61-63).
 * OpenCV 4.11 calib3d: calibration.cpp (cvFindExtrinsicCameraParams2,
 * cvProjectPoints2, cvRodrigues2, cvCalibrateCamera2Internal), compat_ptsetreg
 * (CvLevMarq), levmarq.cpp (LMSolver), ippe.cpp, fundam.cpp (homography).
 * TEST INFRASTRUCTURE ONLY — see oracle.h.
 */
#include "oracle.h"

#include <math.h>
#include <float.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {

/* ---------------------------------------------------------- small linalg */
/* Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major).
 * On return w[] holds eigenvalues (descending) and V rows hold eigenvectors. */
void jacobi_eigen(const double* A_in, int n, double* w, double* V) {
    std::vector<double> A(A_in, A_in + (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double theta = (aqq - app) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; k++) {
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vpk = V[p * n + k], vqk = V[q * n + k];
                    V[p * n + k] = c * vpk - s * vqk;
                    V[q * n + k] = s * vpk + c * vqk;
                }
            }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    for (int i = 0; i < n; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (w[j] > w[m]) m = j;
        if (m != i) {
            std::swap(w[i], w[m]);
            for (int k = 0; k < n; k++) std::swap(V[i * n + k], V[m * n + k]);
        }
    }
}

/* Solve a symmetric positive semi-definite system through its
 * eigen-decomposition (what cv::solve(DECOMP_SVD / DECOMP_EIG) amounts to
 * for JtJ), truncating singular directions like SVD::backSubst. */
struct SymEig {
    int n;
    std::vector<double> w, V;
    double thr;
};

void sym_eig(const double* A, int n, SymEig& e) {
    e.n = n;
    e.w.resize(n);
    e.V.resize((size_t)n * n);
    jacobi_eigen(A, n, e.w.data(), e.V.data());
    double thr = 0;
    for (int i = 0; i < n; i++) thr += fabs(e.w[i]);
    e.thr = thr * (DBL_EPSILON * 2);
}

void sym_backsub(const SymEig& e, const double* b, double* x) {
    const int n = e.n;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int k = 0; k < n; k++) {
        if (fabs(e.w[k]) <= e.thr) continue;
        double s = 0;
        for (int i = 0; i < n; i++) s += e.V[k * n + i] * b[i];
        s /= e.w[k];
        for (int i = 0; i < n; i++) x[i] += s * e.V[k * n + i];
    }
}

void solve_sym(const double* A, const double* b, int n, double* x) {
    SymEig e;
    sym_eig(A, n, e);
    sym_backsub(e, b, x);
}

/* 3x3 SVD R = U diag(w) Vt through the eigen-decomposition of RtR. */
void svd3(const double* A, double* U, double* w, double* Vt) {
    double AtA[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = s;
        }
    double ev[3];
    jacobi_eigen(AtA, 3, ev, Vt);
    for (int i = 0; i < 3; i++) w[i] = sqrt(std::max(ev[i], 0.));
    /* U columns = A v_i / w_i ; complete with cross product when w_i ~ 0 */
    double u[3][3];
    for (int i = 0; i < 3; i++) {
        for (int r = 0; r < 3; r++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[r * 3 + k] * Vt[i * 3 + k];
            u[i][r] = s;
        }
        double nrm = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + u[i][2] * u[i][2]);
        if (nrm > 1e-12 * (w[0] + 1e-300))
            for (int r = 0; r < 3; r++) u[i][r] /= nrm;
        else if (i == 2) {
            u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
            u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
            u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        }
    }
    for (int i = 0; i < 3; i++)
        for (int r = 0; r < 3; r++) U[r * 3 + i] = u[i][r];
}

void mat3_mul(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, t, sizeof t);
}

double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* ------------------------------------------------------------- K13 */
/* calibration.cpp cvRodrigues2, vector -> matrix, J = dR/dr as 3x9. */
void rodrigues_v2m(const double r_in[3], double R[9], double* J) {
    double theta = sqrt(r_in[0] * r_in[0] + r_in[1] * r_in[1] + r_in[2] * r_in[2]);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0);
        if (J) {
            const double j0[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
            memcpy(J, j0, sizeof j0);
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    double r[3] = {r_in[0] * itheta, r_in[1] * itheta, r_in[2] * itheta};
    double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1], r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
    double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    for (int k = 0; k < 9; k++) R[k] = c * (k % 4 == 0) + c1 * rrt[k] + s * rx[k];
    if (J) {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        double drrt[27] = {r[0] + r[0], r[1], r[2], r[1], 0, 0, r[2], 0, 0,
                           0, r[0], 0, r[0], r[1] + r[1], r[2], 0, r[2], 0,
                           0, 0, r[0], 0, 0, r[1], r[0], r[1], r[2] + r[2]};
        const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = r[i];
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx[k] + a4 * drx[i * 9 + k];
        }
    }
}

/* cvRodrigues2, matrix -> vector (R is re-orthonormalised by SVD first). */
void rodrigues_m2v(const double Rin[9], double r[3]) {
    double U[9], w[3], Vt[9], R[9];
    svd3(Rin, U, w, Vt);
    mat3_mul(U, Vt, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) {
            r[0] = r[1] = r[2] = 0;
        } else {
            double t;
            t = (R[0] + 1) * 0.5;
            rx = sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5;
            ry = sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5;
            rz = sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            r[0] = rx * theta;
            r[1] = ry * theta;
            r[2] = rz * theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        r[0] = rx * vth;
        r[1] = ry * vth;
        r[2] = rz * vth;
    }
}

/* ------------------------------------------------------------- K11 */
/* calibration.cpp cvProjectPoints2Internal without tilt / thin prism.
 * k = k1,k2,p1,p2,k3,k4,k5,k6.  Optional Jacobians, row-major per output
 * coordinate: dpdr [2n,3], dpdt [2n,3], dpdf [2n,2], dpdc [2n,2], dpdk [2n,8]. */
void project_points(const double* M, int n, const double r[3], const double t[3], const double fxy[2],
                    const double cxy[2], const double k[8], double* m, double* dpdr, double* dpdt,
                    double* dpdf, double* dpdc, double* dpdk) {
    double R[9], dRdr[27];
    rodrigues_v2m(r, R, dpdr ? dRdr : 0);
    const double fx = fxy[0], fy = fxy[1], cx = cxy[0], cy = cxy[1];
    for (int i = 0; i < n; i++) {
        double X = M[i * 3], Y = M[i * 3 + 1], Z = M[i * 3 + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        double xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2;
        double yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1;
        m[i * 2] = xd * fx + cx;
        m[i * 2 + 1] = yd * fy + cy;
        if (dpdc) {
            dpdc[(2 * i) * 2] = 1; dpdc[(2 * i) * 2 + 1] = 0;
            dpdc[(2 * i + 1) * 2] = 0; dpdc[(2 * i + 1) * 2 + 1] = 1;
        }
        if (dpdf) {
            dpdf[(2 * i) * 2] = xd; dpdf[(2 * i) * 2 + 1] = 0;
            dpdf[(2 * i + 1) * 2] = 0; dpdf[(2 * i + 1) * 2 + 1] = yd;
        }
        if (dpdk) {
            double* d0 = dpdk + (size_t)(2 * i) * 8;
            double* d1 = d0 + 8;
            d0[0] = fx * x * icdist2 * r2; d1[0] = fy * y * icdist2 * r2;
            d0[1] = fx * x * icdist2 * r4; d1[1] = fy * y * icdist2 * r4;
            d0[2] = fx * a1; d1[2] = fy * a3;
            d0[3] = fx * a2; d1[3] = fy * a1;
            d0[4] = fx * x * icdist2 * r6; d1[4] = fy * y * icdist2 * r6;
            d0[5] = fx * x * cdist * (-icdist2) * icdist2 * r2; d1[5] = fy * y * cdist * (-icdist2) * icdist2 * r2;
            d0[6] = fx * x * cdist * (-icdist2) * icdist2 * r4; d1[6] = fy * y * cdist * (-icdist2) * icdist2 * r4;
            d0[7] = fx * x * cdist * (-icdist2) * icdist2 * r6; d1[7] = fy * y * cdist * (-icdist2) * icdist2 * r6;
        }
        if (dpdt || dpdr) {
            /* d(xd,yd)/d(x,y) */
            double dcd = k[0] + 2 * k[1] * r2 + 3 * k[4] * r4;
            double dic = -icdist2 * icdist2 * (k[5] + 2 * k[6] * r2 + 3 * k[7] * r4);
            double f = cdist * icdist2, df = dcd * icdist2 + cdist * dic; /* df/dr2 */
            double dxd_dx = f + 2 * x * x * df + 2 * k[2] * y + 6 * k[3] * x;
            double dxd_dy = 2 * x * y * df + 2 * k[2] * x + 2 * k[3] * y;
            double dyd_dx = 2 * x * y * df + 2 * k[2] * x + 2 * k[3] * y;
            double dyd_dy = f + 2 * y * y * df + 6 * k[2] * y + 2 * k[3] * x;
            /* d(x,y)/d(Xc,Yc,Zc) with z = 1/Zc */
            double dx_dP[3] = {z, 0, -x * z};
            double dy_dP[3] = {0, z, -y * z};
            double du_dP[3], dv_dP[3];
            for (int q = 0; q < 3; q++) {
                du_dP[q] = fx * (dxd_dx * dx_dP[q] + dxd_dy * dy_dP[q]);
                dv_dP[q] = fy * (dyd_dx * dx_dP[q] + dyd_dy * dy_dP[q]);
            }
            if (dpdt)
                for (int q = 0; q < 3; q++) {
                    dpdt[(2 * i) * 3 + q] = du_dP[q];
                    dpdt[(2 * i + 1) * 3 + q] = dv_dP[q];
                }
            if (dpdr)
                for (int q = 0; q < 3; q++) {
                    /* dP/dr_q = dR/dr_q * M */
                    const double* D = dRdr + q * 9;
                    double dP[3] = {D[0] * X + D[1] * Y + D[2] * Z, D[3] * X + D[4] * Y + D[5] * Z, D[6] * X + D[7] * Y + D[8] * Z};
                    dpdr[(2 * i) * 3 + q] = du_dP[0] * dP[0] + du_dP[1] * dP[1] + du_dP[2] * dP[2];
                    dpdr[(2 * i + 1) * 3 + q] = dv_dP[0] * dP[0] + dv_dP[1] * dP[1] + dv_dP[2] * dP[2];
                }
        }
    }
}

/* undistort.dispatch.cpp cvUndistortPointsInternal, default criteria
 * (5 fixed iterations), R = P = I. */
void undistort_points(const double* img, int n, const double K[9], const double k[8], double* out) {
    double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    for (int i = 0; i < n; i++) {
        double x = (img[2 * i] - cx) * ifx, y = (img[2 * i + 1] - cy) * ify;
        double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = x0; y = y0; break; }
            double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        out[2 * i] = x;
        out[2 * i + 1] = y;
    }
}

/* fundam.cpp findHomography(method 0): normalised DLT (runKernel) followed by
 * a reprojection-error polish (HomographyRefineCallback + LM, 10 iters). */
bool find_homography(const double* M /*n x2*/, const double* m /*n x2*/, int n, double H[9]) {
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        cm[0] += m[2 * i]; cm[1] += m[2 * i + 1];
        cM[0] += M[2 * i]; cM[1] += M[2 * i + 1];
    }
    cm[0] /= n; cm[1] /= n; cM[0] /= n; cM[1] /= n;
    for (int i = 0; i < n; i++) {
        sm[0] += fabs(m[2 * i] - cm[0]); sm[1] += fabs(m[2 * i + 1] - cm[1]);
        sM[0] += fabs(M[2 * i] - cM[0]); sM[1] += fabs(M[2 * i + 1] - cM[1]);
    }
    if (fabs(sm[0]) < DBL_EPSILON || fabs(sm[1]) < DBL_EPSILON || fabs(sM[0]) < DBL_EPSILON || fabs(sM[1]) < DBL_EPSILON)
        return false;
    sm[0] = n / sm[0]; sm[1] = n / sm[1]; sM[0] = n / sM[0]; sM[1] = n / sM[1];
    double invHnorm[9] = {1. / sm[0], 0, cm[0], 0, 1. / sm[1], cm[1], 0, 0, 1};
    double Hnorm2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double LtL[81];
    memset(LtL, 0, sizeof LtL);
    for (int i = 0; i < n; i++) {
        double x = (m[2 * i] - cm[0]) * sm[0], y = (m[2 * i + 1] - cm[1]) * sm[1];
        double X = (M[2 * i] - cM[0]) * sM[0], Y = (M[2 * i + 1] - cM[1]) * sM[1];
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = 0; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    double w[9], V[81];
    jacobi_eigen(LtL, 9, w, V);
    double H0[9], T[9];
    memcpy(H0, V + 8 * 9, sizeof H0);
    mat3_mul(invHnorm, H0, T);
    mat3_mul(T, Hnorm2, H);
    if (fabs(H[8]) < DBL_MIN) return false;
    for (int i = 0; i < 8; i++) H[i] /= H[8];
    H[8] = 1;
    /* polish: Gauss-Newton with LM damping on the 8 free entries */
    double lambda = 1e-3;
    auto cost = [&](const double* h) {
        double s = 0;
        for (int i = 0; i < n; i++) {
            double Mx = M[2 * i], My = M[2 * i + 1];
            double ww = h[6] * Mx + h[7] * My + 1.;
            ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            double xi = (h[0] * Mx + h[1] * My + h[2]) * ww, yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
            s += (xi - m[2 * i]) * (xi - m[2 * i]) + (yi - m[2 * i + 1]) * (yi - m[2 * i + 1]);
        }
        return s;
    };
    double cur = cost(H);
    for (int it = 0; it < 10; it++) {
        double JtJ[64], Jte[8];
        memset(JtJ, 0, sizeof JtJ);
        memset(Jte, 0, sizeof Jte);
        for (int i = 0; i < n; i++) {
            double Mx = M[2 * i], My = M[2 * i + 1];
            double ww = H[6] * Mx + H[7] * My + 1.;
            ww = fabs(ww) > DBL_EPSILON ? 1. / ww : 0;
            double xi = (H[0] * Mx + H[1] * My + H[2]) * ww, yi = (H[3] * Mx + H[4] * My + H[5]) * ww;
            double jx[8] = {Mx * ww, My * ww, ww, 0, 0, 0, -Mx * ww * xi, -My * ww * xi};
            double jy[8] = {0, 0, 0, Mx * ww, My * ww, ww, -Mx * ww * yi, -My * ww * yi};
            double ex = xi - m[2 * i], ey = yi - m[2 * i + 1];
            for (int a = 0; a < 8; a++) {
                Jte[a] += jx[a] * ex + jy[a] * ey;
                for (int b = 0; b < 8; b++) JtJ[a * 8 + b] += jx[a] * jx[b] + jy[a] * jy[b];
            }
        }
        bool improved = false;
        for (int tries = 0; tries < 8 && !improved; tries++) {
            double A[64], d[8], Hn[9];
            memcpy(A, JtJ, sizeof A);
            for (int a = 0; a < 8; a++) A[a * 8 + a] *= 1 + lambda;
            solve_sym(A, Jte, 8, d);
            for (int a = 0; a < 8; a++) Hn[a] = H[a] - d[a];
            Hn[8] = 1;
            double c2 = cost(Hn);
            if (c2 <= cur) {
                double rel = cur - c2;
                memcpy(H, Hn, sizeof Hn);
                cur = c2;
                lambda = std::max(lambda * 0.1, 1e-16);
                improved = true;
                if (rel < 1e-30) it = 10;
            } else
                lambda *= 10;
        }
        if (!improved) break;
    }
    return true;
}

/* compat_ptsetreg.cpp CvLevMarq (completeSymmFlag, multiplicative diagonal
 * damping, SVD solve) expressed as a driver around a residual/Jacobian
 * callback.  fcn(p, err[m], J[m*np] or NULL). */
template <class F>
int cv_levmarq(double* param, int np, int m, int max_iter, double epsilon, F fcn) {
    std::vector<double> prevParam(np), J((size_t)m * np), err(m), JtJ((size_t)np * np), JtErr(np), JtJN((size_t)np * np), step(np);
    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX, errNorm = 0;
    auto do_step = [&]() {
        const double lambda = exp(lambdaLg10 * log(10.));
        JtJN = JtJ;
        for (int i = 0; i < np; i++) JtJN[(size_t)i * np + i] *= 1. + lambda;
        solve_sym(JtJN.data(), JtErr.data(), np, step.data());
        for (int i = 0; i < np; i++) param[i] = prevParam[i] - step[i];
    };
    auto norm2 = [&](const std::vector<double>& v) {
        double s = 0;
        for (size_t i = 0; i < v.size(); i++) s += v[i] * v[i];
        return sqrt(s);
    };
    for (;;) {
        /* CALC_J */
        fcn(param, err.data(), J.data());
        for (int a = 0; a < np; a++) {
            for (int b = a; b < np; b++) {
                double s = 0;
                for (int i = 0; i < m; i++) s += J[(size_t)i * np + a] * J[(size_t)i * np + b];
                JtJ[(size_t)a * np + b] = JtJ[(size_t)b * np + a] = s;
            }
            double s = 0;
            for (int i = 0; i < m; i++) s += J[(size_t)i * np + a] * err[i];
            JtErr[a] = s;
        }
        memcpy(prevParam.data(), param, sizeof(double) * np);
        if (iters == 0) prevErrNorm = norm2(err);
        do_step();
        /* CHECK_ERR */
        for (;;) {
            fcn(param, err.data(), (double*)0);
            errNorm = norm2(err);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) {
                    do_step();
                    continue;
                }
            }
            break;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double dn = 0, pn = 0;
        for (int i = 0; i < np; i++) {
            dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]);
            pn += prevParam[i] * prevParam[i];
        }
        double change = sqrt(dn) / (pn > 0 ? sqrt(pn) : DBL_EPSILON);
        if (++iters >= max_iter || change < epsilon) break;
        prevErrNorm = errNorm;
    }
    return iters;
}

/* levmarq.cpp LMSolverImpl::run as used by solvePnPRefineLM. */
template <class F>
int cv_lmsolver(double* x, int np, int m, int max_iter, double eps, F fcn) {
    std::vector<double> xd(np), r(m), rd(m), J((size_t)m * np), A((size_t)np * np), Ap((size_t)np * np), v(np), d(np), D(np), tmp(np);
    auto build = [&]() {
        for (int a = 0; a < np; a++) {
            for (int b = a; b < np; b++) {
                double s = 0;
                for (int i = 0; i < m; i++) s += J[(size_t)i * np + a] * J[(size_t)i * np + b];
                A[(size_t)a * np + b] = A[(size_t)b * np + a] = s;
            }
            double s = 0;
            for (int i = 0; i < m; i++) s += J[(size_t)i * np + a] * r[i];
            v[a] = s;
        }
    };
    auto sq = [&](const std::vector<double>& z) { double s = 0; for (size_t i = 0; i < z.size(); i++) s += z[i] * z[i]; return s; };
    fcn(x, r.data(), J.data());
    double S = sq(r);
    build();
    for (int i = 0; i < np; i++) D[i] = A[(size_t)i * np + i];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    int iter = 0;
    for (;;) {
        Ap = A;
        for (int i = 0; i < np; i++) Ap[(size_t)i * np + i] += lambda * D[i];
        solve_sym(Ap.data(), v.data(), np, d.data());
        for (int i = 0; i < np; i++) xd[i] = x[i] - d[i];
        fcn(xd.data(), rd.data(), (double*)0);
        double Sd = sq(rd);
        /* temp_d = -A d + 2 v ; dS = d . temp_d */
        double dS = 0;
        for (int i = 0; i < np; i++) {
            double s = 0;
            for (int k = 0; k < np; k++) s += A[(size_t)i * np + k] * d[k];
            tmp[i] = -s + 2 * v[i];
            dS += d[i] * tmp[i];
        }
        double R = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (R > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (R < Rlo) {
            double t = 0;
            for (int i = 0; i < np; i++) t += d[i] * v[i];
            double nu = (Sd - S) / (fabs(t) > DBL_EPSILON ? t : 1) + 2;
            nu = std::min(std::max(nu, 2.), 10.);
            if (lambda == 0) {
                /* invert(A) diagonal through the eigen-decomposition */
                std::vector<double> w(np), V((size_t)np * np);
                jacobi_eigen(A.data(), np, w.data(), V.data());
                double maxval = DBL_EPSILON;
                for (int i = 0; i < np; i++) {
                    double s = 0;
                    for (int k = 0; k < np; k++)
                        if (fabs(w[k]) > DBL_EPSILON * fabs(w[0])) s += V[(size_t)k * np + i] * V[(size_t)k * np + i] / w[k];
                    maxval = std::max(maxval, fabs(s));
                }
                lambda = lc = 1. / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd.data(), sizeof(double) * np);
            fcn(x, r.data(), J.data());
            build();
        }
        iter++;
        double dinf = 0, rinf = 0;
        for (int i = 0; i < np; i++) dinf = std::max(dinf, fabs(d[i]));
        for (int i = 0; i < m; i++) rinf = std::max(rinf, fabs(r[i]));
        if (!(iter < max_iter && dinf >= eps && rinf >= eps)) break;
    }
    return iter;
}

void expand_dist(const double* dist, int ndist, double k[8]) {
    for (int i = 0; i < 8; i++) k[i] = (dist && i < ndist) ? dist[i] : 0.;
}

/* calibration.cpp cvFindExtrinsicCameraParams2 (useExtrinsicGuess = 0). */
bool find_extrinsics(const double* obj, const double* img, int n, const double K[9], const double k[8], double param[6]) {
    std::vector<double> mn((size_t)2 * n);
    undistort_points(img, n, K, k, mn.data());
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) { Mc[0] += obj[3 * i]; Mc[1] += obj[3 * i + 1]; Mc[2] += obj[3 * i + 2]; }
    Mc[0] /= n; Mc[1] /= n; Mc[2] /= n;
    double MM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        double d[3] = {obj[3 * i] - Mc[0], obj[3 * i + 1] - Mc[1], obj[3 * i + 2] - Mc[2]};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) MM[a * 3 + b] += d[a] * d[b];
    }
    double W[3], V[9]; /* V rows = eigenvectors = V^T of the SVD */
    jacobi_eigen(MM, 3, W, V);
    double R[9], t[3];
    if (W[2] / W[1] < 1e-3) {
        /* planar structure */
        double Rt[9];
        memcpy(Rt, V, sizeof Rt);
        if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
            for (int i = 0; i < 9; i++) Rt[i] = (i % 4 == 0);
        }
        if (det3(Rt) < 0)
            for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
        double tt[3];
        for (int a = 0; a < 3; a++) tt[a] = -(Rt[a * 3] * Mc[0] + Rt[a * 3 + 1] * Mc[1] + Rt[a * 3 + 2] * Mc[2]);
        std::vector<double> Mxy((size_t)2 * n);
        for (int i = 0; i < n; i++) {
            const double* s = obj + 3 * i;
            Mxy[2 * i] = Rt[0] * s[0] + Rt[1] * s[1] + Rt[2] * s[2] + tt[0];
            Mxy[2 * i + 1] = Rt[3] * s[0] + Rt[4] * s[1] + Rt[5] * s[2] + tt[1];
        }
        double h[9];
        bool ok = find_homography(Mxy.data(), mn.data(), n, h);
        for (int i = 0; i < 9 && ok; i++) ok = std::isfinite(h[i]);
        if (ok) {
            double h1n = sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]);
            double h2n = sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            double s1 = 1. / std::max(h1n, DBL_EPSILON), s2 = 1. / std::max(h2n, DBL_EPSILON);
            double st = 2. / std::max(h1n + h2n, DBL_EPSILON);
            t[0] = h[2] * st; t[1] = h[5] * st; t[2] = h[8] * st;
            double h1[3] = {h[0] * s1, h[3] * s1, h[6] * s1}, h2[3] = {h[1] * s2, h[4] * s2, h[7] * s2};
            double h3[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
            double Hm[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
            double rv[3];
            rodrigues_m2v(Hm, rv);
            rodrigues_v2m(rv, Hm, 0);
            /* t = Hm * T_transform + t ; R = Hm * R_transform */
            double t2[3];
            for (int a = 0; a < 3; a++) t2[a] = Hm[a * 3] * tt[0] + Hm[a * 3 + 1] * tt[1] + Hm[a * 3 + 2] * tt[2] + t[a];
            memcpy(t, t2, sizeof t2);
            mat3_mul(Hm, Rt, R);
        } else {
            for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0);
            t[0] = t[1] = t[2] = 0;
        }
    } else {
        /* non-planar: DLT on the 3x4 projection */
        if (n < 6) return false;
        std::vector<double> LL((size_t)12 * 12, 0.);
        for (int i = 0; i < n; i++) {
            double x = -mn[2 * i], y = -mn[2 * i + 1];
            const double* P = obj + 3 * i;
            double L0[12] = {P[0], P[1], P[2], 1, 0, 0, 0, 0, x * P[0], x * P[1], x * P[2], x};
            double L1[12] = {0, 0, 0, 0, P[0], P[1], P[2], 1, y * P[0], y * P[1], y * P[2], y};
            for (int a = 0; a < 12; a++)
                for (int b = 0; b < 12; b++) LL[a * 12 + b] += L0[a] * L0[b] + L1[a] * L1[b];
        }
        double w[12], Vv[144];
        jacobi_eigen(LL.data(), 12, w, Vv);
        const double* RR = Vv + 11 * 12;
        double Rm[9] = {RR[0], RR[1], RR[2], RR[4], RR[5], RR[6], RR[8], RR[9], RR[10]};
        double tm[3] = {RR[3], RR[7], RR[11]};
        if (det3(Rm) < 0) {
            for (int i = 0; i < 9; i++) Rm[i] = -Rm[i];
            for (int i = 0; i < 3; i++) tm[i] = -tm[i];
        }
        double sc = sqrt(Rm[0] * Rm[0] + Rm[1] * Rm[1] + Rm[2] * Rm[2] + Rm[3] * Rm[3] + Rm[4] * Rm[4] + Rm[5] * Rm[5] + Rm[6] * Rm[6] + Rm[7] * Rm[7] + Rm[8] * Rm[8]);
        double U[9], ww[3], Vt[9];
        svd3(Rm, U, ww, Vt);
        mat3_mul(U, Vt, R);
        double nrmR = 3.; /* ||R||_F^2 = 3 */
        double s = sqrt(nrmR) / sc;
        for (int i = 0; i < 3; i++) t[i] = tm[i] * s;
    }
    rodrigues_m2v(R, param);
    param[3] = t[0]; param[4] = t[1]; param[5] = t[2];

    /* refine: CvLevMarq(6, 2n, (EPS+ITER, 20, FLT_EPSILON), completeSymm) on
     * pixel reprojection error with the full distortion model */
    double fxy[2] = {K[0], K[4]}, cxy[2] = {K[2], K[5]};
    std::vector<double> dpdr((size_t)2 * n * 3), dpdt((size_t)2 * n * 3), proj((size_t)2 * n);
    auto fcn = [&](const double* p, double* err, double* J) {
        project_points(obj, n, p, p + 3, fxy, cxy, k, proj.data(), J ? dpdr.data() : 0, J ? dpdt.data() : 0, 0, 0, 0);
        for (int i = 0; i < 2 * n; i++) err[i] = proj[i] - img[i];
        if (J)
            for (int i = 0; i < 2 * n; i++)
                for (int q = 0; q < 3; q++) {
                    J[(size_t)i * 6 + q] = dpdr[(size_t)i * 3 + q];
                    J[(size_t)i * 6 + 3 + q] = dpdt[(size_t)i * 3 + q];
                }
    };
    cv_levmarq(param, 6, 2 * n, 20, (double)FLT_EPSILON, fcn);
    return true;
}

/* ------------------------------------------------------------- K10 */
/* ippe.cpp PoseSolver::solveSquare: homography of the canonical square,
 * analytic two-fold rotation from its Jacobian at the centre, linear
 * translation, solutions ordered by reprojection error. */
void ippe_rotations(double j00, double j01, double j10, double j11, double p, double q, double R1[9], double R2[9]) {
    double Rv[9];
    double s = sqrt(p * p + q * q + 1), t = sqrt(p * p + q * q);
    double costh = 1. / s, sinth = sqrt(1 - 1. / (s * s));
    if (t <= DBL_EPSILON) {
        for (int i = 0; i < 9; i++) Rv[i] = (i % 4 == 0);
    } else {
        double kx = p / t, ky = q / t; /* Kcrs = [0 0 kx; 0 0 ky; -kx -ky 0] */
        double Kc[9] = {0, 0, kx, 0, 0, ky, -kx, -ky, 0}, K2[9];
        mat3_mul(Kc, Kc, K2);
        for (int i = 0; i < 9; i++) Rv[i] = (i % 4 == 0) + sinth * Kc[i] + (1 - costh) * K2[i];
    }
    double b00 = Rv[0] - p * Rv[6], b01 = Rv[1] - p * Rv[7], b10 = Rv[3] - q * Rv[6], b11 = Rv[4] - q * Rv[7];
    double dtinv = 1.0 / (b00 * b11 - b01 * b10);
    double bi00 = dtinv * b11, bi01 = -dtinv * b01, bi10 = -dtinv * b10, bi11 = dtinv * b00;
    double a00 = bi00 * j00 + bi01 * j10, a01 = bi00 * j01 + bi01 * j11;
    double a10 = bi10 * j00 + bi11 * j10, a11 = bi10 * j01 + bi11 * j11;
    double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    double gamma2 = 0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
    double gamma = sqrt(std::max(gamma2, 0.));
    double r00 = a00 / gamma, r01 = a01 / gamma, r10 = a10 / gamma, r11 = a11 / gamma;
    double b0 = sqrt(std::max(1 - r00 * r00 - r10 * r10, 0.));
    double b1 = sqrt(std::max(1 - r01 * r01 - r11 * r11, 0.));
    double sp = -r00 * r01 - r10 * r11;
    if (sp < 0) b1 = -b1;
    /* third column = col0 x col1 */
    auto build = [&](double sgn, double* R) {
        double c0[3] = {r00, r10, sgn * b0}, c1[3] = {r01, r11, sgn * b1};
        double c2[3] = {c0[1] * c1[2] - c0[2] * c1[1], c0[2] * c1[0] - c0[0] * c1[2], c0[0] * c1[1] - c0[1] * c1[0]};
        double Rt[9] = {c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2]};
        mat3_mul(Rv, Rt, R);
    };
    build(1., R1);
    build(-1., R2);
}

void ippe_translation(const double* objxy, const double* mn, int n, const double R[9], double t[3]) {
    /* least squares for t: tx - x tz = x rz - rx ; ty - y tz = y rz - ry */
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        double X = objxy[2 * i], Y = objxy[2 * i + 1], x = mn[2 * i], y = mn[2 * i + 1];
        double rx = R[0] * X + R[1] * Y, ry = R[3] * X + R[4] * Y, rz = R[6] * X + R[7] * Y;
        double r0[3] = {1, 0, -x}, r1[3] = {0, 1, -y};
        double e0 = x * rz - rx, e1 = y * rz - ry;
        for (int a = 0; a < 3; a++) {
            b[a] += r0[a] * e0 + r1[a] * e1;
            for (int c = 0; c < 3; c++) A[a * 3 + c] += r0[a] * r0[c] + r1[a] * r1[c];
        }
    }
    solve_sym(A, b, 3, t);
}

double ippe_reproj_err(const double* objxy, const double* mn, int n, const double R[9], const double t[3]) {
    double s = 0;
    for (int i = 0; i < n; i++) {
        double X = objxy[2 * i], Y = objxy[2 * i + 1];
        double px = R[0] * X + R[1] * Y + t[0], py = R[3] * X + R[4] * Y + t[1], pz = R[6] * X + R[7] * Y + t[2];
        double dx = px / pz - mn[2 * i], dy = py / pz - mn[2 * i + 1];
        s += dx * dx + dy * dy;
    }
    return sqrt(s / (2.0 * n));
}

/* 4-point homography (exact fit) canonical square -> normalised image. */
bool homography_4pt(const double* src, const double* dst, double H[9]) {
    double a[8][9];
    for (int i = 0; i < 4; i++) {
        double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        double r0[9] = {sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx};
        double r1[9] = {0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy};
        memcpy(a[i], r0, sizeof r0);
        memcpy(a[i + 4], r1, sizeof r1);
    }
    for (int i = 0; i < 8; i++) {
        int k = i;
        for (int j = i + 1; j < 8; j++)
            if (fabs(a[j][i]) > fabs(a[k][i])) k = j;
        if (fabs(a[k][i]) < 1e-300) return false;
        if (k != i)
            for (int j = 0; j < 9; j++) std::swap(a[i][j], a[k][j]);
        for (int j = i + 1; j < 8; j++) {
            double f = a[j][i] / a[i][i];
            for (int c = i; c < 9; c++) a[j][c] -= f * a[i][c];
        }
    }
    double x[8];
    for (int i = 7; i >= 0; i--) {
        double s = a[i][8];
        for (int k = i + 1; k < 8; k++) s -= a[i][k] * x[k];
        x[i] = s / a[i][i];
    }
    for (int i = 0; i < 8; i++) H[i] = x[i];
    H[8] = 1;
    return true;
}

} /* namespace */

extern "C" {

void orc_rodrigues_vec2mat(const double r[3], double R[9], double* jac) { rodrigues_v2m(r, R, jac); }
void orc_rodrigues_mat2vec(const double R[9], double r[3]) { rodrigues_m2v(R, r); }

void orc_project_points(const double* obj, int n, const double rvec[3], const double tvec[3], const double K[9],
                        const double* dist, int ndist, double* img_out) {
    double k[8];
    expand_dist(dist, ndist, k);
    double fxy[2] = {K[0], K[4]}, cxy[2] = {K[2], K[5]};
    project_points(obj, n, rvec, tvec, fxy, cxy, k, img_out, 0, 0, 0, 0, 0);
}

int orc_solve_pnp(const double* obj, const double* img, int n, const double K[9], const double* dist, int ndist,
                  int refine, double rvec[3], double tvec[3]) {
    if (n < 4) return 0;
    double k[8], param[6];
    expand_dist(dist, ndist, k);
    if (!find_extrinsics(obj, img, n, K, k, param)) return 0;
    if (refine) {
        /* solvePnPRefineLM: LMSolver(20 iters, FLT_EPSILON) on (rvec,tvec) */
        double fxy[2] = {K[0], K[4]}, cxy[2] = {K[2], K[5]};
        std::vector<double> dpdr((size_t)2 * n * 3), dpdt((size_t)2 * n * 3), proj((size_t)2 * n);
        auto fcn = [&](const double* p, double* err, double* J) {
            project_points(obj, n, p, p + 3, fxy, cxy, k, proj.data(), J ? dpdr.data() : 0, J ? dpdt.data() : 0, 0, 0, 0);
            for (int i = 0; i < 2 * n; i++) err[i] = proj[i] - img[i];
            if (J)
                for (int i = 0; i < 2 * n; i++)
                    for (int q = 0; q < 3; q++) {
                        J[(size_t)i * 6 + q] = dpdr[(size_t)i * 3 + q];
                        J[(size_t)i * 6 + 3 + q] = dpdt[(size_t)i * 3 + q];
                    }
        };
        cv_lmsolver(param, 6, 2 * n, 20, (double)FLT_EPSILON, fcn);
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
    return 1;
}

int orc_ippe_square(const double* corners, double size, const double K[9], const double* dist, int ndist,
                    double rvec[3], double tvec[3]) {
    double k[8], mn[8];
    expand_dist(dist, ndist, k);
    undistort_points(corners, 4, K, k, mn);
    /* object points as the reference builds them: float32(+-size/2)
     * (aruco.py:97-105) */
    double hl = (double)(float)(size / 2);
    double sq[8] = {-hl, hl, hl, hl, hl, -hl, -hl, -hl};
    double H[9];
    if (!homography_4pt(sq, mn, H)) return 0;
    double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2];
    double j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    double R1[9], R2[9], t1[3], t2[3];
    ippe_rotations(j00, j01, j10, j11, H[2], H[5], R1, R2);
    ippe_translation(sq, mn, 4, R1, t1);
    ippe_translation(sq, mn, 4, R2, t2);
    double e1 = ippe_reproj_err(sq, mn, 4, R1, t1), e2 = ippe_reproj_err(sq, mn, 4, R2, t2);
    const double* R = (e1 > e2) ? R2 : R1;
    const double* t = (e1 > e2) ? t2 : t1;
    rodrigues_m2v(R, rvec);
    tvec[0] = t[0]; tvec[1] = t[1]; tvec[2] = t[2];
    return 1;
}

} /* extern "C" */

/* ------------------------------------------------------------- K12 */
namespace {

/* calibration.cpp cvInitIntrinsicParams2D (aspectRatio unused, flags = 0). */
void init_intrinsics_2d(const double* obj, const double* img, int V, int P, int W, int Hh, double a[9]) {
    memset(a, 0, 9 * sizeof(double));
    a[8] = 1;
    a[2] = (!W) ? 0.5 : (W - 1) * 0.5;
    a[5] = (!Hh) ? 0.5 : (Hh - 1) * 0.5;
    double AtA[4] = {0, 0, 0, 0}, Atb[2] = {0, 0};
    std::vector<double> Mxy((size_t)2 * P);
    for (int i = 0; i < V; i++) {
        for (int j = 0; j < P; j++) { Mxy[2 * j] = obj[((size_t)i * P + j) * 3]; Mxy[2 * j + 1] = obj[((size_t)i * P + j) * 3 + 1]; }
        double H[9];
        if (!find_homography(Mxy.data(), img + (size_t)i * P * 2, P, H)) continue;
        H[0] -= H[6] * a[2]; H[1] -= H[7] * a[2]; H[2] -= H[8] * a[2];
        H[3] -= H[6] * a[5]; H[4] -= H[7] * a[5]; H[5] -= H[8] * a[5];
        double h[3], v[3], d1[3], d2[3], n[4] = {0, 0, 0, 0};
        for (int j = 0; j < 3; j++) {
            double t0 = H[j * 3], t1 = H[j * 3 + 1];
            h[j] = t0; v[j] = t1;
            d1[j] = (t0 + t1) * 0.5;
            d2[j] = (t0 - t1) * 0.5;
            n[0] += t0 * t0; n[1] += t1 * t1;
            n[2] += d1[j] * d1[j]; n[3] += d2[j] * d2[j];
        }
        for (int j = 0; j < 4; j++) n[j] = 1. / sqrt(n[j]);
        for (int j = 0; j < 3; j++) { h[j] *= n[0]; v[j] *= n[1]; d1[j] *= n[2]; d2[j] *= n[3]; }
        double A0[2] = {h[0] * v[0], h[1] * v[1]}, A1[2] = {d1[0] * d2[0], d1[1] * d2[1]};
        double b0 = -h[2] * v[2], b1 = -d1[2] * d2[2];
        for (int r = 0; r < 2; r++) {
            Atb[r] += A0[r] * b0 + A1[r] * b1;
            for (int c = 0; c < 2; c++) AtA[r * 2 + c] += A0[r] * A0[c] + A1[r] * A1[c];
        }
    }
    double f[2];
    solve_sym(AtA, Atb, 2, f);
    a[0] = sqrt(fabs(1. / f[0]));
    a[4] = sqrt(fabs(1. / f[1]));
}

/* Dense symmetric solve by Cholesky (LDLt-free, with eigen fallback). */
bool chol_solve(std::vector<double>& A, std::vector<double>& b, int n) {
    for (int j = 0; j < n; j++) {
        double s = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(s > 0)) return false;
        double d = sqrt(s);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double t = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) t -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = t / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return true;
}

} /* namespace */

extern "C" double orc_calibrate(const double* obj, const double* img, int V, int P, int W, int Hh, int flags,
                                int max_iter, double K_out[9], double* dist_out, double* rvecs, double* tvecs,
                                int32_t* iters_out) {
    /* active intrinsics: fx fy cx cy k1 k2 p1 p2 k3 [k4 k5 k6] */
    const int NI = (flags & 1) ? 12 : 9;
    const int np = NI + 6 * V;
    double a[9];
    init_intrinsics_2d(obj, img, V, P, W, Hh, a);
    std::vector<double> param((size_t)np, 0.);
    param[0] = a[0]; param[1] = a[4]; param[2] = a[2]; param[3] = a[5];
    double k0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < V; i++)
        find_extrinsics(obj + (size_t)i * P * 3, img + (size_t)i * P * 2, P, a, k0, &param[NI + 6 * i]);

    /* CvLevMarq::updateAlt driver (JtJ / JtErr / errNorm = sum of squares),
     * criteria (COUNT+EPS, 30, DBL_EPSILON).  The damped normal equations are
     * solved through the arrow structure: intrinsic block (NI x NI) + one
     * 6x6 block per view + NI x 6 cross blocks — the same linear system
     * OpenCV solves densely. */
    std::vector<double> Aii((size_t)NI * NI), gi(NI), Bie((size_t)V * NI * 6), Aee((size_t)V * 36), ge((size_t)V * 6);
    std::vector<double> prevParam(np), proj((size_t)2 * P), dpdr((size_t)2 * P * 3), dpdt((size_t)2 * P * 3),
        dpdf((size_t)2 * P * 2), dpdc((size_t)2 * P * 2), dpdk((size_t)2 * P * 8);
    auto unpack = [&](const std::vector<double>& p, double fxy[2], double cxy[2], double k[8]) {
        fxy[0] = p[0]; fxy[1] = p[1]; cxy[0] = p[2]; cxy[1] = p[3];
        for (int i = 0; i < 8; i++) k[i] = 0;
        for (int i = 0; i < NI - 4; i++) k[i] = p[4 + i];
    };
    auto eval = [&](const std::vector<double>& p, bool calcJ) {
        double fxy[2], cxy[2], k[8];
        unpack(p, fxy, cxy, k);
        double total = 0;
        if (calcJ) {
            std::fill(Aii.begin(), Aii.end(), 0.);
            std::fill(gi.begin(), gi.end(), 0.);
        }
        std::vector<double> Ji((size_t)2 * P * NI), Je((size_t)2 * P * 6), err((size_t)2 * P);
        for (int v = 0; v < V; v++) {
            const double* rt = &p[NI + 6 * v];
            project_points(obj + (size_t)v * P * 3, P, rt, rt + 3, fxy, cxy, k, proj.data(), calcJ ? dpdr.data() : 0,
                           calcJ ? dpdt.data() : 0, calcJ ? dpdf.data() : 0, calcJ ? dpdc.data() : 0, calcJ ? dpdk.data() : 0);
            for (int i = 0; i < 2 * P; i++) {
                err[i] = proj[i] - img[(size_t)v * P * 2 + i];
                total += err[i] * err[i];
            }
            if (!calcJ) continue;
            for (int i = 0; i < 2 * P; i++) {
                double* ji = &Ji[(size_t)i * NI];
                ji[0] = dpdf[i * 2]; ji[1] = dpdf[i * 2 + 1]; ji[2] = dpdc[i * 2]; ji[3] = dpdc[i * 2 + 1];
                for (int q = 0; q < NI - 4; q++) ji[4 + q] = dpdk[(size_t)i * 8 + q];
                for (int q = 0; q < 3; q++) { Je[(size_t)i * 6 + q] = dpdr[(size_t)i * 3 + q]; Je[(size_t)i * 6 + 3 + q] = dpdt[(size_t)i * 3 + q]; }
            }
            double* A = &Aee[(size_t)v * 36];
            double* B = &Bie[(size_t)v * NI * 6];
            double* g = &ge[(size_t)v * 6];
            memset(A, 0, 36 * sizeof(double));
            memset(B, 0, (size_t)NI * 6 * sizeof(double));
            memset(g, 0, 6 * sizeof(double));
            for (int i = 0; i < 2 * P; i++) {
                const double* ji = &Ji[(size_t)i * NI];
                const double* je = &Je[(size_t)i * 6];
                for (int r = 0; r < NI; r++) {
                    gi[r] += ji[r] * err[i];
                    for (int c = 0; c < NI; c++) Aii[(size_t)r * NI + c] += ji[r] * ji[c];
                    for (int c = 0; c < 6; c++) B[r * 6 + c] += ji[r] * je[c];
                }
                for (int r = 0; r < 6; r++) {
                    g[r] += je[r] * err[i];
                    for (int c = 0; c < 6; c++) A[r * 6 + c] += je[r] * je[c];
                }
            }
        }
        return total;
    };
    auto do_step = [&](int lambdaLg10) {
        const double lambda = exp(lambdaLg10 * log(10.));
        /* Schur: S = Aii' - sum B A'^-1 Bt ; rhs = gi - sum B A'^-1 ge */
        std::vector<double> S(Aii), rhs(gi), AinvBt((size_t)V * 6 * NI), Ainvg((size_t)V * 6);
        for (int r = 0; r < NI; r++) S[(size_t)r * NI + r] *= 1. + lambda;
        for (int v = 0; v < V; v++) {
            double A[36];
            memcpy(A, &Aee[(size_t)v * 36], sizeof A);
            for (int r = 0; r < 6; r++) A[r * 6 + r] *= 1. + lambda;
            const double* B = &Bie[(size_t)v * NI * 6];
            /* solve A X = [Bt | g] column by column (one decomposition of A serves every column) */
            SymEig eA;
            sym_eig(A, 6, eA);
            for (int c = 0; c < NI; c++) {
                double bcol[6], x[6];
                for (int r = 0; r < 6; r++) bcol[r] = B[c * 6 + r];
                sym_backsub(eA, bcol, x);
                for (int r = 0; r < 6; r++) AinvBt[((size_t)v * 6 + r) * NI + c] = x[r];
            }
            sym_backsub(eA, &ge[(size_t)v * 6], &Ainvg[(size_t)v * 6]);
            for (int r = 0; r < NI; r++) {
                double s = 0;
                for (int q = 0; q < 6; q++) s += B[r * 6 + q] * Ainvg[(size_t)v * 6 + q];
                rhs[r] -= s;
                for (int c = 0; c < NI; c++) {
                    double t = 0;
                    for (int q = 0; q < 6; q++) t += B[r * 6 + q] * AinvBt[((size_t)v * 6 + q) * NI + c];
                    S[(size_t)r * NI + c] -= t;
                }
            }
        }
        std::vector<double> di(NI);
        solve_sym(S.data(), rhs.data(), NI, di.data());
        for (int r = 0; r < NI; r++) param[r] = prevParam[r] - di[r];
        for (int v = 0; v < V; v++)
            for (int q = 0; q < 6; q++) {
                double s = Ainvg[(size_t)v * 6 + q];
                for (int c = 0; c < NI; c++) s -= AinvBt[((size_t)v * 6 + q) * NI + c] * di[c];
                param[NI + 6 * v + q] = prevParam[NI + 6 * v + q] - s;
            }
    };

    int lambdaLg10 = -3, iters = 0;
    double prevErrNorm = DBL_MAX, errNorm = 0;
    for (;;) {
        double e0 = eval(param, true);
        prevParam = param;
        if (iters == 0) prevErrNorm = e0;
        do_step(lambdaLg10);
        for (;;) {
            errNorm = eval(param, false);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { do_step(lambdaLg10); continue; }
            }
            break;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double dn = 0, pn = 0;
        for (int i = 0; i < np; i++) { dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]); pn += prevParam[i] * prevParam[i]; }
        double change = sqrt(dn) / sqrt(pn);
        if (++iters >= max_iter || change < DBL_EPSILON) break;
        prevErrNorm = errNorm;
    }
    double total = eval(param, false);
    for (int i = 0; i < 9; i++) K_out[i] = 0;
    K_out[0] = param[0]; K_out[4] = param[1]; K_out[2] = param[2]; K_out[5] = param[3]; K_out[8] = 1;
    for (int i = 0; i < NI - 4; i++) dist_out[i] = param[4 + i];
    for (int v = 0; v < V; v++)
        for (int q = 0; q < 3; q++) {
            if (rvecs) rvecs[3 * v + q] = param[NI + 6 * v + q];
            if (tvecs) tvecs[3 * v + q] = param[NI + 6 * v + 3 + q];
        }
    if (iters_out) *iters_out = iters;
    (void)chol_solve;
    return sqrt(total / ((double)V * P));
}

/* ---------------------------------------------- K12 split per share of views
 * (oracle.h: orc_calib_*).  Same formulas as orc_calibrate above, arranged so that
 * every output is additive over disjoint sets of views. */
namespace {

struct CalibShare {
    int V, P, NI;
    std::vector<double> obj, img, intr, ext, intrT, extT, AinvBt, Ainvg;
};

void unpack_intr(const double* p, int NI, double fxy[2], double cxy[2], double k[8]) {
    fxy[0] = p[0]; fxy[1] = p[1]; cxy[0] = p[2]; cxy[1] = p[3];
    for (int i = 0; i < 8; i++) k[i] = (i < NI - 4) ? p[4 + i] : 0.;
}

double share_error(const CalibShare& s, const double* intr, const double* ext) {
    double fxy[2], cxy[2], k[8], total = 0;
    unpack_intr(intr, s.NI, fxy, cxy, k);
    std::vector<double> proj((size_t)2 * s.P);
    for (int v = 0; v < s.V; v++) {
        const double* rt = ext + 6 * v;
        project_points(&s.obj[(size_t)v * s.P * 3], s.P, rt, rt + 3, fxy, cxy, k, proj.data(), 0, 0, 0, 0, 0);
        for (int i = 0; i < 2 * s.P; i++) {
            const double e = proj[i] - s.img[(size_t)v * s.P * 2 + i];
            total += e * e;
        }
    }
    return total;
}

}  // namespace

extern "C" {

void* orc_calib_begin(const double* obj, const double* img, int V, int P, int flags) {
    CalibShare* s = new CalibShare;
    s->V = V; s->P = P; s->NI = (flags & 1) ? 12 : 9;
    s->obj.assign(obj, obj + (size_t)V * P * 3);
    s->img.assign(img, img + (size_t)V * P * 2);
    s->intr.assign(12, 0.); s->intrT.assign(12, 0.);
    s->ext.assign((size_t)V * 6, 0.); s->extT.assign((size_t)V * 6, 0.);
    s->AinvBt.assign((size_t)V * 6 * s->NI, 0.); s->Ainvg.assign((size_t)V * 6, 0.);
    return s;
}

void orc_calib_end(void* h) { delete static_cast<CalibShare*>(h); }

void orc_calib_focal_acc(void* h, int W, int Hh, double acc6[6]) {
    const CalibShare& s = *static_cast<CalibShare*>(h);
    const double cx = (!W) ? 0.5 : (W - 1) * 0.5, cy = (!Hh) ? 0.5 : (Hh - 1) * 0.5;
    for (int i = 0; i < 6; i++) acc6[i] = 0;
    std::vector<double> Mxy((size_t)2 * s.P);
    for (int v = 0; v < s.V; v++) {
        for (int j = 0; j < s.P; j++) {
            Mxy[2 * j] = s.obj[((size_t)v * s.P + j) * 3];
            Mxy[2 * j + 1] = s.obj[((size_t)v * s.P + j) * 3 + 1];
        }
        double H[9];
        if (!find_homography(Mxy.data(), &s.img[(size_t)v * s.P * 2], s.P, H)) continue;
        H[0] -= H[6] * cx; H[1] -= H[7] * cx; H[2] -= H[8] * cx;
        H[3] -= H[6] * cy; H[4] -= H[7] * cy; H[5] -= H[8] * cy;
        double hh[3], vv[3], d1[3], d2[3], n[4] = {0, 0, 0, 0};
        for (int j = 0; j < 3; j++) {
            const double t0 = H[j * 3], t1 = H[j * 3 + 1];
            hh[j] = t0; vv[j] = t1; d1[j] = (t0 + t1) * 0.5; d2[j] = (t0 - t1) * 0.5;
            n[0] += t0 * t0; n[1] += t1 * t1; n[2] += d1[j] * d1[j]; n[3] += d2[j] * d2[j];
        }
        for (int j = 0; j < 4; j++) n[j] = 1. / sqrt(n[j]);
        for (int j = 0; j < 3; j++) { hh[j] *= n[0]; vv[j] *= n[1]; d1[j] *= n[2]; d2[j] *= n[3]; }
        const double A0[2] = {hh[0] * vv[0], hh[1] * vv[1]}, A1[2] = {d1[0] * d2[0], d1[1] * d2[1]};
        const double b0 = -hh[2] * vv[2], b1 = -d1[2] * d2[2];
        acc6[0] += A0[0] * A0[0] + A1[0] * A1[0];
        acc6[1] += A0[0] * A0[1] + A1[0] * A1[1];
        acc6[2] += A0[1] * A0[1] + A1[1] * A1[1];
        acc6[3] += A0[0] * b0 + A1[0] * b1;
        acc6[4] += A0[1] * b0 + A1[1] * b1;
    }
}

void orc_calib_set_intrinsics(void* h, const double intr[12], int init_extrinsics) {
    CalibShare& s = *static_cast<CalibShare*>(h);
    s.intr.assign(intr, intr + 12);
    if (!init_extrinsics) return;
    const double a[9] = {intr[0], 0, intr[2], 0, intr[1], intr[3], 0, 0, 1};
    const double k0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int v = 0; v < s.V; v++)
        find_extrinsics(&s.obj[(size_t)v * s.P * 3], &s.img[(size_t)v * s.P * 2], s.P, a, k0, &s.ext[(size_t)6 * v]);
}

void orc_calib_accumulate(void* h, double lambda, double* acc) {
    CalibShare& s = *static_cast<CalibShare*>(h);
    const int NI = s.NI, P = s.P;
    double fxy[2], cxy[2], k[8];
    unpack_intr(s.intr.data(), NI, fxy, cxy, k);
    double* S = acc;
    double* rhs = acc + NI * NI;
    for (int i = 0; i < NI * NI + NI + 1; i++) acc[i] = 0;
    std::vector<double> proj((size_t)2 * P), dpdr((size_t)2 * P * 3), dpdt((size_t)2 * P * 3), dpdf((size_t)2 * P * 2),
        dpdc((size_t)2 * P * 2), dpdk((size_t)2 * P * 8), Ji((size_t)2 * P * NI), Je((size_t)2 * P * 6), err((size_t)2 * P);
    for (int v = 0; v < s.V; v++) {
        const double* rt = &s.ext[(size_t)6 * v];
        project_points(&s.obj[(size_t)v * P * 3], P, rt, rt + 3, fxy, cxy, k, proj.data(), dpdr.data(), dpdt.data(),
                       dpdf.data(), dpdc.data(), dpdk.data());
        double Aii[144] = {0}, gi[12] = {0}, B[72] = {0}, A[36] = {0}, g[6] = {0};
        for (int i = 0; i < 2 * P; i++) {
            err[i] = proj[i] - s.img[(size_t)v * P * 2 + i];
            acc[NI * NI + NI] += err[i] * err[i];
            double* ji = &Ji[(size_t)i * NI];
            ji[0] = dpdf[i * 2]; ji[1] = dpdf[i * 2 + 1]; ji[2] = dpdc[i * 2]; ji[3] = dpdc[i * 2 + 1];
            for (int q = 0; q < NI - 4; q++) ji[4 + q] = dpdk[(size_t)i * 8 + q];
            double* je = &Je[(size_t)i * 6];
            for (int q = 0; q < 3; q++) { je[q] = dpdr[(size_t)i * 3 + q]; je[3 + q] = dpdt[(size_t)i * 3 + q]; }
            for (int r = 0; r < NI; r++) {
                gi[r] += ji[r] * err[i];
                for (int c = 0; c < NI; c++) Aii[r * NI + c] += ji[r] * ji[c];
                for (int c = 0; c < 6; c++) B[r * 6 + c] += ji[r] * je[c];
            }
            for (int r = 0; r < 6; r++) {
                g[r] += je[r] * err[i];
                for (int c = 0; c < 6; c++) A[r * 6 + c] += je[r] * je[c];
            }
        }
        for (int r = 0; r < NI; r++) Aii[r * NI + r] *= 1. + lambda;
        for (int r = 0; r < 6; r++) A[r * 6 + r] *= 1. + lambda;
        double* AinvBt = &s.AinvBt[(size_t)v * 6 * NI];
        double* Ainvg = &s.Ainvg[(size_t)v * 6];
        SymEig eA;
        sym_eig(A, 6, eA);
        for (int c = 0; c < NI; c++) {
            double bcol[6], x[6];
            for (int r = 0; r < 6; r++) bcol[r] = B[c * 6 + r];
            sym_backsub(eA, bcol, x);
            for (int r = 0; r < 6; r++) AinvBt[r * NI + c] = x[r];
        }
        sym_backsub(eA, g, Ainvg);
        for (int r = 0; r < NI; r++) {
            double sg = 0;
            for (int q = 0; q < 6; q++) sg += B[r * 6 + q] * Ainvg[q];
            rhs[r] += gi[r] - sg;
            for (int c = 0; c < NI; c++) {
                double t = 0;
                for (int q = 0; q < 6; q++) t += B[r * 6 + q] * AinvBt[q * NI + c];
                S[r * NI + c] += Aii[r * NI + c] - t;
            }
        }
    }
}

double orc_calib_step(void* h, const double* delta) {
    CalibShare& s = *static_cast<CalibShare*>(h);
    const int NI = s.NI;
    s.intrT = s.intr;
    for (int r = 0; r < NI; r++) s.intrT[r] = s.intr[r] - delta[r];
    for (int v = 0; v < s.V; v++)
        for (int q = 0; q < 6; q++) {
            double t = s.Ainvg[(size_t)v * 6 + q];
            for (int c = 0; c < NI; c++) t -= s.AinvBt[((size_t)v * 6 + q) * NI + c] * delta[c];
            s.extT[(size_t)6 * v + q] = s.ext[(size_t)6 * v + q] - t;
        }
    return share_error(s, s.intrT.data(), s.extT.data());
}

void orc_calib_accept(void* h) {
    CalibShare& s = *static_cast<CalibShare*>(h);
    s.intr = s.intrT;
    s.ext = s.extT;
}

void orc_calib_get(void* h, double intr[12], double* rvecs, double* tvecs) {
    const CalibShare& s = *static_cast<CalibShare*>(h);
    if (intr) for (int i = 0; i < 12; i++) intr[i] = s.intr[i];
    for (int v = 0; v < s.V; v++)
        for (int q = 0; q < 3; q++) {
            if (rvecs) rvecs[3 * v + q] = s.ext[(size_t)6 * v + q];
            if (tvecs) tvecs[3 * v + q] = s.ext[(size_t)6 * v + 3 + q];
        }
}

}  /* extern "C" */
