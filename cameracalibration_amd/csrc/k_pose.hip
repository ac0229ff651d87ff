// k_pose.hip — K8/K9/K10: batched pose estimation.
//
//  k_pnp   cv.solvePnP(ITERATIVE) + cv.solvePnPRefineLM (reference
//          camera_sync/aruco.py:368-384) and the MAE/RMSE of aruco.py:386-400:
//          one wavefront per frame, one lane per point.  Planar initialisation
//          (undistort, plane frame, homography, decomposition) then two
//          Levenberg-Marquardt phases on the 6 pose parameters with the two
//          stopping rules of OpenCV's CvLevMarq and LMSolver.  J^T J / J^T r
//          are wave reductions; every lane solves the 6x6 system redundantly.
//  k_ippe  cv.solvePnP(SOLVEPNP_IPPE_SQUARE) (aruco.py:97-117): one lane per
//          marker, closed form.
//  k_gather_anchors  the anchor gather loop of processAruco (aruco.py:308-315).
#include "internal.h"
#include "pose_math.h"
#include "homography.h"

namespace camsync {

namespace {

// ---- 3x3 symmetric eigen-decomposition (cyclic Jacobi, static indices) ----
template <int P, int Q>
__device__ __forceinline__ void jrot3(double* A, double* V) {
    const double apq = A[P * 3 + Q];
    if (fabs(apq) < 1e-300) return;
    const double app = A[P * 3 + P], aqq = A[Q * 3 + Q];
    const double theta = (aqq - app) / (2 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
    const double c = 1 / sqrt(t * t + 1), s = t * c;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double akp = A[k * 3 + P], akq = A[k * 3 + Q];
        A[k * 3 + P] = c * akp - s * akq;
        A[k * 3 + Q] = s * akp + c * akq;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double apk = A[P * 3 + k], aqk = A[Q * 3 + k];
        A[P * 3 + k] = c * apk - s * aqk;
        A[Q * 3 + k] = s * apk + c * aqk;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double vpk = V[P * 3 + k], vqk = V[Q * 3 + k];
        V[P * 3 + k] = c * vpk - s * vqk;
        V[Q * 3 + k] = s * vpk + c * vqk;
    }
}

template <int I, int J>
__device__ __forceinline__ void sort_pair3(double* w, double* V) {
    if (w[J] > w[I]) {
        double t = w[I]; w[I] = w[J]; w[J] = t;
#pragma unroll
        for (int k = 0; k < 3; k++) { t = V[I * 3 + k]; V[I * 3 + k] = V[J * 3 + k]; V[J * 3 + k] = t; }
    }
}

// eigenvalues descending in w, eigenvectors in the rows of V
__device__ __forceinline__ void eigen3(const double* Ain, double* w, double* V) {
    double A[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
#pragma unroll 1
    for (int sweep = 0; sweep < 30; sweep++) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off < 1e-300) break;
        jrot3<0, 1>(A, V);
        jrot3<0, 2>(A, V);
        jrot3<1, 2>(A, V);
    }
    w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
    sort_pair3<0, 1>(w, V);
    sort_pair3<0, 2>(w, V);
    sort_pair3<1, 2>(w, V);
}

// 3x3 SVD A = U diag(w) Vt through the eigen-decomposition of AtA (the restatement's svd3).
__device__ __forceinline__ void svd3(const double* A, double* U, double* w, double* Vt) {
    double AtA[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += A[k * 3 + i] * A[k * 3 + j];
            AtA[i * 3 + j] = s;
        }
    double ev[3];
    eigen3(AtA, ev, Vt);
#pragma unroll
    for (int i = 0; i < 3; i++) w[i] = sqrt(fmax(ev[i], 0.));
    double u[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += A[r * 3 + k] * Vt[i * 3 + k];
            u[i][r] = s;
        }
        const double nrm = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + u[i][2] * u[i][2]);
        if (nrm > 1e-12 * (w[0] + 1e-300)) {
#pragma unroll
            for (int r = 0; r < 3; r++) u[i][r] /= nrm;
        } else if (i == 2) {
            u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
            u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
            u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int r = 0; r < 3; r++) U[r * 3 + i] = u[i][r];
}

// A wavefront's LDS operations execute in order: only the compiler must keep them in order.
__device__ __forceinline__ void pose_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cyclic Jacobi eigen-decomposition of the symmetric 12 x 12 matrix A (LDS, row-major, overwritten), eigenvectors in the
// rows of V (LDS): the rotations in the restatement's order, each applied by twelve lanes (one matrix element of a row
// or column per lane), so every element sees the same arithmetic as on one thread.  Returns the row of V that belongs
// to the smallest eigenvalue.
__device__ __forceinline__ int wave_jacobi12(double* A, double* V, int lane) {
    constexpr int n = 12;
    for (int i = lane; i < n * n; i += 64) V[i] = (i / n == i % n) ? 1.0 : 0.0;
    pose_lds_order();
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double app = A[p * n + p], aqq = A[q * n + q];
                const double theta = (aqq - app) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                const double c = 1 / sqrt(t * t + 1), s = t * c;
                pose_lds_order();
                if (lane < n) {
                    const double akp = A[lane * n + p], akq = A[lane * n + q];
                    A[lane * n + p] = c * akp - s * akq;
                    A[lane * n + q] = s * akp + c * akq;
                }
                pose_lds_order();
                if (lane < n) {
                    const double apk = A[p * n + lane], aqk = A[q * n + lane];
                    A[p * n + lane] = c * apk - s * aqk;
                    A[q * n + lane] = s * apk + c * aqk;
                    const double vpk = V[p * n + lane], vqk = V[q * n + lane];
                    V[p * n + lane] = c * vpk - s * vqk;
                    V[q * n + lane] = s * vpk + c * vqk;
                }
                pose_lds_order();
            }
    }
    // the restatement sorts the eigenvalues in descending order by selection and takes the last row
    int order[n];
    double w[n];
#pragma unroll
    for (int i = 0; i < n; i++) { w[i] = A[i * n + i]; order[i] = i; }
#pragma unroll
    for (int i = 0; i < n; i++) {
        int m = i;
#pragma unroll
        for (int j = i + 1; j < n; j++)
            if (w[j] > w[m]) m = j;
        double tw = w[i]; int to = order[i];
#pragma unroll
        for (int j = 0; j < n; j++) {
            if (j == m) { w[i] = w[j]; order[i] = order[j]; w[j] = tw; order[j] = to; }
        }
    }
    return order[n - 1];
}

struct PoseSums {
    double JtJ[36];
    double Jte[6];
    double err2;
    double rinf;
};

// Residuals (and optionally the normal equations) of the pixel reprojection
// error at pose p, PPL points per lane.
template <bool WITH_J, int PPL>
__device__ __forceinline__ void pose_eval(const Cam& cam, const double* p, const bool (&active)[PPL], const double (&X)[PPL],
                                          const double (&Y)[PPL], const double (&Z)[PPL], const double (&ui)[PPL],
                                          const double (&vi)[PPL], PoseSums& S) {
    double R[9], dR[27];
    rodrigues_v2m(p, R, WITH_J ? dR : nullptr);
    double e2 = 0, m = 0, jte[6], jtj[36];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const double Xc = R[0] * X[k] + R[1] * Y[k] + R[2] * Z[k] + p[3];
        const double Yc = R[3] * X[k] + R[4] * Y[k] + R[5] * Z[k] + p[4];
        const double Zc = R[6] * X[k] + R[7] * Y[k] + R[8] * Z[k] + p[5];
        double u, v, duP[3], dvP[3];
        project_cam<WITH_J, false>(cam, Xc, Yc, Zc, u, v, duP, dvP, nullptr, nullptr);
        const double ex = active[k] ? u - ui[k] : 0.0, ey = active[k] ? v - vi[k] : 0.0;
        const double ek = ex * ex + ey * ey, mk = fmax(fabs(ex), fabs(ey));
        e2 = k == 0 ? ek : e2 + ek;
        m = k == 0 ? mk : fmax(m, mk);
        if (WITH_J) {
            double ju[6], jv[6];
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const double* D = dR + q * 9;
                const double dP0 = D[0] * X[k] + D[1] * Y[k] + D[2] * Z[k], dP1 = D[3] * X[k] + D[4] * Y[k] + D[5] * Z[k],
                             dP2 = D[6] * X[k] + D[7] * Y[k] + D[8] * Z[k];
                ju[q] = active[k] ? duP[0] * dP0 + duP[1] * dP1 + duP[2] * dP2 : 0.0;
                jv[q] = active[k] ? dvP[0] * dP0 + dvP[1] * dP1 + dvP[2] * dP2 : 0.0;
                ju[3 + q] = active[k] ? duP[q] : 0.0;
                jv[3 + q] = active[k] ? dvP[q] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < 6; a++) {
                const double t = ju[a] * ex + jv[a] * ey;
                jte[a] = k == 0 ? t : jte[a] + t;
#pragma unroll
                for (int b = a; b < 6; b++) {
                    const double s = ju[a] * ju[b] + jv[a] * jv[b];
                    jtj[a * 6 + b] = k == 0 ? s : jtj[a * 6 + b] + s;
                }
            }
        }
    }
    S.err2 = wsum(e2);
    S.rinf = wave_allreduce(m, [](double a, double b) { return fmax(a, b); });
    if (WITH_J) {
#pragma unroll
        for (int a = 0; a < 6; a++) {
            S.Jte[a] = wsum(jte[a]);
#pragma unroll
            for (int b = a; b < 6; b++) {
                const double s = wsum(jtj[a * 6 + b]);
                S.JtJ[a * 6 + b] = s;
                S.JtJ[b * 6 + a] = s;
            }
        }
    }
}

// step = (JtJ with damped diagonal)^-1 Jte ; mult: diag *= (1+lambda), else diag += lambda*D
__device__ __forceinline__ bool damped_solve(const PoseSums& S, double lambda, bool mult, const double* D, double* step) {
    double A[36];
#pragma unroll
    for (int i = 0; i < 36; i++) A[i] = S.JtJ[i];
#pragma unroll
    for (int i = 0; i < 6; i++) A[i * 6 + i] = mult ? A[i * 6 + i] * (1. + lambda) : A[i * 6 + i] + lambda * D[i];
    double b[6];
#pragma unroll
    for (int i = 0; i < 6; i++) b[i] = S.Jte[i];
    bool ok = chol_factor<6>(A);
    if (!ok) {
        // nearly singular: retry with a small ridge
        double tr = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) tr += S.JtJ[i * 6 + i];
#pragma unroll
        for (int i = 0; i < 36; i++) A[i] = S.JtJ[i];
#pragma unroll
        for (int i = 0; i < 6; i++) A[i * 6 + i] = A[i * 6 + i] * (1. + lambda) + 1e-12 * tr + 1e-300;
        ok = chol_factor<6>(A);
    }
    chol_backsolve<6>(A, b);
#pragma unroll
    for (int i = 0; i < 6; i++) step[i] = b[i];
    return ok;
}

// Point i of a frame is held by lane i % 64 as its point i / 64 (PPL points per lane: up to 64 PPL points).
template <int PPL>
__global__ __launch_bounds__(64) void k_pnp(PnpArgs a) {
    __shared__ double s_LL[144], s_VV[144];  // the non-planar initialisation's 12 x 12 system and its eigenvectors
    const int lane = threadIdx.x;
    for (int f = blockIdx.x; f < a.B; f += gridDim.x) {
        const int n = min(a.npts[f], min(a.N, 64 * PPL));
        const Cam cam = load_cam(a.K + (a.perFrameCam ? (size_t)f * 9 : 0), a.dist + (a.perFrameCam ? (size_t)f * a.ndist : 0), a.ndist);
        bool active[PPL];
        double X[PPL], Y[PPL], Z[PPL], ui[PPL], vi[PPL];
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            const int i = k * 64 + lane;
            active[k] = i < n;
            X[k] = Y[k] = Z[k] = ui[k] = vi[k] = 0;
            if (active[k]) {
                const float* o = a.obj + ((size_t)f * a.N + i) * 3;
                const float* m = a.img + ((size_t)f * a.N + i) * 2;
                X[k] = o[0]; Y[k] = o[1]; Z[k] = o[2];
                ui[k] = m[0]; vi[k] = m[1];
            }
        }
        int status = 1;
        double p[6] = {0, 0, 0, 0, 0, 0};
        if (n < 4) status = 0;
        if (status == 1) {
            // ---- initial pose (cvFindExtrinsicCameraParams2) ----
            double xn[PPL], yn[PPL];
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                xn[k] = yn[k] = 0;
                undistort_point(cam, ui[k], vi[k], xn[k], yn[k]);
                if (!active[k]) { xn[k] = 0; yn[k] = 0; }
            }
            const double invn = 1.0 / n;
            const double Mc[3] = {wsum(lane_sum<PPL>([&](int k) { return X[k]; })) * invn, wsum(lane_sum<PPL>([&](int k) { return Y[k]; })) * invn,
                                  wsum(lane_sum<PPL>([&](int k) { return Z[k]; })) * invn};
            double d0[PPL], d1[PPL], d2[PPL];
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                d0[k] = active[k] ? X[k] - Mc[0] : 0; d1[k] = active[k] ? Y[k] - Mc[1] : 0; d2[k] = active[k] ? Z[k] - Mc[2] : 0;
            }
            double MM[9];
            MM[0] = wsum(lane_sum<PPL>([&](int k) { return d0[k] * d0[k]; })); MM[1] = wsum(lane_sum<PPL>([&](int k) { return d0[k] * d1[k]; }));
            MM[2] = wsum(lane_sum<PPL>([&](int k) { return d0[k] * d2[k]; })); MM[4] = wsum(lane_sum<PPL>([&](int k) { return d1[k] * d1[k]; }));
            MM[5] = wsum(lane_sum<PPL>([&](int k) { return d1[k] * d2[k]; })); MM[8] = wsum(lane_sum<PPL>([&](int k) { return d2[k] * d2[k]; }));
            MM[3] = MM[1]; MM[6] = MM[2]; MM[7] = MM[5];
            double Wv[3], V[9];
            eigen3(MM, Wv, V);
            if (!(Wv[2] / Wv[1] < 1e-3)) {
                // ---- non-planar object: DLT on the 3x4 projection (at least 6 points) ----
                if (n < 6) {
                    status = 2;  // (cv.solvePnP raises here; the caller reports it)
                } else {
                    // L^T L of the 2n x 12 system [P 1 0 x P x; 0 P 1 y P y] (x, y negated normalised image points): its
                    // blocks are four symmetric 4 x 4 moment matrices of (P, 1)
                    double mA[10], mX[10], mY[10], mC[10];
#pragma unroll
                    for (int k = 0; k < PPL; k++) {
                        const double a0 = active[k] ? 1.0 : 0.0;
                        const double x = -xn[k], y = -yn[k];
                        const double ph[4] = {X[k] * a0, Y[k] * a0, Z[k] * a0, a0};
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < 4; a++)
#pragma unroll
                            for (int b = a; b < 4; b++, e++) {
                                // one row's products as the restatement forms them: L0[a] L0[b] + L1[a] L1[b]
                                const double pa = ph[a] * ph[b];
                                const double px = (x * ph[a]) * ph[b], py = (y * ph[a]) * ph[b];
                                const double pc = (x * ph[a]) * (x * ph[b]) + (y * ph[a]) * (y * ph[b]);
                                mA[e] = k == 0 ? pa : mA[e] + pa;
                                mX[e] = k == 0 ? px : mX[e] + px;
                                mY[e] = k == 0 ? py : mY[e] + py;
                                mC[e] = k == 0 ? pc : mC[e] + pc;
                            }
                    }
                    {
                        int e = 0;
#pragma unroll
                        for (int a = 0; a < 4; a++)
#pragma unroll
                            for (int b = a; b < 4; b++, e++) {
                                const double sa = wsum(mA[e]), sx = wsum(mX[e]), sy = wsum(mY[e]), sc = wsum(mC[e]);
                                if (lane == 0) {
                                    s_LL[a * 12 + b] = s_LL[b * 12 + a] = sa;
                                    s_LL[(4 + a) * 12 + 4 + b] = s_LL[(4 + b) * 12 + 4 + a] = sa;
                                    s_LL[a * 12 + 8 + b] = s_LL[(8 + b) * 12 + a] = sx;
                                    s_LL[b * 12 + 8 + a] = s_LL[(8 + a) * 12 + b] = sx;
                                    s_LL[(4 + a) * 12 + 8 + b] = s_LL[(8 + b) * 12 + 4 + a] = sy;
                                    s_LL[(4 + b) * 12 + 8 + a] = s_LL[(8 + a) * 12 + 4 + b] = sy;
                                    s_LL[(8 + a) * 12 + 8 + b] = s_LL[(8 + b) * 12 + 8 + a] = sc;
                                    s_LL[a * 12 + 4 + b] = s_LL[(4 + b) * 12 + a] = 0.0;
                                    s_LL[b * 12 + 4 + a] = s_LL[(4 + a) * 12 + b] = 0.0;
                                }
                            }
                    }
                    pose_lds_order();
                    const int row = wave_jacobi12(s_LL, s_VV, lane);
                    const double* RR = s_VV + row * 12;
                    double Rm[9] = {RR[0], RR[1], RR[2], RR[4], RR[5], RR[6], RR[8], RR[9], RR[10]};
                    double tm[3] = {RR[3], RR[7], RR[11]};
                    if (det3(Rm) < 0) {
#pragma unroll
                        for (int i = 0; i < 9; i++) Rm[i] = -Rm[i];
#pragma unroll
                        for (int i = 0; i < 3; i++) tm[i] = -tm[i];
                    }
                    const double sc = sqrt(Rm[0] * Rm[0] + Rm[1] * Rm[1] + Rm[2] * Rm[2] + Rm[3] * Rm[3] + Rm[4] * Rm[4] + Rm[5] * Rm[5] +
                                           Rm[6] * Rm[6] + Rm[7] * Rm[7] + Rm[8] * Rm[8]);
                    double U[9], ww[3], Vt[9], R[9];
                    svd3(Rm, U, ww, Vt);
                    mat3_mul(U, Vt, R);
                    const double sN = sqrt(3.) / sc;  // ||R||_F / ||Rm||_F
                    rodrigues_m2v(R, p);
                    p[3] = tm[0] * sN; p[4] = tm[1] * sN; p[5] = tm[2] * sN;
                    pose_lds_order();
                }
            } else {
                double Rt[9];
#pragma unroll
                for (int i = 0; i < 9; i++) Rt[i] = V[i];
                if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
#pragma unroll
                    for (int i = 0; i < 9; i++) Rt[i] = (i % 4 == 0) ? 1.0 : 0.0;
                }
                if (det3(Rt) < 0) {
#pragma unroll
                    for (int i = 0; i < 9; i++) Rt[i] = -Rt[i];
                }
                double tt[3];
#pragma unroll
                for (int q = 0; q < 3; q++) tt[q] = -(Rt[q * 3] * Mc[0] + Rt[q * 3 + 1] * Mc[1] + Rt[q * 3 + 2] * Mc[2]);
                double Mx[PPL], My[PPL];
#pragma unroll
                for (int k = 0; k < PPL; k++) {
                    Mx[k] = active[k] ? Rt[0] * X[k] + Rt[1] * Y[k] + Rt[2] * Z[k] + tt[0] : 0;
                    My[k] = active[k] ? Rt[3] * X[k] + Rt[4] * Y[k] + Rt[5] * Z[k] + tt[1] : 0;
                }
                double H[9];
                const bool hok = wave_homography<PPL>(Mx, My, xn, yn, active, n, H);
                double R[9], t[3];
                bool fin = hok;
#pragma unroll
                for (int i = 0; i < 9; i++) fin = fin && isfinite(H[i]);
                if (fin) {
                    const double tiny = 2.220446049250313e-16;
                    const double h1n = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]);
                    const double h2n = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
                    const double s1 = 1. / fmax(h1n, tiny), s2 = 1. / fmax(h2n, tiny), stt = 2. / fmax(h1n + h2n, tiny);
                    t[0] = H[2] * stt; t[1] = H[5] * stt; t[2] = H[8] * stt;
                    const double h1[3] = {H[0] * s1, H[3] * s1, H[6] * s1}, h2[3] = {H[1] * s2, H[4] * s2, H[7] * s2};
                    const double h3[3] = {h1[1] * h2[2] - h1[2] * h2[1], h1[2] * h2[0] - h1[0] * h2[2], h1[0] * h2[1] - h1[1] * h2[0]};
                    double Hm[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
                    double rv[3];
                    rodrigues_m2v(Hm, rv);
                    rodrigues_v2m(rv, Hm, nullptr);
                    double t2[3];
#pragma unroll
                    for (int q = 0; q < 3; q++) t2[q] = Hm[q * 3] * tt[0] + Hm[q * 3 + 1] * tt[1] + Hm[q * 3 + 2] * tt[2] + t[q];
                    t[0] = t2[0]; t[1] = t2[1]; t[2] = t2[2];
                    mat3_mul(Hm, Rt, R);
                } else {
#pragma unroll
                    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
                    t[0] = t[1] = t[2] = 0;
                }
                rodrigues_m2v(R, p);
                p[3] = t[0]; p[4] = t[1]; p[5] = t[2];
            }
        }
        if (status == 1) {
            PoseSums S;
            // ---- phase 1: CvLevMarq(6, 2n, (EPS+ITER, 20, FLT_EPSILON)) ----
            {
                int lambdaLg10 = -3, iters = 0;
                double prevErr = 1.7976931348623157e308, errNorm = 0;
                double prev[6], step[6];
#pragma unroll 1
                for (;;) {
                    pose_eval<true, PPL>(cam, p, active, X, Y, Z, ui, vi, S);
#pragma unroll
                    for (int i = 0; i < 6; i++) prev[i] = p[i];
                    if (iters == 0) prevErr = sqrt(S.err2);
                    damped_solve(S, exp(lambdaLg10 * 2.302585092994046), true, nullptr, step);
#pragma unroll
                    for (int i = 0; i < 6; i++) p[i] = prev[i] - step[i];
#pragma unroll 1
                    for (;;) {
                        PoseSums E;
                        pose_eval<false, PPL>(cam, p, active, X, Y, Z, ui, vi, E);
                        errNorm = sqrt(E.err2);
                        if (errNorm > prevErr && ++lambdaLg10 <= 16) {
                            damped_solve(S, exp(lambdaLg10 * 2.302585092994046), true, nullptr, step);
#pragma unroll
                            for (int i = 0; i < 6; i++) p[i] = prev[i] - step[i];
                            continue;
                        }
                        break;
                    }
                    lambdaLg10 = max(lambdaLg10 - 1, -16);
                    double dn = 0, pn = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) { dn += (p[i] - prev[i]) * (p[i] - prev[i]); pn += prev[i] * prev[i]; }
                    const double change = sqrt(dn) / (pn > 0 ? sqrt(pn) : 2.220446049250313e-16);
                    if (++iters >= 20 || change < 1.1920928955078125e-07) break;
                    prevErr = errNorm;
                }
            }
            // ---- phase 2: solvePnPRefineLM = LMSolver(20, FLT_EPSILON) ----
            if (a.refine) {
                const double eps = 1.1920928955078125e-07;
                pose_eval<true, PPL>(cam, p, active, X, Y, Z, ui, vi, S);
                double Sc = S.err2, rinf = S.rinf;
                double D[6];
#pragma unroll
                for (int i = 0; i < 6; i++) D[i] = S.JtJ[i * 6 + i];
                const double Rlo = 0.25, Rhi = 0.75;
                double lambda = 1, lc = 0.75;
                int iter = 0;
                double d[6], xd[6];
#pragma unroll 1
                for (;;) {
                    damped_solve(S, lambda, false, D, d);
#pragma unroll
                    for (int i = 0; i < 6; i++) xd[i] = p[i] - d[i];
                    PoseSums E;
                    pose_eval<false, PPL>(cam, xd, active, X, Y, Z, ui, vi, E);
                    const double Sd = E.err2;
                    double dS = 0, tdv = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 6; k++) s += S.JtJ[i * 6 + k] * d[k];
                        dS += d[i] * (-s + 2 * S.Jte[i]);
                        tdv += d[i] * S.Jte[i];
                    }
                    const double Rr = (Sc - Sd) / (fabs(dS) > 2.220446049250313e-16 ? dS : 1);
                    if (Rr > Rhi) {
                        lambda *= 0.5;
                        if (lambda < lc) lambda = 0;
                    } else if (Rr < Rlo) {
                        double nu = (Sd - Sc) / (fabs(tdv) > 2.220446049250313e-16 ? tdv : 1) + 2;
                        nu = fmin(fmax(nu, 2.), 10.);
                        if (lambda == 0) {
                            // 1 / max |diag(A^-1)|
                            double L[36];
#pragma unroll
                            for (int i = 0; i < 36; i++) L[i] = S.JtJ[i];
                            double maxval = 2.220446049250313e-16;
                            if (chol_factor<6>(L)) {
#pragma unroll
                                for (int c = 0; c < 6; c++) {
                                    double e[6];
#pragma unroll
                                    for (int i = 0; i < 6; i++) e[i] = (i == c) ? 1.0 : 0.0;
                                    chol_backsolve<6>(L, e);
                                    maxval = fmax(maxval, fabs(e[c]));
                                }
                            }
                            lambda = lc = 1. / maxval;
                            nu *= 0.5;
                        }
                        lambda *= nu;
                    }
                    double dinf = 0;
#pragma unroll
                    for (int i = 0; i < 6; i++) dinf = fmax(dinf, fabs(d[i]));
                    if (Sd < Sc) {
                        Sc = Sd;
#pragma unroll
                        for (int i = 0; i < 6; i++) p[i] = xd[i];
                        pose_eval<true, PPL>(cam, p, active, X, Y, Z, ui, vi, S);
                        rinf = S.rinf;
                    }
                    iter++;
                    if (!(iter < 20 && dinf >= eps && rinf >= eps)) break;
                }
            }
            bool fin = true;
#pragma unroll
            for (int i = 0; i < 6; i++) fin = fin && isfinite(p[i]);
            if (!fin) status = 0;
        }
        // ---- outputs + metrics ----
        if (status == 1) {
            double R[9];
            rodrigues_v2m(p, R, nullptr);
            double dist2[PPL];
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const double Xc = R[0] * X[k] + R[1] * Y[k] + R[2] * Z[k] + p[3];
                const double Yc = R[3] * X[k] + R[4] * Y[k] + R[5] * Z[k] + p[4];
                const double Zc = R[6] * X[k] + R[7] * Y[k] + R[8] * Z[k] + p[5];
                double u, v;
                project_cam<false, false>(cam, Xc, Yc, Zc, u, v, nullptr, nullptr, nullptr, nullptr);
                dist2[k] = active[k] ? (u - ui[k]) * (u - ui[k]) + (v - vi[k]) * (v - vi[k]) : 0.0;
            }
            const double sd = wsum(lane_sum<PPL>([&](int k) { return sqrt(dist2[k]); })), sd2 = wsum(lane_sum<PPL>([&](int k) { return dist2[k]; }));
            if (lane == 0) {
                for (int i = 0; i < 3; i++) { a.rvec[(size_t)f * 3 + i] = p[i]; a.tvec[(size_t)f * 3 + i] = p[3 + i]; }
                if (a.ok) a.ok[f] = 1;
                if (a.mae) a.mae[f] = sd / n;
                if (a.rmse) a.rmse[f] = sqrt(sd2 / n);
            }
        } else if (lane == 0) {
            for (int i = 0; i < 3; i++) { a.rvec[(size_t)f * 3 + i] = 0; a.tvec[(size_t)f * 3 + i] = 0; }
            if (a.ok) a.ok[f] = (uint8_t)status;
            if (a.mae) a.mae[f] = 0;
            if (a.rmse) a.rmse[f] = 0;
        }
    }
}

// processAruco's anchor loop (aruco.py:308-315): for every anchor in list
// order that the frame sees (last detection of an id wins, aruco.py:270-271),
// append its 4 object corners and 4 image corners.
__global__ void k_gather_anchors(const int32_t* __restrict__ ids, const float* __restrict__ corners,
                                 const int32_t* __restrict__ counts, int B, int maxMarkers,
                                 const int32_t* __restrict__ anchorIds, const float* __restrict__ anchorObj, int nAnchor,
                                 int N, float* __restrict__ obj, float* __restrict__ img, int32_t* __restrict__ npts) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= B) return;
    const int cnt = min(counts[f], maxMarkers);
    int n = 0;
    for (int a = 0; a < nAnchor; a++) {
        int found = -1;
        for (int m = 0; m < cnt; m++)
            if (ids[(size_t)f * maxMarkers + m] == anchorIds[a]) found = m;
        if (found < 0 || n + 4 > N) continue;
        for (int k = 0; k < 4; k++) {
            for (int q = 0; q < 3; q++) obj[((size_t)f * N + n + k) * 3 + q] = anchorObj[((size_t)a * 4 + k) * 3 + q];
            img[((size_t)f * N + n + k) * 2] = corners[((size_t)f * maxMarkers + found) * 8 + 2 * k];
            img[((size_t)f * N + n + k) * 2 + 1] = corners[((size_t)f * maxMarkers + found) * 8 + 2 * k + 1];
        }
        n += 4;
    }
    npts[f] = n;
}

// IPPE for a square (ippe.cpp PoseSolver::solveSquare), one marker per lane.
__global__ void k_ippe(const float* __restrict__ corners, const double* __restrict__ size, int B,
                       const double* __restrict__ K, const double* __restrict__ dist, int ndist, int perFrameCam,
                       double* __restrict__ rvec, double* __restrict__ tvec, uint8_t* __restrict__ okOut) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= B) return;
    const Cam cam = load_cam(K + (perFrameCam ? (size_t)f * 9 : 0), dist + (perFrameCam ? (size_t)f * ndist : 0), ndist);
    double mx[4], my[4];
#pragma unroll
    for (int i = 0; i < 4; i++) undistort_point(cam, (double)corners[(size_t)f * 8 + 2 * i], (double)corners[(size_t)f * 8 + 2 * i + 1], mx[i], my[i]);
    const double hl = (double)(float)(size[f] / 2);
    // homography unit square -> quad (Heckbert), corners TL,TR,BR,BL <-> (0,0),(1,0),(1,1),(0,1)
    const double sx = mx[0] - mx[1] + mx[2] - mx[3], sy = my[0] - my[1] + my[2] - my[3];
    const double dx1 = mx[1] - mx[2], dx2 = mx[3] - mx[2], dy1 = my[1] - my[2], dy2 = my[3] - my[2];
    const double den = dx1 * dy2 - dx2 * dy1;
    bool ok = fabs(den) > 1e-300 && hl > 0;
    const double g = (sx * dy2 - dx2 * sy) / den, h = (dx1 * sy - sx * dy1) / den;
    const double Hu[9] = {mx[1] - mx[0] + g * mx[1], mx[3] - mx[0] + h * mx[3], mx[0],
                          my[1] - my[0] + g * my[1], my[3] - my[0] + h * my[3], my[0], g, h, 1.0};
    // canonical square (X,Y) in [-hl,hl]^2, y up:  u = (X+hl)/(2hl), v = (hl-Y)/(2hl)
    const double i2 = 1.0 / (2 * hl);
    const double A[9] = {i2, 0, 0.5, 0, -i2, 0.5, 0, 0, 1};
    double H[9];
    mat3_mul(Hu, A, H);
    if (fabs(H[8]) < 1e-300) ok = false;
    const double ih = 1.0 / H[8];
#pragma unroll
    for (int i = 0; i < 9; i++) H[i] *= ih;
    const double j00 = H[0] - H[6] * H[2], j01 = H[1] - H[7] * H[2], j10 = H[3] - H[6] * H[5], j11 = H[4] - H[7] * H[5];
    const double p = H[2], q = H[5];
    // rotation taking the optical axis onto the ray through the marker centre
    double Rv[9];
    const double s = sqrt(p * p + q * q + 1), t = sqrt(p * p + q * q);
    const double costh = 1. / s, sinth = sqrt(1 - 1. / (s * s));
    if (t <= 2.220446049250313e-16) {
#pragma unroll
        for (int i = 0; i < 9; i++) Rv[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        const double kx = p / t, ky = q / t;
        const double Kc[9] = {0, 0, kx, 0, 0, ky, -kx, -ky, 0};
        double K2[9];
        mat3_mul(Kc, Kc, K2);
#pragma unroll
        for (int i = 0; i < 9; i++) Rv[i] = ((i % 4 == 0) ? 1.0 : 0.0) + sinth * Kc[i] + (1 - costh) * K2[i];
    }
    const double b00 = Rv[0] - p * Rv[6], b01 = Rv[1] - p * Rv[7], b10 = Rv[3] - q * Rv[6], b11 = Rv[4] - q * Rv[7];
    const double dtinv = 1.0 / (b00 * b11 - b01 * b10);
    const double bi00 = dtinv * b11, bi01 = -dtinv * b01, bi10 = -dtinv * b10, bi11 = dtinv * b00;
    const double a00 = bi00 * j00 + bi01 * j10, a01 = bi00 * j01 + bi01 * j11;
    const double a10 = bi10 * j00 + bi11 * j10, a11 = bi10 * j01 + bi11 * j11;
    const double ata00 = a00 * a00 + a01 * a01, ata01 = a00 * a10 + a01 * a11, ata11 = a10 * a10 + a11 * a11;
    const double gamma2 = 0.5 * (ata00 + ata11 + sqrt((ata00 - ata11) * (ata00 - ata11) + 4.0 * ata01 * ata01));
    const double gamma = sqrt(fmax(gamma2, 0.));
    const double r00 = a00 / gamma, r01 = a01 / gamma, r10 = a10 / gamma, r11 = a11 / gamma;
    const double b0 = sqrt(fmax(1 - r00 * r00 - r10 * r10, 0.));
    double b1 = sqrt(fmax(1 - r01 * r01 - r11 * r11, 0.));
    if (-r00 * r01 - r10 * r11 < 0) b1 = -b1;
    const double sqx[4] = {-hl, hl, hl, -hl}, sqy[4] = {hl, hl, -hl, -hl};
    double bestErr = 0, bestR[9], bestT[3];
#pragma unroll
    for (int sol = 0; sol < 2; sol++) {
        const double sg = sol == 0 ? 1.0 : -1.0;
        const double c0[3] = {r00, r10, sg * b0}, c1[3] = {r01, r11, sg * b1};
        const double c2[3] = {c0[1] * c1[2] - c0[2] * c1[1], c0[2] * c1[0] - c0[0] * c1[2], c0[0] * c1[1] - c0[1] * c1[0]};
        const double Rt[9] = {c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2]};
        double R[9];
        mat3_mul(Rv, Rt, R);
        // translation: least squares on tx - x tz = x rz - rx ; ty - y tz = y rz - ry
        double AA[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double rx = R[0] * sqx[i] + R[1] * sqy[i], ry = R[3] * sqx[i] + R[4] * sqy[i], rz = R[6] * sqx[i] + R[7] * sqy[i];
            const double e0 = mx[i] * rz - rx, e1 = my[i] * rz - ry;
            const double q0[3] = {1, 0, -mx[i]}, q1[3] = {0, 1, -my[i]};
#pragma unroll
            for (int r = 0; r < 3; r++) {
                bb[r] += q0[r] * e0 + q1[r] * e1;
#pragma unroll
                for (int c = 0; c < 3; c++) AA[r * 3 + c] += q0[r] * q0[c] + q1[r] * q1[c];
            }
        }
        bool sok = chol_factor<3>(AA);
        chol_backsolve<3>(AA, bb);
        double err = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double px = R[0] * sqx[i] + R[1] * sqy[i] + bb[0], py = R[3] * sqx[i] + R[4] * sqy[i] + bb[1], pz = R[6] * sqx[i] + R[7] * sqy[i] + bb[2];
            const double ex = px / pz - mx[i], ey = py / pz - my[i];
            err += ex * ex + ey * ey;
        }
        err = sqrt(err / 8.0);
        if (!sok) err = 1e300;
        if (sol == 0 || err < bestErr) {
            bestErr = err;
#pragma unroll
            for (int i = 0; i < 9; i++) bestR[i] = R[i];
            bestT[0] = bb[0]; bestT[1] = bb[1]; bestT[2] = bb[2];
        }
    }
    double rv[3];
    rodrigues_m2v(bestR, rv);
    bool fin = ok;
#pragma unroll
    for (int i = 0; i < 3; i++) fin = fin && isfinite(rv[i]) && isfinite(bestT[i]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        rvec[(size_t)f * 3 + i] = fin ? rv[i] : 0.0;
        tvec[(size_t)f * 3 + i] = fin ? bestT[i] : 0.0;
    }
    if (okOut) okOut[f] = fin ? 1 : 0;
}

// ---- processArucoFromMultipleCameras' per-marker work on the device (aruco.py:417-470) ----
// entry e = frame * nMoving + m: the image corners of moving marker m in that frame (the last of several detections
// with that id: detectAruco's dictionary keeps the last, aruco.py:270-271), with the frame's camera, for k_ippe.
// valid[e] = the frame's camera was located (its anchor PnP succeeded) and the marker was seen.
__global__ void k_gather_moving(const int32_t* __restrict__ ids, const float* __restrict__ corners, const int32_t* __restrict__ counts,
                                int markerOut, const uint8_t* __restrict__ poseOk, int B, const int32_t* __restrict__ movingIds,
                                const double* __restrict__ movingSizes, int M, const double* __restrict__ K,
                                const double* __restrict__ dist, int ndist, int perFrameCam, float* __restrict__ outCorners,
                                double* __restrict__ outSize, double* __restrict__ outK, double* __restrict__ outDist,
                                uint8_t* __restrict__ valid) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * M) return;
    const int f = e / M, m = e - f * M;
    const int n = min(max(counts[f], 0), markerOut);
    int at = -1;
    for (int i = 0; i < n; i++)
        if (ids[(size_t)f * markerOut + i] == movingIds[m]) at = i;
#pragma unroll
    for (int k = 0; k < 8; k++) outCorners[(size_t)e * 8 + k] = at >= 0 ? corners[((size_t)f * markerOut + at) * 8 + k] : 0.f;
    outSize[e] = movingSizes[m];
    const double* Kf = K + (perFrameCam ? (size_t)f * 9 : 0);
    for (int k = 0; k < 9; k++) outK[(size_t)e * 9 + k] = Kf[k];
    for (int k = 0; k < 8; k++) outDist[(size_t)e * 8 + k] = k < ndist ? dist[(perFrameCam ? (size_t)f * ndist : 0) + k] : 0.0;
    valid[e] = (at >= 0 && poseOk[f] == 1) ? 1 : 0;
}

// One lane per moving marker: aruco2world = cam2world @ aruco2cam for every camera that located it (cam2world = the
// inverse of the camera's world2cam: R^T, -R^T t; Transform.invert / combine), applied to the marker's corners
// (-s/2, s/2, 0), (s/2, s/2, 0), (s/2, -s/2, 0), (-s/2, -s/2, 0) (aruco.py:434-441), summed over the cameras in frame
// order and divided by their number (np.mean(axis=0), aruco.py:468).
__global__ void k_fuse_moving(const double* __restrict__ camRvec, const double* __restrict__ camTvec, int B, int M,
                              const double* __restrict__ mRvec, const double* __restrict__ mTvec, const uint8_t* __restrict__ mOk,
                              const uint8_t* __restrict__ valid, const double* __restrict__ movingSizes,
                              double* __restrict__ fused, int32_t* __restrict__ seen, uint8_t* __restrict__ located) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const double h = movingSizes[m] / 2;
    const double lx[4] = {-h, h, h, -h}, ly[4] = {h, h, -h, -h};
    double acc[12];
#pragma unroll
    for (int i = 0; i < 12; i++) acc[i] = 0;
    int cnt = 0;
    for (int f = 0; f < B; f++) {
        const int e = f * M + m;
        const bool ok = valid[e] && mOk[e] == 1;
        located[e] = ok ? 1 : 0;
        if (!ok) continue;
        double Rc[9], Rm[9];
        rodrigues_v2m(camRvec + (size_t)f * 3, Rc, nullptr);
        rodrigues_v2m(mRvec + (size_t)e * 3, Rm, nullptr);
        const double* tc = camTvec + (size_t)f * 3;
        const double* tm = mTvec + (size_t)e * 3;
        // aruco2world: rotation Rc^T Rm, translation Rc^T tm - Rc^T tc
        double Rw[9], tw[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int c2 = 0; c2 < 3; c2++) Rw[r * 3 + c2] = Rc[0 * 3 + r] * Rm[0 * 3 + c2] + Rc[1 * 3 + r] * Rm[1 * 3 + c2] + Rc[2 * 3 + r] * Rm[2 * 3 + c2];
            const double inv_t = -(Rc[0 * 3 + r] * tc[0] + Rc[1 * 3 + r] * tc[1] + Rc[2 * 3 + r] * tc[2]);
            tw[r] = Rc[0 * 3 + r] * tm[0] + Rc[1 * 3 + r] * tm[1] + Rc[2 * 3 + r] * tm[2] + inv_t;
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int r = 0; r < 3; r++) acc[k * 3 + r] += Rw[r * 3 + 0] * lx[k] + Rw[r * 3 + 1] * ly[k] + tw[r];
        cnt++;
    }
    seen[m] = cnt;
#pragma unroll
    for (int i = 0; i < 12; i++) fused[(size_t)m * 12 + i] = cnt ? acc[i] / cnt : 0.0;
}

}  // namespace

void launch_pnp(const PnpArgs& a, hipStream_t st) {
    if (a.B <= 0) return;
    const dim3 grid(min(a.B, 256 * 16));
    if (a.N <= 64) hipLaunchKernelGGL(k_pnp<1>, grid, dim3(64), 0, st, a);
    else if (a.N <= 128) hipLaunchKernelGGL(k_pnp<2>, grid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL(k_pnp<4>, grid, dim3(64), 0, st, a);
}

void launch_gather_anchors(const int32_t* ids, const float* corners, const int32_t* counts, int B, int markerOut,
                           const int32_t* anchorIds, const float* anchorObj, int nAnchor, int N, float* obj,
                           float* img, int32_t* npts, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_gather_anchors, dim3((B + 63) / 64), dim3(64), 0, st, ids, corners, counts, B, markerOut,
                       anchorIds, anchorObj, nAnchor, N, obj, img, npts);
}

void launch_ippe(const float* corners, const double* size, int B, const double* K, const double* dist, int ndist,
                 int perFrameCam, double* rvec, double* tvec, uint8_t* ok, hipStream_t st) {
    if (B <= 0) return;
    hipLaunchKernelGGL(k_ippe, dim3((B + 63) / 64), dim3(64), 0, st, corners, size, B, K, dist, ndist, perFrameCam, rvec,
                       tvec, ok);
}

void launch_fuse_moving(const int32_t* ids, const float* corners, const int32_t* counts, int markerOut, const uint8_t* poseOk,
                        const double* camRvec, const double* camTvec, int B, const int32_t* movingIds, const double* movingSizes,
                        int M, const double* K, const double* dist, int ndist, int perFrameCam, float* eCorners, double* eSize,
                        double* eK, double* eDist, uint8_t* eValid, double* mRvec, double* mTvec, uint8_t* mOk, double* fused,
                        int32_t* seen, uint8_t* located, hipStream_t st) {
    const int E = B * M;
    if (E <= 0) return;
    hipLaunchKernelGGL(k_gather_moving, dim3((E + 63) / 64), dim3(64), 0, st, ids, corners, counts, markerOut, poseOk, B, movingIds, movingSizes, M,
                       K, dist, ndist, perFrameCam, eCorners, eSize, eK, eDist, eValid);
    hipLaunchKernelGGL(k_ippe, dim3((E + 63) / 64), dim3(64), 0, st, eCorners, eSize, E, eK, eDist, 8, 1, mRvec, mTvec, mOk);
    hipLaunchKernelGGL(k_fuse_moving, dim3((M + 63) / 64), dim3(64), 0, st, camRvec, camTvec, B, M, mRvec, mTvec, mOk, eValid, movingSizes,
                       fused, seen, located);
}

}  // namespace camsync
