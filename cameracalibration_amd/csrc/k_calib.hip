// k_calib.hip — K12: the Levenberg-Marquardt refinement inside
// cv.calibrateCamera (reference camera_sync/camera.py:61-63; OpenCV
// calibration.cpp cvCalibrateCamera2Internal).
//
// The unknowns are NI intrinsics (fx fy cx cy k1 k2 p1 p2 k3 [k4 k5 k6]) shared
// by all views plus 6 extrinsics per view, so the damped normal equations have
// arrow structure.  One wavefront per view (one lane per board point, P <= 64)
// forms that view's blocks by wave reductions
//      Aee = Je^T Je (6x6)   Bie = Ji^T Je (NI x 6)   Aii_v = Ji^T Ji   g_e, g_i
// eliminates the view's extrinsics (Schur complement)
//      S_v = Aii_v' - Bie Aee'^-1 Bie^T      r_v = g_i - Bie Aee'^-1 g_e
// (' = diagonal scaled by 1 + lambda, OpenCV's damping) and stores S_v, r_v and
// the view's squared error.  A second kernel sums the per-view records in a
// fixed order (bitwise reproducible).  Only that (NI^2 + NI + 1)-double record
// crosses GPUs: the caller all-reduces it, solves the NI x NI system and the
// views back-substitute locally.  It is the same linear system OpenCV solves
// densely, so the iterates agree to rounding.
#include "internal.h"
#include "pose_math.h"
#include "homography.h"

namespace camsync {

namespace {

template <int NI>
__global__ __launch_bounds__(64) void k_calib_accumulate(CalibState s, double lambda) {
    constexpr int L = NI * NI + NI + 1;
    const int lane = threadIdx.x;
    for (int v = blockIdx.x; v < s.V; v += gridDim.x) {
        const bool active = lane < s.P;
        Cam cam;
        cam.fx = s.intr[0]; cam.fy = s.intr[1]; cam.cx = s.intr[2]; cam.cy = s.intr[3];
#pragma unroll
        for (int i = 0; i < 8; i++) cam.k[i] = (i < NI - 4) ? s.intr[4 + i] : 0.0;
        double X = 0, Y = 0, Z = 0, ui = 0, vi = 0;
        if (active) {
            const float* o = s.obj + ((size_t)v * s.P + lane) * 3;
            const float* m = s.img + ((size_t)v * s.P + lane) * 2;
            X = o[0]; Y = o[1]; Z = o[2];
            ui = m[0]; vi = m[1];
        }
        const double* rt = s.ext + (size_t)v * 6;
        double R[9], dR[27];
        rodrigues_v2m(rt, R, dR);
        const double Xc = R[0] * X + R[1] * Y + R[2] * Z + rt[3];
        const double Yc = R[3] * X + R[4] * Y + R[5] * Z + rt[4];
        const double Zc = R[6] * X + R[7] * Y + R[8] * Z + rt[5];
        double u, w, duP[3], dvP[3], duI[12], dvI[12];
        project_cam<true, true>(cam, Xc, Yc, Zc, u, w, duP, dvP, duI, dvI);
        const double a0 = active ? 1.0 : 0.0;
        const double ex = (u - ui) * a0, ey = (w - vi) * a0;
        double ju[6], jv[6];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const double* D = dR + q * 9;
            const double dP0 = D[0] * X + D[1] * Y + D[2] * Z, dP1 = D[3] * X + D[4] * Y + D[5] * Z, dP2 = D[6] * X + D[7] * Y + D[8] * Z;
            ju[q] = (duP[0] * dP0 + duP[1] * dP1 + duP[2] * dP2) * a0;
            jv[q] = (dvP[0] * dP0 + dvP[1] * dP1 + dvP[2] * dP2) * a0;
            ju[3 + q] = duP[q] * a0;
            jv[3 + q] = dvP[q] * a0;
        }
#pragma unroll
        for (int q = 0; q < NI; q++) { duI[q] *= a0; dvI[q] *= a0; }

        double A[36], ge[6], B[NI * 6], S[NI * NI], gi[NI];
        const double err = wsum(ex * ex + ey * ey);
#pragma unroll
        for (int a = 0; a < 6; a++) {
            ge[a] = wsum(ju[a] * ex + jv[a] * ey);
#pragma unroll
            for (int b = a; b < 6; b++) {
                const double t = wsum(ju[a] * ju[b] + jv[a] * jv[b]);
                A[a * 6 + b] = t;
                A[b * 6 + a] = t;
            }
        }
#pragma unroll
        for (int r = 0; r < NI; r++) {
            gi[r] = wsum(duI[r] * ex + dvI[r] * ey);
#pragma unroll
            for (int c = 0; c < 6; c++) B[r * 6 + c] = wsum(duI[r] * ju[c] + dvI[r] * jv[c]);
#pragma unroll
            for (int c = r; c < NI; c++) {
                const double t = wsum(duI[r] * duI[c] + dvI[r] * dvI[c]);
                S[r * NI + c] = t;
                S[c * NI + r] = t;
            }
        }
        // damping, then eliminate the extrinsics
#pragma unroll
        for (int r = 0; r < NI; r++) S[r * NI + r] *= 1. + lambda;
#pragma unroll
        for (int r = 0; r < 6; r++) A[r * 6 + r] *= 1. + lambda;
        chol_factor<6>(A);
        double* rec = s.viewBlocks + (size_t)v * (6 * NI + 6);  // Aee'^-1 Bie^T (6 x NI), Aee'^-1 g_e (6)
        double x[6];
#pragma unroll
        for (int q = 0; q < 6; q++) x[q] = ge[q];
        chol_backsolve<6>(A, x);
        double Ainvg[6];
#pragma unroll
        for (int q = 0; q < 6; q++) Ainvg[q] = x[q];
#pragma unroll
        for (int r = 0; r < NI; r++) {
            double t = 0;
#pragma unroll
            for (int q = 0; q < 6; q++) t += B[r * 6 + q] * Ainvg[q];
            gi[r] -= t;
        }
#pragma unroll
        for (int c = 0; c < NI; c++) {
#pragma unroll
            for (int q = 0; q < 6; q++) x[q] = B[c * 6 + q];
            chol_backsolve<6>(A, x);  // column c of Aee'^-1 Bie^T
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 6; q++) rec[q * NI + c] = x[q];
            }
#pragma unroll
            for (int r = 0; r < NI; r++) {
                double t = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) t += B[r * 6 + q] * x[q];
                S[r * NI + c] -= t;
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 6; q++) rec[6 * NI + q] = Ainvg[q];
        }
        // per-view record; lanes share the stores
        double* out = s.partial + (size_t)v * L;
#pragma unroll
        for (int i = 0; i < NI * NI; i++)
            if (lane == (i & 63)) out[i] = S[i];
#pragma unroll
        for (int i = 0; i < NI; i++)
            if (lane == i) out[NI * NI + i] = gi[i];
        if (lane == 0) out[NI * NI + NI] = err;
    }
}

// acc[l] = sum over views of partial[v][l], fixed pairwise order
__global__ __launch_bounds__(256) void k_reduce_views(const double* __restrict__ partial, int V, int L, double* __restrict__ acc) {
    __shared__ double sm[256];
    const int l = blockIdx.x;
    double t = 0;
    for (int v = threadIdx.x; v < V; v += 256) t += partial[(size_t)v * L + l];
    sm[threadIdx.x] = t;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) sm[threadIdx.x] += sm[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) acc[l] = sm[0];
}

// trial parameters: intrinsics from the solved delta, extrinsics by back-substitution
template <int NI>
__global__ void k_calib_backsub(CalibState s, const double* __restrict__ di) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) {
        for (int i = 0; i < 12; i++) s.intrTrial[i] = (i < NI) ? s.intr[i] - di[i] : 0.0;
    }
    if (v >= s.V) return;
    const double* rec = s.viewBlocks + (size_t)v * (6 * NI + 6);
    for (int q = 0; q < 6; q++) {
        double t = rec[6 * NI + q];
        for (int c = 0; c < NI; c++) t -= rec[q * NI + c] * di[c];
        s.extTrial[(size_t)v * 6 + q] = s.ext[(size_t)v * 6 + q] - t;
    }
}

// squared reprojection error of every local view at the trial parameters
template <int NI>
__global__ __launch_bounds__(64) void k_calib_error(CalibState s, int useTrial) {
    const int lane = threadIdx.x;
    const double* intr = useTrial ? s.intrTrial : s.intr;
    const double* ext = useTrial ? s.extTrial : s.ext;
    for (int v = blockIdx.x; v < s.V; v += gridDim.x) {
        const bool active = lane < s.P;
        Cam cam;
        cam.fx = intr[0]; cam.fy = intr[1]; cam.cx = intr[2]; cam.cy = intr[3];
#pragma unroll
        for (int i = 0; i < 8; i++) cam.k[i] = (i < NI - 4) ? intr[4 + i] : 0.0;
        double X = 0, Y = 0, Z = 0, ui = 0, vi = 0;
        if (active) {
            const float* o = s.obj + ((size_t)v * s.P + lane) * 3;
            const float* m = s.img + ((size_t)v * s.P + lane) * 2;
            X = o[0]; Y = o[1]; Z = o[2];
            ui = m[0]; vi = m[1];
        }
        const double* rt = ext + (size_t)v * 6;
        double R[9];
        rodrigues_v2m(rt, R, nullptr);
        const double Xc = R[0] * X + R[1] * Y + R[2] * Z + rt[3];
        const double Yc = R[3] * X + R[4] * Y + R[5] * Z + rt[4];
        const double Zc = R[6] * X + R[7] * Y + R[8] * Z + rt[5];
        double u, w;
        project_cam<false, false>(cam, Xc, Yc, Zc, u, w, nullptr, nullptr, nullptr, nullptr);
        const double ex = active ? u - ui : 0.0, ey = active ? w - vi : 0.0;
        const double e = wsum(ex * ex + ey * ey);
        if (lane == 0) s.errPartial[v] = e;
    }
}

// cvInitIntrinsicParams2D: per view, the homography board -> pixels and its two
// rows of the 2-unknown least squares for (1/fx^2, 1/fy^2); the per-view
// record is [a00 a01 a11 b0 b1 valid].
__global__ __launch_bounds__(64) void k_calib_focal_rows(CalibState s, double cx, double cy, double* __restrict__ rows) {
    const int lane = threadIdx.x;
    for (int v = blockIdx.x; v < s.V; v += gridDim.x) {
        const bool active = lane < s.P;
        double Mx = 0, My = 0, mx = 0, my = 0;
        if (active) {
            const float* o = s.obj + ((size_t)v * s.P + lane) * 3;
            const float* m = s.img + ((size_t)v * s.P + lane) * 2;
            Mx = o[0]; My = o[1];
            mx = m[0]; my = m[1];
        }
        double H[9];
        const bool ok = wave_homography(Mx, My, mx, my, active, s.P, H);
        double rec[6] = {0, 0, 0, 0, 0, 0};
        if (ok) {
            H[0] -= H[6] * cx; H[1] -= H[7] * cx; H[2] -= H[8] * cx;
            H[3] -= H[6] * cy; H[4] -= H[7] * cy; H[5] -= H[8] * cy;
            double h[3], w[3], d1[3], d2[3], n[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double t0 = H[j * 3], t1 = H[j * 3 + 1];
                h[j] = t0; w[j] = t1;
                d1[j] = (t0 + t1) * 0.5;
                d2[j] = (t0 - t1) * 0.5;
                n[0] += t0 * t0; n[1] += t1 * t1; n[2] += d1[j] * d1[j]; n[3] += d2[j] * d2[j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) n[j] = 1. / sqrt(n[j]);
#pragma unroll
            for (int j = 0; j < 3; j++) { h[j] *= n[0]; w[j] *= n[1]; d1[j] *= n[2]; d2[j] *= n[3]; }
            const double A0[2] = {h[0] * w[0], h[1] * w[1]}, A1[2] = {d1[0] * d2[0], d1[1] * d2[1]};
            const double b0 = -h[2] * w[2], b1 = -d1[2] * d2[2];
            rec[0] = A0[0] * A0[0] + A1[0] * A1[0];
            rec[1] = A0[0] * A0[1] + A1[0] * A1[1];
            rec[2] = A0[1] * A0[1] + A1[1] * A1[1];
            rec[3] = A0[0] * b0 + A1[0] * b1;
            rec[4] = A0[1] * b0 + A1[1] * b1;
            rec[5] = 1;
        }
        if (lane < 6) rows[(size_t)v * 6 + lane] = rec[lane];
    }
}

// Solve the reduced NI x NI system S d = rhs on the device (one lane; 12 x 12 at most): Cholesky, with a ridge retry
// when a pivot vanishes -- the arithmetic of camsync_calib_solve, in the same order.  flag[0] = 1 when both fail.
template <int NI>
__global__ void k_calib_solve(const double* __restrict__ acc, double* __restrict__ delta, double* __restrict__ flag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double A[NI * NI], b[NI];
    for (int attempt = 0; attempt < 2; attempt++) {
        double tr = 0;
        for (int i = 0; i < NI; i++) tr += acc[i * NI + i];
        for (int i = 0; i < NI * NI; i++) A[i] = acc[i];
        if (attempt)
            for (int i = 0; i < NI; i++) A[i * NI + i] += 1e-14 * tr + 1e-300;
        for (int i = 0; i < NI; i++) b[i] = acc[NI * NI + i];
        bool ok = true;
        for (int j = 0; j < NI && ok; j++) {
            double s = A[j * NI + j];
            for (int k = 0; k < j; k++) s -= A[j * NI + k] * A[j * NI + k];
            if (!(s > 0)) { ok = false; break; }
            const double d = sqrt(s);
            A[j * NI + j] = d;
            for (int i = j + 1; i < NI; i++) {
                double t = A[i * NI + j];
                for (int k = 0; k < j; k++) t -= A[i * NI + k] * A[j * NI + k];
                A[i * NI + j] = t / d;
            }
        }
        if (!ok) continue;
        for (int i = 0; i < NI; i++) {
            double s = b[i];
            for (int k = 0; k < i; k++) s -= A[i * NI + k] * b[k];
            b[i] = s / A[i * NI + i];
        }
        for (int i = NI - 1; i >= 0; i--) {
            double s = b[i];
            for (int k = i + 1; k < NI; k++) s -= A[k * NI + i] * b[k];
            b[i] = s / A[i * NI + i];
        }
        for (int i = 0; i < NI; i++) delta[i] = b[i];
        flag[0] = 0.0;
        return;
    }
    for (int i = 0; i < NI; i++) delta[i] = 0.0;
    flag[0] = 1.0;
}

// Parameter change of a trial step, for the LM stopping rule ||p_trial - p|| / ||p||: rec[1] = sum over the local
// extrinsics of (trial - current)^2, rec[2] = sum of current^2 (both to be summed over ranks), rec[3], rec[4] = the same
// over the intrinsics (replicated on every rank: not to be summed).  rec[0] already holds the trial error.
__global__ __launch_bounds__(256) void k_calib_change(CalibState s, double* __restrict__ rec) {
    __shared__ double sd[256], sp[256];
    double dn = 0, pn = 0;
    const int n = s.V * 6;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double c = s.ext[i], d = s.extTrial[i] - c;
        dn += d * d;
        pn += c * c;
    }
    sd[threadIdx.x] = dn;
    sp[threadIdx.x] = pn;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { sd[threadIdx.x] += sd[threadIdx.x + d]; sp[threadIdx.x] += sp[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        rec[1] = sd[0];
        rec[2] = sp[0];
        double di = 0, pi = 0;
        for (int i = 0; i < 12; i++) {
            const double c = s.intr[i], d = s.intrTrial[i] - c;
            di += d * d;
            pi += c * c;
        }
        rec[3] = di;
        rec[4] = pi;
    }
}

__global__ void k_copy_f64(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

}  // namespace

void launch_calib_accumulate(const CalibState& s, double lambda, hipStream_t st) {
    const int L = s.NI * s.NI + s.NI + 1;
    const int grid = min(max(s.V, 1), 256 * 8);
    if (s.NI == 9) hipLaunchKernelGGL(k_calib_accumulate<9>, dim3(grid), dim3(64), 0, st, s, lambda);
    else hipLaunchKernelGGL(k_calib_accumulate<12>, dim3(grid), dim3(64), 0, st, s, lambda);
    hipLaunchKernelGGL(k_reduce_views, dim3(L), dim3(256), 0, st, s.partial, s.V, L, s.acc);
}

void launch_calib_backsub(const CalibState& s, const double* di, hipStream_t st) {
    const int grid = (max(s.V, 1) + 63) / 64;
    if (s.NI == 9) hipLaunchKernelGGL(k_calib_backsub<9>, dim3(grid), dim3(64), 0, st, s, di);
    else hipLaunchKernelGGL(k_calib_backsub<12>, dim3(grid), dim3(64), 0, st, s, di);
}

void launch_calib_error(const CalibState& s, int useTrial, hipStream_t st) {
    const int grid = min(max(s.V, 1), 256 * 8);
    if (s.NI == 9) hipLaunchKernelGGL(k_calib_error<9>, dim3(grid), dim3(64), 0, st, s, useTrial);
    else hipLaunchKernelGGL(k_calib_error<12>, dim3(grid), dim3(64), 0, st, s, useTrial);
    hipLaunchKernelGGL(k_reduce_views, dim3(1), dim3(256), 0, st, s.errPartial, s.V, 1, s.errOut);
}

void launch_calib_focal_rows(const CalibState& s, double cx, double cy, double* rows, double* acc6, hipStream_t st) {
    const int grid = min(max(s.V, 1), 256 * 8);
    hipLaunchKernelGGL(k_calib_focal_rows, dim3(grid), dim3(64), 0, st, s, cx, cy, rows);
    hipLaunchKernelGGL(k_reduce_views, dim3(6), dim3(256), 0, st, rows, s.V, 6, acc6);
}

void launch_calib_solve(const CalibState& s, double* delta, double* flag, hipStream_t st) {
    if (s.NI == 9) hipLaunchKernelGGL(k_calib_solve<9>, dim3(1), dim3(1), 0, st, s.acc, delta, flag);
    else hipLaunchKernelGGL(k_calib_solve<12>, dim3(1), dim3(1), 0, st, s.acc, delta, flag);
}

void launch_calib_change(const CalibState& s, double* rec, hipStream_t st) {
    hipLaunchKernelGGL(k_calib_change, dim3(1), dim3(256), 0, st, s, rec);
}

void launch_calib_accept(const CalibState& s, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_f64, dim3(1), dim3(64), 0, st, s.intrTrial, s.intr, 12);
    const int n = s.V * 6;
    if (n > 0) hipLaunchKernelGGL(k_copy_f64, dim3((n + 255) / 256), dim3(256), 0, st, s.extTrial, s.ext, n);
}

}  // namespace camsync
