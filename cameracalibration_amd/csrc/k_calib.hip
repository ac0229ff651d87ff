// k_calib.hip — K12: the Levenberg-Marquardt refinement inside
// cv.calibrateCamera (reference camera_sync/camera.py:61-63; OpenCV
// calibration.cpp cvCalibrateCamera2Internal).
//
// The unknowns are NI intrinsics (fx fy cx cy k1 k2 p1 p2 k3 [k4 k5 k6]) shared
// by all views plus 6 extrinsics per view, so the damped normal equations have
// arrow structure.  One wavefront per view (one lane per board point, P <= 64):
// the lanes write the rows of the view's Jacobian [Je | Ji | e] (2P x (NI + 7))
// to LDS and the wave forms M = J^T J with v_mfma_f64_16x16x4_f64 (the rows are
// the k dimension), which holds every block of the view at once:
//      Aee = Je^T Je (6x6)   Bie = Ji^T Je (NI x 6)   Aii_v = Ji^T Ji   g_e, g_i, e^T e
// The view's extrinsics are eliminated (Schur complement)
//      S_v = Aii_v' - Bie Aee'^-1 Bie^T      r_v = g_i - Bie Aee'^-1 g_e
// (' = diagonal scaled by 1 + lambda, OpenCV's damping), S_v, r_v and the squared
// error are summed over the wave's views in registers, over the workgroup's waves
// in LDS and over the workgroups by the last one to finish, all in a fixed order
// (bitwise reproducible).  Only that (NI^2 + NI + 3)-double record crosses GPUs.
//
// ONE kernel is a whole trial step of the LM loop: solve the reduced NI x NI
// system of the record in place (every workgroup, redundantly), back-substitute
// its views' extrinsics, evaluate the trial parameters -- error AND the next
// record, damped for the case that the trial is accepted -- and reduce.  It is
// the same linear system OpenCV solves densely, so the iterates agree to rounding.
#include "internal.h"
#include "pose_math.h"
#include "homography.h"

namespace camsync {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// A wavefront's LDS operations execute in order: exchanging data between its lanes needs only the compiler to keep
// the order.
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double lane_bcast(double v, int src) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)u, src), hi = __builtin_amdgcn_readlane((unsigned int)(u >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Wait until this wave's outstanding vector memory operations (the write-through record stores) are acknowledged.
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int kLmWaves = 4;       // views in flight per workgroup
constexpr int kLmRecord = 160;    // doubles per partial record (>= NI^2 + NI + 3)
constexpr int kLmGroup = 32;      // records summed per level of the final reduction (at most kLmGroup^2 workgroups)

// One LM trial step (see the head of the file).
//   mode 0: evaluate the current parameters (intr, ext)
//   mode 1: solve the record in work[0..L), trial = current - step, evaluate the trial parameters
//   mode 2: as 1 with the intrinsic step given in work[kWDelta..]
// Evaluating parameters writes, for those parameters: the damped record [S | rhs | err] to work[0..L) (in place: the
// last workgroup writes it after every workgroup has read the old one), the views' elimination blocks, and for
// modes 1/2 the trial parameters, the parameter-change sums work[L] = sum (ext_trial - ext)^2, work[L+1] = sum ext^2
// over the local views (to be summed over ranks with the record) and, replicated on every rank, work[kWRec..] =
// [sum (intr_trial - intr)^2, sum intr^2, solver flag (1: singular), err of the record that was solved] and the step
// in work[kWDelta..].
template <int NI>
__global__ __launch_bounds__(64 * kLmWaves) void k_calib_lm(CalibState s, int mode, double lambda, double* __restrict__ work,
                                                            double* __restrict__ report, unsigned long long seq) {
    constexpr int L = NI * NI + NI + 1;
    constexpr int NC = NI + 7;           // columns of [Je | Ji | e]
    constexpr int LD = (NC | 1) + 1;     // row stride: even (16-byte rows), not a multiple of 16 doubles
    constexpr int kJ = 64 * LD, kM = NC * LD, kX = (NI + 1) * 6;
    constexpr int kPerWave = kJ + kM + kX;
    static_assert(L + 2 <= kLmRecord, "record");
    __shared__ double lds[kLmWaves * kPerWave];
    __shared__ double sol[(NI + 1) * NI + NI + 4];  // the system being solved, then the step; flag; old error
    __shared__ double comb[kLmWaves][kLmRecord];
    __shared__ unsigned int lastBlock;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* const J = lds + wave * kPerWave;
    double* const M = J + kJ;
    double* const X = M + kM;
    double* const step = sol + (NI + 1) * NI;  // [NI] step, [NI] flag, [NI + 1] error of the record solved

    // ---- the intrinsic step ----
    if (mode == 1 && wave == 0) {
        // Cholesky of the reduced system, one lane per row, the right-hand side as row NI (its factor row is the
        // forward substitution); a ridge retry when a pivot vanishes.  Element for element the arithmetic of
        // camsync_calib_solve, up to the order of the back substitution's sums.
        bool ok = false;
        for (int attempt = 0; attempt < 2 && !ok; attempt++) {
            double tr = 0;
            for (int i = 0; i < NI; i++) tr += work[i * NI + i];
            if (lane <= NI) {
                for (int j = 0; j < NI; j++) {
                    double v = lane < NI ? work[lane * NI + j] : work[NI * NI + j];
                    if (attempt && lane == j) v += 1e-14 * tr + 1e-300;
                    sol[lane * NI + j] = v;
                }
            }
            wave_lds_order();
            ok = true;
            for (int j = 0; j < NI; j++) {
                double p = sol[j * NI + j];
                for (int k = 0; k < j; k++) p -= sol[j * NI + k] * sol[j * NI + k];
                if (!(p > 0)) { ok = false; break; }
                const double d = sqrt(p);
                double t = 0;
                if (lane > j && lane <= NI) {
                    t = sol[lane * NI + j];
                    for (int k = 0; k < j; k++) t -= sol[lane * NI + k] * sol[j * NI + k];
                    t = t / d;
                }
                wave_lds_order();
                if (lane > j && lane <= NI) sol[lane * NI + j] = t;
                if (lane == j) sol[j * NI + j] = d;
                wave_lds_order();
            }
        }
        double x[NI];
#pragma unroll
        for (int i = NI - 1; i >= 0; i--) {
            double v = sol[NI * NI + i];
#pragma unroll
            for (int k = i + 1; k < NI; k++) v -= sol[k * NI + i] * x[k];
            x[i] = v / sol[i * NI + i];
        }
        wave_lds_order();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NI; i++) step[i] = ok ? x[i] : 0.0;
            step[NI] = ok ? 0.0 : 1.0;
        }
    } else if (wave == 0 && lane == 0) {
        for (int i = 0; i < NI; i++) step[i] = mode == 2 ? work[kWDelta + i] : 0.0;
        step[NI] = 0.0;
    }
    if (tid == 64) step[NI + 1] = work[L - 1];
    __syncthreads();

    const bool trial = mode != 0;
    Cam cam;
    {
        double in[12];
#pragma unroll
        for (int i = 0; i < 12; i++) in[i] = i < NI ? (trial ? s.intr[i] - step[i] : s.intr[i]) : 0.0;
        cam.fx = in[0]; cam.fy = in[1]; cam.cx = in[2]; cam.cy = in[3];
#pragma unroll
        for (int i = 0; i < 8; i++) cam.k[i] = in[4 + i];
        if (trial && blockIdx.x == 0 && tid < 12) {
            double v = 0;
#pragma unroll
            for (int i = 0; i < 12; i++) v = tid == i ? in[i] : v;
            s.intrTrial[tid] = v;
        }
    }
    double* const blocksOut = trial ? s.viewBlocksTrial : s.viewBlocks;

    double acc0 = 0, acc1 = 0, acc2 = 0, dn = 0, pn = 0;
    const int ksteps = (s.P + 3) >> 2;
    for (int v = blockIdx.x * kLmWaves + wave; v < s.V; v += gridDim.x * kLmWaves) {
        // ---- this view's extrinsics ----
        double e6 = 0;
        if (lane < 6) {
            const double cur = s.ext[(size_t)v * 6 + lane];
            e6 = cur;
            if (trial) {
                const double* rec = s.viewBlocks + (size_t)v * (6 * NI + 6);
                double t = rec[6 * NI + lane];
                for (int c = 0; c < NI; c++) t -= rec[lane * NI + c] * step[c];
                e6 = cur - t;
                s.extTrial[(size_t)v * 6 + lane] = e6;
                const double d = e6 - cur;
                dn += d * d;
                pn += cur * cur;
            }
        }
        double rt[6];
#pragma unroll
        for (int q = 0; q < 6; q++) rt[q] = lane_bcast(e6, q);

        const bool active = lane < s.P;
        double X0 = 0, Y0 = 0, Z0 = 0, ui = 0, vi = 0;
        if (active) {
            const float* o = s.obj + ((size_t)v * s.P + lane) * 3;
            const float* m = s.img + ((size_t)v * s.P + lane) * 2;
            X0 = o[0]; Y0 = o[1]; Z0 = o[2];
            ui = m[0]; vi = m[1];
        }
        double R[9], dR[27];
        rodrigues_v2m(rt, R, dR);
        const double Xc = R[0] * X0 + R[1] * Y0 + R[2] * Z0 + rt[3];
        const double Yc = R[3] * X0 + R[4] * Y0 + R[5] * Z0 + rt[4];
        const double Zc = R[6] * X0 + R[7] * Y0 + R[8] * Z0 + rt[5];
        double u, w, duP[3], dvP[3], duI[12], dvI[12];
        project_cam<true, true>(cam, Xc, Yc, Zc, u, w, duP, dvP, duI, dvI);
        const double a0 = active ? 1.0 : 0.0;
        const double ex = (u - ui) * a0, ey = (w - vi) * a0;
        double ju[6], jv[6];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const double* D = dR + q * 9;
            const double dP0 = D[0] * X0 + D[1] * Y0 + D[2] * Z0, dP1 = D[3] * X0 + D[4] * Y0 + D[5] * Z0, dP2 = D[6] * X0 + D[7] * Y0 + D[8] * Z0;
            ju[q] = (duP[0] * dP0 + duP[1] * dP1 + duP[2] * dP2) * a0;
            jv[q] = (dvP[0] * dP0 + dvP[1] * dP1 + dvP[2] * dP2) * a0;
            ju[3 + q] = duP[q] * a0;
            jv[3 + q] = dvP[q] * a0;
        }

        // ---- M = J^T J: the u rows, then the v rows, through the same LDS tile ----
        f64x4 t00 = {0, 0, 0, 0}, t01 = {0, 0, 0, 0}, t11 = {0, 0, 0, 0};
        const int col = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            double* row = J + lane * LD;
#pragma unroll
            for (int q = 0; q < 6; q++) row[q] = half ? jv[q] : ju[q];
#pragma unroll
            for (int q = 0; q < NI; q++) row[6 + q] = (half ? dvI[q] : duI[q]) * a0;
            row[6 + NI] = half ? ey : ex;
#pragma unroll
            for (int q = NC; q < LD; q++) row[q] = 0.0;
            wave_lds_order();
            for (int k = 0; k < ksteps; k++) {
                const double* r4 = J + (4 * k + kq) * LD;
                const double a = r4[col];
                t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, t00, 0, 0, 0);
                if (NC > 16) {
                    const double b = (16 + col < NC) ? r4[16 + col] : 0.0;
                    t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, t01, 0, 0, 0);
                    t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, t11, 0, 0, 0);
                }
            }
            wave_lds_order();
        }
        // lane l, register r of a tile: row (l >> 4) + 4 r, column l & 15
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = kq + 4 * r;
            M[i * LD + col] = t00[r];
            if (NC > 16) {
                if (16 + col < NC) {
                    M[i * LD + 16 + col] = t01[r];
                    M[(16 + col) * LD + i] = t01[r];
                    if (16 + i < NC) M[(16 + i) * LD + 16 + col] = t11[r];
                }
            }
        }
        wave_lds_order();

        // ---- eliminate the extrinsics ----
        double A[36];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) A[a * 6 + b] = M[a * LD + b] * (a == b ? 1. + lambda : 1.);
        chol_factor<6>(A);
        if (lane <= NI) {
            // column `lane` of Aee'^-1 [Bie^T | g_e]
            double x[6];
#pragma unroll
            for (int q = 0; q < 6; q++) x[q] = M[q * LD + 6 + lane];
            chol_backsolve<6>(A, x);
            double* rec = blocksOut + (size_t)v * (6 * NI + 6);
#pragma unroll
            for (int q = 0; q < 6; q++) {
                X[lane * 6 + q] = x[q];
                if (lane < NI) rec[q * NI + lane] = x[q];
                else rec[6 * NI + q] = x[q];
            }
        }
        wave_lds_order();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int e = lane + 64 * k;
            double val = 0;
            if (e < NI * NI + NI) {
                const int r = e < NI * NI ? e / NI : e - NI * NI, c = e < NI * NI ? e % NI : NI;
                double t = 0;
#pragma unroll
                for (int q = 0; q < 6; q++) t += M[(6 + r) * LD + q] * X[c * 6 + q];
                val = M[(6 + r) * LD + 6 + c] * (r == c ? 1. + lambda : 1.) - t;
            } else if (e == L - 1) {
                val = M[(NC - 1) * LD + NC - 1];
            }
            if (k == 0) acc0 += val;
            else if (k == 1) acc1 += val;
            else acc2 += val;
        }
        wave_lds_order();
    }

    // ---- sum over the workgroup's waves, then over the workgroups ----
    dn = wsum(dn);
    pn = wsum(pn);
    comb[wave][lane] = acc0;
    if (lane + 64 < L) comb[wave][lane + 64] = acc1;
    if (lane + 128 < L) comb[wave][lane + 128] = acc2;
    if (lane == 0) { comb[wave][L] = dn; comb[wave][L + 1] = pn; }
    __syncthreads();
    if (tid < L + 2) {
        double t = comb[0][tid];
#pragma unroll
        for (int w = 1; w < kLmWaves; w++) t += comb[w][tid];
        __hip_atomic_store(s.partial + (size_t)blockIdx.x * kLmRecord + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stores_done();
    }
    // Two levels, kLmGroup records each, so that the loads of a level are all in flight at once: the last workgroup
    // of a group sums the group's records, the last of those sums the groups.
    const unsigned int group = blockIdx.x / kLmGroup, nGroups = (gridDim.x + kLmGroup - 1) / kLmGroup;
    const unsigned int inGroup = min((unsigned int)kLmGroup, gridDim.x - group * kLmGroup);
    double* const groupRec = s.partial + (size_t)gridDim.x * kLmRecord;
    // The records cross workgroups as device-scope atomic stores and loads (write-through / read-through past this
    // XCD's L2).  Every thread waits for its own record stores to be acknowledged (stores_done) before the barrier,
    // so all of the workgroup's record is at the device's point of coherence before thread 0 takes its ticket; the
    // reader's loads are issued after its ticket's value has come back.  That orders the hand-off without a release
    // fence, which would write back the XCD's whole L2 per workgroup (40 us per launch).
    __syncthreads();
    if (tid == 0) lastBlock = atomicAdd(s.ticket + 1 + group, 1u) == inGroup - 1 ? 1u : 0u;
    __syncthreads();
    if (!lastBlock) return;
    if (tid < L + 2) {
        double v[kLmGroup];
#pragma unroll
        for (int i = 0; i < kLmGroup; i++)
            v[i] = (unsigned int)i < inGroup ? __hip_atomic_load(s.partial + ((size_t)group * kLmGroup + i) * kLmRecord + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        double t = v[0];
#pragma unroll
        for (int i = 1; i < kLmGroup; i++) t += v[i];
        __hip_atomic_store(groupRec + (size_t)group * kLmRecord + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        stores_done();
    }
    __syncthreads();
    if (tid == 0) {
        s.ticket[1 + group] = 0u;
        lastBlock = atomicAdd(s.ticket, 1u) == nGroups - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!lastBlock) return;
    if (tid < L + 2) {
        double v[kLmGroup];
#pragma unroll
        for (int i = 0; i < kLmGroup; i++)
            v[i] = (unsigned int)i < nGroups ? __hip_atomic_load(groupRec + (size_t)i * kLmRecord + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        double t = v[0];
#pragma unroll
        for (int i = 1; i < kLmGroup; i++) t += v[i];
        work[tid] = t;
        if (report && tid >= L - 1) comb[0][tid - (L - 1)] = t;  // [err, change sums of the extrinsics]: for the host, below
    }
    if (report) __syncthreads();
    if (tid == 0) {
        double di = 0, pi = 0;
        for (int i = 0; i < 12; i++) {
            const double c = i < NI ? s.intr[i] : 0.0;
            const double d = trial && i < NI ? (c - step[i]) - c : 0.0;
            di += d * d;
            pi += c * c;
        }
        work[kWRec + 0] = di;
        work[kWRec + 1] = pi;
        work[kWRec + 2] = step[NI];
        work[kWRec + 3] = step[NI + 1];
        for (int i = 0; i < 12; i++) work[kWDelta + i] = i < NI ? step[i] : 0.0;
        *s.ticket = 0u;
        if (report) {
            // One GPU: what the host decides the next step from goes straight into its (page-locked, coherent) memory,
            // and a sequence number after it -- the host polls that instead of enqueueing a copy and waiting for the
            // stream, which cost several times the kernel's 33 us.
            report[0] = comb[0][0]; report[1] = comb[0][1]; report[2] = comb[0][2];
            report[3] = di; report[4] = pi; report[5] = step[NI]; report[6] = step[NI + 1];
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(report + 8), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
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

__global__ void k_fill_i32(int32_t* __restrict__ dst, int n, int32_t value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

}  // namespace

void launch_fill_i32(int32_t* dst, int n, int32_t value, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, st, dst, n, value);
}

int calib_lm_grid(int V) { return min(max((V + kLmWaves - 1) / kLmWaves, 1), kLmGroup * kLmGroup); }

void launch_calib_lm(const CalibState& s, int mode, double lambda, double* work, hipStream_t st, double* report, unsigned long long seq) {
    const int grid = calib_lm_grid(s.V);
    if (s.NI == 9) hipLaunchKernelGGL(k_calib_lm<9>, dim3(grid), dim3(64 * kLmWaves), 0, st, s, mode, lambda, work, report, seq);
    else hipLaunchKernelGGL(k_calib_lm<12>, dim3(grid), dim3(64 * kLmWaves), 0, st, s, mode, lambda, work, report, seq);
}

void launch_calib_focal_rows(const CalibState& s, double cx, double cy, double* rows, double* acc6, hipStream_t st) {
    const int grid = min(max(s.V, 1), 256 * 8);
    hipLaunchKernelGGL(k_calib_focal_rows, dim3(grid), dim3(64), 0, st, s, cx, cy, rows);
    hipLaunchKernelGGL(k_reduce_views, dim3(6), dim3(256), 0, st, rows, s.V, 6, acc6);
}

}  // namespace camsync
