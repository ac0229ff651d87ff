// k_identify.hip — K6: _identifyOneCandidate of ArucoDetector.detectMarkers
// step 2 (reference camera_sync/aruco.py:256-264): perspective-normalise the
// quad to (markerSize+2)*cell pixels with INTER_NEAREST, mean/stddev gate,
// Otsu threshold, per-cell majority vote, border test, dictionary lookup over
// the four rotations.
//
// One wavefront per quad candidate.  Lane 0 solves the 8x8 system for the
// perspective map (same elimination order as the CPU restatement, so the
// nearest-neighbour sample positions round identically), the 64 lanes gather
// the samples, build the histogram in LDS and vote the cells; the dictionary
// is searched one entry per lane.
#include "internal.h"

namespace camsync {

namespace {

constexpr int kMaxSide = 32;  // (markerSize + 2*border) * cell <= 32

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Code of a markerSize x markerSize bit grid (given as a row-major MSB-first
// integer) seen under rotation r, as Dictionary::getByteListFromBits lays the
// four rotations out.
__device__ __forceinline__ uint32_t rotate_code(uint32_t code, int n, int r) {
    if (r == 0) return code;
    uint32_t out = 0;
    for (int row = 0; row < n; row++)
        for (int col = 0; col < n; col++) {
            int sr, sc;
            if (r == 1) { sr = col; sc = n - 1 - row; }
            else if (r == 2) { sr = n - 1 - row; sc = n - 1 - col; }
            else { sr = n - 1 - col; sc = row; }
            const uint32_t bit = (code >> (n * n - 1 - (sr * n + sc))) & 1u;
            out = (out << 1) | bit;
        }
    return out;
}

__global__ __launch_bounds__(64) void k_identify(const uint8_t* __restrict__ gray, QuadRec* __restrict__ quads,
                                                 const int* __restrict__ qcount, const uint16_t* __restrict__ dict,
                                                 DetCfg cfg) {
    __shared__ double s_a[8][9];
    __shared__ double s_mi[9];
    __shared__ int s_ok;
    __shared__ uint8_t s_img[kMaxSide * kMaxSide];
    __shared__ int s_hist[256];
    __shared__ int s_thr;
    __shared__ double s_q1[256], s_mu1[256];

    const int lane = threadIdx.x;
    const int f = blockIdx.y;
    const int nq = min(qcount[f], kQuadCap);
    const int H = cfg.H, W = cfg.W;
    const int nb = cfg.markerSize + 2 * cfg.borderBits;
    const int cell = cfg.cellSize;
    const int S = nb * cell;
    const uint8_t* __restrict__ img = gray + (size_t)f * H * W;

    for (int qi = blockIdx.x; qi < nq; qi += gridDim.x) {
        __syncthreads();
        QuadRec* qr = &quads[(size_t)f * kQuadCap + qi];
        if (lane == 0) {
            // getPerspectiveTransform(corners -> (0,0),(S-1,0),(S-1,S-1),(0,S-1)) and its inverse
            const double dstx[4] = {0, (double)(float)(S - 1), (double)(float)(S - 1), 0};
            const double dsty[4] = {0, 0, (double)(float)(S - 1), (double)(float)(S - 1)};
            for (int i = 0; i < 4; i++) {
                const double sx = qr->c[2 * i], sy = qr->c[2 * i + 1], dx = dstx[i], dy = dsty[i];
                s_a[i][0] = sx; s_a[i][1] = sy; s_a[i][2] = 1; s_a[i][3] = 0; s_a[i][4] = 0; s_a[i][5] = 0;
                s_a[i][6] = -sx * dx; s_a[i][7] = -sy * dx; s_a[i][8] = dx;
                s_a[i + 4][0] = 0; s_a[i + 4][1] = 0; s_a[i + 4][2] = 0; s_a[i + 4][3] = sx; s_a[i + 4][4] = sy; s_a[i + 4][5] = 1;
                s_a[i + 4][6] = -sx * dy; s_a[i + 4][7] = -sy * dy; s_a[i + 4][8] = dy;
            }
            bool ok = true;
            for (int i = 0; i < 8 && ok; i++) {
                int k = i;
                for (int j = i + 1; j < 8; j++)
                    if (fabs(s_a[j][i]) > fabs(s_a[k][i])) k = j;
                if (fabs(s_a[k][i]) < 2.220446049250313e-16) { ok = false; break; }
                if (k != i)
                    for (int j = i; j < 9; j++) { double t = s_a[i][j]; s_a[i][j] = s_a[k][j]; s_a[k][j] = t; }
                const double d = -1 / s_a[i][i];
                for (int j = i + 1; j < 8; j++) {
                    const double alpha = s_a[j][i] * d;
                    for (int c = i + 1; c < 9; c++) s_a[j][c] += alpha * s_a[i][c];
                }
            }
            double M[9];
            if (ok) {
                double x[8];
                for (int i = 7; i >= 0; i--) {
                    double s = s_a[i][8];
                    for (int k = i + 1; k < 8; k++) s -= s_a[i][k] * x[k];
                    x[i] = s / s_a[i][i];
                }
                for (int i = 0; i < 8; i++) M[i] = x[i];
                M[8] = 1.0;
                const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                                   M[2] * (M[3] * M[7] - M[4] * M[6]);
                if (det == 0) ok = false;
                else {
                    const double id = 1. / det;
                    s_mi[0] = (M[4] * M[8] - M[5] * M[7]) * id;
                    s_mi[1] = (M[2] * M[7] - M[1] * M[8]) * id;
                    s_mi[2] = (M[1] * M[5] - M[2] * M[4]) * id;
                    s_mi[3] = (M[5] * M[6] - M[3] * M[8]) * id;
                    s_mi[4] = (M[0] * M[8] - M[2] * M[6]) * id;
                    s_mi[5] = (M[2] * M[3] - M[0] * M[5]) * id;
                    s_mi[6] = (M[3] * M[7] - M[4] * M[6]) * id;
                    s_mi[7] = (M[1] * M[6] - M[0] * M[7]) * id;
                    s_mi[8] = (M[0] * M[4] - M[1] * M[3]) * id;
                }
            }
            s_ok = ok ? 1 : 0;
        }
        for (int i = lane; i < 256; i += 64) s_hist[i] = 0;
        __syncthreads();

        // warpPerspective(INTER_NEAREST, constant 0 border)
        const bool ok = s_ok != 0;
        const int m2 = cell / 2;
        int sum = 0;
        long long sq = 0;
        for (int p = lane; p < S * S; p += 64) {
            const int y = p / S, x = p - y * S;
            int v = 0;
            if (ok) {
                const double X0 = s_mi[0] * 0 + s_mi[1] * y + s_mi[2];
                const double Y0 = s_mi[3] * 0 + s_mi[4] * y + s_mi[5];
                const double W0 = s_mi[6] * 0 + s_mi[7] * y + s_mi[8];
                double Wd = W0 + s_mi[6] * x;
                Wd = Wd ? 1. / Wd : 0;
                const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + s_mi[0] * x) * Wd));
                const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + s_mi[3] * x) * Wd));
                int X = (int)rint(fX), Y = (int)rint(fY);
                X = min(max(X, -32768), 32767);
                Y = min(max(Y, -32768), 32767);
                if (X >= 0 && X < W && Y >= 0 && Y < H) v = img[(size_t)Y * W + X];
            }
            s_img[p] = (uint8_t)v;
            atomicAdd(&s_hist[v], 1);
            if (y >= m2 && y < S - m2 && x >= m2 && x < S - m2) {
                sum += v;
                sq += v * v;
            }
        }
        sum = wave_sum_i(sum);
        sq = wave_sum_ll(sq);
        __syncthreads();

        const int nin = (S - 2 * m2) * (S - 2 * m2);
        const double mean = (double)sum / nin;
        const double var = fmax((double)sq / nin - mean * mean, 0.);
        const double stddev = sqrt(var);
        uint64_t bitsMask;  // bit (y*nb+x) of the nb x nb grid
        if (stddev < cfg.minOtsuStdDev) {
            bitsMask = (mean > 127) ? ~0ull : 0ull;
        } else {
            // getThreshVal_Otsu_8u.  Its loop carries (q1, mu1) from bin to bin through a division, so that part
            // stays sequential on one lane, in the reference's operation order; everything that only feeds the
            // arg-max (q2, mu2, sigma) is evaluated afterwards by all lanes.  Bins before the first and after
            // the last occupied one cannot change the result (q1 is 0 resp. ~1 there: the loop `continue`s) and
            // are skipped; the histogram moments are sums of integers, exact in any order.
            const int N = S * S;
            const double scale = 1. / N;
            int h4[4];
            long long part = 0;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                h4[k] = s_hist[4 * lane + k];
                part += (long long)(4 * lane + k) * h4[k];
                any = any || h4[k] != 0;
            }
            const unsigned long long occ = __ballot(any);
            const int firstBin = 4 * (__ffsll((long long)occ) - 1), lastBin = 4 * (63 - __clzll(occ)) + 3;
            const double mu = (double)wave_sum_ll(part) * scale;
            if (lane == 0) {
                double mu1 = 0, q1 = 0;
                for (int i = firstBin; i <= lastBin; i++) {
                    const double p_i = s_hist[i] * scale;
                    mu1 *= q1;
                    q1 += p_i;
                    const double q2 = 1. - q1;
                    const bool skip = fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1. - 1.1920928955078125e-07;
                    if (!skip) mu1 = (mu1 + i * p_i) / q1;
                    s_q1[i] = skip ? -1. : q1;
                    s_mu1[i] = mu1;
                }
            }
            __syncthreads();
            double best = 0;
            int bestBin = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = 4 * lane + k;
                if (i >= firstBin && i <= lastBin) {
                    const double q1 = s_q1[i];
                    if (q1 >= 0) {
                        const double mu1 = s_mu1[i], q2 = 1. - q1;
                        const double mu2 = (mu - q1 * mu1) / q2;
                        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
                        if (sigma > best) { best = sigma; bestBin = i; }
                    }
                }
            }
            // first bin reaching the overall maximum (the reference keeps the earlier bin on ties: strict >)
            double wmax = best;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, d, 64));
            const unsigned long long who = __ballot(best == wmax && best > 0);
            if (lane == 0) s_thr = 0;
            __syncthreads();
            if (who && lane == __ffsll((long long)who) - 1) s_thr = bestBin;
            __syncthreads();
            const int thr = s_thr;
            const int side = cell - 2 * cfg.cellMargin;
            bool bit = false;
            if (lane < nb * nb) {
                const int cy = lane / nb, cx = lane - cy * nb;
                const int xs = cx * cell + cfg.cellMargin, ys = cy * cell + cfg.cellMargin;
                int nz = 0;
                for (int yy = 0; yy < side; yy++)
                    for (int xx = 0; xx < side; xx++) nz += s_img[(ys + yy) * S + xs + xx] > thr;
                bit = nz > side * side / 2;
            }
            bitsMask = __ballot(bit);
        }
        // border errors and inner code (nb*nb <= 64)
        int borderErrors = 0;
        uint32_t code = 0;
        const int bb = cfg.borderBits, ms = cfg.markerSize;
        for (int y = 0; y < nb; y++)
            for (int x = 0; x < nb; x++) {
                const uint32_t b = (uint32_t)((bitsMask >> (y * nb + x)) & 1ull);
                const bool inner = (y >= bb && y < nb - bb && x >= bb && x < nb - bb);
                if (inner) code = (code << 1) | b;
                else borderErrors += b;
            }
        int id = -1, rot = 0;
        if (borderErrors <= cfg.maxBorderErrors) {
            // Dictionary::identify: first entry (lowest id) whose best rotation is within maxCorrection
            for (int m0 = 0; m0 < cfg.dictN && id < 0; m0 += 64) {
                const int m = m0 + lane;
                int best = ms * ms + 1, bestRot = -1;
                if (m < cfg.dictN) {
                    const uint32_t dc = dict[m];
                    for (int r = 0; r < 4; r++) {
                        const int hd = __popc(rotate_code(dc, ms, r) ^ code);
                        if (hd < best) { best = hd; bestRot = r; }
                    }
                }
                const unsigned long long hit = __ballot(m < cfg.dictN && best <= cfg.maxCorrection);
                if (hit) {
                    const int src = __ffsll((long long)hit) - 1;
                    id = m0 + src;
                    rot = __shfl(bestRot, src, 64);
                }
            }
        }
        if (lane == 0) {
            qr->valid = (id >= 0) ? 1 : 0;
            qr->id = id;
            qr->rot = rot;
        }
    }
}

}  // namespace

void launch_identify(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    hipLaunchKernelGGL(k_identify, dim3(64, B), dim3(64), 0, st, b.gray, b.quads, b.qcount, b.dict, c);
}

}  // namespace camsync
