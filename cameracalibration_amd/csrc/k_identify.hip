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
#include "wave_reduce.h"

namespace camsync {

namespace {

// (markerSize + 2*border) * cell <= 32 is checked on the host (check_params): a patch has at most 1024 samples

// value of lane `src` (wave-uniform, mostly a compile-time constant here) without a trip through LDS
__device__ __forceinline__ int lane_get(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double lane_get(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int wave_sum_i(int v) {
    return wave_allreduce(v, [](int a, int b) { return a + b; });
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
    return wave_allreduce(v, [](long long a, long long b) { return a + b; });
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

// Quads are taken eight at a time: getThreshVal_Otsu's loop carries (q1, mu1) from bin to bin through a division,
// so it is sequential -- a few thousand dependent instructions that used to run on one lane of the wave while 63
// waited, more than everything else in this kernel together.  With a batch, eight lanes run eight of those loops
// side by side; sampling, histogram, votes and dictionary search stay 64 lanes per quad.
constexpr int kIdBatch = 8;

__global__ __launch_bounds__(64) void k_identify(const uint8_t* __restrict__ gray, int grayC, QuadRec* __restrict__ quads,
                                                 const int* __restrict__ qcount, const uint16_t* __restrict__ dict,
                                                 DetCfg cfg) {
    extern __shared__ __align__(16) uint8_t s_imgs[];  // kIdBatch patches of imgStride bytes (LDS decides the occupancy)
    __shared__ int s_histw[256];                       // the histogram being built (atomics)
    __shared__ uint16_t s_hist[kIdBatch][256];         // the batch's histograms (counts <= 1024)
    __shared__ double s_mu[kIdBatch];
    __shared__ int s_first[kIdBatch], s_last[kIdBatch], s_mode[kIdBatch], s_thr[kIdBatch];

    const int lane = threadIdx.x;
    const int f = blockIdx.y;
    const int nq = min(qcount[f], kQuadCap);
    const int H = cfg.H, W = cfg.W;
    const int nb = cfg.markerSize + 2 * cfg.borderBits;
    const int cell = cfg.cellSize;
    const int S = nb * cell;
    const GraySrc img{gray + (size_t)f * H * W * grayC, grayC, (uint32_t)(H * W - 1)};
    const int imgStride = (S * S + 15) & ~15;
    const int sy0 = lane / S, sx0 = lane - sy0 * S, sqy = 64 / S, sqx = 64 - sqy * S;

    for (int base = blockIdx.x * kIdBatch; base < nq; base += gridDim.x * kIdBatch) {
        const int nbq = min(kIdBatch, nq - base);
        // ---- phase 1, quad by quad: perspective map, samples, histogram, mean / standard deviation ----
        for (int b = 0; b < nbq; b++) {
            __syncthreads();
            const QuadRec* qr = &quads[(size_t)f * kQuadCap + base + b];
            // getPerspectiveTransform(corners -> (0,0),(S-1,0),(S-1,S-1),(0,S-1)) and its inverse.
            // The 8x9 system is eliminated with lane c holding column c in registers: every element update of the
            // reference's row operations is independent, so the arithmetic (and hence the rounding of the sample
            // positions) is the same as on one thread, without the serial trips through LDS.
            double mi[9];
            bool ok = true;
            {
                const double S1 = (double)(float)(S - 1);
                double col[8];
    #pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double sx = qr->c[2 * r], sy = qr->c[2 * r + 1];
                    const double dx = (r == 1 || r == 2) ? S1 : 0., dy = (r >= 2) ? S1 : 0.;
                    double top, bot;  // rows r and r + 4, column `lane`
                    switch (lane) {
                        case 0: top = sx; bot = 0; break;
                        case 1: top = sy; bot = 0; break;
                        case 2: top = 1; bot = 0; break;
                        case 3: top = 0; bot = sx; break;
                        case 4: top = 0; bot = sy; break;
                        case 5: top = 0; bot = 1; break;
                        case 6: top = -sx * dx; bot = -sx * dy; break;
                        case 7: top = -sy * dx; bot = -sy * dy; break;
                        case 8: top = dx; bot = dy; break;
                        default: top = 0; bot = 0; break;
                    }
                    col[r] = top;
                    col[r + 4] = bot;
                }
    #pragma unroll
                for (int i = 0; i < 8; i++) {
                    // partial pivoting on column i (held by lane i): first row with the largest magnitude
                    int k = i;
                    double big = fabs(col[i]);
    #pragma unroll
                    for (int j = i + 1; j < 8; j++)
                        if (fabs(col[j]) > big) { big = fabs(col[j]); k = j; }
                    k = lane_get(k, i);
                    big = lane_get(big, i);
                    if (big < 2.220446049250313e-16) { ok = false; break; }
                    // swap rows i and k
                    double rk = col[i];
    #pragma unroll
                    for (int j = i + 1; j < 8; j++) rk = (j == k) ? col[j] : rk;
    #pragma unroll
                    for (int j = i + 1; j < 8; j++) col[j] = (j == k) ? col[i] : col[j];
                    col[i] = rk;
                    const double d = -1 / lane_get(col[i], i);
    #pragma unroll
                    for (int j = i + 1; j < 8; j++) {
                        const double alpha = lane_get(col[j], i) * d;
                        if (lane > i) col[j] += alpha * col[i];
                    }
                }
                if (ok) {
                    double x[8];
    #pragma unroll
                    for (int i = 7; i >= 0; i--) {
                        double sacc = lane_get(col[i], 8);
    #pragma unroll
                        for (int k = i + 1; k < 8; k++) sacc -= lane_get(col[i], k) * x[k];
                        x[i] = sacc / lane_get(col[i], i);
                    }
                    const double M8 = 1.0;
                    const double det = x[0] * (x[4] * M8 - x[5] * x[7]) - x[1] * (x[3] * M8 - x[5] * x[6]) +
                                       x[2] * (x[3] * x[7] - x[4] * x[6]);
                    if (det == 0) ok = false;
                    else {
                        const double id = 1. / det;
                        mi[0] = (x[4] * M8 - x[5] * x[7]) * id;
                        mi[1] = (x[2] * x[7] - x[1] * M8) * id;
                        mi[2] = (x[1] * x[5] - x[2] * x[4]) * id;
                        mi[3] = (x[5] * x[6] - x[3] * M8) * id;
                        mi[4] = (x[0] * M8 - x[2] * x[6]) * id;
                        mi[5] = (x[2] * x[3] - x[0] * x[5]) * id;
                        mi[6] = (x[3] * x[7] - x[4] * x[6]) * id;
                        mi[7] = (x[1] * x[6] - x[0] * x[7]) * id;
                        mi[8] = (x[0] * x[4] - x[1] * x[3]) * id;
                    }
                }
                if (!ok) {
    #pragma unroll
                    for (int i = 0; i < 9; i++) mi[i] = 0;
                }
            }
            for (int i = lane; i < 256; i += 64) s_histw[i] = 0;
            uint8_t* const s_img = s_imgs + b * imgStride;
            __syncthreads();

            // warpPerspective(INTER_NEAREST, constant 0 border)
            const int m2 = cell / 2;
            int sum = 0;
            long long sq = 0;
            // sample p = lane + 64 k is (y, x); stepping by 64 samples without dividing
            for (int p = lane, y = sy0, x = sx0; p < S * S; p += 64, y += sqy, x += sqx) {
                if (x >= S) { x -= S; y++; }
                int v = 0;
                if (ok) {
                    const double X0 = mi[0] * 0 + mi[1] * y + mi[2];
                    const double Y0 = mi[3] * 0 + mi[4] * y + mi[5];
                    const double W0 = mi[6] * 0 + mi[7] * y + mi[8];
                    double Wd = W0 + mi[6] * x;
                    Wd = Wd ? 1. / Wd : 0;
                    const double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + mi[0] * x) * Wd));
                    const double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + mi[3] * x) * Wd));
                    int X = (int)rint(fX), Y = (int)rint(fY);
                    X = min(max(X, -32768), 32767);
                    Y = min(max(Y, -32768), 32767);
                    if (X >= 0 && X < W && Y >= 0 && Y < H) v = img.at((size_t)Y * W + X);
                }
                s_img[p] = (uint8_t)v;
                atomicAdd(&s_histw[v], 1);
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
            // mode 0: Otsu threshold needed; 1 / 2: flat patch, all cells white / black
            int mode = 0, firstBin = 0, lastBin = 0;
            double mu = 0;
            if (stddev < cfg.minOtsuStdDev) {
                mode = (mean > 127) ? 1 : 2;
            } else {
                // Bins before the first and after the last occupied one cannot change the result (q1 is 0 resp. ~1
                // there: the loop `continue`s) and are skipped; the first moment is a sum of integers, exact in
                // any order.
                long long part = 0;
                bool any = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int h = s_histw[4 * lane + k];
                    s_hist[b][4 * lane + k] = (uint16_t)h;
                    part += (long long)(4 * lane + k) * h;
                    any = any || h != 0;
                }
                const unsigned long long occ = __ballot(any);
                firstBin = 4 * (__ffsll((long long)occ) - 1);
                lastBin = 4 * (63 - __clzll(occ)) + 3;
                mu = (double)wave_sum_ll(part) * (1. / (S * S));
            }
            if (lane == 0) {
                s_mode[b] = mode;
                s_first[b] = firstBin;
                s_last[b] = lastBin;
                s_mu[b] = mu;
            }
        }
        __syncthreads();
        // ---- phase 2, one lane per quad: getThreshVal_Otsu_8u in the reference's operation order ----
        if (lane < nbq && s_mode[lane] == 0) {
            const double scale = 1. / (S * S), mu = s_mu[lane];
            const uint16_t* __restrict__ h = s_hist[lane];
            double mu1 = 0, q1 = 0, max_sigma = 0;
            int thr = 0;
            for (int i = s_first[lane]; i <= s_last[lane]; i++) {
                const double p_i = h[i] * scale;
                mu1 *= q1;
                q1 += p_i;
                const double q2 = 1. - q1;
                if (fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1. - 1.1920928955078125e-07) continue;
                mu1 = (mu1 + i * p_i) / q1;
                const double mu2 = (mu - q1 * mu1) / q2;
                const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
                if (sigma > max_sigma) {  // strict: the earlier bin wins ties
                    max_sigma = sigma;
                    thr = i;
                }
            }
            s_thr[lane] = thr;
        }
        __syncthreads();
        // ---- phase 3, quad by quad: cell votes, border test, dictionary ----
        for (int b = 0; b < nbq; b++) {
            QuadRec* qr = &quads[(size_t)f * kQuadCap + base + b];
            const int mode = s_mode[b];
            uint64_t bitsMask;  // bit (y*nb+x) of the nb x nb grid
            if (mode) {
                bitsMask = (mode == 1) ? ~0ull : 0ull;
            } else {
                const int thr = s_thr[b];
                const uint8_t* const s_img = s_imgs + b * imgStride;
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
                    const uint32_t bv = (uint32_t)((bitsMask >> (y * nb + x)) & 1ull);
                    const bool inner = (y >= bb && y < nb - bb && x >= bb && x < nb - bb);
                    if (inner) code = (code << 1) | bv;
                    else borderErrors += bv;
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
                        rot = __builtin_amdgcn_readlane(bestRot, src);
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
}

}  // namespace

void launch_identify(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    // eight waves per frame, eight quads per wave and round (a frame of the benchmark scene has about 60)
    const int S = (c.markerSize + 2 * c.borderBits) * c.cellSize;
    const size_t lds = (size_t)kIdBatch * ((S * S + 15) & ~15);
    hipLaunchKernelGGL(k_identify, dim3(8, B), dim3(64), lds, st, b.graySrc, b.graySrcC, b.quads, b.qcount, b.dict, c);
}

}  // namespace camsync
