// k_identify.hip — K6: _identifyOneCandidate of ArucoDetector.detectMarkers
// step 2 (reference camera_sync/aruco.py:256-264): perspective-normalise the
// quad to (markerSize+2)*cell pixels with INTER_NEAREST, mean/stddev gate,
// Otsu threshold, per-cell majority vote, border test, dictionary lookup over
// the four rotations.
//
// k_quad_persp solves the 8x8 systems for the perspective maps, one lane per
// quad (same elimination order as the CPU restatement, so the nearest-neighbour
// sample positions round identically); k_identify takes the quads in batches:
// the 64 lanes of a wave gather a quad's samples, build the histogram in LDS and
// vote the cells, Otsu's sequential loop runs one lane per quad, the dictionary
// (all four rotations, laid out by the host) is searched one entry per lane.
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

// a wavefront's LDS operations execute in order: between its own phases only the compiler must keep the order
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int wave_sum_i(int v) {
    return wave_allreduce(v, [](int a, int b) { return a + b; });
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
    return wave_allreduce(v, [](long long a, long long b) { return a + b; });
}

// Perspective maps, one lane per quad: getPerspectiveTransform(corners -> (0,0),(S-1,0),(S-1,S-1),(0,S-1)) and its
// inverse.  The 8x9 system is eliminated in registers in the reference's order (partial pivoting, first row with the
// largest magnitude), so the nearest-neighbour sample positions round as on the CPU.  It used to run inside
// k_identify on nine lanes of a wave per quad; here a wave does 64 quads, and k_identify keeps its registers.
// persp[q] = the nine entries of the inverse map, then 1.0 / 0.0 = usable / degenerate.
__global__ __launch_bounds__(64) void k_quad_persp(const QuadRec* __restrict__ quads, const int* __restrict__ qcount, DetCfg cfg,
                                                   double* __restrict__ persp) {
    // (frame = the fast grid index: consecutive workgroups go to different XCDs, and a frame's first workgroup has most
    // of its quads -- with the batch index fast, the busy workgroups all landed on half of the XCDs)
    const int f = blockIdx.x;
    const int nq = min(qcount[f], cfg.quadCap);
    const int S = (cfg.markerSize + 2 * cfg.borderBits) * cfg.cellSize;
    const double S1 = (double)(float)(S - 1);
    for (int q = blockIdx.y * 64 + threadIdx.x; q < nq; q += gridDim.y * 64) {
        const QuadRec* qr = &quads[(size_t)f * cfg.quadCap + q];
        double A[8][9];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double sx = qr->c[2 * r], sy = qr->c[2 * r + 1];
            const double dx = (r == 1 || r == 2) ? S1 : 0., dy = (r >= 2) ? S1 : 0.;
            A[r][0] = sx; A[r][1] = sy; A[r][2] = 1; A[r][3] = 0; A[r][4] = 0; A[r][5] = 0; A[r][6] = -sx * dx; A[r][7] = -sy * dx; A[r][8] = dx;
            A[r + 4][0] = 0; A[r + 4][1] = 0; A[r + 4][2] = 0; A[r + 4][3] = sx; A[r + 4][4] = sy; A[r + 4][5] = 1;
            A[r + 4][6] = -sx * dy; A[r + 4][7] = -sy * dy; A[r + 4][8] = dy;
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int k = i;
            double big = fabs(A[i][i]);
#pragma unroll
            for (int j = i + 1; j < 8; j++)
                if (fabs(A[j][i]) > big) { big = fabs(A[j][i]); k = j; }
            if (big < 2.220446049250313e-16) ok = false;  // (the lane goes on with numbers nobody reads)
            // swap rows i and k (columns before i are not read again)
#pragma unroll
            for (int c = i; c < 9; c++) {
                double rk = A[i][c];
#pragma unroll
                for (int j = i + 1; j < 8; j++) rk = (j == k) ? A[j][c] : rk;
#pragma unroll
                for (int j = i + 1; j < 8; j++) A[j][c] = (j == k) ? A[i][c] : A[j][c];
                A[i][c] = rk;
            }
            const double d = -1 / A[i][i];
#pragma unroll
            for (int j = i + 1; j < 8; j++) {
                const double alpha = A[j][i] * d;
#pragma unroll
                for (int c = i + 1; c < 9; c++) A[j][c] += alpha * A[i][c];
            }
        }
        double x[8];
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            double sacc = A[i][8];
#pragma unroll
            for (int k = i + 1; k < 8; k++) sacc -= A[i][k] * x[k];
            x[i] = sacc / A[i][i];
        }
        const double M8 = 1.0;
        const double det = x[0] * (x[4] * M8 - x[5] * x[7]) - x[1] * (x[3] * M8 - x[5] * x[6]) + x[2] * (x[3] * x[7] - x[4] * x[6]);
        if (!(det != 0)) ok = false;
        double mi[9];
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
        double* out = persp + ((size_t)f * cfg.quadCap + q) * 10;
#pragma unroll
        for (int i = 0; i < 9; i++) out[i] = ok ? mi[i] : 0.0;
        out[9] = ok ? 1.0 : 0.0;
    }
}

// Quads are taken 32 at a time by a workgroup of four waves.  Sampling, histogram, cell votes and dictionary search
// are 64 lanes per quad (the batch's quads interleaved over the waves, one after the other in each).  getThreshVal_Otsu's loop carries (q1, mu1) from
// bin to bin through a division, so it is sequential: a few thousand dependent instructions per quad, which one wave
// runs for the 32 quads side by side, one lane each, while the other three wait.
constexpr int kIdBatch = 32, kIdWaves = 4;

// ROUNDS x 64 >= samples of a patch (9: the reference's 24 x 24 patch; 16: the largest patch check_params lets through)
template <int kIdRounds>
__global__ __launch_bounds__(64 * kIdWaves) void k_identify(const uint8_t* __restrict__ gray, int grayC, QuadRec* __restrict__ quads,
                                                            const int* __restrict__ qcount, const uint16_t* __restrict__ dictRot,
                                                            const double* __restrict__ persp, DetCfg cfg,
                                                            const uint32_t* __restrict__ idList, const int* __restrict__ idCount, int pass) {
    extern __shared__ __align__(16) uint8_t s_imgs[];  // kIdBatch patches of imgStride bytes
    __shared__ int s_histw[kIdWaves][256];             // the histograms being built (atomics), one per wave
    __shared__ uint16_t s_hist[kIdBatch][256];         // the batch's histograms (counts <= 1024)
    __shared__ double s_mu[kIdBatch];
    __shared__ int s_first[kIdBatch], s_last[kIdBatch], s_mode[kIdBatch], s_thr[kIdBatch], s_quad[kIdBatch];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x;  // (fast index: see k_quad_persp)
    // pass 0: every quad of the frame; 1: the candidates the grouping selected; 2: the fall-backs of the selected ones
    // that pass 1 found invalid (k_group lists both kinds)
    const int nq = pass == 0 ? min(qcount[f], cfg.quadCap) : min(idCount[2 * f + pass - 1], cfg.quadCap);
    const uint32_t* const list = idList + (size_t)f * cfg.quadCap;
    const int H = cfg.H, W = cfg.W;
    const int nb = cfg.markerSize + 2 * cfg.borderBits;
    const int cell = cfg.cellSize;
    const int S = nb * cell;
    const GraySrc img{gray + (size_t)f * H * W * grayC, grayC, (uint32_t)(H * W - 1)};
    const int imgStride = (S * S + 15) & ~15;
    const int sy0 = lane / S, sx0 = lane - sy0 * S, sqy = 64 / S, sqx = 64 - sqy * S;
    int* const histw = s_histw[wave];
    // cells of the nb x nb grid that belong to the border (bit y * nb + x)
    const int bb = cfg.borderBits, ms = cfg.markerSize;
    uint64_t borderMask = 0;
    for (int y = 0; y < nb; y++)
        for (int x = 0; x < nb; x++)
            if (!(y >= bb && y < nb - bb && x >= bb && x < nb - bb)) borderMask |= 1ull << (y * nb + x);

    for (int base = blockIdx.y * kIdBatch; base < nq; base += gridDim.y * kIdBatch) {
        const int nbq = min(kIdBatch, nq - base);
        __syncthreads();  // (the previous round's phase 3 has read what phase 1 overwrites)
        // ---- phase 1, quad by quad in every wave: samples, histogram, mean / standard deviation ----
        for (int b = wave; b < nbq; b += kIdWaves) {  // (interleaved: a short list still occupies all the waves)
            const uint32_t entry = pass == 0 ? (uint32_t)(base + b) : pass == 1 ? list[base + b] : list[cfg.quadCap - 1 - (base + b)];
            const int q = (int)(entry & 0xFFFFu);
            if (lane == 0) s_quad[b] = q;
            if (pass == 2 && quads[(size_t)f * cfg.quadCap + (entry >> 16)].valid) {
                // the representative was identified: its fall-backs are never asked for
                if (lane == 0) s_mode[b] = 3;
                continue;
            }
            const double* mp = persp + ((size_t)f * cfg.quadCap + q) * 10;
            double mi[9];
#pragma unroll
            for (int i = 0; i < 9; i++) mi[i] = mp[i];
            const bool ok = mp[9] != 0.0;
            for (int i = lane; i < 256; i += 64) histw[i] = 0;
            uint8_t* const s_img = s_imgs + b * imgStride;
            wave_lds_fence();

            // warpPerspective(INTER_NEAREST, constant 0 border)
            const int m2 = cell / 2;
            int sum = 0;
            long long sq = 0;
            // sample p = lane + 64 k is (y, x); stepping by 64 samples without dividing.  All the samples of a quad are
            // fetched before any is used: with one fetch per trip round the loop the wave waited for memory once per
            // 64 samples, and those waits were most of this kernel's time.
            int vals[kIdRounds];
            {
                int y = sy0, x = sx0;
#pragma unroll
                for (int k = 0; k < kIdRounds; k++) {
                    if (x >= S) { x -= S; y++; }
                    int v = 0;
                    if (ok && lane + 64 * k < S * S) {
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
                    vals[k] = v;
                    y += sqy;
                    x += sqx;
                }
            }
            {
                int y = sy0, x = sx0;
#pragma unroll
                for (int k = 0; k < kIdRounds; k++) {
                    if (x >= S) { x -= S; y++; }
                    const int p = lane + 64 * k;
                    if (p < S * S) {
                        const int v = vals[k];
                        s_img[p] = (uint8_t)v;
                        atomicAdd(&histw[v], 1);
                        if (y >= m2 && y < S - m2 && x >= m2 && x < S - m2) {
                            sum += v;
                            sq += v * v;
                        }
                    }
                    y += sqy;
                    x += sqx;
                }
            }
            sum = wave_sum_i(sum);
            sq = wave_sum_ll(sq);
            wave_lds_fence();

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
                    const int h = histw[4 * lane + k];
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
            wave_lds_fence();
        }
        __syncthreads();
        // ---- phase 2, one lane per quad: getThreshVal_Otsu_8u in the reference's operation order ----
        if (wave == 0 && lane < nbq && s_mode[lane] == 0) {
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
        // ---- phase 3, quad by quad in every wave: cell votes, border test, dictionary ----
        for (int b = wave; b < nbq; b += kIdWaves) {  // (interleaved: a short list still occupies all the waves)
            const int mode = s_mode[b];
            if (mode == 3) continue;
            QuadRec* qr = &quads[(size_t)f * cfg.quadCap + s_quad[b]];
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
            if (nb * nb < 64) bitsMask &= (1ull << (nb * nb)) - 1ull;
            // border errors and inner code, rows top to bottom, the left cell of a row in the higher bit
            const int borderErrors = __popcll(bitsMask & borderMask);
            uint32_t code = 0;
            for (int y = bb; y < nb - bb; y++) {
                const uint32_t row = (uint32_t)(bitsMask >> (y * nb + bb)) & ((1u << ms) - 1u);
                code = (code << ms) | (__brev(row) >> (32 - ms));
            }
            int id = -1, rot = 0;
            if (borderErrors <= cfg.maxBorderErrors) {
                // Dictionary::identify: first entry (lowest id) whose best rotation is within maxCorrection
                for (int m0 = 0; m0 < cfg.dictN && id < 0; m0 += 64) {
                    const int m = m0 + lane;
                    int best = ms * ms + 1, bestRot = -1;
                    if (m < cfg.dictN) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int hd = __popc((uint32_t)dictRot[r * cfg.dictN + m] ^ code);
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

// Code of a markerSize x markerSize bit grid (a row-major MSB-first integer) seen under rotation r, as
// Dictionary::getByteListFromBits lays the four rotations out.
uint32_t rotate_marker_code(uint32_t code, int n, int r) {
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

void launch_identify(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st) {
    const int S = (c.markerSize + 2 * c.borderBits) * c.cellSize;
    const size_t lds = (size_t)kIdBatch * ((S * S + 15) & ~15);
    hipLaunchKernelGGL(k_quad_persp, dim3(B, 2), dim3(64), 0, st, b.quads, b.qcount, c, b.persp);
    // As the reference detector does, only the candidates the grouping of near-duplicates selects are identified, and
    // the other members of a group only when its representative is not a marker: a marker is found at every threshold
    // scale and as an inner and an outer border, so a frame has several times as many candidates as groups, and the
    // bit sampling is bound by the scattered reads of the frame (one 64-byte sector per sample).
    auto run = [&](int pass, int gridY) {
        if (S * S <= 9 * 64)
            hipLaunchKernelGGL(k_identify<9>, dim3(B, gridY), dim3(64 * kIdWaves), lds, st, b.graySrc, b.graySrcC, b.quads, b.qcount, b.dict,
                               b.persp, c, b.idList, b.idCount, pass);
        else
            hipLaunchKernelGGL(k_identify<16>, dim3(B, gridY), dim3(64 * kIdWaves), lds, st, b.graySrc, b.graySrcC, b.quads, b.qcount, b.dict,
                               b.persp, c, b.idList, b.idCount, pass);
    };
    if (b.identifyAll) {
        run(0, 4);  // (four workgroups per frame, 32 quads per workgroup and round)
        return;
    }
    launch_group(c, b, B, st);
    run(1, 2);  // (two workgroups per frame: the lists are short; smaller batches or more waves were slower)
    run(2, 2);
}

}  // namespace camsync
