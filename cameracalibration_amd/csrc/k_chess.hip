// k_chess.hip — chessboard inner-corner candidates: the front end of cv.findChessboardCorners as Camera.calibrate
// uses it (reference camera_sync/camera.py:48-52).
//
// OpenCV's detector (quad extraction and linking) cannot be restated without its source; what the reference needs
// from it is the set of inner corners, which cornerSubPix((11,11)) then pulls onto the saddle points whatever the
// starting pixel (camera.py:58).  The candidates here come from the ChESS response (Bennett & Lasenby, "ChESS -
// quick and robust detection of chess-board features", 2014) in integer arithmetic on 3x3 box sums of the gray image:
//   ring of 16 samples I_n at radius 5 s,
//   SR = sum_{n<4} |(I_n + I_{n+8}) - (I_{n+4} + I_{n+12})|     large at an X-corner
//   DR = sum_{n<8} |I_n - I_{n+8}|                              large on an edge
//   MR = |mean(ring) - mean(centre cross)| * 16                 large on a blob
//   R  = SR - DR - MR
// followed by non-maximum suppression in a (10 s + 1)^2 window and a threshold of max(R) / 20.  The lattice is
// assembled on the host (camera_sync/chessboard.py).  All integers, so the tests' CPU twin gives the same candidates
// bit for bit; the end-to-end pin is the reference's own calibration result (tests/test_chessboard.py).
#include "internal.h"

#include <algorithm>

namespace camsync {

namespace {

__constant__ int8_t kRingDx[16] = {5, 5, 4, 2, 0, -2, -4, -5, -5, -5, -4, -2, 0, 2, 4, 5};
__constant__ int8_t kRingDy[16] = {0, 2, 4, 5, 5, 5, 4, 2, 0, -2, -4, -5, -5, -5, -4, -2};

// gray u8 [B,H,W] from frames [B,H,W,C] (cv.cvtColor BGR2GRAY integer luma), and its 3x3 box sums with a replicated border
__global__ __launch_bounds__(256) void k_chess_gray(const uint8_t* __restrict__ in, int C, uint8_t* __restrict__ gray, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (C == 1) {
        gray[i] = in[i];
    } else {
        const uint8_t* p = in + i * 3;
        gray[i] = (uint8_t)((p[0] * 3735u + p[1] * 19235u + p[2] * 9798u + 16384u) >> 15);
    }
}

__global__ __launch_bounds__(256) void k_chess_box3(const uint8_t* __restrict__ gray, uint16_t* __restrict__ S, int B, int H, int W) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t px = (size_t)H * W;
    if (i >= px * B) return;
    const size_t f = i / px, r = i - f * px;
    const int y = (int)(r / W), x = (int)(r - (size_t)y * W);
    const uint8_t* g = gray + f * px;
    int s = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++) {
        const int yy = min(max(y + dy, 0), H - 1);
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) s += g[(size_t)yy * W + min(max(x + dx, 0), W - 1)];
    }
    S[i] = (uint16_t)s;
}

__global__ __launch_bounds__(256) void k_chess_response(const uint16_t* __restrict__ S, int32_t* __restrict__ R, int B, int H, int W,
                                                        int s, int* __restrict__ frameMax) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t px = (size_t)H * W;
    int r = 0;
    size_t f = 0;
    if (i < px * B) {
        f = i / px;
        const size_t q = i - f * px;
        const int y = (int)(q / W), x = (int)(q - (size_t)y * W), pad = 5 * s;
        if (x >= pad && x < W - pad && y >= pad && y < H - pad) {
            const uint16_t* P = S + f * px + q;
            int I[16];
#pragma unroll
            for (int n = 0; n < 16; n++) I[n] = P[(ptrdiff_t)kRingDy[n] * s * W + kRingDx[n] * s];
            int SR = 0, DR = 0, ring = 0;
#pragma unroll
            for (int n = 0; n < 4; n++) SR += abs((I[n] + I[n + 8]) - (I[n + 4] + I[n + 12]));
#pragma unroll
            for (int n = 0; n < 8; n++) DR += abs(I[n] - I[n + 8]);
#pragma unroll
            for (int n = 0; n < 16; n++) ring += I[n];
            const int loc = P[0] + P[-(ptrdiff_t)s * W] + P[(ptrdiff_t)s * W] + P[-s] + P[s];
            const int MR = abs(ring * 5 - loc * 16);
            r = SR - DR - MR / 5;
        }
        R[i] = r;
    }
    // per-frame maximum: blocks do not straddle frames often; one atomic per wave that has something to say
    if (i < px * B && r > 0) atomicMax(&frameMax[f], r);
}

__global__ __launch_bounds__(256) void k_chess_nms(const int32_t* __restrict__ R, int B, int H, int W, int m,
                                                   const int* __restrict__ frameMax, int maxCand, int32_t* __restrict__ cand,
                                                   int* __restrict__ counts) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t px = (size_t)H * W;
    if (i >= px * B) return;
    const size_t f = i / px, q = i - f * px;
    const int v = R[i];
    if (v <= frameMax[f] / 20 || v <= 0) return;
    const int y = (int)(q / W), x = (int)(q - (size_t)y * W);
    const int32_t* Rf = R + f * px;
    for (int yy = max(y - m, 0); yy <= min(y + m, H - 1); yy++)
        for (int xx = max(x - m, 0); xx <= min(x + m, W - 1); xx++) {
            const int w = Rf[(size_t)yy * W + xx];
            // the first maximum in raster order keeps the window
            if (w > v || (w == v && (yy < y || (yy == y && xx < x)))) return;
        }
    const int slot = atomicAdd(&counts[f], 1);
    if (slot < maxCand) {
        int32_t* c = cand + ((size_t)f * maxCand + slot) * 3;
        c[0] = x; c[1] = y; c[2] = v;
    }
}

}  // namespace

void launch_chess(const uint8_t* in, int C, int B, int H, int W, int scale, uint8_t* gray, uint16_t* S, int32_t* R,
                  int* frameMax, int maxCand, int32_t* cand, int* counts, hipStream_t st) {
    const size_t total = (size_t)B * H * W;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_chess_gray, dim3(blocks), dim3(256), 0, st, in, C, gray, total);
    hipLaunchKernelGGL(k_chess_box3, dim3(blocks), dim3(256), 0, st, gray, S, B, H, W);
    hipLaunchKernelGGL(k_chess_response, dim3(blocks), dim3(256), 0, st, S, R, B, H, W, scale, frameMax);
    hipLaunchKernelGGL(k_chess_nms, dim3(blocks), dim3(256), 0, st, R, B, H, W, 5 * scale, frameMax, maxCand, cand, counts);
}

}  // namespace camsync
