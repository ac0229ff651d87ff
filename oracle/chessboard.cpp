/*
 * oracle/chessboard.cpp — CPU twin of the chessboard corner-candidate stage (cameracalibration_amd/csrc/k_chess.hip).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference calls cv.findChessboardCorners (camera_sync/camera.py:52), whose algorithm lives in OpenCV and cannot
 * be restated without its source.  This file therefore does NOT restate OpenCV: it restates this repository's own
 * candidate detector (integer ChESS response + non-maximum suppression) so that the HIP kernels can be compared bit
 * for bit on the CPU.  What ties the front end to the reference is the end-to-end pin of tests/test_chessboard.py:
 * the reference's chessboard photos pics/calib/frameN.jpg -> cornerSubPix((11,11)) -> calibrateCamera reproduce the
 * reference's stored calibration/robot_640-480_500.npz.
 */
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "oracle.h"

extern "C" {

/* gray u8 [H,W] -> candidates sorted by strength (descending), then raster order: xy i32 [n,2], strength i32 [n].
 * Returns the number of candidates found (may exceed max_cand). */
int orc_chess_candidates(const uint8_t* gray, int H, int W, int scale, int max_cand, int32_t* xy, int32_t* strength) {
    static const int dx[16] = {5, 5, 4, 2, 0, -2, -4, -5, -5, -5, -4, -2, 0, 2, 4, 5};
    static const int dy[16] = {0, 2, 4, 5, 5, 5, 4, 2, 0, -2, -4, -5, -5, -5, -4, -2};
    const int s = scale, pad = 5 * s, m = 5 * s;
    std::vector<int> S((size_t)H * W), R((size_t)H * W, 0);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int t = 0;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++)
                    t += gray[(size_t)std::min(std::max(y + j, 0), H - 1) * W + std::min(std::max(x + i, 0), W - 1)];
            S[(size_t)y * W + x] = t;
        }
    int mx = 0;
    for (int y = pad; y < H - pad; y++)
        for (int x = pad; x < W - pad; x++) {
            int I[16], SR = 0, DR = 0, ring = 0;
            for (int n = 0; n < 16; n++) I[n] = S[(size_t)(y + dy[n] * s) * W + x + dx[n] * s];
            for (int n = 0; n < 4; n++) SR += abs((I[n] + I[n + 8]) - (I[n + 4] + I[n + 12]));
            for (int n = 0; n < 8; n++) DR += abs(I[n] - I[n + 8]);
            for (int n = 0; n < 16; n++) ring += I[n];
            const int loc = S[(size_t)y * W + x] + S[(size_t)(y - s) * W + x] + S[(size_t)(y + s) * W + x] + S[(size_t)y * W + x - s] +
                            S[(size_t)y * W + x + s];
            const int r = SR - DR - abs(ring * 5 - loc * 16) / 5;
            R[(size_t)y * W + x] = r;
            mx = std::max(mx, r);
        }
    struct Cand { int v, x, y; };
    std::vector<Cand> out;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int v = R[(size_t)y * W + x];
            if (v <= mx / 20 || v <= 0) continue;
            bool keep = true;
            for (int yy = std::max(y - m, 0); yy <= std::min(y + m, H - 1) && keep; yy++)
                for (int xx = std::max(x - m, 0); xx <= std::min(x + m, W - 1); xx++) {
                    const int w = R[(size_t)yy * W + xx];
                    if (w > v || (w == v && (yy < y || (yy == y && xx < x)))) { keep = false; break; }
                }
            if (keep) out.push_back(Cand{v, x, y});
        }
    std::stable_sort(out.begin(), out.end(), [](const Cand& a, const Cand& b) { return a.v > b.v; });
    for (size_t i = 0; i < out.size() && (int)i < max_cand; i++) {
        xy[2 * i] = out[i].x;
        xy[2 * i + 1] = out[i].y;
        strength[i] = out[i].v;
    }
    return (int)out.size();
}

}  // extern "C"
