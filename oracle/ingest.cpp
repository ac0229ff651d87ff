/*
 * oracle/ingest.cpp — CPU restatement of the colour conversion behind the reference's frame ingest
 * (reference camera_sync/camera.py:104-118, 150-154: cv.VideoCapture(...).read()).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * A UVC camera delivers packed YUYV 4:2:2 (V4L2_PIX_FMT_YUYV: Y0 U Y1 V per pixel pair); VideoCapture hands the
 * reference BGR frames, converted on the CPU as cv.cvtColor(COLOR_YUV2BGR_YUYV) does.  Restated from OpenCV 4.11's
 * published algorithm (imgproc color_yuv: ITU-R BT.601 limited range, 20-bit fixed point, chroma shared by the pixel
 * pair, no interpolation).  The reference holds no captured raw frame and no output of this conversion: PARITY
 * UNPINNED against OpenCV; the known answers of the BT.601 colour bars are checked in tests/test_ingest.py and the
 * GPU path is compared with this file bit for bit.
 */
#include <stdint.h>

#include <algorithm>

#include "oracle.h"

namespace {

const int kShift = 20, kCY = 1220542, kCUB = 2116026, kCUG = -409993, kCVG = -852492, kCVR = 1673527;

inline uint8_t sat8(int v) { return (uint8_t)std::min(255, std::max(0, v)); }

}  // namespace

extern "C" {

/* yuyv u8 [H, W/2, 4] -> bgr u8 [H, W, 3]; W must be even.  Returns 0, or -1 on bad arguments. */
int orc_yuyv_to_bgr(const uint8_t* yuyv, int H, int W, uint8_t* bgr) {
    if (!yuyv || !bgr || H <= 0 || W <= 0 || (W & 1)) return -1;
    for (int y = 0; y < H; y++) {
        const uint8_t* s = yuyv + (size_t)y * W * 2;
        uint8_t* d = bgr + (size_t)y * W * 3;
        for (int x = 0; x < W; x += 2, s += 4, d += 6) {
            const int u = (int)s[1] - 128, v = (int)s[3] - 128;
            const int ruv = (1 << (kShift - 1)) + kCVR * v;
            const int guv = (1 << (kShift - 1)) + kCVG * v + kCUG * u;
            const int buv = (1 << (kShift - 1)) + kCUB * u;
            for (int k = 0; k < 2; k++) {
                const int yy = std::max(0, (int)s[2 * k] - 16) * kCY;
                d[3 * k + 0] = sat8((yy + buv) >> kShift);
                d[3 * k + 1] = sat8((yy + guv) >> kShift);
                d[3 * k + 2] = sat8((yy + ruv) >> kShift);
            }
        }
    }
    return 0;
}

} /* extern "C" */
