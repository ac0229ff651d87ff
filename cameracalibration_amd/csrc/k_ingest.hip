// k_ingest.hip — frame ingest: packed YUYV 4:2:2 (what a UVC camera delivers, V4L2_PIX_FMT_YUYV) -> BGR frames in
// device memory, the layout every other entry point takes.  Replaces the conversion cv.VideoCapture.read() does on
// the CPU before the reference sees a frame (reference camera_sync/camera.py:104-118, 150-154); the raw frame is two
// thirds of the BGR one, so the host copy shrinks by a third as well.
//
// Arithmetic of cv.cvtColor(COLOR_YUV2BGR_YUYV): ITU-R BT.601 limited range in 20-bit fixed point, the chroma pair
// shared by two pixels.  HBM-bound: 2 B/px read + 3 B/px written; a lane converts 8 pixels (one 16-byte load, three
// 8-byte stores; the lanes of a wavefront cover 1 KiB in and 1.5 KiB out, contiguous).
#include "internal.h"

namespace camsync {

namespace {

constexpr int kShift = 20, kCY = 1220542, kCUB = 2116026, kCUG = -409993, kCVG = -852492, kCVR = 1673527;

// saturate_cast<uchar>(v >> kShift): clamped before the shift.  (Written as a shift followed by a clamp, the compiler
// fuses shift, clamp and byte packing into gfx950's v_ashr_pk_u8_i32, and the bytes that came out of that were wrong
// -- R of the first and B of the second pixel of every pair; found by the bit-exact test against the CPU restatement.)
__device__ __forceinline__ uint32_t sat8s(int v) {
    return (uint32_t)min(max(v, 0), (255 << kShift) | ((1 << kShift) - 1)) >> kShift;
}

// one pixel pair (Y0 U Y1 V) -> six bytes B0 G0 R0 B1 G1 R1, returned in the low 48 bits
__device__ __forceinline__ uint64_t pair_bgr(uint32_t p) {
    const int y0 = (int)(p & 255u), u = (int)((p >> 8) & 255u) - 128, y1 = (int)((p >> 16) & 255u), v = (int)(p >> 24) - 128;
    const int ruv = (1 << (kShift - 1)) + kCVR * v;
    const int guv = (1 << (kShift - 1)) + kCVG * v + kCUG * u;
    const int buv = (1 << (kShift - 1)) + kCUB * u;
    const int a = max(0, y0 - 16) * kCY, b = max(0, y1 - 16) * kCY;
    const uint32_t lo = sat8s(a + buv) | (sat8s(a + guv) << 8) | (sat8s(a + ruv) << 16) | (sat8s(b + buv) << 24);
    const uint32_t hi = sat8s(b + guv) | (sat8s(b + ruv) << 8);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__global__ __launch_bounds__(256) void k_yuyv_bgr(const uint4* __restrict__ yuyv, uint2* __restrict__ bgr, size_t groups) {
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (size_t)gridDim.x * 256) {
        const uint4 in = yuyv[g];  // 8 pixels
        const uint64_t p0 = pair_bgr(in.x), p1 = pair_bgr(in.y), p2 = pair_bgr(in.z), p3 = pair_bgr(in.w);
        // 24 bytes: p0 (6) p1 (6) p2 (6) p3 (6)
        const uint64_t w0 = p0 | (p1 << 48), w1 = (p1 >> 16) | (p2 << 32), w2 = (p2 >> 32) | (p3 << 16);
        uint2* out = bgr + g * 3;
        out[0] = make_uint2((uint32_t)w0, (uint32_t)(w0 >> 32));
        out[1] = make_uint2((uint32_t)w1, (uint32_t)(w1 >> 32));
        out[2] = make_uint2((uint32_t)w2, (uint32_t)(w2 >> 32));
    }
}

}  // namespace

// px must be a multiple of 8; yuyv 16-byte and bgr 8-byte aligned (checked by the caller)
void launch_yuyv_bgr(const uint8_t* d_yuyv, uint8_t* d_bgr, size_t px, hipStream_t st) {
    const size_t groups = px / 8;
    const unsigned int grid = (unsigned int)std::min<size_t>((groups + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(k_yuyv_bgr, dim3(grid), dim3(256), 0, st, reinterpret_cast<const uint4*>(d_yuyv),
                       reinterpret_cast<uint2*>(d_bgr), groups);
}

}  // namespace camsync
