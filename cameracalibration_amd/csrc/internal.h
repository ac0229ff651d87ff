// internal.h — shared declarations of libcamsync_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#include <string>
#include "../../include/camsync.h"

namespace camsync {

constexpr int kMaxScales = 8;
// Per-frame lists: quad candidates (DetCfg::quadCap slots, also the rejected candidates reported) and accepted
// markers (DetCfg::markerOut slots).  They start at the sizes below -- 40 markers x 3 scales x border pairs fit, and the
// selection of a frame with up to kQuadLds candidates runs in LDS -- and double when a frame overflows them
// (camsync_api.hip, grow_lists) up to kQuadCeiling: the selection keeps candidate indices in 16 bits.
constexpr int kQuadCap0 = 1024, kMarkerOut0 = 64;
constexpr int kQuadLds = 1024;
constexpr int kQuadCeiling = 32768;
constexpr int kMaxAnchors = 64;

// Detection configuration, derived from camsync_params + frame size and passed
// to kernels by value.
struct DetCfg {
    int H, W;
    uint32_t magicW;         // ceil(2^36 / W): pos / W = umulhi(pos, magicW) >> 4 for every raster position pos < 2^24 (row_of)
    int WS;                  // 32-bit words per bit-plane row = ceil(W/32)
    int nScales;
    int radius[kMaxScales];  // adaptive threshold half windows
    int idelta;              // floor(adaptiveThreshConstant)
    int minPerim, maxPerim;  // contour length bounds (points)
    double accuracyRate;     // polygonalApproxAccuracyRate
    double minCornerDistanceRate;
    int minDistToBorder;
    int borderBits, markerSize, cellSize, cellMargin;
    int maxBorderErrors;     // int(markerSize^2 * maxErroneousBitsInBorderRate)
    double minOtsuStdDev;
    int maxCorrection;       // int(maxCorrectionBits * errorCorrectionRate)
    float minMarkerDistanceRate, minGroupDistance;
    int subpix;              // cornerRefinementMethod == SUBPIX
    int subpixWin, subpixMaxIter;
    float relWin;
    double subpixEps;
    int dictN;
    int quadCap;             // per-frame slots of the quad-candidate lists (multiple of 64)
    int markerOut;           // per-frame slots of the accepted-marker outputs
};

// Bit planes are block-linear: a plane is a grid of WS x HB blocks, each block = 32 rows x 32 pixels =
// 32 words = one 128-byte line (word r of a block holds row r, bit i = pixel i).  A border follower then
// touches one or two lines per neighbourhood instead of three rows 1024 pixels long.
__host__ __device__ inline int plane_hb(int H) { return (H + 31) >> 5; }
#ifdef __HIPCC__
// Row of raster position pos = y * W + x without a division: with m = ceil(2^36 / W) = (2^36 + e) / W, e < W < 2^12,
// pos * m / 2^36 exceeds pos / W by pos * e / (W * 2^36) < 1 / W for pos < 2^24 (frames are at most 4095 a side).
__device__ __forceinline__ int row_of(const DetCfg& cfg, uint32_t pos) { return (int)(__umulhi(pos, cfg.magicW) >> 4); }
#endif
__host__ __device__ inline size_t plane_words(int H, int WS) { return (size_t)plane_hb(H) * WS * 32; }
__host__ __device__ inline size_t plane_word_index(int WS, int xw, int y) {
    return ((size_t)(y >> 5) * WS + xw) * 32 + (y & 31);
}

// 64-bit contour-start candidate: frame<<40 | hole<<39 | scale<<36 | pos(y*W+x)
__host__ __device__ inline uint64_t pack_cand(uint32_t frame, uint32_t hole, uint32_t scale, uint32_t pos) {
    return ((uint64_t)frame << 40) | ((uint64_t)hole << 39) | ((uint64_t)scale << 36) | (uint64_t)pos;
}

struct ContourRec {
    uint32_t frame;
    uint32_t pos;      // start pixel y*W+x
    uint32_t len;      // number of points
    uint32_t off;      // offset into the point pool
    uint16_t scale;
    uint16_t hole;
    uint32_t scanKey;  // raster position at which the sequential scan discovers it
};

struct QuadRec {
    int16_t c[8];      // corners after the clockwise reorder, x0,y0,..x3,y3
    uint32_t scanKey;
    uint16_t scale;
    uint16_t valid;    // identification result
    int32_t clen;      // contour length
    int32_t id, rot;
};

struct MarkerRec {      // accepted marker, compact list for sub-pixel refinement
    uint32_t frame;
    uint32_t slot;      // index in the frame's output
};

// One piece of an accepted contour for the store pass: follow the border from `state` for `count` points
// and write them at pool[dst...].  state = x | y << 12 | s << 24 where s is the direction (0..7) from the
// pixel to its predecessor on the border; s = 8 + hole asks for the initial search of a fresh start.
struct StoreTask {
    uint32_t plane;  // frame * nScales + scale
    uint32_t state;
    uint32_t dst;
    uint32_t count;
};
// A piece (at most kCkStride points) of an accepted long border for the copy pass: the border followers record
// the direction of every step (4 bits) in 128-byte chunks of kCkStride steps, so the points of a piece are its
// start pixel plus a prefix sum over one chunk, written forwards or (backward half of a border) backwards.
struct CopyTask {
    uint32_t chunk;  // index of the 32-word chain chunk
    uint32_t start;  // x | y << 16 of the piece's first pixel
    uint32_t dst;    // pool index of that pixel
    uint32_t n;      // points; bit 31: the following points go to dst - 1, dst - 2, ...
};
// Long borders are followed in segments.  A *seed* is a place on a border that can be recognised from the 3x3
// neighbourhood alone and where a follower can start without knowing where the border began: a foreground pixel on a
// row y = 0 (mod 32) whose west neighbour is background (the follower starts as if it had just swept that crack), on
// a row y = 16 (mod 32) whose east neighbour is background, or, on the other rows, on a column x = 0 / 16 (mod 32)
// whose north / south neighbour is background; in all cases only where the border goes on on both sides of the crack
// (foreground in the three pixels above and below, or left and right), which keeps noise specks out.  A pixel can
// carry one seed at most, so (plane, pixel) names it.  Every seed's follower walks until its sweep passes the crack of the next seed: that is one
// segment (at most kSegLen points; a longer stretch is cut by continuation segments).  The segments of a border form a
// ring through their `next` links, so a border of any length costs kSegLen dependent steps instead of its length.
struct SegRec {
    uint32_t next;     // segment that continues the border, or one of the kSeg* marks
    uint32_t len;      // points of this segment
    uint32_t minAll;   // smallest raster key y * W + x among its points
    uint32_t minEast;  // the same among the points whose east crack lies on the border (hole borders start there)
    uint32_t chunk;    // chain chunk holding the direction of every step
    uint32_t start;    // x | y << 16 of its first point
    uint32_t pad0, pad1;
};
constexpr uint32_t kSegSelf = 0xFFFFFFFFu;   // (start candidates only) the border closed on the candidate's own start state
constexpr uint32_t kSegAbort = 0xFFFFFFFEu;  // (start candidates only) not the canonical start, or longer than allowed
constexpr uint32_t kSegNone = 0xFFFFFFFDu;   // no record: a list overflowed
constexpr uint32_t kSegPending = 0xFFFFFFFCu;  // ends at a seed whose id k_seg_link still has to look up (pad0 / pad1 say which)
constexpr unsigned int kResolveLogHops = 128;  // records per border the log holds (longer rings are walked twice)
constexpr int kSegLen = 256;                  // = kCkStride: one chain chunk per segment

constexpr int kTraceWaves = 256 * 16;  // persistent single-wave workgroups of the long-border kernels
constexpr int kCkStride = 256;         // points per copy task and chain chunk
constexpr int kMaxBorder = 16380;       // longest border kept (maxMarkerPerimeterRate x the larger frame side must not exceed it)

// Compute units of the current device (256 on MI355X), asked once per device: grids that are sized to what the chip
// holds at once follow it.
inline int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// Debugging aid (environment variable CAMSYNC_DEBUG_STAGES): wait for the stream after every kernel of the detection
// sequence and report it on stderr, so that a failing kernel is named.  Off: one getenv at the first call.
inline bool debug_stages() {
    static const bool on = std::getenv("CAMSYNC_DEBUG_STAGES") != nullptr;
    return on;
}
inline void debug_stage(hipStream_t st, const char* name) {
    if (!debug_stages()) return;
    const hipError_t e = hipStreamSynchronize(st);
    std::fprintf(stderr, "[camsync] %s: %s\n", name, hipGetErrorString(e));
    std::fflush(stderr);
}

// Every counter sits in a cache line of its own: they are the targets of all the list appends and work-queue pops of
// the pipeline, and atomics on one line are served one after the other by one L2 channel -- with all of them in one
// 72-byte struct, the three allocators of k_resolve alone queued ~80 k atomics on that line per launch.
struct Counters {
    alignas(128) unsigned int nCand;
    alignas(128) unsigned int nContours;
    alignas(128) unsigned int nPool;
    alignas(128) unsigned int nMarkers;
    alignas(128) unsigned int overflow;  // 1 starts, 2 contours, 4 points, 8 quads, 32 markers, 64 store/copy tasks, 128 bad contour record, 512 border seeds / segments
    alignas(128) unsigned int nLong;     // candidates handed from the short-trace pass to the long-trace pass
    alignas(128) unsigned int qDecide;   // work-queue head of the segment walkers' start candidates (qWalk: their seeds)
    alignas(128) unsigned int qStore;
    alignas(128) unsigned int nTasks;    // store tasks appended by the short-border pass
    alignas(128) unsigned int qShort;    // work-queue head of the short-border kernel
    alignas(128) unsigned int nCopy;     // copy tasks appended by k_resolve
    alignas(128) unsigned int nSeeds;    // border seeds found by the start pass (see SegRec)
    alignas(128) unsigned int nVirt;     // continuation segments handed out by the segment walkers (ids count down from segCap - 1)
    alignas(128) unsigned int qWalk;     // work-queue heads of the segment walkers (seeds) and of the resolve pass
    alignas(128) unsigned int qResolve;
    alignas(128) unsigned int nBin[5 * 32];  // contours per length bin of the quad fit, one cache line each: nBin[32 k]
    alignas(128) unsigned int needQuads;    // after overflow 8 / 32: the most quad candidates / accepted markers a frame had
    alignas(128) unsigned int needMarkers;
    alignas(128) unsigned int pad;
};

// The detection pipeline does not write the gray image when the fused K1+K2 kernel runs: it is a fifth of that
// kernel's memory traffic and more of its time, and only K6 (marker bits) and K7 (sub-pixel corners) look at gray
// values again, a few thousand per marker.  They convert them from the input frame as they read them, with K1's
// integer luma (cv.cvtColor BGR2GRAY: (3735 B + 19235 G + 9798 R + 2^14) >> 15).
struct GraySrc {
    const uint8_t* p;  // first byte of the frame
    int C;             // bytes per pixel: 1 = gray values, 3 = BGR
    uint32_t last;     // index of the frame's last pixel
#ifdef __HIPCC__
    __device__ __forceinline__ int at(size_t idx) const {
        if (C == 1) return p[idx];
        const uint8_t* q = p + idx * 3;
        uint32_t bgr;  // B | G << 8 | R << 16
        if (idx < last) {
            // one (unaligned) 4-byte load; its fourth byte belongs to the next pixel of the frame
            uint32_t w;
            __builtin_memcpy(&w, q, 4);
            bgr = w & 0x00FFFFFFu;
        } else {
            bgr = q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16);
        }
        // the weights split as w = 256 wh + wl, each half a byte dot product (as in K1)
        const uint32_t lo = __builtin_amdgcn_udot4(bgr, 0x00462397u, 16384u, false);
        const uint32_t hi = __builtin_amdgcn_udot4(bgr, 0x00264B0Eu, 0u, false);
        return (int)(((hi << 8) + lo) >> 15);
    }
#endif
};

struct DetBuffers {
    const uint8_t* in;   // [B,H,W,C]
    uint8_t* gray;       // [B,H,W]; null when the gray image is not to be written (see GraySrc)
    const uint8_t* graySrc;  // what K6/K7 read gray values from: `gray` (1 byte per pixel) or `in` itself
    int graySrcC;            // bytes per pixel of graySrc (1 or 3)
    uint8_t* bin;        // [B,H,W] packed scale bits (only when a caller asks for the threshold images), else null
    uint32_t* planes;    // [B,nScales,H,WS] one bit per pixel (bit x&31 of word x>>5), image frame zeroed
    uint64_t* cand;
    uint64_t* candLong;  // same capacity as cand
    unsigned int candCap;
    ContourRec* contours;
    unsigned int contourCap;
    uint32_t* pool;
    unsigned int poolCap;
    StoreTask* tasks;
    unsigned int taskCap;
    uint32_t* chain;     // [chainCap][32] step directions, 8 per word
    unsigned int chainCap;
    CopyTask* copies;
    unsigned int copyCap;
    uint64_t* seeds;     // border seeds, packed like start candidates (bit 39: 0 = west crack, 1 = north crack)
    unsigned int seedCap;
    SegRec* seg;         // [segCap] ids: [0, nSeeds) seeds, [nSeeds, nSeeds + nLong) start candidates, continuations from the top down
    unsigned int segCap;
    uint32_t* seedTable;  // seed -> id, one table per bit plane indexed by the seed's place on the grid lines (k_contours.hip)
    uint32_t* binIdx;    // [5][contourCap] contour indices by length bin (k_quad_bin)
    QuadRec* quads;      // [B,quadCap]
    int* qcount;         // [B]
    int quadCap;         // per-frame slots of quads, persp, idList, selState, outRejected
    int markerOut;       // per-frame slots of outIds, outCorners, rawCorners
    uint8_t* selWork;    // [B, sel_work_bytes(quadCap)] working set of k_select for frames with more than kQuadLds candidates, or null
    Counters* counters;
    MarkerRec* markers;  // compact accepted list
    unsigned int markerCap;
    // per-frame outputs
    int32_t* outIds;     // [B,markerOut]
    float* outCorners;   // [B,markerOut,8] final (sub-pixel) corners
    float* rawCorners;   // [B,markerOut,8] integer corners before refinement
    int32_t* outCounts;  // [B]
    float* outRejected;  // [B,quadCap,8]
    int32_t* outRejCounts;
    // tables
    const uint16_t* dict;   // dictN codes under each of the four rotations, [rotation][entry]
    uint4* resolveLog;      // [resolveWaves, kResolveLogHops, 64] records visited by k_resolve's first pass (chunk, start, len)
    unsigned int resolveWaves;  // waves of k_resolve (each owns a slice of the log)
    double* persp;          // [B, quadCap, 10] inverse perspective map of every quad + usable flag (k_quad_persp)
    uint32_t* idList;       // [B, quadCap] quads to identify (k_group): from the front the candidates selected by the grouping
                            // (quad index | 0xFFFF << 16), from the back their fall-backs (quad index | representative << 16)
    int* idCount;           // [B, 2] entries at the front / at the back
    int16_t* selState;      // [B, 3, quadCap] k_group's order, selection flags and fall-back relation for the final k_select
    int identifyAll;        // debug: identify every quad candidate, not only the ones the selection can ask for
    const float* expTab;    // [21][41] separable cornerSubPix weights
};

// launchers (each enqueues on `st`, no sync)
void launch_gray_thresh(const DetCfg& c, const DetBuffers& b, int B, int C, hipStream_t st);
// true when launch_gray_thresh runs the fused kernel, which can do without writing the gray image
bool gray_thresh_is_fused(const DetCfg& c);
void launch_yuyv_bgr(const uint8_t* d_yuyv, uint8_t* d_bgr, size_t px, hipStream_t st);
void launch_contour_starts(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st);
void launch_trace(const DetCfg& c, const DetBuffers& b, hipStream_t st);
void launch_quad_fit(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st);
void launch_identify(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st);
void launch_select(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st);
void launch_group(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st);
void launch_subpix(const DetCfg& c, const DetBuffers& b, int B, hipStream_t st);
size_t sel_work_bytes(int quadCap);
void launch_export_markers(const DetBuffers& b, int B, int maxMarkers, int32_t* dIds, float* dCorners, int32_t* dCounts,
                           hipStream_t st);

// pose
struct PnpArgs {
    const float* obj;      // [B,N,3]
    const float* img;      // [B,N,2]
    const int32_t* npts;   // [B]
    int B, N;
    const double* K;       // [B or 1, 9]
    const double* dist;    // [B or 1, ndist]
    int ndist, perFrameCam;
    int refine;            // run the solvePnPRefineLM phase
    double* rvec;          // [B,3]
    double* tvec;          // [B,3]
    uint8_t* ok;           // [B]
    double* mae;           // [B] or null
    double* rmse;          // [B] or null
};
void launch_pnp(const PnpArgs& a, hipStream_t st);
void launch_gather_anchors(const int32_t* ids, const float* corners, const int32_t* counts, int B, int markerOut,
                           const int32_t* anchorIds, const float* anchorObj, int nAnchor, int N,
                           float* obj, float* img, int32_t* npts, hipStream_t st);
void launch_ippe(const float* corners, const double* size, int B, const double* K, const double* dist,
                 int ndist, int perFrameCam, double* rvec, double* tvec, uint8_t* ok, hipStream_t st);

void launch_fuse_moving(const int32_t* ids, const float* corners, const int32_t* counts, int markerOut, const uint8_t* poseOk,
                        const double* camRvec, const double* camTvec, int B, const int32_t* movingIds, const double* movingSizes,
                        int M, const double* K, const double* dist, int ndist, int perFrameCam, float* eCorners, double* eSize,
                        double* eK, double* eDist, uint8_t* eValid, double* mRvec, double* mTvec, uint8_t* mOk, double* fused,
                        int32_t* seen, uint8_t* located, hipStream_t st);

// chessboard corner candidates (k_chess.hip) and cornerSubPix on caller-supplied points (k_subpix.hip)
void launch_chess(const uint8_t* in, int C, int B, int H, int W, int scale, uint8_t* gray, uint16_t* S, int32_t* R,
                  int* frameMax, int maxCand, int32_t* cand, int* counts, hipStream_t st);
void launch_subpix_points(const uint8_t* gray, int grayC, int B, int H, int W, const float* pts, const int32_t* npts, int N,
                          int win, int maxIters, double eps, const float* expTab, float* out, hipStream_t st);

// undistortion (k_undistort.hip)
void optimal_new_camera_matrix(const double* K, const double* k8, int W, int H, double alpha, int newW, int newH,
                               double* newK, int32_t* roi);
bool undistort_camera_ok(const double* newK);
void launch_undistort_map(const double* K, const double* newK, const double* k8, int H, int W, uint2* map, hipStream_t st);
void launch_remap(const uint8_t* src, const uint2* map, uint8_t* dst, int B, int H, int W, int C, hipStream_t st);

// compressed frame ingest (k_jpeg.hip)
struct JpegState;
JpegState* jpeg_create();
void jpeg_destroy(JpegState* s);
int jpeg_info(const uint8_t* data, size_t n, int32_t* height, int32_t* width, int32_t* components);
int jpeg_decode_batch(JpegState* s, hipStream_t st, const uint8_t* const* jpegs, const size_t* sizes, int B, int H, int W,
                      uint8_t* out, int outOnDevice, std::string& err);

// calibration
struct CalibState {
    int V, P, NI;             // local views, points per view, active intrinsics (9 or 12)
    const float* obj;         // [V,P,3]
    const float* img;         // [V,P,2]
    double* intr;             // [12] fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6 (current)
    double* ext;              // [V,6] current
    double* intrTrial;        // [12]
    double* extTrial;         // [V,6]
    double* viewBlocks;       // [V, 6*NI + 6] Aee'^-1 Bie^T and Aee'^-1 g_e of the current parameters
    double* viewBlocksTrial;  // the same of the trial parameters (accepting a trial swaps the three pairs)
    double* partial;          // [workgroups + 32, 160] per-workgroup records of k_calib_lm, then its per-group records
    unsigned int* ticket;     // [1 + 32] groups done, workgroups done per group (k_calib_lm leaves them at 0)
};
// offsets into the CAMSYNC_CALIB_WORK_DOUBLES work buffer: the record [S | rhs | err | change sums] (summed over
// ranks), the focal-length normal equations, the replicated values of a trial step, the intrinsic step
constexpr int kWAcc = 0, kWFocal = 160, kWRec = 168, kWDelta = 176;
uint32_t rotate_marker_code(uint32_t code, int n, int r);
size_t seed_table_entries(int H, int W);
int calib_lm_grid(int V);
void launch_fill_i32(int32_t* dst, int n, int32_t value, hipStream_t st);
// report / seq: see k_calib_lm (host-visible copy of what the LM loop decides from; null = none)
void launch_calib_lm(const CalibState& s, int mode, double lambda, double* work, hipStream_t st, double* report = nullptr,
                     unsigned long long seq = 0);
void launch_calib_focal_rows(const CalibState& s, double cx, double cy, double* rows, double* acc6, hipStream_t st);

}  // namespace camsync
