// camsync_api.hip — the extern "C" boundary of libcamsync_hip.so
// (include/camsync.h).  Owns the context: one HIP stream, the scratch lists of
// the detection pipeline sized for max_batch frames, the dictionary and the
// cornerSubPix weight table, the calibration state.  Batches larger than
// max_batch are processed in chunks on the same stream.
#include "internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace camsync;

struct camsync_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int maxH = 0, maxW = 0, maxB = 0;
    camsync_params params;
    int dictN = 0, markerSize = 4, maxCorrectionBits = 1;
    std::string err;
    unsigned char* naScratch = nullptr;  // this GPU's copy of the no-allocation scratch array (see scratch_for_device)
    unsigned int allocContourCap = 0, allocPoolCap = 0, allocTaskCap = 0;  // set once camsync_debug_limit_capacity lowered them

    // device scratch
    uint8_t* d_in = nullptr;   // staging for host frames (lazy)
    size_t inBytes = 0;
    uint8_t* d_bin = nullptr;  // byte-per-pixel threshold images for camsync_gray_threshold (lazy)
    int planeScales = 0;       // scales the bit-plane buffer is sized for
    DetBuffers buf{};
    uint16_t* d_dict = nullptr;
    float* d_expTab = nullptr;
    // pose scratch
    float *d_obj = nullptr, *d_img = nullptr;
    int32_t* d_npts = nullptr;
    double *d_K = nullptr, *d_dist = nullptr, *d_rvec = nullptr, *d_tvec = nullptr, *d_mae = nullptr, *d_rmse = nullptr;
    uint8_t* d_ok = nullptr;
    int32_t* d_anchorIds = nullptr;
    float* d_anchorObj = nullptr;
    size_t poseCap = 0;  // frames the pose scratch is sized for
    int poseN = 0;

    // undistortion: the map of the last (camera, new camera, size) and the output staging (both allocated on first use)
    uint2* d_umap = nullptr;
    size_t umapPx = 0;
    double umapKey[9 + 9 + 8] = {};
    int umapH = 0, umapW = 0;
    uint8_t* d_uout = nullptr;

    // chessboard candidates / cornerSubPix on points: scratch for chunks of chessChunk frames (allocated on first use)
    int chessChunk = 0, chessMaxCand = 0, subpixCap = 0;
    uint8_t* d_chessGray = nullptr;
    uint16_t* d_chessS = nullptr;
    int32_t *d_chessR = nullptr, *d_chessCand = nullptr;
    int *d_chessMax = nullptr, *d_chessCount = nullptr;
    float *d_spIn = nullptr, *d_spOut = nullptr;
    int32_t* d_spN = nullptr;

    // last detect chunk (for the debug getters)
    int lastB = 0, lastH = 0, lastW = 0;
    int hostQuadCap = 0;  // stride the pinned host buffer h_rej was sized for
    int lastMarkers = 0, lastRejected = 0;   // most markers / rejected candidates a frame of the last detection call had
    JpegState* jpeg = nullptr;  // compressed frame ingest (created on first use)
    int regrown = 0;  // times a scratch list was doubled after an overflow (grow_lists)
    // the last camsync_detect_pnp_batch / _frames call, when it was one chunk: its detections and camera poses are still
    // on the device for camsync_fuse_moving_markers (poseFrames = 0: nothing to fuse from)
    int poseFrames = 0, poseNdist = 0, posePerFrameCam = 0;
    // scratch of camsync_fuse_moving_markers, for fuseCap (frame, marker) entries and fuseCapM markers
    size_t fuseCap = 0, fuseCapM = 0;
    float* d_fCorners = nullptr;
    double *d_fSize = nullptr, *d_fK = nullptr, *d_fDist = nullptr, *d_fRvec = nullptr, *d_fTvec = nullptr, *d_fFused = nullptr, *d_fSizes = nullptr;
    uint8_t *d_fValid = nullptr, *d_fOk = nullptr, *d_fLocated = nullptr;
    int32_t *d_fIds = nullptr, *d_fSeen = nullptr;

    // calibration
    CalibState cal{};
    float *d_calObj = nullptr, *d_calImg = nullptr;
    double* d_calWork = nullptr;   // [CAMSYNC_CALIB_WORK_DOUBLES]: acc | focal sums | trial record | delta | flag (see kW*)
    double* d_focalRows = nullptr;
    size_t calCapPts = 0, calCapV = 0;  // what the calibration buffers were sized for
    int calCapNI = 0;
    double* h_pin = nullptr;       // pinned host doubles for the small read-backs of the LM loop
    // host frames on their way to the device (upload_host): page-locked pieces, an event per piece
    uint8_t* upPinned = nullptr;
    hipEvent_t upEv[33] = {};  // one per place in the buffer, and one for the whole of the last upload
    bool upPending = false;
    double* h_report = nullptr;    // coherent host memory the LM kernel reports to on one GPU: 7 doubles, sequence number at [8]
    unsigned long long calSeq = 0;
    // pinned host copies of a chunk's detection results (allocated on first use, kept for the context's life)
    int32_t *h_rejCounts = nullptr;
    float *h_rej = nullptr;
    // a chunk's results (counters, counts, poses, ids, corners) gathered into one block on the device (k_pack) and copied
    // to page-locked host memory in one transfer
    uint8_t *d_pack = nullptr, *h_pack = nullptr;
    size_t packBytes = 0;
    uint8_t* d_calOk = nullptr;
    double* d_calK = nullptr;
    int32_t* d_calNpts = nullptr;
    bool calReady = false;

    // timers
    bool timing = false;
    hipEvent_t ev[CAMSYNC_N_STAGES + 1] = {};
    double ms[CAMSYNC_N_STAGES] = {};
    int64_t launches[CAMSYNC_N_STAGES] = {};
};

// ---- scratch of the entry point that allocates nothing (camsync_pnp_batch_noalloc) ------------------------------
// camsync_pnp_batch stages its host inputs and its results in device buffers that it (re)allocates when a batch is
// larger than any before.  The no-allocation variant stages them in this file-scope device array instead: every GPU
// the library runs on holds one copy (400 KiB) until the process ends.  In host code the array's NAME is only a
// handle; its device address is resolved once per device, with that device current, by hipGetSymbolAddress
// (scratch_for_device below, called from camsync_create), and only that pointer is ever passed to copies and kernels.
namespace {
constexpr size_t kNaMaxFrames = CAMSYNC_NOALLOC_MAX_FRAMES, kNaMaxPoints = CAMSYNC_NOALLOC_MAX_POINTS;
constexpr size_t kNaObj = 0;                                        // f32 [points, 3]
constexpr size_t kNaImg = kNaObj + kNaMaxPoints * 3 * 4;            // f32 [points, 2]
constexpr size_t kNaK = kNaImg + kNaMaxPoints * 2 * 4;              // f64 [frames, 9]
constexpr size_t kNaDist = kNaK + kNaMaxFrames * 9 * 8;             // f64 [frames, 8]
constexpr size_t kNaRvec = kNaDist + kNaMaxFrames * 8 * 8;          // f64 [frames, 3]
constexpr size_t kNaTvec = kNaRvec + kNaMaxFrames * 3 * 8;          // f64 [frames, 3]
constexpr size_t kNaMae = kNaTvec + kNaMaxFrames * 3 * 8;           // f64 [frames]
constexpr size_t kNaRmse = kNaMae + kNaMaxFrames * 8;               // f64 [frames]
constexpr size_t kNaNpts = kNaRmse + kNaMaxFrames * 8;              // i32 [frames]
constexpr size_t kNaOk = kNaNpts + kNaMaxFrames * 4;                // u8  [frames]
constexpr size_t kNaBytes = ((kNaOk + kNaMaxFrames + 255) / 256) * 256;
static_assert(kNaImg % 8 == 0 && kNaK % 8 == 0 && kNaNpts % 8 == 0, "scratch fields are 8-byte aligned");
}  // namespace
static __device__ __align__(256) unsigned char scratch[kNaBytes];

namespace {

constexpr int kMaxDevices = 64;
unsigned char* g_scratchPtr[kMaxDevices] = {};  // device address of `scratch` on each GPU (each GPU has its own copy)
std::mutex g_scratchMutex;

// Resolve the address of this device's copy of `scratch` (the device must be current).  Returns null on failure.
unsigned char* scratch_for_device(int device, hipError_t* err) {
    *err = hipSuccess;
    if (device < 0 || device >= kMaxDevices) { *err = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    if (!g_scratchPtr[device]) {
        void* p = nullptr;
        *err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (*err != hipSuccess || !p) return nullptr;
        g_scratchPtr[device] = static_cast<unsigned char*>(p);
    }
    return g_scratchPtr[device];
}

// layout of the calibration work buffer (doubles); everything that is summed over ranks lives here, so that a
// caller-provided buffer (e.g. a torch tensor) can be handed to the collective
static_assert(kWDelta + 12 <= CAMSYNC_CALIB_WORK_DOUBLES && 12 * 12 + 12 + 3 <= kWFocal, "work buffer layout");

int fail(camsync_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        char tmp[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(tmp, sizeof tmp, fmt, ap);
        va_end(ap);
        c->err = tmp;
    }
    return code;
}

#define HIPCHK(c, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail(c, CAMSYNC_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <class T>
hipError_t dalloc(T** p, size_t n) {
    return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
}

// DICT_4X4_50 codes are supplied by the caller (camsync_set_dictionary); the
// Python mirror ships the table (camera_sync/_dict4x4_50.py).

DetCfg make_cfg(const camsync_ctx* c, int H, int W) {
    const camsync_params& p = c->params;
    DetCfg g{};
    g.quadCap = c->buf.quadCap;
    g.markerOut = c->buf.markerOut;
    g.H = H;
    g.W = W;
    g.WS = (W + 31) / 32;
    g.magicW = (uint32_t)(((1ull << 36) + (unsigned long long)W - 1ull) / (unsigned long long)W);
    g.nScales = (p.adaptiveThreshWinSizeMax - p.adaptiveThreshWinSizeMin) / p.adaptiveThreshWinSizeStep + 1;
    g.nScales = std::min(g.nScales, kMaxScales);
    for (int i = 0; i < g.nScales; i++) {
        int win = p.adaptiveThreshWinSizeMin + i * p.adaptiveThreshWinSizeStep;
        if (win % 2 == 0) win++;
        g.radius[i] = win / 2;
    }
    g.idelta = (int)floor(p.adaptiveThreshConstant);
    const int maxdim = std::max(W, H);
    g.minPerim = (int)(unsigned)(p.minMarkerPerimeterRate * maxdim);
    g.maxPerim = (int)(unsigned)(p.maxMarkerPerimeterRate * maxdim);
    g.accuracyRate = p.polygonalApproxAccuracyRate;
    g.minCornerDistanceRate = p.minCornerDistanceRate;
    g.minDistToBorder = p.minDistanceToBorder;
    g.borderBits = p.markerBorderBits;
    g.markerSize = c->markerSize;
    g.cellSize = p.perspectiveRemovePixelPerCell;
    g.cellMargin = (int)(p.perspectiveRemoveIgnoredMarginPerCell * p.perspectiveRemovePixelPerCell);
    g.maxBorderErrors = (int)(c->markerSize * c->markerSize * p.maxErroneousBitsInBorderRate);
    g.minOtsuStdDev = p.minOtsuStdDev;
    g.maxCorrection = (int)((double)c->maxCorrectionBits * p.errorCorrectionRate);
    g.minMarkerDistanceRate = (float)p.minMarkerDistanceRate;
    g.minGroupDistance = p.minGroupDistance;
    g.subpix = p.cornerRefinementMethod == 1;
    g.subpixWin = p.cornerRefinementWinSize;
    g.subpixMaxIter = p.cornerRefinementMaxIterations;
    g.relWin = p.relativeCornerRefinmentWinSize;
    g.subpixEps = p.cornerRefinementMinAccuracy;
    g.dictN = c->dictN;
    return g;
}

int check_params(camsync_ctx* c, const camsync_params* p) {
    if (p->adaptiveThreshWinSizeMin < 3 || p->adaptiveThreshWinSizeMax < p->adaptiveThreshWinSizeMin ||
        p->adaptiveThreshWinSizeStep <= 0)
        return fail(c, CAMSYNC_E_ARG, "adaptive threshold window parameters out of range");
    const int nScales = (p->adaptiveThreshWinSizeMax - p->adaptiveThreshWinSizeMin) / p->adaptiveThreshWinSizeStep + 1;
    if (nScales > kMaxScales) return fail(c, CAMSYNC_E_ARG, "more than %d threshold scales", kMaxScales);
    if (p->markerBorderBits < 1 || p->perspectiveRemovePixelPerCell < 1)
        return fail(c, CAMSYNC_E_ARG, "markerBorderBits / perspectiveRemovePixelPerCell must be >= 1");
    if ((c->markerSize + 2 * p->markerBorderBits) * p->perspectiveRemovePixelPerCell > 32 ||
        (c->markerSize + 2 * p->markerBorderBits) * (c->markerSize + 2 * p->markerBorderBits) > 64)
        return fail(c, CAMSYNC_E_ARG, "normalised marker image larger than 32x32 pixels / 64 cells");
    if (p->cornerRefinementMethod != 0 && p->cornerRefinementMethod != 1)
        return fail(c, CAMSYNC_E_ARG, "cornerRefinementMethod must be NONE(0) or SUBPIX(1)");
    if (p->cornerRefinementMethod == 1 &&
        (p->cornerRefinementWinSize < 1 || p->cornerRefinementWinSize > 20 || p->cornerRefinementMaxIterations < 1 ||
         p->cornerRefinementMinAccuracy <= 0))
        return fail(c, CAMSYNC_E_ARG, "corner refinement parameters out of range (winSize 1..20)");
    return CAMSYNC_OK;
}

struct StageTimer {
    camsync_ctx* c;
    int idx = 0;
    explicit StageTimer(camsync_ctx* c_) : c(c_) {
        if (c->timing) (void)hipEventRecord(c->ev[0], c->stream);
    }
    void mark(int stage) {  // call after enqueuing `stage`
        if (!c->timing) return;
        (void)hipEventRecord(c->ev[stage + 1], c->stream);
    }
};

int ensure_staging(camsync_ctx* c) {
    if (c->d_in) return CAMSYNC_OK;
    c->inBytes = (size_t)c->maxB * c->maxH * c->maxW * 3;
    HIPCHK(c, dalloc(&c->d_in, c->inBytes));
    return CAMSYNC_OK;
}

// Frames in ordinary (pageable) host memory to the device.  hipMemcpyAsync stages such memory itself, copying and
// sending a megabyte at a time in turn on the calling thread: 50 MB (eight 1080p frames) in 1.86 ms, against 0.87 ms from
// page-locked memory.  Here several threads copy pieces into page-locked memory and each sends its piece as soon as it is
// there, so that the link is busy while the rest is still being copied.  Pieces of 2 MB with nothing between their
// transfers: 48 pieces of 1 MB with an event after each took 1.5 ms for the transfers alone.  An upload of more than
// the page-locked buffer (64 MB) takes its pieces round the buffer, and a piece then waits for the transfer that last
// used its place.  Everything is enqueued on the context's stream: what follows on it sees the frames.
#ifndef UP_BIG_PIECE
#define UP_BIG_PIECE 4
#endif
constexpr size_t kUpPiece = 2u << 20, kUpBuffer = 64u << 20;
constexpr int kUpThreads = 8, kUpSlots = (int)(kUpBuffer / kUpPiece);
static_assert(kUpSlots % kUpThreads == 0 && (kUpSlots / 4) % kUpThreads == 0, "a place in the buffer is always filled by the same thread");
struct UpSeg { uint8_t* dst; const uint8_t* src; size_t n; };

int upload_host(camsync_ctx* c, const UpSeg* segs, int nseg) {
    size_t total = 0;
    for (int i = 0; i < nseg; i++) total += segs[i].n;
    // page-locked frames (camsync_host_alloc, hipHostRegister) go as they are: the link's full rate without a copy
    bool locked = true;
    for (int i = 0; i < nseg && locked; i++) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, segs[i].src) != hipSuccess) {
            (void)hipGetLastError();  // (ordinary memory is unknown to the runtime)
            locked = false;
        } else {
            locked = at.type == hipMemoryTypeHost;
        }
    }
    if (locked || total < 2 * kUpPiece) {
        for (int i = 0; i < nseg; i++) HIPCHK(c, hipMemcpyAsync(segs[i].dst, segs[i].src, segs[i].n, hipMemcpyHostToDevice, c->stream));
        return CAMSYNC_OK;
    }
    if (!c->upPinned) {
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->upPinned), kUpBuffer, hipHostMallocDefault));
        for (int i = 0; i <= kUpSlots; i++) HIPCHK(c, hipEventCreateWithFlags(&c->upEv[i], hipEventDisableTiming));
    }
    // (every caller waits for the stream before it returns; this is for the one that did not get that far)
    if (c->upPending) HIPCHK(c, hipEventSynchronize(c->upEv[kUpSlots]));
    c->upPending = false;
    // (an upload that goes round the buffer pays for an event per piece: larger pieces then)
    const size_t piece = total > kUpBuffer ? UP_BIG_PIECE * kUpPiece : kUpPiece;
    const size_t slots = kUpBuffer / piece;
    // a piece = up to `piece` bytes of consecutive destination: part of a large frame, or many small frames that follow
    // one another on the device (seg, off: where it starts; n: its bytes)
    struct Piece { uint8_t* dst; int seg; size_t off, n; };
    std::vector<Piece> pieces;
    for (int i = 0; i < nseg;) {
        size_t o = 0;
        while (i < nseg) {
            Piece pc{segs[i].dst + o, i, o, 0};
            // fill the piece from this segment on, across segments whose destinations are adjacent
            int j = i;
            size_t oj = o;
            while (j < nseg && pc.n < piece) {
                const size_t take = std::min(piece - pc.n, segs[j].n - oj);
                pc.n += take;
                oj += take;
                if (oj == segs[j].n) {
                    const bool adjacent = j + 1 < nseg && segs[j + 1].dst == segs[j].dst + segs[j].n;
                    j++;
                    oj = 0;
                    if (!adjacent) break;
                }
            }
            pieces.push_back(pc);
            i = j;
            o = oj;
            if (o == 0) break;
        }
    }
    const size_t nP = pieces.size();
    const int T = (int)std::min<size_t>(kUpThreads, nP);
    std::atomic<int> err{(int)hipSuccess};
    auto work = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess) { err = (int)hipErrorInvalidDevice; return; }
        for (size_t p = (size_t)t; p < nP && err.load() == (int)hipSuccess; p += (size_t)T) {
            const size_t slot = p % slots;
            uint8_t* pin = c->upPinned + slot * piece;
            hipError_t e = p >= slots ? hipEventSynchronize(c->upEv[slot]) : hipSuccess;  // the piece sent from this place before
            if (e == hipSuccess) {
                size_t done = 0, off = pieces[p].off;
                for (int sg = pieces[p].seg; done < pieces[p].n; sg++, off = 0) {
                    const size_t take = std::min(pieces[p].n - done, segs[sg].n - off);
                    memcpy(pin + done, segs[sg].src + off, take);
                    done += take;
                }
                e = hipMemcpyAsync(pieces[p].dst, pin, pieces[p].n, hipMemcpyHostToDevice, c->stream);
            }
            if (e == hipSuccess && p + slots < nP) e = hipEventRecord(c->upEv[slot], c->stream);  // (someone will wait for it)
            if (e != hipSuccess) err = (int)e;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    HIPCHK(c, (hipError_t)err.load());
    HIPCHK(c, hipEventRecord(c->upEv[kUpSlots], c->stream));
    c->upPending = true;
    return CAMSYNC_OK;
}

// The gray image is only written for camsync_gray_threshold and for detector parameters that the fused K1+K2
// kernel does not cover: allocated on first use (8.5 GB for 4096 1080p frames).
int ensure_gray(camsync_ctx* c) {
    if (c->buf.gray) return CAMSYNC_OK;
    HIPCHK(c, dalloc(&c->buf.gray, (size_t)c->maxB * c->maxH * c->maxW));
    return CAMSYNC_OK;
}

// Where the results of a chunk of nb frames lie in the packed block (offsets are multiples of 8).
struct PackLayout {
    size_t counters, counts, ok, rvec, tvec, mae, rmse, ids, corners, total;
    PackLayout(size_t nb, size_t mo) {
        auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
        size_t o = 0;
        counters = o; o += up8(sizeof(Counters));
        counts = o; o += up8(nb * 4);
        ok = o; o += up8(nb);
        rvec = o; o += nb * 24;
        tvec = o; o += nb * 24;
        mae = o; o += nb * 8;
        rmse = o; o += nb * 8;
        ids = o; o += up8(nb * mo * 4);
        corners = o; o += nb * mo * 32;
        total = o;
    }
};

// up to ten arrays -> one block: the results of a chunk leave the device in one transfer instead of nine (each small
// transfer costs ~30 us of an 8-camera set's 2.3 ms)
struct PackSegs {
    const void* src[10];
    unsigned long long off[10], bytes[10];
    int n;
};
__global__ __launch_bounds__(256) void k_pack(PackSegs s, uint8_t* __restrict__ dst) {
    for (int k = 0; k < s.n; k++) {
        const size_t words = (size_t)(s.bytes[k] >> 2);
        const uint32_t* __restrict__ src = static_cast<const uint32_t*>(s.src[k]);
        uint32_t* __restrict__ d = reinterpret_cast<uint32_t*>(dst + s.off[k]);
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) d[i] = src[i];
        if (blockIdx.x == 0 && threadIdx.x < (s.bytes[k] & 3ull))
            dst[s.off[k] + words * 4 + threadIdx.x] = static_cast<const uint8_t*>(s.src[k])[words * 4 + threadIdx.x];
    }
}

int ensure_host_out(camsync_ctx* c, bool wantRejected) {
    const size_t B = (size_t)c->maxB;
    const int qc = c->buf.quadCap;
    if (!c->h_rejCounts) HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_rejCounts), sizeof(int32_t) * B, hipHostMallocDefault));
    const size_t need = PackLayout(B, (size_t)c->buf.markerOut).total;
    if (c->packBytes < need) {  // (first use, or the per-frame marker slots have grown: grow_lists)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->h_pack) (void)hipHostFree(c->h_pack);
        if (c->d_pack) (void)hipFree(c->d_pack);
        c->h_pack = nullptr;
        c->d_pack = nullptr;
        c->packBytes = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pack), need, hipHostMallocDefault));
        HIPCHK(c, dalloc(&c->d_pack, need));
        c->packBytes = need;
    }
    if (wantRejected && c->hostQuadCap != qc) {
        if (c->h_rej) (void)hipHostFree(c->h_rej);
        c->h_rej = nullptr;
        c->hostQuadCap = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_rej), sizeof(float) * B * qc * 8, hipHostMallocDefault));
        c->hostQuadCap = qc;
    }
    return CAMSYNC_OK;
}

int ensure_bin(camsync_ctx* c) {
    if (c->d_bin) return CAMSYNC_OK;
    HIPCHK(c, dalloc(&c->d_bin, (size_t)c->maxB * c->maxH * c->maxW));
    return CAMSYNC_OK;
}

int ensure_planes(camsync_ctx* c, int nScales) {
    if (nScales <= c->planeScales) return CAMSYNC_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->buf.planes) (void)hipFree(c->buf.planes);
    if (c->buf.seedTable) (void)hipFree(c->buf.seedTable);
    c->buf.planes = nullptr;
    c->buf.seedTable = nullptr;
    c->planeScales = 0;
    HIPCHK(c, dalloc(&c->buf.planes, (size_t)c->maxB * nScales * plane_words(c->maxH, (c->maxW + 31) / 32)));
    HIPCHK(c, dalloc(&c->buf.seedTable, (size_t)c->maxB * nScales * seed_table_entries(c->maxH, c->maxW)));
    c->planeScales = nScales;
    return CAMSYNC_OK;
}

// Enqueue the detection pipeline for one chunk already resident at d_in.
int enqueue_detect(camsync_ctx* c, const uint8_t* d_in, int B, int H, int W, int C) {
    DetCfg cfg = make_cfg(c, H, W);
    int rcp = ensure_planes(c, cfg.nScales);
    if (rcp) return rcp;
    if (cfg.maxPerim > kMaxBorder)
        return fail(c, CAMSYNC_E_ARG, "maxMarkerPerimeterRate gives contours of up to %d points, the border follower handles %d",
                    cfg.maxPerim, kMaxBorder);
    const bool fused = gray_thresh_is_fused(cfg);
    if (!fused) {
        rcp = ensure_gray(c);
        if (rcp) return rcp;
    }
    DetBuffers b = c->buf;
    b.in = d_in;
    b.bin = nullptr;
    if (fused) {  // no gray image: K6 and K7 read the input frame (GraySrc)
        b.gray = nullptr;
        b.graySrc = d_in;
        b.graySrcC = C;
    } else {
        b.graySrc = b.gray;
        b.graySrcC = 1;
    }
    HIPCHK(c, hipMemsetAsync(b.counters, 0, sizeof(Counters), c->stream));
    HIPCHK(c, hipMemsetAsync(b.qcount, 0, sizeof(int) * B, c->stream));
    StageTimer t(c);
    launch_gray_thresh(cfg, b, B, C, c->stream);
    t.mark(0);
    debug_stage(c->stream, "gray_threshold");
    launch_contour_starts(cfg, b, B, c->stream);
    t.mark(1);
    debug_stage(c->stream, "contour_starts");
    launch_trace(cfg, b, c->stream);
    t.mark(2);
    debug_stage(c->stream, "trace");
    launch_quad_fit(cfg, b, B, c->stream);
    t.mark(3);
    debug_stage(c->stream, "quad_fit");
    launch_identify(cfg, b, B, c->stream);
    t.mark(4);
    debug_stage(c->stream, "identify");
    launch_select(cfg, b, B, c->stream);
    t.mark(5);
    debug_stage(c->stream, "select");
    launch_subpix(cfg, b, B, c->stream);
    t.mark(6);
    debug_stage(c->stream, "subpix");
    c->lastB = B;
    c->lastH = H;
    c->lastW = W;
    HIPCHK(c, hipGetLastError());
    return CAMSYNC_OK;
}

int collect_timers(camsync_ctx* c, int lastStage) {
    if (!c->timing) return CAMSYNC_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float total = 0;
    for (int s = 0; s <= lastStage; s++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[s], c->ev[s + 1]) == hipSuccess) {
            c->ms[s] += ms;
            c->launches[s]++;
            total += ms;
        }
    }
    c->ms[CAMSYNC_N_STAGES - 1] += total;
    c->launches[CAMSYNC_N_STAGES - 1]++;
    return CAMSYNC_OK;
}

// The scratch lists of the detection (start candidates, contour records, contour points, store / copy tasks, border
// seeds and segments) are sized per frame by rule of thumb at camsync_create; cv.aruco.detectMarkers has no such limits
// (aruco.py:256-264).  A chunk that overflows one of them is therefore run again with the lists concerned doubled, until
// it fits, the list's index range ends (2^31 entries; 2^25 seeds: ids are packed with marks above that) or the device
// has no memory left -- only then does the call fail with CAMSYNC_E_CAPACITY.  The per-frame lists of quad candidates
// (1024 slots at first) and of accepted markers (64) grow the same way, to what the fullest frame of the chunk needs
// (at least twice the size, at most kQuadCeiling: the selection keeps candidate indices in 16 bits).
constexpr unsigned int kGrowable = 1u | 2u | 4u | 8u | 32u | 64u | 512u;

template <typename T>
bool regrow(T*& ptr, size_t oldCount, size_t newCount) {
    T* fresh = nullptr;
    if (newCount <= oldCount || hipMalloc(reinterpret_cast<void**>(&fresh), newCount * sizeof(T)) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    (void)hipFree(ptr);
    ptr = fresh;
    return true;
}

// Doubles the lists named in `mask`; false when none of them could grow (then the overflow is final).
bool grow_lists(camsync_ctx* c, const Counters& h) {
    const unsigned int mask = h.overflow;
    if (c->allocContourCap) return false;  // capacities pinned by camsync_debug_limit_capacity (tests of the loud path)
    if (!(mask & kGrowable) || (mask & ~kGrowable & ~128u)) return false;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
    DetBuffers& b = c->buf;
    auto twice = [](unsigned int cap, unsigned int ceiling) { return (unsigned int)std::min<unsigned long long>(2ull * cap, ceiling); };
    bool grown = false;
    if (mask & 1u) {
        const unsigned int n = twice(b.candCap, 0x7FFFFFFFu);
        // (two lists of one capacity: both or neither)
        uint64_t* a = nullptr;
        if (n > b.candCap && hipMalloc(reinterpret_cast<void**>(&a), (size_t)n * sizeof(uint64_t)) == hipSuccess) {
            if (regrow(b.candLong, b.candCap, n)) {
                (void)hipFree(b.cand);
                b.cand = a;
                b.candCap = n;
                grown = true;
            } else {
                (void)hipFree(a);
            }
        }
        (void)hipGetLastError();
    }
    if (mask & 2u) {
        const unsigned int n = twice(b.contourCap, 0x7FFFFFFFu);
        uint32_t* idx = nullptr;
        if (n > b.contourCap && hipMalloc(reinterpret_cast<void**>(&idx), (size_t)n * 5 * sizeof(uint32_t)) == hipSuccess) {
            if (regrow(b.contours, b.contourCap, n)) {
                (void)hipFree(b.binIdx);
                b.binIdx = idx;
                b.contourCap = n;
                grown = true;
            } else {
                (void)hipFree(idx);
            }
        }
        (void)hipGetLastError();
    }
    if (mask & 4u) {
        const unsigned int n = twice(b.poolCap, 0x7FFFFFFFu);
        if (regrow(b.pool, b.poolCap, n)) { b.poolCap = n; grown = true; }
    }
    if (mask & 64u) {
        const unsigned int n = twice(b.taskCap, 0x7FFFFFFFu);
        CopyTask* cp = nullptr;
        if (n > b.taskCap && hipMalloc(reinterpret_cast<void**>(&cp), (size_t)n * sizeof(CopyTask)) == hipSuccess) {
            if (regrow(b.tasks, b.taskCap, n)) {
                (void)hipFree(b.copies);
                b.copies = cp;
                b.taskCap = b.copyCap = n;
                grown = true;
            } else {
                (void)hipFree(cp);
            }
        }
        (void)hipGetLastError();
    }
    if (mask & 512u) {
        const unsigned int nSeed = twice(b.seedCap, 0x01FFFFFFu), nSeg = (unsigned int)std::min<size_t>(2 * (size_t)nSeed, 0x03FFFFFFu);
        uint64_t* sd = nullptr;
        SegRec* sg = nullptr;
        uint32_t* ch = nullptr;
        if (nSeed > b.seedCap && hipMalloc(reinterpret_cast<void**>(&sd), (size_t)nSeed * sizeof(uint64_t)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&sg), (size_t)nSeg * sizeof(SegRec)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&ch), (size_t)nSeg * 32 * sizeof(uint32_t)) == hipSuccess) {
            (void)hipFree(b.seeds); (void)hipFree(b.seg); (void)hipFree(b.chain);
            b.seeds = sd; b.seg = sg; b.chain = ch;
            b.seedCap = nSeed;
            b.segCap = b.chainCap = nSeg;
            grown = true;
        } else {
            if (sd) (void)hipFree(sd);
            if (sg) (void)hipFree(sg);
            if (ch) (void)hipFree(ch);
        }
        (void)hipGetLastError();
    }
    const size_t B = (size_t)c->maxB;
    if (mask & 8u) {
        // per-frame quad candidates: the candidate records, their perspective maps, the identification's list, the
        // selection's state, the rejected candidates reported, and the selection's working set beyond the LDS size
        const int n = (int)std::min<unsigned long long>(std::max<unsigned long long>(2ull * b.quadCap, ((unsigned long long)h.needQuads + 63) & ~63ull), kQuadCeiling);
        QuadRec* q = nullptr; double* pm = nullptr; uint32_t* il = nullptr; int16_t* ss = nullptr; float* rj = nullptr; uint8_t* sw = nullptr;
        if (n > b.quadCap && hipMalloc(reinterpret_cast<void**>(&q), B * n * sizeof(QuadRec)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&pm), B * n * 10 * sizeof(double)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&il), B * n * sizeof(uint32_t)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&ss), B * 3 * n * sizeof(int16_t)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&rj), B * n * 8 * sizeof(float)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&sw), B * sel_work_bytes(n)) == hipSuccess) {
            (void)hipFree(b.quads); (void)hipFree(b.persp); (void)hipFree(b.idList); (void)hipFree(b.selState); (void)hipFree(b.outRejected);
            if (b.selWork) (void)hipFree(b.selWork);
            b.quads = q; b.persp = pm; b.idList = il; b.selState = ss; b.outRejected = rj; b.selWork = sw;
            b.quadCap = n;
            grown = true;
        } else {
            void* fresh[] = {q, pm, il, ss, rj, sw};
            for (void* p : fresh)
                if (p) (void)hipFree(p);
        }
        (void)hipGetLastError();
    }
    if (mask & 32u) {
        const int n = (int)std::min<unsigned long long>(std::max<unsigned long long>(2ull * b.markerOut, ((unsigned long long)h.needMarkers + 63) & ~63ull), (unsigned long long)b.quadCap);
        MarkerRec* mk = nullptr; int32_t* oi = nullptr; float* oc = nullptr; float* rc = nullptr;
        if (n > b.markerOut && hipMalloc(reinterpret_cast<void**>(&mk), B * n * sizeof(MarkerRec)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&oi), B * n * sizeof(int32_t)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&oc), B * n * 8 * sizeof(float)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&rc), B * n * 8 * sizeof(float)) == hipSuccess) {
            (void)hipFree(b.markers); (void)hipFree(b.outIds); (void)hipFree(b.outCorners); (void)hipFree(b.rawCorners);
            b.markers = mk; b.outIds = oi; b.outCorners = oc; b.rawCorners = rc;
            b.markerOut = n;
            b.markerCap = (unsigned)std::min<size_t>(B * n, 0x7FFFFFFFu);
            grown = true;
        } else {
            void* fresh[] = {mk, oi, oc, rc};
            for (void* p : fresh)
                if (p) (void)hipFree(p);
        }
        (void)hipGetLastError();
    }
    if (grown) c->regrown++;
    return grown;
}

int overflow_error(camsync_ctx* c, const Counters& h) {
    return fail(c, CAMSYNC_E_CAPACITY,
                "detection scratch overflow (mask 0x%x: 1 starts, 2 contours, 4 points, 8 quads, 32 markers, 64 store/copy tasks, "
                "128 bad contour record, 512 border seeds / segments); candidates %u contours %u points %u seeds %u segments %u copies %u; "
                "quad candidates of the fullest frame %u (slots %d), markers %u (slots %d)",
                h.overflow, h.nCand, h.nContours, h.nPool, h.nSeeds, h.nSeeds + h.nLong + h.nVirt, h.nCopy, h.needQuads, c->buf.quadCap,
                h.needMarkers, c->buf.markerOut);
}

// The chunk's counters after everything enqueued for it has run.  0: it fits; 1: a list overflowed and has been grown,
// run the chunk again; negative: error.
int verdict_of(camsync_ctx* c, const Counters& h) {
    if (!h.overflow) return 0;
    if (grow_lists(c, h)) return 1;
    return overflow_error(c, h);
}
int chunk_verdict(camsync_ctx* c) {
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, c->buf.counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return verdict_of(c, h);
}

int check_overflow(camsync_ctx* c) {
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, c->buf.counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h.overflow) return overflow_error(c, h);
    return CAMSYNC_OK;
}

int check_frame_args(camsync_ctx* c, const void* frames, int B, int H, int W, int C) {
    if (!c) return CAMSYNC_E_ARG;
    if (!frames || B <= 0) return fail(c, CAMSYNC_E_ARG, "frames is NULL or batch <= 0");
    if (C != 1 && C != 3) return fail(c, CAMSYNC_E_ARG, "channels must be 1 or 3, got %d", C);
    if (H > c->maxH || W > c->maxW || H < 32 || W < 32)
        return fail(c, CAMSYNC_E_ARG, "frame %dx%d outside the context's range (32..%d x 32..%d)", H, W, c->maxH, c->maxW);
    if (W % 8) return fail(c, CAMSYNC_E_ARG, "frame width must be a multiple of 8 (got %d)", W);
    if (H > 4095 || W > 4095) return fail(c, CAMSYNC_E_ARG, "frames larger than 4095 pixels a side are not supported");
    return CAMSYNC_OK;
}

// Points per frame of the anchor PnP: 4 per visible anchor; 64 (one point per lane) covers the reference's sheets v1, v3
// and v4, 128 / 256 (two / four per lane) a sheet of up to 32 / 64 anchors such as v2 (25 markers).
static int anchor_points(int n_anchor) { return n_anchor <= 16 ? 64 : n_anchor <= 32 ? 128 : 256; }

int ensure_pose_scratch(camsync_ctx* c, size_t B, int N) {
    if (B <= c->poseCap && N <= c->poseN) return CAMSYNC_OK;
    const size_t nB = std::max(B, c->poseCap);
    const int nN = std::max(N, c->poseN);
    (void)hipFree(c->d_obj); (void)hipFree(c->d_img); (void)hipFree(c->d_npts); (void)hipFree(c->d_K); (void)hipFree(c->d_dist);
    (void)hipFree(c->d_rvec); (void)hipFree(c->d_tvec); (void)hipFree(c->d_mae); (void)hipFree(c->d_rmse); (void)hipFree(c->d_ok);
    c->d_obj = c->d_img = nullptr;
    c->poseCap = 0;
    HIPCHK(c, dalloc(&c->d_obj, nB * nN * 3));
    HIPCHK(c, dalloc(&c->d_img, nB * nN * 2));
    HIPCHK(c, dalloc(&c->d_npts, nB));
    HIPCHK(c, dalloc(&c->d_K, nB * 9));
    HIPCHK(c, dalloc(&c->d_dist, nB * 8));
    HIPCHK(c, dalloc(&c->d_rvec, nB * 3));
    HIPCHK(c, dalloc(&c->d_tvec, nB * 3));
    HIPCHK(c, dalloc(&c->d_mae, nB));
    HIPCHK(c, dalloc(&c->d_rmse, nB));
    HIPCHK(c, dalloc(&c->d_ok, nB));
    c->poseCap = nB;
    c->poseN = nN;
    return CAMSYNC_OK;
}

}  // namespace

extern "C" {

int camsync_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return CAMSYNC_E_NODEVICE;
    return n;
}

void camsync_default_params(camsync_params* p) {
    p->adaptiveThreshWinSizeMin = 3;
    p->adaptiveThreshWinSizeMax = 23;
    p->adaptiveThreshWinSizeStep = 10;
    p->adaptiveThreshConstant = 7;
    p->minMarkerPerimeterRate = 0.03;
    p->maxMarkerPerimeterRate = 4.0;
    p->polygonalApproxAccuracyRate = 0.03;
    p->minCornerDistanceRate = 0.05;
    p->minDistanceToBorder = 3;
    p->markerBorderBits = 1;
    p->minMarkerDistanceRate = 0.125;
    p->minGroupDistance = 0.21f;
    p->cornerRefinementMethod = 1;
    p->cornerRefinementWinSize = 20;
    p->relativeCornerRefinmentWinSize = 0.3f;
    p->cornerRefinementMaxIterations = 100;
    p->cornerRefinementMinAccuracy = 0.01;
    p->perspectiveRemovePixelPerCell = 4;
    p->perspectiveRemoveIgnoredMarginPerCell = 0.13;
    p->maxErroneousBitsInBorderRate = 0.35;
    p->minOtsuStdDev = 5.0;
    p->errorCorrectionRate = 0.6;
}

int camsync_create(camsync_ctx** out, int device, int max_h, int max_w, int max_batch) {
    if (!out || max_h < 32 || max_w < 32 || max_batch < 1) return CAMSYNC_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CAMSYNC_E_NODEVICE;
    camsync_ctx* c = new camsync_ctx();
    *out = c;
    c->device = device;
    c->maxH = max_h;
    c->maxW = max_w;
    c->maxB = max_batch;
    camsync_default_params(&c->params);
    HIPCHK(c, hipSetDevice(device));
    {
        hipError_t e = hipSuccess;
        c->naScratch = scratch_for_device(device, &e);
        if (!c->naScratch) return fail(c, CAMSYNC_E_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
    }
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const size_t px = (size_t)max_h * max_w, B = (size_t)max_batch;
    DetBuffers& b = c->buf;
    // the host-input staging buffer, the gray image and the byte-per-pixel threshold image are allocated on first use
    b.gray = nullptr;
    b.bin = nullptr;
    c->planeScales = 3;
    HIPCHK(c, dalloc(&b.planes, B * c->planeScales * plane_words(max_h, (max_w + 31) / 32)));
    b.candCap = (unsigned)std::min<size_t>(std::max<size_t>(B * (px / 32), 65536), 0x7FFFFFFFu);
    HIPCHK(c, dalloc(&b.cand, b.candCap));
    HIPCHK(c, dalloc(&b.candLong, b.candCap));
    // (+ slack for the blocks the persistent waves of the short-border and resolve passes reserve and may leave partly
    // unused: kSlackWaves bounds the waves of one launch)
    constexpr size_t kSlackWaves = 16384;
    b.contourCap = (unsigned)std::min<size_t>(B * 4096 + kSlackWaves * 32, 0x7FFFFFFFu);
    HIPCHK(c, dalloc(&b.contours, b.contourCap));
    HIPCHK(c, dalloc(&b.binIdx, (size_t)b.contourCap * 5));
    b.poolCap = (unsigned)std::min<size_t>(B * std::max<size_t>(262144, px / 8) + kSlackWaves * 8192, 0x7FFFFFFFu);
    HIPCHK(c, dalloc(&b.pool, b.poolCap));
    b.taskCap = (unsigned)std::min<size_t>((size_t)b.contourCap * 2 + b.poolCap / kCkStride + kSlackWaves * 64, 0x7FFFFFFFu);
    HIPCHK(c, dalloc(&b.tasks, b.taskCap));
    // long borders in segments (internal.h, SegRec): seeds, one record and one 128-byte chain chunk per segment, the
    // seed -> id table (one entry per pixel of the grid lines of every bit plane).  Measured: ~1000 seeds and ~1500 records per 1080p frame of the
    // bench scene, four times that on the 4K scene of configs[2], 1700 seeds per 1280x720 frame of the reference's
    // photo pic_2.png (px / 540: textured scenes are what the budget of px / 256 is for).
    b.seedCap = (unsigned)std::min<size_t>(B * std::max<size_t>(2048, px / 256) + 65536, 0x01FFFFFFu);
    HIPCHK(c, dalloc(&b.seeds, b.seedCap));
    b.segCap = (unsigned)std::min<size_t>(2 * (size_t)b.seedCap, 0x03FFFFFFu);
    HIPCHK(c, dalloc(&b.seg, b.segCap));
    HIPCHK(c, dalloc(&b.seedTable, B * c->planeScales * seed_table_entries(max_h, max_w)));  // (reallocated with the planes, ensure_planes)
    b.chainCap = b.segCap;
    HIPCHK(c, dalloc(&b.chain, (size_t)b.chainCap * 32));
    b.copyCap = b.taskCap;
    HIPCHK(c, dalloc(&b.copies, b.copyCap));
    b.resolveWaves = (unsigned)std::min<size_t>(std::max<size_t>(B * 8, 64), 256 * 32);
    HIPCHK(c, dalloc(&b.resolveLog, (size_t)b.resolveWaves * kResolveLogHops * 64));
    b.quadCap = kQuadCap0;
    b.markerOut = kMarkerOut0;
    b.selWork = nullptr;  // (allocated with the lists once a frame has more than kQuadLds candidates)
    HIPCHK(c, dalloc(&b.quads, B * b.quadCap));
    HIPCHK(c, dalloc(&b.persp, B * b.quadCap * 10));
    HIPCHK(c, dalloc(&b.idList, B * b.quadCap));
    HIPCHK(c, dalloc(&b.idCount, B * 2));
    HIPCHK(c, dalloc(&b.selState, B * 3 * b.quadCap));
    b.identifyAll = 0;
    HIPCHK(c, dalloc(&b.qcount, B));
    HIPCHK(c, dalloc(&b.counters, 1));
    b.markerCap = (unsigned)(B * b.markerOut);
    HIPCHK(c, dalloc(&b.markers, b.markerCap));
    HIPCHK(c, dalloc(&b.outIds, B * b.markerOut));
    HIPCHK(c, dalloc(&b.outCorners, B * b.markerOut * 8));
    HIPCHK(c, dalloc(&b.rawCorners, B * b.markerOut * 8));
    HIPCHK(c, dalloc(&b.outCounts, B));
    HIPCHK(c, dalloc(&b.outRejected, B * b.quadCap * 8));
    HIPCHK(c, dalloc(&b.outRejCounts, B));
    HIPCHK(c, dalloc(&c->d_dict, 4 * 1024));  // the codes under the four rotations
    HIPCHK(c, dalloc(&c->d_expTab, 21 * 41));
    HIPCHK(c, dalloc(&c->d_anchorIds, kMaxAnchors));
    HIPCHK(c, dalloc(&c->d_anchorObj, kMaxAnchors * 12));
    b.dict = c->d_dict;
    b.expTab = c->d_expTab;
    // cornerSubPix separable weights exp(-((i-win)/win)^2), float arithmetic as in OpenCV
    std::vector<float> tab(21 * 41, 0.f);
    for (int win = 1; win <= 20; win++)
        for (int i = 0; i <= 2 * win; i++) {
            float y = (float)(i - win) / win;
            tab[win * 41 + i] = expf(-y * y);
        }
    HIPCHK(c, hipMemcpy(c->d_expTab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    for (int i = 0; i <= CAMSYNC_N_STAGES; i++) HIPCHK(c, hipEventCreate(&c->ev[i]));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pin), sizeof(double) * 256, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_report), sizeof(double) * 16, hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->h_report, 0, sizeof(double) * 16);
    return CAMSYNC_OK;
}

int camsync_host_alloc(void** out, size_t bytes) {
    if (!out || !bytes) return CAMSYNC_E_ARG;
    *out = nullptr;
    // portable: page-locked for every GPU of the process, not only the current one
    return hipHostMalloc(out, bytes, hipHostMallocPortable) == hipSuccess ? CAMSYNC_OK : CAMSYNC_E_HIP;
}

int camsync_host_free(void* p) {
    if (!p) return CAMSYNC_OK;
    return hipHostFree(p) == hipSuccess ? CAMSYNC_OK : CAMSYNC_E_HIP;
}

void camsync_destroy(camsync_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    DetBuffers& b = c->buf;
    void* ptrs[] = {c->d_chessGray, c->d_chessS, c->d_chessR, c->d_chessCand, c->d_chessMax, c->d_chessCount, c->d_spIn, c->d_spOut, c->d_spN,
                    c->d_umap, c->d_uout, c->d_in, c->d_bin, b.planes, b.gray, b.cand, b.candLong, b.contours, b.binIdx, b.pool, b.tasks, b.seeds, b.seg, b.seedTable, b.chain, b.copies, b.resolveLog, b.quads, b.persp, b.idList, b.idCount, b.selState, b.selWork, b.qcount, b.counters, b.markers,
                    b.outIds, b.outCorners, b.rawCorners, b.outCounts, b.outRejected, b.outRejCounts, c->d_dict, c->d_expTab,
                    c->d_obj, c->d_img, c->d_npts, c->d_K, c->d_dist, c->d_rvec, c->d_tvec, c->d_mae, c->d_rmse, c->d_ok,
                    c->d_anchorIds, c->d_anchorObj, c->d_fCorners, c->d_fSize, c->d_fK, c->d_fDist, c->d_fRvec, c->d_fTvec, c->d_fFused, c->d_fSizes, c->d_fValid, c->d_fOk, c->d_fLocated, c->d_fIds, c->d_fSeen, c->d_calObj, c->d_calImg, c->cal.intr, c->cal.ext, c->cal.intrTrial,
                    c->cal.extTrial, c->cal.viewBlocks, c->cal.viewBlocksTrial, c->cal.partial, c->cal.ticket,
                    c->d_calWork, c->d_focalRows, c->d_calOk, c->d_calK, c->d_calNpts};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    jpeg_destroy(c->jpeg);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_report) (void)hipHostFree(c->h_report);
    if (c->upPinned) {
        (void)hipHostFree(c->upPinned);
        for (hipEvent_t e : c->upEv)
            if (e) (void)hipEventDestroy(e);
    }
    if (c->d_pack) (void)hipFree(c->d_pack);
    void* hostPtrs[] = {c->h_pack, c->h_rejCounts, c->h_rej};
    for (void* p : hostPtrs)
        if (p) (void)hipHostFree(p);
    for (int i = 0; i <= CAMSYNC_N_STAGES; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* camsync_last_error(const camsync_ctx* c) { return c ? c->err.c_str() : "null context"; }

int camsync_set_params(camsync_ctx* c, const camsync_params* p) {
    if (!c || !p) return CAMSYNC_E_ARG;
    int rc = check_params(c, p);
    if (rc) return rc;
    c->params = *p;
    return CAMSYNC_OK;
}

int camsync_get_params(const camsync_ctx* c, camsync_params* p) {
    if (!c || !p) return CAMSYNC_E_ARG;
    *p = c->params;
    return CAMSYNC_OK;
}

int camsync_set_dictionary(camsync_ctx* c, const uint16_t* codes, int n, int marker_size, int max_correction_bits) {
    if (!c) return CAMSYNC_E_ARG;
    if (!codes || n < 1 || n > 1024) return fail(c, CAMSYNC_E_ARG, "dictionary must hold 1..1024 codes");
    if (marker_size < 1 || marker_size > 4) return fail(c, CAMSYNC_E_ARG, "marker_size must be 1..4 (16-bit codes)");
    HIPCHK(c, hipSetDevice(c->device));
    // the dictionary search compares with every entry under its four rotations: laid out once here, [rotation][entry]
    std::vector<uint16_t> rot((size_t)4 * n);
    for (int r = 0; r < 4; r++)
        for (int i = 0; i < n; i++) rot[(size_t)r * n + i] = (uint16_t)rotate_marker_code(codes[i], marker_size, r);
    HIPCHK(c, hipMemcpy(c->d_dict, rot.data(), sizeof(uint16_t) * rot.size(), hipMemcpyHostToDevice));
    const int oldSize = c->markerSize;
    c->markerSize = marker_size;
    int rc = check_params(c, &c->params);
    if (rc) { c->markerSize = oldSize; return rc; }
    c->dictN = n;
    c->maxCorrectionBits = max_correction_bits;
    return CAMSYNC_OK;
}

int camsync_gray_threshold(camsync_ctx* c, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                           uint8_t* gray_out, uint8_t* bin_out) {
    int rc = check_frame_args(c, frames, B, H, W, C);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t px = (size_t)H * W;
    const hipMemcpyKind outKind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    for (int b0 = 0; b0 < B; b0 += c->maxB) {
        const int nb = std::min(c->maxB, B - b0);
        const uint8_t* src = frames + (size_t)b0 * px * C;
        const uint8_t* d_in = src;
        if (!on_device) {
            int rcs = ensure_staging(c);
            if (rcs) return rcs;
            HIPCHK(c, hipMemcpyAsync(c->d_in, src, (size_t)nb * px * C, hipMemcpyHostToDevice, c->stream));
            d_in = c->d_in;
        } else if (((uintptr_t)src) % 8) {
            return fail(c, CAMSYNC_E_ARG, "device frames must be 8-byte aligned");
        }
        DetCfg cfg = make_cfg(c, H, W);
        rc = ensure_planes(c, cfg.nScales);
        if (rc) return rc;
        rc = ensure_gray(c);
        if (rc) return rc;
        if (bin_out) {
            rc = ensure_bin(c);
            if (rc) return rc;
        }
        DetBuffers b = c->buf;
        b.in = d_in;
        b.bin = bin_out ? c->d_bin : nullptr;
        StageTimer t(c);
        launch_gray_thresh(cfg, b, nb, C, c->stream);
        t.mark(0);
        HIPCHK(c, hipGetLastError());
        if (gray_out) HIPCHK(c, hipMemcpyAsync(gray_out + (size_t)b0 * px, b.gray, (size_t)nb * px, outKind, c->stream));
        if (bin_out) HIPCHK(c, hipMemcpyAsync(bin_out + (size_t)b0 * px, b.bin, (size_t)nb * px, outKind, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        rc = collect_timers(c, 0);
        if (rc) return rc;
    }
    return CAMSYNC_OK;
}

// `frames` is one contiguous batch, or (frame_ptrs != null) B separate frames, e.g. one per camera
static int detect_impl(camsync_ctx* c, const uint8_t* frames, const uint8_t* const* frame_ptrs, int on_device, int B, int H, int W, int C,
                       int max_markers, int32_t* ids, float* corners, int32_t* counts, int max_rejected, float* rejected,
                       int32_t* rej_counts, const int32_t* anchor_ids, const float* anchor_obj, int n_anchor,
                       const double* K, const double* dist, int ndist, int per_frame_cam, double* rvec, double* tvec,
                       uint8_t* pose_ok, double* mae, double* rmse, bool withPose) {
    int rc = check_frame_args(c, frame_ptrs ? (const void*)frame_ptrs : (const void*)frames, B, H, W, C);
    if (rc) return rc;
    if (frame_ptrs)
        for (int i = 0; i < B; i++)
            if (!frame_ptrs[i]) return fail(c, CAMSYNC_E_ARG, "frame %d is NULL", i);
    if (c->dictN <= 0) return fail(c, CAMSYNC_E_ARG, "no dictionary set (camsync_set_dictionary)");
    if (max_markers < 1 || !ids || !corners || !counts) return fail(c, CAMSYNC_E_ARG, "detection outputs missing");
    if (withPose) {
        if (n_anchor < 1 || n_anchor > kMaxAnchors || !anchor_ids || !anchor_obj)
            return fail(c, CAMSYNC_E_ARG, "1..64 anchors required");
        if (!K || ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return fail(c, CAMSYNC_E_ARG, "camera matrix / distortion missing");
        if (!rvec || !tvec || !pose_ok) return fail(c, CAMSYNC_E_ARG, "pose outputs missing");
    }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t px = (size_t)H * W;
    c->lastMarkers = c->lastRejected = 0;
    if (withPose) {
        rc = ensure_pose_scratch(c, (size_t)c->maxB, anchor_points(n_anchor));
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->d_anchorIds, anchor_ids, sizeof(int32_t) * n_anchor, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_anchorObj, anchor_obj, sizeof(float) * n_anchor * 12, hipMemcpyHostToDevice, c->stream));
    }
    c->poseFrames = 0;
    for (int b0 = 0; b0 < B; b0 += c->maxB) {
        const int nb = std::min(c->maxB, B - b0);
        // (per chunk: the per-frame slots, and the pinned buffers with them, grow when a chunk overflows them)
        rc = ensure_host_out(c, rejected != nullptr);
        if (rc) return rc;
        const size_t mo = (size_t)c->buf.markerOut, qc = (size_t)c->buf.quadCap;
        const PackLayout L((size_t)nb, mo);
        const int32_t *const hIds = reinterpret_cast<const int32_t*>(c->h_pack + L.ids), *const hCounts = reinterpret_cast<const int32_t*>(c->h_pack + L.counts);
        const float* const hCorners = reinterpret_cast<const float*>(c->h_pack + L.corners);
        int32_t* const hRejCounts = c->h_rejCounts;
        float* const hRej = c->h_rej;
        const uint8_t* src = frame_ptrs ? nullptr : frames + (size_t)b0 * px * C;
        const uint8_t* d_in = src;
        if (frame_ptrs) {  // separate frames (host or device) are gathered into the staging buffer
            int rcs = ensure_staging(c);
            if (rcs) return rcs;
            if (on_device) {
                for (int i = 0; i < nb; i++)
                    HIPCHK(c, hipMemcpyAsync(c->d_in + (size_t)i * px * C, frame_ptrs[b0 + i], px * C, hipMemcpyDeviceToDevice, c->stream));
            } else {
                std::vector<UpSeg> segs((size_t)nb);
                for (int i = 0; i < nb; i++) segs[(size_t)i] = UpSeg{c->d_in + (size_t)i * px * C, frame_ptrs[b0 + i], px * C};
                rcs = upload_host(c, segs.data(), nb);
                if (rcs) return rcs;
            }
            d_in = c->d_in;
        } else if (!on_device) {
            int rcs = ensure_staging(c);
            if (rcs) return rcs;
            const UpSeg seg{c->d_in, src, (size_t)nb * px * C};
            rcs = upload_host(c, &seg, 1);
            if (rcs) return rcs;
            d_in = c->d_in;
        } else if (((uintptr_t)src) % 8) {
            return fail(c, CAMSYNC_E_ARG, "device frames must be 8-byte aligned");
        }
        rc = enqueue_detect(c, d_in, nb, H, W, C);
        if (rc) return rc;
        int lastStage = 6;
        if (withPose) {
            const size_t kOff = per_frame_cam ? (size_t)b0 : 0, kCnt = per_frame_cam ? (size_t)nb : 1;
            HIPCHK(c, hipMemcpyAsync(c->d_K, K + kOff * 9, sizeof(double) * 9 * kCnt, hipMemcpyHostToDevice, c->stream));
            if (ndist) HIPCHK(c, hipMemcpyAsync(c->d_dist, dist + kOff * ndist, sizeof(double) * ndist * kCnt, hipMemcpyHostToDevice, c->stream));
            launch_gather_anchors(c->buf.outIds, c->buf.outCorners, c->buf.outCounts, nb, c->buf.markerOut, c->d_anchorIds, c->d_anchorObj,
                                  n_anchor, anchor_points(n_anchor), c->d_obj, c->d_img, c->d_npts, c->stream);
            PnpArgs a{};
            a.obj = c->d_obj; a.img = c->d_img; a.npts = c->d_npts; a.B = nb; a.N = anchor_points(n_anchor);
            a.K = c->d_K; a.dist = c->d_dist; a.ndist = ndist; a.perFrameCam = per_frame_cam; a.refine = 1;
            a.rvec = c->d_rvec; a.tvec = c->d_tvec; a.ok = c->d_ok; a.mae = c->d_mae; a.rmse = c->d_rmse;
            launch_pnp(a, c->stream);
            if (c->timing) (void)hipEventRecord(c->ev[8], c->stream);
            lastStage = 7;
        }
        {
            PackSegs ps{};
            auto seg = [&](const void* src, size_t off, size_t bytes) {
                ps.src[ps.n] = src; ps.off[ps.n] = off; ps.bytes[ps.n] = bytes; ps.n++;
            };
            seg(c->buf.counters, L.counters, sizeof(Counters));
            seg(c->buf.outCounts, L.counts, (size_t)nb * 4);
            seg(c->buf.outIds, L.ids, (size_t)nb * mo * 4);
            seg(c->buf.outCorners, L.corners, (size_t)nb * mo * 32);
            if (withPose) {
                seg(c->d_ok, L.ok, (size_t)nb);
                seg(c->d_rvec, L.rvec, (size_t)nb * 24);
                seg(c->d_tvec, L.tvec, (size_t)nb * 24);
                seg(c->d_mae, L.mae, (size_t)nb * 8);
                seg(c->d_rmse, L.rmse, (size_t)nb * 8);
            }
            const unsigned blocks = (unsigned)std::min<size_t>(1024, (L.total / 4 + 255) / 256);
            hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, c->stream, ps, c->d_pack);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(c->h_pack, c->d_pack, L.total, hipMemcpyDeviceToHost, c->stream));
        }
        if (rejected) {
            HIPCHK(c, hipMemcpyAsync(hRej, c->buf.outRejected, sizeof(float) * nb * qc * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(hRejCounts, c->buf.outRejCounts, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        {
            Counters hc;
            memcpy(&hc, c->h_pack + L.counters, sizeof hc);
            rc = verdict_of(c, hc);
        }
        if (rc == 1) {  // a scratch list overflowed and has been doubled: this chunk again
            b0 -= c->maxB;
            continue;
        }
        if (rc) return rc;
        rc = collect_timers(c, lastStage);
        if (rc) return rc;
        if (withPose) {
            memcpy(rvec + (size_t)b0 * 3, c->h_pack + L.rvec, sizeof(double) * 3 * nb);
            memcpy(tvec + (size_t)b0 * 3, c->h_pack + L.tvec, sizeof(double) * 3 * nb);
            memcpy(pose_ok + b0, c->h_pack + L.ok, (size_t)nb);
            if (mae) memcpy(mae + b0, c->h_pack + L.mae, sizeof(double) * nb);
            if (rmse) memcpy(rmse + b0, c->h_pack + L.rmse, sizeof(double) * nb);
        }
        for (int f = 0; f < nb; f++) {
            const int n = std::min(hCounts[f], max_markers);
            counts[b0 + f] = n;
            c->lastMarkers = std::max(c->lastMarkers, (int)hCounts[f]);
            for (int m = 0; m < n; m++) {
                ids[(size_t)(b0 + f) * max_markers + m] = hIds[(size_t)f * mo + m];
                memcpy(corners + ((size_t)(b0 + f) * max_markers + m) * 8, &hCorners[((size_t)f * mo + m) * 8], 8 * sizeof(float));
            }
            if (rejected && rej_counts) {
                const int nr = std::min(hRejCounts[f], max_rejected);
                rej_counts[b0 + f] = nr;
                c->lastRejected = std::max(c->lastRejected, (int)hRejCounts[f]);
                for (int m = 0; m < nr; m++)
                    memcpy(rejected + ((size_t)(b0 + f) * max_rejected + m) * 8, &hRej[((size_t)f * qc + m) * 8], 8 * sizeof(float));
            }
        }
    }
    if (withPose && B <= c->maxB) {
        c->poseFrames = B;
        c->poseNdist = ndist;
        c->posePerFrameCam = per_frame_cam;
    }
    return CAMSYNC_OK;
}

int camsync_detect_batch(camsync_ctx* c, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                         int max_markers, int32_t* ids, float* corners, int32_t* counts, int max_rejected,
                         float* rejected, int32_t* rej_counts) {
    return detect_impl(c, frames, nullptr, on_device, B, H, W, C, max_markers, ids, corners, counts, max_rejected, rejected,
                       rej_counts, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, false);
}

int camsync_detect_pnp_batch(camsync_ctx* c, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                             const int32_t* anchor_ids, const float* anchor_obj, int n_anchor, const double* K,
                             const double* dist, int ndist, int per_frame_cam, int max_markers, int32_t* ids,
                             float* corners, int32_t* counts, double* rvec, double* tvec, uint8_t* pose_ok, double* mae,
                             double* rmse) {
    return detect_impl(c, frames, nullptr, on_device, B, H, W, C, max_markers, ids, corners, counts, 0, nullptr, nullptr,
                       anchor_ids, anchor_obj, n_anchor, K, dist, ndist, per_frame_cam, rvec, tvec, pose_ok, mae, rmse, true);
}

int camsync_detect_pnp_frames(camsync_ctx* c, const uint8_t* const* frame_ptrs, int on_device, int B, int H, int W, int C,
                              const int32_t* anchor_ids, const float* anchor_obj, int n_anchor, const double* K,
                              const double* dist, int ndist, int per_frame_cam, int max_markers, int32_t* ids,
                              float* corners, int32_t* counts, double* rvec, double* tvec, uint8_t* pose_ok, double* mae,
                              double* rmse) {
    if (!c) return CAMSYNC_E_ARG;
    if (!frame_ptrs) return fail(c, CAMSYNC_E_ARG, "frame_ptrs is NULL");
    return detect_impl(c, nullptr, frame_ptrs, on_device, B, H, W, C, max_markers, ids, corners, counts, 0, nullptr, nullptr,
                       anchor_ids, anchor_obj, n_anchor, K, dist, ndist, per_frame_cam, rvec, tvec, pose_ok, mae, rmse, true);
}

int camsync_fuse_moving_markers(camsync_ctx* c, int n_frames, const int32_t* moving_ids, const double* moving_sizes, int n_moving,
                                double* fused_corners, int32_t* seen, double* marker_rvec, double* marker_tvec, uint8_t* located) {
    if (!c) return CAMSYNC_E_ARG;
    if (n_frames < 1 || n_frames != c->poseFrames)
        return fail(c, CAMSYNC_E_ARG, "camsync_fuse_moving_markers works on the frames of the camsync_detect_pnp_batch / _frames call just "
                    "before it on this context (that call had %d frames in one chunk, this one names %d)", c->poseFrames, n_frames);
    if (n_moving < 1 || n_moving > 1024 || !moving_ids || !moving_sizes || !fused_corners || !seen)
        return fail(c, CAMSYNC_E_ARG, "1..1024 moving markers with ids and sizes, and the fused_corners / seen outputs, are required");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t E = (size_t)n_frames * n_moving, M = (size_t)n_moving;
    if (E > c->fuseCap || M > c->fuseCapM) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        void* old[] = {c->d_fCorners, c->d_fSize, c->d_fK, c->d_fDist, c->d_fRvec, c->d_fTvec, c->d_fFused, c->d_fSizes, c->d_fValid, c->d_fOk,
                       c->d_fLocated, c->d_fIds, c->d_fSeen};
        for (void* p : old)
            if (p) (void)hipFree(p);
        c->fuseCap = c->fuseCapM = 0;
        c->d_fCorners = nullptr; c->d_fSize = c->d_fK = c->d_fDist = c->d_fRvec = c->d_fTvec = c->d_fFused = c->d_fSizes = nullptr;
        c->d_fValid = c->d_fOk = c->d_fLocated = nullptr; c->d_fIds = c->d_fSeen = nullptr;
        const size_t nE = std::max<size_t>(E, 64), nM = std::max<size_t>(M, 16);
        HIPCHK(c, dalloc(&c->d_fCorners, nE * 8));
        HIPCHK(c, dalloc(&c->d_fSize, nE));
        HIPCHK(c, dalloc(&c->d_fK, nE * 9));
        HIPCHK(c, dalloc(&c->d_fDist, nE * 8));
        HIPCHK(c, dalloc(&c->d_fRvec, nE * 3));
        HIPCHK(c, dalloc(&c->d_fTvec, nE * 3));
        HIPCHK(c, dalloc(&c->d_fValid, nE));
        HIPCHK(c, dalloc(&c->d_fOk, nE));
        HIPCHK(c, dalloc(&c->d_fLocated, nE));
        HIPCHK(c, dalloc(&c->d_fFused, nM * 12));
        HIPCHK(c, dalloc(&c->d_fSizes, nM));
        HIPCHK(c, dalloc(&c->d_fIds, nM));
        HIPCHK(c, dalloc(&c->d_fSeen, nM));
        c->fuseCap = nE;
        c->fuseCapM = nM;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_fIds, moving_ids, sizeof(int32_t) * M, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_fSizes, moving_sizes, sizeof(double) * M, hipMemcpyHostToDevice, c->stream));
    launch_fuse_moving(c->buf.outIds, c->buf.outCorners, c->buf.outCounts, c->buf.markerOut, c->d_ok, c->d_rvec, c->d_tvec, n_frames, c->d_fIds, c->d_fSizes,
                       n_moving, c->d_K, c->d_dist, c->poseNdist, c->posePerFrameCam, c->d_fCorners, c->d_fSize, c->d_fK, c->d_fDist,
                       c->d_fValid, c->d_fRvec, c->d_fTvec, c->d_fOk, c->d_fFused, c->d_fSeen, c->d_fLocated, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(fused_corners, c->d_fFused, sizeof(double) * M * 12, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(seen, c->d_fSeen, sizeof(int32_t) * M, hipMemcpyDeviceToHost, c->stream));
    if (marker_rvec) HIPCHK(c, hipMemcpyAsync(marker_rvec, c->d_fRvec, sizeof(double) * E * 3, hipMemcpyDeviceToHost, c->stream));
    if (marker_tvec) HIPCHK(c, hipMemcpyAsync(marker_tvec, c->d_fTvec, sizeof(double) * E * 3, hipMemcpyDeviceToHost, c->stream));
    if (located) HIPCHK(c, hipMemcpyAsync(located, c->d_fLocated, E, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

int camsync_detect_pnp_resident(camsync_ctx* c, const uint8_t* d_frames, int B, int H, int W, int C,
                                const int32_t* anchor_ids, const float* anchor_obj, int n_anchor, const double* K,
                                const double* dist, int ndist, int max_markers, int32_t* d_ids, float* d_corners,
                                int32_t* d_counts, double* d_rvec, double* d_tvec, uint8_t* d_pose_ok) {
    int rc = check_frame_args(c, d_frames, B, H, W, C);
    if (rc) return rc;
    if (c->dictN <= 0) return fail(c, CAMSYNC_E_ARG, "no dictionary set (camsync_set_dictionary)");
    if (n_anchor < 1 || n_anchor > kMaxAnchors || !anchor_ids || !anchor_obj) return fail(c, CAMSYNC_E_ARG, "1..64 anchors required");
    if (!K || ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return fail(c, CAMSYNC_E_ARG, "camera matrix / distortion missing");
    if (max_markers < 1 || !d_ids || !d_corners || !d_counts || !d_rvec || !d_tvec || !d_pose_ok)
        return fail(c, CAMSYNC_E_ARG, "output pointers missing");
    if (((uintptr_t)d_frames) % 8) return fail(c, CAMSYNC_E_ARG, "device frames must be 8-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    rc = ensure_pose_scratch(c, (size_t)c->maxB, anchor_points(n_anchor));
    if (rc) return rc;
    // anchors and camera are host pointers (small); frames and outputs live on the GPU
    HIPCHK(c, hipMemcpyAsync(c->d_anchorIds, anchor_ids, sizeof(int32_t) * n_anchor, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_anchorObj, anchor_obj, sizeof(float) * n_anchor * 12, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_K, K, sizeof(double) * 9, hipMemcpyHostToDevice, c->stream));
    if (ndist) HIPCHK(c, hipMemcpyAsync(c->d_dist, dist, sizeof(double) * ndist, hipMemcpyHostToDevice, c->stream));
    const size_t px = (size_t)H * W;
    for (int b0 = 0; b0 < B; b0 += c->maxB) {
        const int nb = std::min(c->maxB, B - b0);
        rc = enqueue_detect(c, d_frames + (size_t)b0 * px * C, nb, H, W, C);
        if (rc) return rc;
        launch_gather_anchors(c->buf.outIds, c->buf.outCorners, c->buf.outCounts, nb, c->buf.markerOut, c->d_anchorIds, c->d_anchorObj,
                              n_anchor, anchor_points(n_anchor), c->d_obj, c->d_img, c->d_npts, c->stream);
        PnpArgs a{};
        a.obj = c->d_obj; a.img = c->d_img; a.npts = c->d_npts; a.B = nb; a.N = anchor_points(n_anchor);
        a.K = c->d_K; a.dist = c->d_dist; a.ndist = ndist; a.perFrameCam = 0; a.refine = 1;
        a.rvec = d_rvec + (size_t)b0 * 3; a.tvec = d_tvec + (size_t)b0 * 3; a.ok = d_pose_ok + b0; a.mae = nullptr; a.rmse = nullptr;
        debug_stage(c->stream, "gather_anchors");
        launch_pnp(a, c->stream);
        debug_stage(c->stream, "pnp");
        if (c->timing) (void)hipEventRecord(c->ev[8], c->stream);
        // ids, corners and counts (clamped to max_markers) in the caller's layout
        launch_export_markers(c->buf, nb, max_markers, d_ids + (size_t)b0 * max_markers, d_corners + (size_t)b0 * max_markers * 8,
                              d_counts + b0, c->stream);
        // the scratch lists are reused by the next chunk: the stream orders that; overflow is checked per chunk
        rc = chunk_verdict(c);
        if (rc == 1) {  // a scratch list overflowed and has been doubled: this chunk again
            b0 -= c->maxB;
            continue;
        }
        if (rc) return rc;
        rc = collect_timers(c, 7);
        if (rc) return rc;
    }
    return CAMSYNC_OK;
}

int camsync_last_extent(const camsync_ctx* c, int32_t* markers, int32_t* rejected) {
    if (!c) return CAMSYNC_E_ARG;
    if (markers) *markers = c->lastMarkers;
    if (rejected) *rejected = c->lastRejected;
    return CAMSYNC_OK;
}

int camsync_debug_limit_capacity(camsync_ctx* c, unsigned int contours, unsigned int points, unsigned int tasks) {
    if (!c) return CAMSYNC_E_ARG;
    DetBuffers& b = c->buf;
    if (!c->allocContourCap) {
        c->allocContourCap = b.contourCap;
        c->allocPoolCap = b.poolCap;
        c->allocTaskCap = b.taskCap;
    }
    if (contours) b.contourCap = std::min(contours, c->allocContourCap);
    if (points) b.poolCap = std::min(points, c->allocPoolCap);
    if (tasks) {
        b.taskCap = std::min(tasks, c->allocTaskCap);
        b.copyCap = b.taskCap;
    }
    return CAMSYNC_OK;
}

int camsync_debug_shrink_lists(camsync_ctx* c, unsigned int starts, unsigned int contours, unsigned int points, unsigned int tasks,
                               unsigned int seeds) {
    if (!c || c->allocContourCap) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    DetBuffers& b = c->buf;
    // (smaller buffers, really: an overflow must find the lists' ends where the capacities say they are)
    if (starts && starts < b.candCap) {
        (void)hipFree(b.cand); (void)hipFree(b.candLong);
        b.cand = b.candLong = nullptr;
        b.candCap = starts;
        HIPCHK(c, dalloc(&b.cand, b.candCap));
        HIPCHK(c, dalloc(&b.candLong, b.candCap));
    }
    if (contours && contours < b.contourCap) {
        (void)hipFree(b.contours); (void)hipFree(b.binIdx);
        b.contours = nullptr; b.binIdx = nullptr;
        b.contourCap = contours;
        HIPCHK(c, dalloc(&b.contours, b.contourCap));
        HIPCHK(c, dalloc(&b.binIdx, (size_t)b.contourCap * 5));
    }
    if (points && points < b.poolCap) {
        (void)hipFree(b.pool);
        b.pool = nullptr;
        b.poolCap = points;
        HIPCHK(c, dalloc(&b.pool, b.poolCap));
    }
    if (tasks && tasks < b.taskCap) {
        (void)hipFree(b.tasks); (void)hipFree(b.copies);
        b.tasks = nullptr; b.copies = nullptr;
        b.taskCap = b.copyCap = tasks;
        HIPCHK(c, dalloc(&b.tasks, b.taskCap));
        HIPCHK(c, dalloc(&b.copies, b.copyCap));
    }
    if (seeds && seeds < b.seedCap) {
        (void)hipFree(b.seeds); (void)hipFree(b.seg); (void)hipFree(b.chain);
        b.seeds = nullptr; b.seg = nullptr; b.chain = nullptr;
        b.seedCap = seeds;
        b.segCap = b.chainCap = 2 * seeds;
        HIPCHK(c, dalloc(&b.seeds, b.seedCap));
        HIPCHK(c, dalloc(&b.seg, b.segCap));
        HIPCHK(c, dalloc(&b.chain, (size_t)b.chainCap * 32));
    }
    return CAMSYNC_OK;
}

int camsync_debug_regrown(const camsync_ctx* c) { return c ? c->regrown : CAMSYNC_E_ARG; }

int camsync_debug_identify_all(camsync_ctx* c, int on) {
    if (!c) return CAMSYNC_E_ARG;
    c->buf.identifyAll = on ? 1 : 0;
    return CAMSYNC_OK;
}

int camsync_debug_restore_capacity(camsync_ctx* c) {
    if (!c) return CAMSYNC_E_ARG;
    if (c->allocContourCap) {
        c->buf.contourCap = c->allocContourCap;
        c->buf.poolCap = c->allocPoolCap;
        c->buf.taskCap = c->allocTaskCap;
        c->buf.copyCap = c->allocTaskCap;
    }
    return CAMSYNC_OK;
}

int camsync_debug_counts(camsync_ctx* c, int frame, int32_t out[8]) {
    if (!c || !out) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Counters h;
    HIPCHK(c, hipMemcpy(&h, c->buf.counters, sizeof h, hipMemcpyDeviceToHost));
    int q = 0;
    if (frame >= 0 && frame < c->lastB) HIPCHK(c, hipMemcpy(&q, c->buf.qcount + frame, sizeof q, hipMemcpyDeviceToHost));
    out[0] = (int32_t)h.nCand; out[1] = (int32_t)h.nContours; out[2] = (int32_t)h.nPool; out[3] = (int32_t)h.nMarkers;
    out[4] = (int32_t)h.overflow; out[5] = q; out[6] = (int32_t)h.nLong; out[7] = (int32_t)h.qDecide;
    return CAMSYNC_OK;
}

int camsync_debug_starts(camsync_ctx* c, const uint8_t* frames, int B, int H, int W, int C, int max_starts, uint64_t* out) {
    int rc = check_frame_args(c, frames, B, H, W, C);
    if (rc) return rc;
    if (B > c->maxB || max_starts < 0 || (max_starts > 0 && !out)) return fail(c, CAMSYNC_E_ARG, "batch larger than the context's chunk, or no output");
    HIPCHK(c, hipSetDevice(c->device));
    rc = ensure_staging(c);
    if (rc) return rc;
    const DetCfg cfg = make_cfg(c, H, W);
    rc = ensure_planes(c, cfg.nScales);
    if (rc) return rc;
    if (!gray_thresh_is_fused(cfg)) {
        rc = ensure_gray(c);
        if (rc) return rc;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_in, frames, (size_t)B * H * W * C, hipMemcpyHostToDevice, c->stream));
    DetBuffers b = c->buf;
    b.in = c->d_in;
    b.bin = nullptr;
    if (gray_thresh_is_fused(cfg)) b.gray = nullptr;
    HIPCHK(c, hipMemsetAsync(b.counters, 0, sizeof(Counters), c->stream));
    launch_gray_thresh(cfg, b, B, C, c->stream);
    launch_contour_starts(cfg, b, B, c->stream);
    HIPCHK(c, hipGetLastError());
    Counters h;
    HIPCHK(c, hipMemcpyAsync(&h, b.counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const unsigned int n = std::min(h.nCand, b.candCap);
    const unsigned int m = std::min<unsigned int>(n, (unsigned int)max_starts);
    if (m) HIPCHK(c, hipMemcpy(out, b.cand, sizeof(uint64_t) * m, hipMemcpyDeviceToHost));
    c->lastB = B; c->lastH = H; c->lastW = W;
    return (int)std::min<unsigned int>(h.nCand, 0x7FFFFFFFu);
}

int camsync_debug_planes(camsync_ctx* c, int frame, int scale, uint8_t* out_bits) {
    if (!c || !out_bits || frame < 0 || frame >= c->lastB) return CAMSYNC_E_ARG;
    const DetCfg cfg = make_cfg(c, c->lastH, c->lastW);
    if (scale < 0 || scale >= cfg.nScales) return fail(c, CAMSYNC_E_ARG, "scale %d outside 0..%d", scale, cfg.nScales - 1);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t per = plane_words(cfg.H, cfg.WS);
    std::vector<uint32_t> h(per);
    HIPCHK(c, hipMemcpy(h.data(), c->buf.planes + ((size_t)frame * cfg.nScales + scale) * per, per * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int y = 0; y < cfg.H; y++)
        for (int x = 0; x < cfg.W; x++)
            out_bits[(size_t)y * cfg.W + x] = (uint8_t)((h[plane_word_index(cfg.WS, x >> 5, y)] >> (x & 31)) & 1u);
    return CAMSYNC_OK;
}

int camsync_debug_lists(camsync_ctx* c, uint32_t used[8], uint32_t cap[8]) {
    if (!c || !used || !cap) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Counters h;
    HIPCHK(c, hipMemcpy(&h, c->buf.counters, sizeof h, hipMemcpyDeviceToHost));
    const DetBuffers& b = c->buf;
    const uint32_t u[8] = {h.nCand, h.nLong, h.nContours, h.nPool, h.nTasks, h.nSeeds + h.nLong + h.nVirt, h.nCopy, h.nSeeds};
    const uint32_t k[8] = {b.candCap, b.candCap, b.contourCap, b.poolCap, b.taskCap, b.segCap, b.copyCap, b.seedCap};
    for (int i = 0; i < 8; i++) { used[i] = u[i]; cap[i] = k[i]; }
    return CAMSYNC_OK;
}

int camsync_debug_quads(camsync_ctx* c, int frame, int max_quads, float* corners, int32_t* scale, int32_t* scan_key,
                        int32_t* contour_len, int32_t* id, int32_t* rot) {
    if (!c || frame < 0 || frame >= c->lastB) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int q = 0;
    HIPCHK(c, hipMemcpy(&q, c->buf.qcount + frame, sizeof q, hipMemcpyDeviceToHost));
    q = std::min(q, c->buf.quadCap);
    std::vector<QuadRec> h(q);
    if (q) HIPCHK(c, hipMemcpy(h.data(), c->buf.quads + (size_t)frame * c->buf.quadCap, sizeof(QuadRec) * q, hipMemcpyDeviceToHost));
    for (int i = 0; i < q && i < max_quads; i++) {
        for (int k = 0; k < 8; k++) corners[i * 8 + k] = (float)h[i].c[k];
        if (scale) scale[i] = h[i].scale;
        if (scan_key) scan_key[i] = (int32_t)h[i].scanKey;
        if (contour_len) contour_len[i] = h[i].clen;
        if (id) id[i] = h[i].valid ? h[i].id : -1;
        if (rot) rot[i] = h[i].rot;
    }
    return q;
}

int camsync_debug_contours(camsync_ctx* c, int frame, int max_contours, int32_t* scale, int32_t* start_key,
                           int32_t* is_hole, int32_t* length) {
    if (!c || frame < 0 || frame >= c->lastB) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Counters hc;
    HIPCHK(c, hipMemcpy(&hc, c->buf.counters, sizeof hc, hipMemcpyDeviceToHost));
    const unsigned n = std::min(hc.nContours, c->buf.contourCap);
    std::vector<ContourRec> h(n);
    if (n) HIPCHK(c, hipMemcpy(h.data(), c->buf.contours, sizeof(ContourRec) * n, hipMemcpyDeviceToHost));
    int k = 0;
    for (unsigned i = 0; i < n; i++) {
        if ((int)h[i].frame != frame) continue;
        if (k < max_contours) {
            if (scale) scale[k] = h[i].scale;
            if (start_key) start_key[k] = (int32_t)h[i].pos;
            if (is_hole) is_hole[k] = h[i].hole;
            if (length) length[k] = (int32_t)h[i].len;
        }
        k++;
    }
    return k;
}

int camsync_debug_points(camsync_ctx* c, int frame, int max_points, int32_t* xy) {
    if (!c || frame < 0 || frame >= c->lastB || !xy) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    Counters hc;
    HIPCHK(c, hipMemcpy(&hc, c->buf.counters, sizeof hc, hipMemcpyDeviceToHost));
    const unsigned n = std::min(hc.nContours, c->buf.contourCap);
    std::vector<ContourRec> h(n);
    if (n) HIPCHK(c, hipMemcpy(h.data(), c->buf.contours, sizeof(ContourRec) * n, hipMemcpyDeviceToHost));
    int k = 0;
    std::vector<uint32_t> pts;
    for (unsigned i = 0; i < n; i++) {
        if ((int)h[i].frame != frame) continue;
        if (h[i].off >= c->buf.poolCap || h[i].len > c->buf.poolCap - h[i].off)
            return fail(c, CAMSYNC_E_CAPACITY, "contour record %u points outside the pool (off %u len %u)", i, h[i].off, h[i].len);
        pts.resize(h[i].len);
        if (h[i].len) HIPCHK(c, hipMemcpy(pts.data(), c->buf.pool + h[i].off, sizeof(uint32_t) * h[i].len, hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < h[i].len; j++, k++)
            if (k < max_points) {
                xy[2 * k] = (int32_t)(pts[j] & 0xFFFFu);
                xy[2 * k + 1] = (int32_t)(pts[j] >> 16);
            }
    }
    return k;
}

int camsync_pnp_batch(camsync_ctx* c, const float* obj, const float* img, const int32_t* npts, int B, int N,
                      const double* K, const double* dist, int ndist, int per_frame_cam, double* rvec, double* tvec,
                      uint8_t* ok, double* mae, double* rmse) {
    if (!c) return CAMSYNC_E_ARG;
    if (!obj || !img || !npts || !K || !rvec || !tvec || !ok || B <= 0) return fail(c, CAMSYNC_E_ARG, "missing argument");
    if (N < 4 || N > 256) return fail(c, CAMSYNC_E_ARG, "N must be 4..256 points per frame");
    if (ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return fail(c, CAMSYNC_E_ARG, "ndist must be 0..8");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_pose_scratch(c, (size_t)B, N);
    if (rc) return rc;
    const size_t kc = per_frame_cam ? (size_t)B : 1;
    HIPCHK(c, hipMemcpyAsync(c->d_obj, obj, sizeof(float) * 3 * N * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_img, img, sizeof(float) * 2 * N * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_npts, npts, sizeof(int32_t) * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_K, K, sizeof(double) * 9 * kc, hipMemcpyHostToDevice, c->stream));
    if (ndist) HIPCHK(c, hipMemcpyAsync(c->d_dist, dist, sizeof(double) * ndist * kc, hipMemcpyHostToDevice, c->stream));
    PnpArgs a{};
    a.obj = c->d_obj; a.img = c->d_img; a.npts = c->d_npts; a.B = B; a.N = N;
    a.K = c->d_K; a.dist = c->d_dist; a.ndist = ndist; a.perFrameCam = per_frame_cam; a.refine = 1;
    a.rvec = c->d_rvec; a.tvec = c->d_tvec; a.ok = c->d_ok; a.mae = c->d_mae; a.rmse = c->d_rmse;
    if (c->timing) (void)hipEventRecord(c->ev[7], c->stream);
    launch_pnp(a, c->stream);
    if (c->timing) (void)hipEventRecord(c->ev[8], c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(rvec, c->d_rvec, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(tvec, c->d_tvec, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ok, c->d_ok, B, hipMemcpyDeviceToHost, c->stream));
    if (mae) HIPCHK(c, hipMemcpyAsync(mae, c->d_mae, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    if (rmse) HIPCHK(c, hipMemcpyAsync(rmse, c->d_rmse, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->timing) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev[7], c->ev[8]) == hipSuccess) { c->ms[7] += ms; c->launches[7]++; }
    }
    return CAMSYNC_OK;
}

int camsync_pnp_batch_noalloc(camsync_ctx* c, const float* obj, const float* img, const int32_t* npts, int B, int N,
                              const double* K, const double* dist, int ndist, int per_frame_cam, double* rvec,
                              double* tvec, uint8_t* ok, double* mae, double* rmse) {
    if (!c) return CAMSYNC_E_ARG;
    if (!obj || !img || !npts || !K || !rvec || !tvec || !ok || B <= 0) return fail(c, CAMSYNC_E_ARG, "missing argument");
    if (N < 4 || N > 256) return fail(c, CAMSYNC_E_ARG, "N must be 4..256 points per frame");
    if (ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return fail(c, CAMSYNC_E_ARG, "ndist must be 0..8");
    if ((size_t)B > kNaMaxFrames || (size_t)B * (size_t)N > kNaMaxPoints)
        return fail(c, CAMSYNC_E_CAPACITY,
                    "camsync_pnp_batch_noalloc handles at most %d frames and %d points (B * N) per call, got B = %d, N = %d; "
                    "use camsync_pnp_batch", CAMSYNC_NOALLOC_MAX_FRAMES, CAMSYNC_NOALLOC_MAX_POINTS, B, N);
    HIPCHK(c, hipSetDevice(c->device));
    unsigned char* const sp = c->naScratch;  // device address of this GPU's scratch array, never the symbol itself
    if (!sp) return fail(c, CAMSYNC_E_HIP, "no scratch address for device %d", c->device);
    float* d_obj = reinterpret_cast<float*>(sp + kNaObj);
    float* d_img = reinterpret_cast<float*>(sp + kNaImg);
    double* d_K = reinterpret_cast<double*>(sp + kNaK);
    double* d_dist = reinterpret_cast<double*>(sp + kNaDist);
    double* d_rvec = reinterpret_cast<double*>(sp + kNaRvec);
    double* d_tvec = reinterpret_cast<double*>(sp + kNaTvec);
    double* d_mae = reinterpret_cast<double*>(sp + kNaMae);
    double* d_rmse = reinterpret_cast<double*>(sp + kNaRmse);
    int32_t* d_npts = reinterpret_cast<int32_t*>(sp + kNaNpts);
    uint8_t* d_ok = sp + kNaOk;
    const size_t kc = per_frame_cam ? (size_t)B : 1;
    // every field the kernel reads is overwritten for the B frames of this call, and it writes every output of those
    // frames: nothing has to start at zero, and what an earlier (larger) call left behind is never read
    HIPCHK(c, hipMemcpyAsync(d_obj, obj, sizeof(float) * 3 * N * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_img, img, sizeof(float) * 2 * N * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_npts, npts, sizeof(int32_t) * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_K, K, sizeof(double) * 9 * kc, hipMemcpyHostToDevice, c->stream));
    if (ndist) HIPCHK(c, hipMemcpyAsync(d_dist, dist, sizeof(double) * ndist * kc, hipMemcpyHostToDevice, c->stream));
    PnpArgs a{};
    a.obj = d_obj; a.img = d_img; a.npts = d_npts; a.B = B; a.N = N;
    a.K = d_K; a.dist = d_dist; a.ndist = ndist; a.perFrameCam = per_frame_cam; a.refine = 1;
    a.rvec = d_rvec; a.tvec = d_tvec; a.ok = d_ok; a.mae = d_mae; a.rmse = d_rmse;
    launch_pnp(a, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(rvec, d_rvec, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(tvec, d_tvec, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(ok, d_ok, B, hipMemcpyDeviceToHost, c->stream));
    if (mae) HIPCHK(c, hipMemcpyAsync(mae, d_mae, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    if (rmse) HIPCHK(c, hipMemcpyAsync(rmse, d_rmse, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

int camsync_ippe_square_batch(camsync_ctx* c, const float* corners, const double* size, int B, const double* K,
                              const double* dist, int ndist, int per_frame_cam, double* rvec, double* tvec, uint8_t* ok) {
    if (!c) return CAMSYNC_E_ARG;
    if (!corners || !size || !K || !rvec || !tvec || B <= 0) return fail(c, CAMSYNC_E_ARG, "missing argument");
    if (ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return fail(c, CAMSYNC_E_ARG, "ndist must be 0..8");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_pose_scratch(c, (size_t)B, 4);
    if (rc) return rc;
    const size_t kc = per_frame_cam ? (size_t)B : 1;
    // reuse the pose scratch: corners -> d_img (B*4*2 floats), sizes -> d_mae
    HIPCHK(c, hipMemcpyAsync(c->d_img, corners, sizeof(float) * 8 * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_mae, size, sizeof(double) * B, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_K, K, sizeof(double) * 9 * kc, hipMemcpyHostToDevice, c->stream));
    if (ndist) HIPCHK(c, hipMemcpyAsync(c->d_dist, dist, sizeof(double) * ndist * kc, hipMemcpyHostToDevice, c->stream));
    launch_ippe(c->d_img, c->d_mae, B, c->d_K, c->d_dist, ndist, per_frame_cam, c->d_rvec, c->d_tvec, c->d_ok, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(rvec, c->d_rvec, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(tvec, c->d_tvec, sizeof(double) * 3 * B, hipMemcpyDeviceToHost, c->stream));
    if (ok) HIPCHK(c, hipMemcpyAsync(ok, c->d_ok, B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

/* ---- chessboard front end ---- */

int camsync_chess_candidates(camsync_ctx* c, const uint8_t* frames, int on_device, int B, int H, int W, int C, int scale,
                             int max_cand, int32_t* xy, int32_t* strength, int32_t* counts) {
    if (!c) return CAMSYNC_E_ARG;
    if (!frames || B <= 0 || !xy || !strength || !counts) return fail(c, CAMSYNC_E_ARG, "missing argument");
    if (C != 1 && C != 3) return fail(c, CAMSYNC_E_ARG, "channels must be 1 or 3, got %d", C);
    if (scale < 1 || scale > 16 || max_cand < 1 || max_cand > 65536) return fail(c, CAMSYNC_E_ARG, "scale must be 1..16, max_cand 1..65536");
    if (H > c->maxH || W > c->maxW || H <= 10 * scale || W <= 10 * scale)
        return fail(c, CAMSYNC_E_ARG, "frame %dx%d outside the context's range, or smaller than the detector's ring", H, W);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t px = (size_t)H * W, maxPx = (size_t)c->maxH * c->maxW;
    if (!c->chessChunk) {
        const int chunk = std::min(c->maxB, 16);
        HIPCHK(c, dalloc(&c->d_chessGray, chunk * maxPx));
        HIPCHK(c, dalloc(&c->d_chessS, chunk * maxPx));
        HIPCHK(c, dalloc(&c->d_chessR, chunk * maxPx));
        HIPCHK(c, dalloc(&c->d_chessMax, (size_t)chunk));
        HIPCHK(c, dalloc(&c->d_chessCount, (size_t)chunk));
        c->chessChunk = chunk;
    }
    if (c->chessMaxCand < max_cand) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_chessCand) (void)hipFree(c->d_chessCand);
        c->d_chessCand = nullptr;
        c->chessMaxCand = 0;
        HIPCHK(c, dalloc(&c->d_chessCand, (size_t)c->chessChunk * max_cand * 3));
        c->chessMaxCand = max_cand;
    }
    if (!on_device) {
        int rc = ensure_staging(c);
        if (rc) return rc;
    }
    std::vector<int32_t> hc((size_t)c->chessChunk * max_cand * 3);
    std::vector<int> hn(c->chessChunk);
    struct Cd { int32_t x, y, v; };
    for (int b0 = 0; b0 < B; b0 += c->chessChunk) {
        const int nb = std::min(c->chessChunk, B - b0);
        const uint8_t* src = frames + (size_t)b0 * px * C;
        if (!on_device) {
            HIPCHK(c, hipMemcpyAsync(c->d_in, src, (size_t)nb * px * C, hipMemcpyHostToDevice, c->stream));
            src = c->d_in;
        }
        HIPCHK(c, hipMemsetAsync(c->d_chessMax, 0, sizeof(int) * nb, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_chessCount, 0, sizeof(int) * nb, c->stream));
        launch_chess(src, C, nb, H, W, scale, c->d_chessGray, c->d_chessS, c->d_chessR, c->d_chessMax, max_cand, c->d_chessCand,
                     c->d_chessCount, c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hn.data(), c->d_chessCount, sizeof(int) * nb, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hc.data(), c->d_chessCand, sizeof(int32_t) * 3 * (size_t)nb * max_cand, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int f = 0; f < nb; f++) {
            if (hn[f] > max_cand)
                return fail(c, CAMSYNC_E_CAPACITY, "frame %d has %d corner candidates, more than max_cand = %d", b0 + f, hn[f], max_cand);
            Cd* cd = reinterpret_cast<Cd*>(hc.data() + (size_t)f * max_cand * 3);
            // strongest first; equal strengths in raster order (the kernel appends in no particular order)
            std::sort(cd, cd + hn[f], [](const Cd& a, const Cd& b) { return a.v != b.v ? a.v > b.v : (a.y != b.y ? a.y < b.y : a.x < b.x); });
            counts[b0 + f] = hn[f];
            for (int k = 0; k < hn[f]; k++) {
                xy[((size_t)(b0 + f) * max_cand + k) * 2] = cd[k].x;
                xy[((size_t)(b0 + f) * max_cand + k) * 2 + 1] = cd[k].y;
                strength[(size_t)(b0 + f) * max_cand + k] = cd[k].v;
            }
        }
    }
    return CAMSYNC_OK;
}

int camsync_corner_subpix(camsync_ctx* c, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                          const float* pts, const int32_t* npts, int N, int win, int max_iter, double eps, float* out) {
    if (!c) return CAMSYNC_E_ARG;
    if (!frames || !pts || !out || B <= 0 || N <= 0) return fail(c, CAMSYNC_E_ARG, "missing argument");
    if (C != 1 && C != 3) return fail(c, CAMSYNC_E_ARG, "channels must be 1 or 3, got %d", C);
    if (H > c->maxH || W > c->maxW || H < 2 || W < 2) return fail(c, CAMSYNC_E_ARG, "frame %dx%d outside the context's range", H, W);
    if (win < 1 || win > 20) return fail(c, CAMSYNC_E_ARG, "cornerSubPix window half-size must be 1..20, got %d", win);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t px = (size_t)H * W;
    const int chunk = c->maxB;
    if (c->subpixCap < chunk * N) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_spIn); (void)hipFree(c->d_spOut); (void)hipFree(c->d_spN);
        c->d_spIn = c->d_spOut = nullptr;
        c->d_spN = nullptr;
        c->subpixCap = 0;
        HIPCHK(c, dalloc(&c->d_spIn, (size_t)chunk * N * 2));
        HIPCHK(c, dalloc(&c->d_spOut, (size_t)chunk * N * 2));
        HIPCHK(c, dalloc(&c->d_spN, (size_t)chunk));
        c->subpixCap = chunk * N;
    }
    if (!on_device) {
        int rc = ensure_staging(c);
        if (rc) return rc;
    }
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        const uint8_t* src = frames + (size_t)b0 * px * C;
        if (!on_device) {
            HIPCHK(c, hipMemcpyAsync(c->d_in, src, (size_t)nb * px * C, hipMemcpyHostToDevice, c->stream));
            src = c->d_in;
        }
        HIPCHK(c, hipMemcpyAsync(c->d_spIn, pts + (size_t)b0 * N * 2, sizeof(float) * 2 * (size_t)nb * N, hipMemcpyHostToDevice, c->stream));
        if (npts) HIPCHK(c, hipMemcpyAsync(c->d_spN, npts + b0, sizeof(int32_t) * nb, hipMemcpyHostToDevice, c->stream));
        // points the kernel skips (beyond npts) come back unchanged
        HIPCHK(c, hipMemcpyAsync(c->d_spOut, c->d_spIn, sizeof(float) * 2 * (size_t)nb * N, hipMemcpyDeviceToDevice, c->stream));
        launch_subpix_points(src, C, nb, H, W, c->d_spIn, npts ? c->d_spN : nullptr, N, win, max_iter, eps, c->d_expTab, c->d_spOut, c->stream);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(out + (size_t)b0 * N * 2, c->d_spOut, sizeof(float) * 2 * (size_t)nb * N, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return CAMSYNC_OK;
}

/* ---- undistortion ---- */

int camsync_optimal_new_camera_matrix(const double* K, const double* dist, int ndist, int W, int H, double alpha,
                                      int new_w, int new_h, double* new_K, int32_t* roi) {
    if (!K || !new_K || W < 2 || H < 2 || ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return CAMSYNC_E_ARG;
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist; i++) k[i] = dist[i];
    if (new_w <= 0 || new_h <= 0) { new_w = W; new_h = H; }
    optimal_new_camera_matrix(K, k, W, H, alpha, new_w, new_h, new_K, roi);
    return CAMSYNC_OK;
}

int camsync_undistort_batch(camsync_ctx* c, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                            const double* K, const double* dist, int ndist, const double* new_K, uint8_t* out) {
    if (!c) return CAMSYNC_E_ARG;
    if (!frames || !out || B <= 0) return fail(c, CAMSYNC_E_ARG, "frames / out is NULL or batch <= 0");
    if (C != 1 && C != 3) return fail(c, CAMSYNC_E_ARG, "channels must be 1 or 3, got %d", C);
    if (H > c->maxH || W > c->maxW || H < 2 || W < 4 || (W % 4))
        return fail(c, CAMSYNC_E_ARG, "frame %dx%d outside the context's range, or width not a multiple of 4", H, W);
    if (!K || ndist < 0 || ndist > 8 || (ndist > 0 && !dist)) return fail(c, CAMSYNC_E_ARG, "camera matrix / distortion missing (ndist 0..8)");
    if (on_device && ((((uintptr_t)frames) | ((uintptr_t)out)) % 4)) return fail(c, CAMSYNC_E_ARG, "device frames must be 4-byte aligned");
    HIPCHK(c, hipSetDevice(c->device));
    double key[26];
    for (int i = 0; i < 9; i++) { key[i] = K[i]; key[9 + i] = new_K ? new_K[i] : K[i]; }
    for (int i = 0; i < 8; i++) key[18 + i] = i < ndist ? dist[i] : 0.0;
    if (!undistort_camera_ok(key + 9)) return fail(c, CAMSYNC_E_NUMERIC, "the new camera matrix is singular");
    const size_t px = (size_t)H * W;
    if (c->umapPx < px) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_umap) (void)hipFree(c->d_umap);
        c->d_umap = nullptr;
        c->umapPx = 0;
        HIPCHK(c, dalloc(&c->d_umap, px));
        c->umapPx = px;
        c->umapH = 0;
    }
    if (c->umapH != H || c->umapW != W || memcmp(key, c->umapKey, sizeof key) != 0) {
        launch_undistort_map(key, key + 9, key + 18, H, W, c->d_umap, c->stream);
        HIPCHK(c, hipGetLastError());
        memcpy(c->umapKey, key, sizeof key);
        c->umapH = H;
        c->umapW = W;
    }
    if (!on_device) {
        int rc = ensure_staging(c);
        if (rc) return rc;
        if (!c->d_uout) HIPCHK(c, dalloc(&c->d_uout, (size_t)c->maxB * c->maxH * c->maxW * 3));
    }
    for (int b0 = 0; b0 < B; b0 += c->maxB) {
        const int nb = std::min(c->maxB, B - b0);
        const uint8_t* src = frames + (size_t)b0 * px * C;
        uint8_t* dst = out + (size_t)b0 * px * C;
        if (!on_device) {
            HIPCHK(c, hipMemcpyAsync(c->d_in, src, (size_t)nb * px * C, hipMemcpyHostToDevice, c->stream));
            launch_remap(c->d_in, c->d_umap, c->d_uout, nb, H, W, C, c->stream);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(dst, c->d_uout, (size_t)nb * px * C, hipMemcpyDeviceToHost, c->stream));
        } else {
            launch_remap(src, c->d_umap, dst, nb, H, W, C, c->stream);
            HIPCHK(c, hipGetLastError());
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

/* ---- frame ingest ---- */

int camsync_ingest_yuyv(camsync_ctx* c, const uint8_t* yuyv, int in_on_device, int B, int H, int W, uint8_t* bgr, int out_on_device) {
    if (!c) return CAMSYNC_E_ARG;
    if (!yuyv || !bgr || B <= 0) return fail(c, CAMSYNC_E_ARG, "yuyv / bgr is NULL or batch <= 0");
    if (H <= 0 || W <= 0 || (W % 8)) return fail(c, CAMSYNC_E_ARG, "frame width must be a positive multiple of 8 (got %dx%d)", H, W);
    if (H > c->maxH || W > c->maxW) return fail(c, CAMSYNC_E_ARG, "frame %dx%d larger than the context's %dx%d", H, W, c->maxH, c->maxW);
    if ((out_on_device && ((uintptr_t)bgr % 8)) || (in_on_device && ((uintptr_t)yuyv % 16)))
        return fail(c, CAMSYNC_E_ARG, "device buffers must be aligned: yuyv to 16 bytes, bgr to 8");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t px = (size_t)H * W;
    if (!in_on_device) {
        int rc = ensure_staging(c);  // (maxB frames of 3 bytes a pixel: the raw frames need two)
        if (rc) return rc;
    }
    if (!out_on_device && !c->d_uout) HIPCHK(c, dalloc(&c->d_uout, (size_t)c->maxB * c->maxH * c->maxW * 3));
    // chunks of max_batch frames only where a staging buffer of the context is involved
    const int step = (in_on_device && out_on_device) ? B : c->maxB;
    for (int b0 = 0; b0 < B; b0 += step) {
        const int nb = std::min(step, B - b0);
        const uint8_t* src = yuyv + (size_t)b0 * px * 2;
        uint8_t* dst = bgr + (size_t)b0 * px * 3;
        if (!in_on_device) {
            HIPCHK(c, hipMemcpyAsync(c->d_in, src, (size_t)nb * px * 2, hipMemcpyHostToDevice, c->stream));
            src = c->d_in;
        }
        launch_yuyv_bgr(src, out_on_device ? dst : c->d_uout, (size_t)nb * px, c->stream);
        HIPCHK(c, hipGetLastError());
        if (!out_on_device) HIPCHK(c, hipMemcpyAsync(dst, c->d_uout, (size_t)nb * px * 3, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

int camsync_jpeg_info(const uint8_t* data, size_t n, int32_t* height, int32_t* width, int32_t* components) {
    return jpeg_info(data, n, height, width, components);
}

int camsync_decode_jpeg_batch(camsync_ctx* c, const uint8_t* const* jpegs, const size_t* sizes, int B, int H, int W, uint8_t* bgr,
                              int out_on_device) {
    if (!c) return CAMSYNC_E_ARG;
    if (!jpegs || !sizes || !bgr || B <= 0) return fail(c, CAMSYNC_E_ARG, "jpegs / sizes / bgr is NULL or batch <= 0");
    if (H < 8 || W < 8 || H > c->maxH || W > c->maxW)
        return fail(c, CAMSYNC_E_ARG, "frame %dx%d outside the context's range (8..%d x 8..%d)", H, W, c->maxH, c->maxW);
    if (out_on_device && ((uintptr_t)bgr % 4)) return fail(c, CAMSYNC_E_ARG, "the device output must be 4-byte aligned");
    for (int i = 0; i < B; i++)
        if (!jpegs[i] || sizes[i] < 4) return fail(c, CAMSYNC_E_ARG, "frame %d is NULL or shorter than a JPEG header", i);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->jpeg) c->jpeg = jpeg_create();
    std::string err;
    const int rc = jpeg_decode_batch(c->jpeg, c->stream, jpegs, sizes, B, H, W, bgr, out_on_device, err);
    if (rc) return fail(c, rc, "camsync_decode_jpeg_batch: %s", err.c_str());
    return CAMSYNC_OK;
}

/* ---- calibration ---- */

int camsync_calib_begin(camsync_ctx* c, const float* obj, const float* img, int V, int P, int flags) {
    if (!c) return CAMSYNC_E_ARG;
    if (V < 0 || P < 4 || P > 64 || (V > 0 && (!obj || !img)))
        return fail(c, CAMSYNC_E_ARG, "calibration needs 4..64 points per view (one wavefront lane per point)");
    HIPCHK(c, hipSetDevice(c->device));
    CalibState& s = c->cal;
    const int NI = (flags & 1) ? 12 : 9;
    const size_t nv = (size_t)std::max(V, 1);
    // the buffers of the last problem are kept while they are large enough (repeated calibrations allocate nothing)
    if (!(c->calReady && nv * P <= c->calCapPts && nv <= c->calCapV && NI <= c->calCapNI)) {
        void* old[] = {c->d_calObj, c->d_calImg, s.intr, s.ext, s.intrTrial, s.extTrial, s.viewBlocks, s.viewBlocksTrial, s.partial,
                       s.ticket, c->d_calWork, c->d_focalRows, c->d_calOk, c->d_calK, c->d_calNpts};
        for (void* p : old)
            if (p) (void)hipFree(p);
        c->d_calObj = c->d_calImg = nullptr;
        c->d_calWork = c->d_focalRows = c->d_calK = nullptr;
        c->d_calOk = nullptr;
        c->d_calNpts = nullptr;
        c->calReady = false;
        s = CalibState{};
        HIPCHK(c, dalloc(&c->d_calObj, nv * P * 3));
        HIPCHK(c, dalloc(&c->d_calImg, nv * P * 2));
        HIPCHK(c, dalloc(&s.intr, 12));
        HIPCHK(c, dalloc(&s.ext, nv * 6));
        HIPCHK(c, dalloc(&s.intrTrial, 12));
        HIPCHK(c, dalloc(&s.extTrial, nv * 6));
        HIPCHK(c, dalloc(&s.viewBlocks, nv * (6 * NI + 6)));
        HIPCHK(c, dalloc(&s.viewBlocksTrial, nv * (6 * NI + 6)));
        HIPCHK(c, dalloc(&s.partial, (size_t)(calib_lm_grid((int)nv) + 32) * 160));
        HIPCHK(c, dalloc(&s.ticket, 33));
        HIPCHK(c, hipMemset(s.ticket, 0, sizeof(unsigned int) * 33));
        HIPCHK(c, dalloc(&c->d_calWork, (size_t)CAMSYNC_CALIB_WORK_DOUBLES));
        HIPCHK(c, dalloc(&c->d_focalRows, nv * 6));
        HIPCHK(c, dalloc(&c->d_calOk, nv));
        HIPCHK(c, dalloc(&c->d_calK, 9 + 8));
        HIPCHK(c, dalloc(&c->d_calNpts, nv));
        c->calCapPts = nv * P;
        c->calCapV = nv;
        c->calCapNI = NI;
    }
    s.V = V;
    s.P = P;
    s.NI = NI;
    HIPCHK(c, hipMemsetAsync(c->d_calWork, 0, sizeof(double) * CAMSYNC_CALIB_WORK_DOUBLES, c->stream));
    if (V > 0) {
        HIPCHK(c, hipMemcpyAsync(c->d_calObj, obj, sizeof(float) * (size_t)V * P * 3, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_calImg, img, sizeof(float) * (size_t)V * P * 2, hipMemcpyHostToDevice, c->stream));
        launch_fill_i32(c->d_calNpts, V, P, c->stream);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemsetAsync(s.intr, 0, sizeof(double) * 12, c->stream));
    HIPCHK(c, hipMemsetAsync(s.ext, 0, sizeof(double) * nv * 6, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // obj / img are the caller's again
    s.obj = c->d_calObj;
    s.img = c->d_calImg;
    c->calReady = true;
    return CAMSYNC_OK;
}

int camsync_calib_n_intrinsics(const camsync_ctx* c) { return (c && c->calReady) ? c->cal.NI : CAMSYNC_E_ARG; }

int camsync_calib_focal_acc(camsync_ctx* c, int W, int H, double acc6[6]) {
    if (!c || !c->calReady || !acc6) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    for (int i = 0; i < 6; i++) acc6[i] = 0;
    if (c->cal.V == 0) return CAMSYNC_OK;
    const double cx = (!W) ? 0.5 : (W - 1) * 0.5, cy = (!H) ? 0.5 : (H - 1) * 0.5;
    launch_calib_focal_rows(c->cal, cx, cy, c->d_focalRows, c->d_calWork + kWFocal, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(acc6, c->d_calWork + kWFocal, sizeof(double) * 6, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

int camsync_calib_set_intrinsics(camsync_ctx* c, const double intr[12], int init_extrinsics) {
    if (!c || !c->calReady || !intr) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    CalibState& s = c->cal;
    HIPCHK(c, hipMemcpyAsync(s.intr, intr, sizeof(double) * 12, hipMemcpyHostToDevice, c->stream));
    if (init_extrinsics && s.V > 0) {
        // cvFindExtrinsicCameraParams2 per view with the initial camera matrix and zero distortion
        double Kd[17] = {intr[0], 0, intr[2], 0, intr[1], intr[3], 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0};
        HIPCHK(c, hipMemcpyAsync(c->d_calK, Kd, sizeof Kd, hipMemcpyHostToDevice, c->stream));
        int rc = ensure_pose_scratch(c, (size_t)s.V, 4);
        if (rc) return rc;
        PnpArgs a{};
        a.obj = s.obj; a.img = s.img; a.npts = c->d_calNpts; a.B = s.V; a.N = s.P;
        a.K = c->d_calK; a.dist = c->d_calK + 9; a.ndist = 0; a.perFrameCam = 0; a.refine = 0;
        a.rvec = c->d_rvec; a.tvec = c->d_tvec; a.ok = c->d_calOk; a.mae = nullptr; a.rmse = nullptr;
        launch_pnp(a, c->stream);
        HIPCHK(c, hipGetLastError());
        // interleave rvec/tvec into ext [V,6]
        HIPCHK(c, hipMemcpy2DAsync(s.ext, sizeof(double) * 6, c->d_rvec, sizeof(double) * 3, sizeof(double) * 3, s.V, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(s.ext + 3, sizeof(double) * 6, c->d_tvec, sizeof(double) * 3, sizeof(double) * 3, s.V, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

int camsync_calib_accumulate(camsync_ctx* c, double lambda, double* acc, int acc_on_device) {
    if (!c || !c->calReady || !acc) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    CalibState& s = c->cal;
    const int L = s.NI * s.NI + s.NI + 1;
    launch_calib_lm(s, 0, lambda, c->d_calWork, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(acc, c->d_calWork + kWAcc, sizeof(double) * L, acc_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

int camsync_calib_solve(const double* acc, int ni, double* delta) {
    if (!acc || !delta || ni < 1 || ni > 12) return CAMSYNC_E_ARG;
    // symmetric positive definite ni x ni system S d = r; Cholesky, falling back to a ridge when a pivot vanishes
    double A[144], b[12];
    for (int attempt = 0; attempt < 2; attempt++) {
        double tr = 0;
        for (int i = 0; i < ni; i++) tr += acc[i * ni + i];
        for (int i = 0; i < ni * ni; i++) A[i] = acc[i];
        if (attempt)
            for (int i = 0; i < ni; i++) A[i * ni + i] += 1e-14 * tr + 1e-300;
        for (int i = 0; i < ni; i++) b[i] = acc[ni * ni + i];
        bool ok = true;
        for (int j = 0; j < ni && ok; j++) {
            double s = A[j * ni + j];
            for (int k = 0; k < j; k++) s -= A[j * ni + k] * A[j * ni + k];
            if (!(s > 0)) { ok = false; break; }
            const double d = sqrt(s);
            A[j * ni + j] = d;
            for (int i = j + 1; i < ni; i++) {
                double t = A[i * ni + j];
                for (int k = 0; k < j; k++) t -= A[i * ni + k] * A[j * ni + k];
                A[i * ni + j] = t / d;
            }
        }
        if (!ok) continue;
        for (int i = 0; i < ni; i++) {
            double s = b[i];
            for (int k = 0; k < i; k++) s -= A[i * ni + k] * b[k];
            b[i] = s / A[i * ni + i];
        }
        for (int i = ni - 1; i >= 0; i--) {
            double s = b[i];
            for (int k = i + 1; k < ni; k++) s -= A[k * ni + i] * b[k];
            b[i] = s / A[i * ni + i];
        }
        for (int i = 0; i < ni; i++) delta[i] = b[i];
        return CAMSYNC_OK;
    }
    return CAMSYNC_E_NUMERIC;
}

int camsync_calib_step(camsync_ctx* c, const double* delta, double* err_out) {
    if (!c || !c->calReady || !delta || !err_out) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    CalibState& s = c->cal;
    const int L = s.NI * s.NI + s.NI + 1;
    HIPCHK(c, hipMemcpyAsync(c->d_calWork + kWDelta, delta, sizeof(double) * s.NI, hipMemcpyHostToDevice, c->stream));
    // the trial parameters' record is formed as well (undamped, unused here); the error is its last entry
    launch_calib_lm(s, 2, 0.0, c->d_calWork, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(err_out, c->d_calWork + kWAcc + L - 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CAMSYNC_OK;
}

// The trial parameters become the current ones: the three buffer pairs change roles.
static void calib_swap(CalibState& s) {
    std::swap(s.intr, s.intrTrial);
    std::swap(s.ext, s.extTrial);
    std::swap(s.viewBlocks, s.viewBlocksTrial);
}

int camsync_calib_accept(camsync_ctx* c) {
    if (!c || !c->calReady) return CAMSYNC_E_ARG;
    calib_swap(c->cal);
    return CAMSYNC_OK;
}

int camsync_calib_get(camsync_ctx* c, double intr[12], double* rvecs, double* tvecs) {
    if (!c || !c->calReady) return CAMSYNC_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    CalibState& s = c->cal;
    if (intr) HIPCHK(c, hipMemcpy(intr, s.intr, sizeof(double) * 12, hipMemcpyDeviceToHost));
    if ((rvecs || tvecs) && s.V > 0) {
        std::vector<double> e((size_t)s.V * 6);
        HIPCHK(c, hipMemcpy(e.data(), s.ext, sizeof(double) * e.size(), hipMemcpyDeviceToHost));
        for (int v = 0; v < s.V; v++)
            for (int q = 0; q < 3; q++) {
                if (rvecs) rvecs[3 * v + q] = e[(size_t)v * 6 + q];
                if (tvecs) tvecs[3 * v + q] = e[(size_t)v * 6 + 3 + q];
            }
    }
    return CAMSYNC_OK;
}

// cvCalibrateCamera2Internal's flow (focal init, per-view extrinsic init, CvLevMarq::updateAlt loop with criteria
// (COUNT+EPS, max_iter, DBL_EPSILON)) with the whole iteration on the device: accumulate -> [sum over ranks] ->
// solve (one lane) -> back-substitute -> trial error + parameter change -> [sum over ranks] -> ONE read-back of six
// doubles, on which the host takes the accept / raise-lambda / stop decision.  `ar` is the only thing that differs
// between one GPU (null) and many.
static int calibrate_impl(camsync_ctx* c, const float* obj, const float* img, int V, int P, long long Vtotal, int W, int H,
                          int flags, int max_iter, camsync_allreduce_fn ar, void* user, double* d_work, double* K,
                          double* dist, double* rvecs, double* tvecs, double* rms, int32_t* iters) {
    if (!c) return CAMSYNC_E_ARG;
    if (V < 0 || Vtotal < 1 || Vtotal < V || !K || !dist) return fail(c, CAMSYNC_E_ARG, "need at least one view overall and K/dist outputs");
    if (max_iter <= 0) max_iter = 30;
    int rc = camsync_calib_begin(c, obj, img, V, P, flags);
    if (rc) return rc;
    CalibState& s = c->cal;
    const int NI = s.NI, L = NI * NI + NI + 1;
    double* const work = d_work ? d_work : c->d_calWork;  // the records that cross ranks live in the caller's buffer
    HIPCHK(c, hipMemsetAsync(work, 0, sizeof(double) * CAMSYNC_CALIB_WORK_DOUBLES, c->stream));
    auto allreduce = [&](double* p, int n) -> int {
        if (!ar) return CAMSYNC_OK;
        const int r = ar(user, p, n, (void*)c->stream);
        return r ? fail(c, CAMSYNC_E_COMM, "the all-reduce callback failed with %d", r) : CAMSYNC_OK;
    };
    double* const hp = c->h_pin;

    // ---- cvInitIntrinsicParams2D: 2x2 normal equations for (1/fx^2, 1/fy^2), summed over ranks ----
    const double cx0 = (!W) ? 0.5 : (W - 1) * 0.5, cy0 = (!H) ? 0.5 : (H - 1) * 0.5;
    if (V > 0) launch_calib_focal_rows(s, cx0, cy0, c->d_focalRows, work + kWFocal, c->stream);
    else HIPCHK(c, hipMemsetAsync(work + kWFocal, 0, sizeof(double) * 6, c->stream));
    HIPCHK(c, hipGetLastError());
    rc = allreduce(work + kWFocal, 6);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(hp, work + kWFocal, sizeof(double) * 6, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double det = hp[0] * hp[2] - hp[1] * hp[1];
    if (!(fabs(det) > 0)) return fail(c, CAMSYNC_E_NUMERIC, "focal length initialisation is singular");
    const double f0 = (hp[3] * hp[2] - hp[4] * hp[1]) / det, f1 = (hp[0] * hp[4] - hp[1] * hp[3]) / det;
    double intr[12] = {sqrt(fabs(1. / f0)), sqrt(fabs(1. / f1)), cx0, cy0, 0, 0, 0, 0, 0, 0, 0, 0};
    rc = camsync_calib_set_intrinsics(c, intr, 1);
    if (rc) return rc;

    // One kernel and one all-reduce per trial step (k_calib.hip): the kernel that evaluates a trial also forms the
    // trial parameters' record, damped as the next iteration will want it if the trial is accepted.
    auto lambda_of = [](int lg10) { return exp(lg10 * log(10.)); };
    // On one GPU (no all-reduce) the kernel's last workgroup reports what the loop decides from straight into host
    // memory and the host polls a sequence number: no copy to enqueue, no stream to wait for.
    double* d_report = nullptr;
    if (!ar) HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&d_report), c->h_report, 0));
    auto evaluate = [&](int mode, int lg10) -> int {
        launch_calib_lm(s, mode, lambda_of(lg10), work, c->stream, d_report, ++c->calSeq);
        HIPCHK(c, hipGetLastError());
        return allreduce(work + kWAcc, L + 2);
    };
    // hp <- [err(trial), change sums of the extrinsics, | change sums of the intrinsics, flag, old err] of the last evaluate
    auto read_back = [&](double* out7) -> int {
        if (d_report) {
            volatile unsigned long long* const seq = reinterpret_cast<volatile unsigned long long*>(c->h_report + 8);
            for (unsigned long spins = 0; *seq != c->calSeq; spins++) {
                if ((spins & 0xFFFF) == 0xFFFF) {  // (a kernel that failed never reports: ask the stream now and then)
                    const hipError_t q = hipStreamQuery(c->stream);
                    if (q != hipErrorNotReady && *seq != c->calSeq)
                        return fail(c, CAMSYNC_E_HIP, "the calibration kernel ended without its report: %s", hipGetErrorString(q));
                }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            for (int i = 0; i < 7; i++) out7[i] = c->h_report[i];
            return CAMSYNC_OK;
        }
        HIPCHK(c, hipMemcpyAsync(hp, work + kWAcc + L - 1, sizeof(double) * (kWRec + 4 - (L - 1)), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < 3; i++) out7[i] = hp[i];
        for (int i = 0; i < 4; i++) out7[3 + i] = hp[kWRec - (L - 1) + i];
        return CAMSYNC_OK;
    };
    int lambdaLg10 = -3, it = 0;
    double prevErr = 0, errNorm = 0;
    rc = evaluate(0, lambdaLg10);
    if (rc) return rc;
    for (;;) {
        // trial step from the record at the current parameters; hp <- [err(trial), change sums | replicated values]
        const int nextLg10 = std::max(lambdaLg10 - 1, -16);
        rc = evaluate(1, nextLg10);
        if (rc) return rc;
        double r7[7];
        rc = read_back(r7);
        if (rc) return rc;
        const double* const rep = r7 + 3;
        if (rep[2] != 0.0) return fail(c, CAMSYNC_E_NUMERIC, "intrinsic block is singular");
        if (it == 0) prevErr = rep[3];
        errNorm = r7[0];
        if (errNorm > prevErr && ++lambdaLg10 <= 16) {
            // rejected: the same parameters with more damping
            rc = evaluate(0, lambdaLg10);
            if (rc) return rc;
            continue;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        // relative parameter change ||p - p_prev|| / ||p_prev|| (extrinsics of all ranks, intrinsics once)
        const double change = sqrt(r7[1] + rep[0]) / sqrt(r7[2] + rep[1]);
        calib_swap(s);
        if (++it >= max_iter || change < 2.220446049250313e-16) break;
        prevErr = errNorm;
        if (lambdaLg10 != nextLg10) {  // (only when the damping had reached its ceiling)
            rc = evaluate(0, lambdaLg10);
            if (rc) return rc;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double out[12];
    rc = camsync_calib_get(c, out, rvecs, tvecs);
    if (rc) return rc;
    for (int i = 0; i < 9; i++) K[i] = 0;
    K[0] = out[0]; K[4] = out[1]; K[2] = out[2]; K[5] = out[3]; K[8] = 1;
    for (int i = 0; i < NI - 4; i++) dist[i] = out[4 + i];
    if (rms) *rms = sqrt(errNorm / ((double)Vtotal * P));
    if (iters) *iters = it;
    return CAMSYNC_OK;
}

int camsync_calibrate(camsync_ctx* c, const float* obj, const float* img, int V, int P, int W, int H, int flags,
                      int max_iter, double* K, double* dist, double* rvecs, double* tvecs, double* rms, int32_t* iters) {
    if (c && V < 1) return fail(c, CAMSYNC_E_ARG, "need at least one view and K/dist outputs");
    return calibrate_impl(c, obj, img, V, P, V, W, H, flags, max_iter, nullptr, nullptr, nullptr, K, dist, rvecs, tvecs, rms, iters);
}

int camsync_calibrate_dist(camsync_ctx* c, const float* obj, const float* img, int V_local, int P, int64_t V_total, int W,
                           int H, int flags, int max_iter, camsync_allreduce_fn allreduce, void* user, double* d_work,
                           double* K, double* dist, double* rvecs, double* tvecs, double* rms, int32_t* iters) {
    return calibrate_impl(c, obj, img, V_local, P, (long long)V_total, W, H, flags, max_iter, allreduce, user, d_work, K, dist,
                          rvecs, tvecs, rms, iters);
}

/* ---- timers ---- */

int camsync_enable_timers(camsync_ctx* c, int on) {
    if (!c) return CAMSYNC_E_ARG;
    c->timing = on != 0;
    return CAMSYNC_OK;
}

int camsync_get_timers(camsync_ctx* c, double ms[CAMSYNC_N_STAGES], int64_t launches[CAMSYNC_N_STAGES]) {
    if (!c) return CAMSYNC_E_ARG;
    for (int i = 0; i < CAMSYNC_N_STAGES; i++) {
        if (ms) ms[i] = c->ms[i];
        if (launches) launches[i] = c->launches[i];
    }
    return CAMSYNC_OK;
}

int camsync_reset_timers(camsync_ctx* c) {
    if (!c) return CAMSYNC_E_ARG;
    for (int i = 0; i < CAMSYNC_N_STAGES; i++) { c->ms[i] = 0; c->launches[i] = 0; }
    return CAMSYNC_OK;
}

}  // extern "C"
