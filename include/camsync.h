/*
 * camsync.h — C ABI of libcamsync_hip.so, the MI355X (gfx950) implementation of
 * camera_sync's per-frame vision hot path.
 *
 * Each entry point replaces one OpenCV call made by the reference
 * (6figuress/CameraCalibration, package camera_sync); the citation next to it
 * is the reference call site it is a drop-in for.  The reference-side binding
 * is a ctypes stub; cameracalibration_amd/camera_sync/ is the
 * Python mirror of the reference modules built on that stub.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative CAMSYNC_E_* code; camsync_last_error() gives the text.
 * A ctx owns one HIP stream and all scratch memory; use one ctx per host
 * thread / GPU.  Calls are synchronous (they return after the stream drained)
 * unless stated.  Buffers flagged `on_device` are device pointers on the
 * ctx's GPU (e.g. torch tensor .data_ptr()); otherwise host pointers.
 */
#ifndef CAMSYNC_H
#define CAMSYNC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAMSYNC_OK 0
#define CAMSYNC_E_ARG (-1)      /* bad argument */
#define CAMSYNC_E_HIP (-2)      /* HIP runtime error */
#define CAMSYNC_E_CAPACITY (-3) /* a per-frame scratch list overflowed */
#define CAMSYNC_E_NODEVICE (-4) /* no usable GPU */
#define CAMSYNC_E_NUMERIC (-5)  /* solver failed (reference raises "Could not solve PnP") */
#define CAMSYNC_E_COMM (-6)     /* the caller's all-reduce callback reported an error */

typedef struct camsync_ctx camsync_ctx;

/* cv.aruco.DetectorParameters as the reference configures it
 * (camera_sync/aruco.py:249-254); defaults of OpenCV 4.11 for the rest. */
typedef struct camsync_params {
    int32_t adaptiveThreshWinSizeMin;      /* 3  */
    int32_t adaptiveThreshWinSizeMax;      /* 23 */
    int32_t adaptiveThreshWinSizeStep;     /* 10 */
    double  adaptiveThreshConstant;        /* 7  */
    double  minMarkerPerimeterRate;        /* 0.03 */
    double  maxMarkerPerimeterRate;        /* 4.0  */
    double  polygonalApproxAccuracyRate;   /* 0.03 */
    double  minCornerDistanceRate;         /* 0.05 */
    int32_t minDistanceToBorder;           /* 3 */
    int32_t markerBorderBits;              /* 1 */
    double  minMarkerDistanceRate;         /* 0.125 */
    float   minGroupDistance;              /* 0.21 */
    int32_t cornerRefinementMethod;        /* 1 = CORNER_REFINE_SUBPIX (aruco.py:251) */
    int32_t cornerRefinementWinSize;       /* 20  (aruco.py:252) */
    float   relativeCornerRefinmentWinSize;/* 0.3 */
    int32_t cornerRefinementMaxIterations; /* 100 (aruco.py:254) */
    double  cornerRefinementMinAccuracy;   /* 0.01 (aruco.py:253) */
    int32_t perspectiveRemovePixelPerCell; /* 4 */
    double  perspectiveRemoveIgnoredMarginPerCell; /* 0.13 */
    double  maxErroneousBitsInBorderRate;  /* 0.35 */
    double  minOtsuStdDev;                 /* 5.0 */
    double  errorCorrectionRate;           /* 0.6 */
} camsync_params;

/* ---- context -------------------------------------------------------- */

/* Library / device probe: number of visible GPUs, <0 on error. */
int camsync_device_count(void);

/* Create a context on `device` sized for frames up to max_h x max_w and
 * chunks of max_batch frames (larger batches are processed in chunks).
 * Replaces the per-call construction of cv.aruco.ArucoDetector
 * (aruco.py:247-262). */
int camsync_create(camsync_ctx** out, int device, int max_h, int max_w, int max_batch);
void camsync_destroy(camsync_ctx* ctx);
const char* camsync_last_error(const camsync_ctx* ctx);

/* Frame ingest (camera.py:104-118, :150-154: cv.VideoCapture hands the reference pageable numpy frames).  Host
 * buffers from camsync_host_alloc are page-locked: a capture or decode library that writes its frames there (V4L2
 * user pointers, a decoder's output surface) spares the staging copy the driver makes of pageable memory, and the
 * host-to-device copies of the *_batch entry points run at the link's rate (measured: DESIGN.md section 6).  Any
 * host pointer is accepted by those entry points; these two calls only provide the fast kind (frames in ordinary
 * memory are staged by the detection entry points themselves: up to eight short-lived threads of the calling process
 * copy them piecewise into a page-locked buffer of 64 MB that the context keeps).  The memory is not
 * tied to a context (it may be shared by the contexts of several GPUs). */
int camsync_host_alloc(void** out, size_t bytes);
int camsync_host_free(void* p);

/* What VideoCapture.read() does before the reference sees a frame (camera.py:150-154, capture opened at :98-118): a
 * UVC camera delivers packed YUYV 4:2:2 (V4L2_PIX_FMT_YUYV, bytes Y0 U Y1 V per pixel pair) and OpenCV converts it to
 * BGR on the CPU (cvtColor COLOR_YUV2BGR_YUYV: BT.601 limited range, 20-bit fixed point).  Here the raw frames are
 * converted on the GPU: yuyv u8 [B,H,W,2] on the host or (in_on_device) on the GPU, 16-byte aligned there; bgr u8
 * [B,H,W,3] on the GPU (out_on_device, 8-byte aligned: ready for camsync_detect_pnp_resident and the on_device form
 * of every *_batch entry point) or on the host.  W must be a multiple of 8.  A raw frame is two thirds of a BGR one,
 * so the host-to-device copy shrinks by a third as well. */
int camsync_ingest_yuyv(camsync_ctx* ctx, const uint8_t* yuyv, int in_on_device, int B, int H, int W,
                        uint8_t* bgr, int out_on_device);

/* Compressed frames: what cv.VideoCapture(...).read() does on a camera that delivers MJPG, and cv.imread on the
 * reference's stored photos (camera.py:98-118, 150-154; pics/ *.jpg): baseline JPEG -> BGR, decoded as libjpeg-turbo
 * does with its defaults (accurate integer IDCT, triangle-filter chroma upsampling, fixed-point YCbCr -> RGB).
 * jpegs: B host pointers to one JPEG file / MJPEG frame each, sizes[B] their lengths; all frames H x W (the SOF size;
 * camsync_jpeg_info reads it from a header, host only, no context).  bgr u8 [B,H,W,3], on the host or (out_on_device,
 * 4-byte aligned) left on the GPU for the on_device / _resident entry points.  Taken: SOF0 / SOF1, 8-bit, one
 * interleaved scan, 1 or 3 components (Y / YCbCr), 4:4:4 / 4:2:2 / 4:2:0, restart intervals, frames without DHT segments
 * (the Annex K tables); anything else (progressive, arithmetic coding, CMYK, ...) is CAMSYNC_E_ARG with a message naming
 * the frame, and so is corrupt entropy-coded data.  A frame's Huffman stream is decoded by many lanes at once
 * (subsequences that synchronise themselves, or its restart intervals side by side), so a single frame takes about a
 * millisecond at 1080p; batches of 256 frames and more are decoded in four passes, the next pass's bytes travelling
 * while this one's kernels run.  An output on the device is written fastest when it is 8-byte aligned. */
int camsync_jpeg_info(const uint8_t* data, size_t n, int32_t* height, int32_t* width, int32_t* components);
int camsync_decode_jpeg_batch(camsync_ctx* ctx, const uint8_t* const* jpegs, const size_t* sizes, int B, int H, int W,
                              uint8_t* bgr, int out_on_device);

/* cv.aruco.DetectorParameters() (aruco.py:249-254). */
void camsync_default_params(camsync_params* p);
int camsync_set_params(camsync_ctx* ctx, const camsync_params* p);
int camsync_get_params(const camsync_ctx* ctx, camsync_params* p);

/* cv.aruco.getPredefinedDictionary (aruco.py:247): n codes, each the
 * marker_size*marker_size inner bits of rotation 0, row-major, MSB first.
 * Rotations are derived on device.  marker_size <= 4 (16-bit codes). */
int camsync_set_dictionary(camsync_ctx* ctx, const uint16_t* codes, int n, int marker_size,
                           int max_correction_bits);

/* ---- detection ------------------------------------------------------ */

/* cv.cvtColor(BGR2GRAY) + the adaptiveThreshold passes of detectMarkers
 * (aruco.py:246, :256-264 step 1a).  frames u8 [B,H,W,C], C in {1,3}.
 * gray_out u8 [B,H,W]; bin_out u8 [B,H,W], bit s = threshold image of scale s
 * (window = WinSizeMin + s*Step).  Either output may be NULL.  Outputs follow
 * `on_device` like the input. */
int camsync_gray_threshold(camsync_ctx* ctx, const uint8_t* frames, int on_device,
                           int B, int H, int W, int C, uint8_t* gray_out, uint8_t* bin_out);

/* cv.cvtColor + ArucoDetector.detectMarkers for a batch (aruco.py:246-264).
 * ids i32 [B,max_markers], corners f32 [B,max_markers,4,2], counts i32 [B];
 * rejected f32 [B,max_rejected,4,2] and rej_counts i32 [B] may be NULL.
 * Marker order per frame = the reference detector's output order.
 * Outputs are host pointers. */
int camsync_detect_batch(camsync_ctx* ctx, const uint8_t* frames, int on_device,
                         int B, int H, int W, int C,
                         int max_markers, int32_t* ids, float* corners, int32_t* counts,
                         int max_rejected, float* rejected, int32_t* rej_counts);

/* The outputs of camsync_detect_batch / camsync_detect_pnp_batch / _frames hold at most max_markers (max_rejected)
 * entries per frame and the counts are clamped to that.  camsync_last_extent reports the largest unclamped count of
 * any frame of the last such call on this context, so that a caller who must not lose a marker (the reference's
 * detectMarkers returns every one, aruco.py:256-264) can call again with wider outputs.  Either pointer may be NULL. */
int camsync_last_extent(const camsync_ctx* ctx, int32_t* markers, int32_t* rejected);

/* Intermediate results of the last chunk processed (parity tests and tuning, no counterpart in the
 * reference: OpenCV does not expose them).
 *   camsync_debug_counts   chunk-wide counters: [0] contour-start candidates, [1] contours kept by the
 *                          perimeter filter, [2] contour points stored, [3] markers accepted, [4] overflow
 *                          flags (0 = no scratch list overflowed), [5] quads of `frame`, [6] candidates that
 *                          went to the long-border pass, [7] work-queue head of that pass
 *   camsync_debug_quads    quad candidates of `frame` after approxPolyDP and the size/convexity filters,
 *                          before duplicate removal: corners f32 [n,4,2] (clockwise), threshold scale, raster
 *                          key of the contour start, contour length, identified id (-1 = rejected), rotation
 *   camsync_debug_contours contours of `frame`: threshold scale, raster key of the start pixel, hole flag,
 *                          length in points
 *   camsync_debug_points   the points of those contours, concatenated in the same order: xy i32 [n,2]
 * The quad/contour/point calls return the number of elements available (may exceed the capacity passed), or <0. */
int camsync_debug_counts(camsync_ctx* ctx, int frame, int32_t out[8]);
/* Fill level and capacity of the chunk-wide scratch lists after the last chunk (capacity planning; a list that is
 * fuller than its capacity made the call fail with CAMSYNC_E_CAPACITY): [0] contour-start candidates, [1] candidates
 * of the long-border pass, [2] contour records, [3] contour points, [4] store tasks (short borders), [5] border
 * segments (a record and a 128-byte chain chunk each), [6] copy tasks, [7] border seeds.  The counters include the
 * blocks the border-follower waves reserve and leave partly unused. */
int camsync_debug_lists(camsync_ctx* ctx, uint32_t used[8], uint32_t cap[8]);
/* The threshold image of `frame` at `scale` as the detection pipeline of the last chunk produced it (one bit per
 * pixel on the device; here one byte per pixel, 0 or 1, [H,W]): cv.adaptiveThreshold(MEAN_C, BINARY_INV) with the
 * image's outermost pixel frame cleared, which is what findContours sees. */
int camsync_debug_planes(camsync_ctx* ctx, int frame, int scale, uint8_t* out_bits);
/* The first two stages alone (gray + thresholds, contour-start candidates) on B host frames (B <= max_batch): the
 * candidate list, unordered, each entry frame << 40 | hole << 39 | scale << 36 | (y * W + x) of the start pixel.
 * Returns the number of candidates found (may exceed max_starts), or < 0. */
int camsync_debug_starts(camsync_ctx* ctx, const uint8_t* frames, int B, int H, int W, int C, int max_starts, uint64_t* out);
int camsync_debug_quads(camsync_ctx* ctx, int frame, int max_quads, float* corners,
                        int32_t* scale, int32_t* scan_key, int32_t* contour_len,
                        int32_t* id, int32_t* rot);
int camsync_debug_contours(camsync_ctx* ctx, int frame, int max_contours, int32_t* scale,
                           int32_t* start_key, int32_t* is_hole, int32_t* length);
int camsync_debug_points(camsync_ctx* ctx, int frame, int max_points, int32_t* xy);

/* Testing aid, no counterpart in the reference: lower the capacities of the contour-record, contour-point and
 * store-task lists of a context (0 = leave as is; values above the allocated capacity are clamped to it), so that
 * the overflow paths, which otherwise need gigabytes of contours, can be exercised with one small frame.
 * camsync_debug_restore_capacity puts the allocated capacities back. */
int camsync_debug_limit_capacity(camsync_ctx* ctx, unsigned int contours, unsigned int points, unsigned int tasks);
int camsync_debug_restore_capacity(camsync_ctx* ctx);
/* The pipeline identifies only the quad candidates the selection can ask for (as the reference detector does: the
 * representative of each group of near-duplicates, its fall-backs only when it is not a marker).  on != 0 makes the
 * following calls identify every candidate, so that camsync_debug_quads reports an id for each. */
int camsync_debug_identify_all(camsync_ctx* ctx, int on);

/* The detection's scratch lists (start candidates, contour records, contour points, store / copy tasks, border seeds;
 * per frame: 1024 quad candidates and 64 accepted markers) are sized at camsync_create; a chunk that overflows one is
 * run again with that list grown (doubled; the per-frame lists to what the fullest frame needs) -- the reference's
 * cv.aruco.detectMarkers has no such limits, aruco.py:256-264 -- and CAMSYNC_E_CAPACITY is left for a list that cannot
 * grow any more (2^31 entries, 32768 quad candidates in one frame, or no device memory).  Frames with more than 1024
 * quad candidates take a slower selection kernel (working set in global memory instead of LDS).
 * camsync_debug_shrink_lists replaces the lists by smaller ones (entries; 0 = leave) so that tests can make that
 * happen; camsync_debug_regrown counts the growths since camsync_create. */
int camsync_debug_shrink_lists(camsync_ctx* ctx, unsigned int starts, unsigned int contours, unsigned int points,
                               unsigned int tasks, unsigned int seeds);
int camsync_debug_regrown(const camsync_ctx* ctx);

/* ---- pose ----------------------------------------------------------- */

/* cv.solvePnP(ITERATIVE) followed by cv.solvePnPRefineLM, batched
 * (aruco.py:368-384), plus the MAE/RMSE of aruco.py:386-400.
 * obj f32 [B,N,3], img f32 [B,N,2] (4 <= N <= 256: a wavefront per frame, up to
 * four points per lane), npts i32 [B] (<= N), K f64 [B or 1,3,3],
 * dist f64 [B or 1, ndist] (ndist <= 8: k1,k2,p1,p2,k3,k4,k5,k6),
 * per_frame_cam: 1 if K/dist are per frame.  Coplanar points start from a
 * homography, other point sets (at least 6) from a DLT of the projection, as
 * cvFindExtrinsicCameraParams2 does.  Outputs (host): rvec,tvec f64 [B,3],
 * ok u8 [B] (1 solved, 0 no solution / fewer than 4 points, 2 fewer than 6
 * points that are not coplanar), mae,rmse f64 [B] (may be NULL). */
int camsync_pnp_batch(camsync_ctx* ctx, const float* obj, const float* img, const int32_t* npts,
                      int B, int N, const double* K, const double* dist, int ndist,
                      int per_frame_cam, double* rvec, double* tvec, uint8_t* ok,
                      double* mae, double* rmse);

/* camsync_pnp_batch for callers that must not allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and results as
 * camsync_pnp_batch, but the inputs and results are staged in a fixed device array that lives in the library image
 * (about 400 KiB per GPU, held until the process ends) instead of buffers that grow with the batch, so the call makes
 * no hipMalloc / hipFree.  Limits: B <= CAMSYNC_NOALLOC_MAX_FRAMES and B * N <= CAMSYNC_NOALLOC_MAX_POINTS; larger
 * batches fail with CAMSYNC_E_CAPACITY (use camsync_pnp_batch).  All contexts of one GPU share that array: calls that
 * overlap in time on the same GPU race (serialise them); calls on different GPUs are independent. */
#define CAMSYNC_NOALLOC_MAX_FRAMES 256
#define CAMSYNC_NOALLOC_MAX_POINTS 16384
int camsync_pnp_batch_noalloc(camsync_ctx* ctx, const float* obj, const float* img, const int32_t* npts,
                              int B, int N, const double* K, const double* dist, int ndist,
                              int per_frame_cam, double* rvec, double* tvec, uint8_t* ok,
                              double* mae, double* rmse);

/* The per-marker half of processArucoFromMultipleCameras (aruco.py:417-470) for the cameras of ONE batch, on the
 * device: works on the detections and camera poses that the camsync_detect_pnp_batch / camsync_detect_pnp_frames call
 * just before it left on this context (n_frames must be that call's batch, which must have been one chunk:
 * B <= max_batch).  For every frame whose camera was located (pose_ok == 1) and every moving marker it sees (the last
 * of several detections with that id, as detectAruco's dictionary keeps it, aruco.py:270-271): the marker's pose by
 * IPPE_SQUARE with that frame's camera (estimate_pose_single_markers, aruco.py:82-119), aruco2world = cam2world @
 * aruco2cam (locateAruco, aruco.py:288-290), applied to the corners (-s/2, s/2, 0), (s/2, s/2, 0), (s/2, -s/2, 0),
 * (-s/2, -s/2, 0) (aruco.py:434-441); per marker the mean over the cameras (np.mean(axis=0), aruco.py:468).
 * fused_corners f64 [n_moving,4,3] (zero where seen == 0), seen i32 [n_moving] = cameras that located the marker;
 * optional: marker_rvec / marker_tvec f64 [n_frames,n_moving,3] (aruco2cam) and located u8 [n_frames,n_moving].
 * The caller keeps the stateful part, the low-pass filter of Aruco.corners (aruco.py:469-470, position.py:34-37). */
int camsync_fuse_moving_markers(camsync_ctx* ctx, int n_frames, const int32_t* moving_ids, const double* moving_sizes,
                                int n_moving, double* fused_corners, int32_t* seen, double* marker_rvec,
                                double* marker_tvec, uint8_t* located);

/* cv.solvePnP(SOLVEPNP_IPPE_SQUARE) for B squares (aruco.py:97-117 inside
 * estimate_pose_single_markers).  corners f32 [B,4,2] (TL,TR,BR,BL),
 * size f64 [B], K/dist as above. */
int camsync_ippe_square_batch(camsync_ctx* ctx, const float* corners, const double* size, int B,
                              const double* K, const double* dist, int ndist, int per_frame_cam,
                              double* rvec, double* tvec, uint8_t* ok);

/* detectAruco + anchor gather + PnP fused on device: the body of processAruco
 * up to camera.updateWorldPosition (aruco.py:305-336) for B frames.
 * anchors: n_anchor <= 64 ids with their 4 object corners f32 [n_anchor,4,3]
 * (Aruco.getCornersAsList order, aruco.py:58-67).  Frames that see < 1 anchor
 * (fewer than 4 points) get pose_ok = 0 (processAruco returns False / raises).
 * Detection outputs as camsync_detect_batch. */
int camsync_detect_pnp_batch(camsync_ctx* ctx, const uint8_t* frames, int on_device,
                             int B, int H, int W, int C,
                             const int32_t* anchor_ids, const float* anchor_obj, int n_anchor,
                             const double* K, const double* dist, int ndist, int per_frame_cam,
                             int max_markers, int32_t* ids, float* corners, int32_t* counts,
                             double* rvec, double* tvec, uint8_t* pose_ok, double* mae, double* rmse);

/* camsync_detect_pnp_batch for B frames that do not lie in one buffer -- one frame per camera, as
 * processArucoFromMultipleCameras receives them (aruco.py:407-417: `for c, f in zip(cameras, frame)`): frame_ptrs[i]
 * is frame i, [H,W,C] u8, host or device memory (pinned host memory copies fastest).  The frames are gathered on the
 * GPU and go through one launch sequence; with per_frame_cam = 1 every frame has its own camera matrix and distortion.
 * Results as camsync_detect_pnp_batch. */
int camsync_detect_pnp_frames(camsync_ctx* ctx, const uint8_t* const* frame_ptrs, int on_device,
                              int B, int H, int W, int C,
                              const int32_t* anchor_ids, const float* anchor_obj, int n_anchor,
                              const double* K, const double* dist, int ndist, int per_frame_cam,
                              int max_markers, int32_t* ids, float* corners, int32_t* counts,
                              double* rvec, double* tvec, uint8_t* pose_ok, double* mae, double* rmse);

/* Same pipeline, inputs already resident on the GPU and results left on the
 * GPU (no host copies, returns after the stream drained): the benchmark /
 * serving entry point.  d_* are device pointers laid out as above. */
int camsync_detect_pnp_resident(camsync_ctx* ctx, const uint8_t* d_frames, int B, int H, int W, int C,
                                const int32_t* anchor_ids, const float* anchor_obj, int n_anchor,
                                const double* K, const double* dist, int ndist,
                                int max_markers, int32_t* d_ids, float* d_corners, int32_t* d_counts,
                                double* d_rvec, double* d_tvec, uint8_t* d_pose_ok);

/* ---- chessboard front end of Camera.calibrate ----------------------- */

/* Inner-corner candidates of a chessboard: the part of cv.findChessboardCorners(gray, (7,7)) (camera.py:49-52) that
 * touches pixels.  frames u8 [B,H,W,C] (C = 3 is converted with cv.cvtColor's luma first).  Candidates are X-corner
 * maxima of an integer ChESS response at ring radius 5 * scale, strongest first: xy i32 [B,max_cand,2], strength i32
 * [B,max_cand], counts i32 [B]; more than max_cand candidates in a frame is CAMSYNC_E_CAPACITY.  Assembling the
 * pattern's lattice from them is host logic (cameracalibration_amd/camera_sync/chessboard.py). */
int camsync_chess_candidates(camsync_ctx* ctx, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                             int scale, int max_cand, int32_t* xy, int32_t* strength, int32_t* counts);

/* cv.cornerSubPix(gray, corners, (win,win), (-1,-1), (EPS + MAX_ITER, max_iter, eps)) (camera.py:58) for B frames:
 * pts / out f32 [B,N,2] (host), npts i32 [B] or NULL (N points in every frame); win 1..20.  frames may be BGR: the
 * gray values are converted as they are read. */
int camsync_corner_subpix(camsync_ctx* ctx, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                          const float* pts, const int32_t* npts, int N, int win, int max_iter, double eps, float* out);

/* ---- undistortion --------------------------------------------------- */

/* cv.getOptimalNewCameraMatrix(K, dist, (W,H), alpha, (new_w,new_h)) with centerPrincipalPoint = false
 * (camera.py:246-248; the reference passes alpha = 1 and the same size).  Host arithmetic, no context needed.
 * new_K f64 [9]; roi (may be NULL) = x, y, w, h of validPixROI.  new_w / new_h <= 0 mean W / H. */
int camsync_optimal_new_camera_matrix(const double* K, const double* dist, int ndist, int W, int H, double alpha,
                                      int new_w, int new_h, double* new_K, int32_t* roi);

/* cv.undistort(img, K, dist, None, new_K) (camera.py:251) for B frames of one camera: frames / out u8 [B,H,W,C],
 * C in {1,3}, W a multiple of 4, both on the host or (on_device) both on the GPU; new_K NULL = K.  Bilinear with
 * OpenCV's 5 fractional bits and constant (0) border.  The map of the last (K, dist, new_K, size) is kept in the
 * context, so a stream of frames from one camera pays for it once. */
int camsync_undistort_batch(camsync_ctx* ctx, const uint8_t* frames, int on_device, int B, int H, int W, int C,
                            const double* K, const double* dist, int ndist, const double* new_K, uint8_t* out);

/* ---- calibration ---------------------------------------------------- */

/* cv.calibrateCamera(objpoints, imgpoints, size, None, None)
 * (camera.py:61-63).  obj f32 [V,P,3], img f32 [V,P,2]; flags bit0 =
 * CALIB_RATIONAL_MODEL.  K f64 [9], dist f64 [5 or 8], rvecs/tvecs f64 [V,3]
 * (may be NULL), rms f64, iters i32 (LM iterations run). */
int camsync_calibrate(camsync_ctx* ctx, const float* obj, const float* img, int V, int P,
                      int W, int H, int flags, int max_iter,
                      double* K, double* dist, double* rvecs, double* tvecs,
                      double* rms, int32_t* iters);

/* cv.calibrateCamera (camera.py:61-63) with the views sharded over ranks, one rank per GPU: this rank passes its
 * V_local views (may be 0) and the global count V_total; the Levenberg-Marquardt loop runs on the device (one kernel per
 * trial step) and sums one small record per trial step over the ranks through `allreduce`:
 *     int allreduce(void* user, double* d_buf, int n, void* hip_stream)
 * must replace d_buf[0..n) (DEVICE memory) by its sum over all ranks, ordered after the work already enqueued on
 * hip_stream (a hipStream_t) and before work enqueued on it later, and return 0 -- e.g.
 * ncclAllReduce(d_buf, d_buf, n, ncclDouble, ncclSum, comm, (hipStream_t)hip_stream) with RCCL, or
 * torch.distributed.all_reduce on a tensor aliasing the buffer (cameracalibration_amd/calibrate_dist.py).  n is 6
 * (focal initialisation) or ni*ni + ni + 3 <= 159 (normal equations, squared error, parameter change).  Every rank
 * makes the same sequence of calls.  d_work: optional caller-owned device buffer of CAMSYNC_CALIB_WORK_DOUBLES doubles
 * that then holds every record handed to `allreduce` (so a framework can register it once); NULL = the context's own.
 * allreduce == NULL is the single-GPU case (camsync_calibrate).  Outputs as camsync_calibrate; rvecs/tvecs are this
 * rank's V_local views; K, dist, rms, iters are identical on every rank. */
#define CAMSYNC_CALIB_WORK_DOUBLES 192
typedef int (*camsync_allreduce_fn)(void* user, double* d_buf, int n, void* hip_stream);
int camsync_calibrate_dist(camsync_ctx* ctx, const float* obj, const float* img, int V_local, int P, int64_t V_total,
                           int W, int H, int flags, int max_iter, camsync_allreduce_fn allreduce, void* user,
                           double* d_work, double* K, double* dist, double* rvecs, double* tvecs,
                           double* rms, int32_t* iters);

/* Multi-GPU calibration building blocks (one rank per GPU, views sharded
 * contiguously).  The caller owns the LM loop (cvCalibrateCamera2Internal's
 * CvLevMarq::updateAlt state machine) and all-reduces (sum) the small record
 * between accumulate and step; cameracalibration_amd/calibrate_dist.py is the
 * torch.distributed (RCCL) driver, camsync_calibrate the single-GPU one.
 *   camsync_calib_begin       upload this rank's views (obj f32 [V,P,3], img
 *                             f32 [V,P,2], P <= 64); flags bit0 = rational
 *   camsync_calib_focal_acc   this rank's sums for the closed-form focal
 *                             initialisation (cvInitIntrinsicParams2D):
 *                             [a00 a01 a11 b0 b1 nviews], to be summed over ranks
 *   camsync_calib_set_intrinsics  set fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6; with
 *                             init_extrinsics, also initialise every local
 *                             view's pose (cvFindExtrinsicCameraParams2)
 *   camsync_calib_accumulate  acc f64 [ni*ni + ni + 1] = this rank's Schur
 *                             complement, reduced right-hand side and squared
 *                             error at the current parameters, damping lambda
 *   camsync_calib_solve       host: solve the all-reduced ni x ni system
 *   camsync_calib_step        apply the intrinsic delta, back-substitute the
 *                             local extrinsics into the trial parameters and
 *                             return the local squared error there
 *   camsync_calib_accept      trial parameters become current */
int camsync_calib_begin(camsync_ctx* ctx, const float* obj, const float* img, int V, int P, int flags);
int camsync_calib_focal_acc(camsync_ctx* ctx, int W, int H, double acc6[6]);
int camsync_calib_set_intrinsics(camsync_ctx* ctx, const double intr[12], int init_extrinsics);
int camsync_calib_accumulate(camsync_ctx* ctx, double lambda, double* acc, int acc_on_device);
int camsync_calib_solve(const double* acc, int ni, double* delta);
int camsync_calib_step(camsync_ctx* ctx, const double* delta, double* err_out);
int camsync_calib_accept(camsync_ctx* ctx);
int camsync_calib_get(camsync_ctx* ctx, double intr[12], double* rvecs, double* tvecs);
int camsync_calib_n_intrinsics(const camsync_ctx* ctx);

/* ---- measurement ---------------------------------------------------- */

/* Accumulated hipEvent time (ms) and launch count per stage since the last
 * reset: stage 0 gray+threshold, 1 contour starts, 2 trace, 3 quad fit,
 * 4 identify, 5 per-frame select, 6 subpix, 7 pnp, 8 whole chunk.  Timing is
 * only collected after camsync_enable_timers(ctx, 1). */
#define CAMSYNC_N_STAGES 9
int camsync_enable_timers(camsync_ctx* ctx, int on);
int camsync_get_timers(camsync_ctx* ctx, double ms[CAMSYNC_N_STAGES], int64_t launches[CAMSYNC_N_STAGES]);
int camsync_reset_timers(camsync_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
