/*
 * oracle/oracle.h — CPU restatement of the camera_sync vision hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cameracalibration_amd/ may include,
 * link or load this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * What it restates: the arithmetic *inside* the OpenCV calls made by the
 * reference (camera_sync/aruco.py:246-264 detectAruco, :352-404 PnP,
 * :82-119 estimate_pose_single_markers; camera_sync/camera.py:61-63
 * calibrateCamera).  That arithmetic lives in the un-vendored third-party wheel
 * opencv-contrib-python 4.11.0.86 (reference uv.lock:925-926), absent here, so
 * this file restates OpenCV 4.11's published algorithms (objdetect/aruco,
 * imgproc findContours / approxPolyDP / cornerSubPix / warpPerspective,
 * calib3d solvePnP / calibrateCamera) from their documented behaviour.
 *
 * Pins (tests/test_oracle_goldens.py, tests/test_chessboard.py):
 *   P1  tests/golden/pic_{1,2}.png -> ids {10..15} + 96 corner floats, 1e-3 px
 *       (reference camera_sync/test_aruco.py:203-326)                              PINNED
 *   P2  pic_2 + Logitec_A_1280-720_0.npz -> rvec/tvec, 1e-4
 *       (reference camera_sync/test_aruco.py:329-359)                              PINNED
 *   P3  Rodrigues both ways (reference camera_sync/test_referential.py)            PINNED
 *   P4  dictionary ids 10..15, 20..47 from the reference's printable sheets        PINNED
 *   P5  pics/calib/ *.jpg -> calibration/robot_640-480_500.npz (chessboard corners,
 *       cornerSubPix, calibrateCamera end to end)                                  PINNED
 *   P6  pics/blender/calib/ *.png -> calibration/blender_1920-1080_0.npz            PINNED
 *   P7  dictionary ids 0..3 from pics/robot_calib (the v1 sheet's small markers)   PINNED
 *   IPPE_SQUARE numerics, undistort pixels, YUYV conversion, the `rejected` list,
 *   output order, dictionary ids 4..9, 16..19, 48, 49:                            parity unpinned
 */
#ifndef CAMSYNC_ORACLE_H
#define CAMSYNC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Detector parameters in effect for aruco.py:249-254 (OpenCV 4.11 defaults +
 * the reference's four overrides).  Same field order as include/camsync.h. */
typedef struct orc_params {
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
    int32_t cornerRefinementMethod;        /* 1 = SUBPIX (aruco.py:251) */
    int32_t cornerRefinementWinSize;       /* 20 (aruco.py:252) */
    float   relativeCornerRefinmentWinSize;/* 0.3 */
    int32_t cornerRefinementMaxIterations; /* 100 (aruco.py:254) */
    double  cornerRefinementMinAccuracy;   /* 0.01 (aruco.py:253) */
    int32_t perspectiveRemovePixelPerCell; /* 4 */
    double  perspectiveRemoveIgnoredMarginPerCell; /* 0.13 */
    double  maxErroneousBitsInBorderRate;  /* 0.35 */
    double  minOtsuStdDev;                 /* 5.0 */
    double  errorCorrectionRate;           /* 0.6 */
} orc_params;

void orc_default_params(orc_params* p);

/* K1: cv.cvtColor(BGR2GRAY) (aruco.py:246). */
void orc_bgr2gray(const uint8_t* bgr, int h, int w, uint8_t* gray);

/* K2: adaptiveThreshold(MEAN_C, BINARY_INV, win, C) inside detectMarkers. */
void orc_adaptive_threshold(const uint8_t* gray, int h, int w, int win, double C, uint8_t* out);

/* K3: findContours(RETR_LIST, CHAIN_APPROX_NONE).  Returns the number of
 * contours; fills (up to the caps) starts[i] = offset of contour i in pts,
 * lens[i] = point count, pts = x,y int32 pairs.  Order = OpenCV's. */
int orc_find_contours(const uint8_t* bin, int h, int w,
                      int32_t* starts, int32_t* lens, int max_contours,
                      int32_t* pts, int max_pts, int32_t* total_pts);

/* approxPolyDP(closed) on one int contour; returns vertex count. */
int orc_approx_poly_dp(const int32_t* pts, int n, double eps, int32_t* out, int max_out);

/* K3+K4 for one threshold image: quad candidates (4 corners, f32) in OpenCV's
 * order, before the clockwise reorder.  Also returns each candidate's contour
 * start pixel key (y*w+x of the first contour point) and contour length. */
int orc_quad_candidates(const uint8_t* bin, int h, int w, const orc_params* p,
                        float* quads, int32_t* start_key, int32_t* clen, int max_quads);

/* Whole detectMarkers (K1..K7) on a u8 frame, channels = 1 (gray) or 3 (BGR).
 * ids[i], corners[i*8..] for accepted markers in OpenCV's output order;
 * rejected likewise.  Returns the number accepted (or <0 on error). */
int orc_detect(const uint8_t* img, int h, int w, int channels, const orc_params* p,
               const uint16_t* dict_codes, int dict_n,
               int32_t* ids, float* corners, int max_markers,
               float* rejected, int max_rejected, int32_t* n_rejected);

/* Marker identification only (K6) for given corners on a gray image.
 * Returns 1 when identified; id / rotation written. */
int orc_identify(const uint8_t* gray, int h, int w, const float* corners4,
                 const orc_params* p, const uint16_t* dict_codes, int dict_n,
                 int32_t* id, int32_t* rotation, uint8_t* bits36);

/* cornerSubPix (K7) on n points, in place. */
void orc_corner_subpix(const uint8_t* gray, int h, int w, float* pts, int n,
                       int win, int max_iter, double eps);

/* K13 */
void orc_rodrigues_vec2mat(const double r[3], double R[9], double* jac /*9x3 dR/dr or NULL*/);
void orc_rodrigues_mat2vec(const double R[9], double r[3]);

/* K11: projectPoints, dist = k1,k2,p1,p2,k3,k4,k5,k6 (ndist of them used, rest 0). */
void orc_project_points(const double* obj, int n, const double rvec[3], const double tvec[3],
                        const double K[9], const double* dist, int ndist, double* img_out);

/* K8+K9: cv.solvePnP(ITERATIVE) + cv.solvePnPRefineLM (aruco.py:368-384).
 * obj f64 [n,3], img f64 [n,2] (caller converts from f32, as cv does).
 * Returns 1 on success. */
int orc_solve_pnp(const double* obj, const double* img, int n, const double K[9],
                  const double* dist, int ndist, int refine,
                  double rvec[3], double tvec[3]);

/* K10: cv.solvePnP(SOLVEPNP_IPPE_SQUARE) for a square of side `size`
 * (aruco.py:97-117); corners f64 [4,2]. */
int orc_ippe_square(const double* corners, double size, const double K[9],
                    const double* dist, int ndist, double rvec[3], double tvec[3]);

/* K12: cv.calibrateCamera (camera.py:61-63).  obj [V,P,3], img [V,P,2] f64.
 * flags: 0 = 5-coef model, 1 = CALIB_RATIONAL_MODEL (8 coef).
 * dist_out has 5 or 8 entries.  Returns RMS reprojection error (px). */
double orc_calibrate(const double* obj, const double* img, int V, int P, int W, int H,
                     int flags, int max_iter, double K_out[9], double* dist_out,
                     double* rvecs, double* tvecs, int32_t* iters_out);

/* K12 in pieces, for the multi-process tests: the same arithmetic as orc_calibrate split the way
 * the distributed driver (cameracalibration_amd/calibrate_dist.py) needs it.  Each handle owns a
 * contiguous share of the views; everything returned is a SUM over the handle's views, so summing over
 * handles (the all-reduce) gives the global quantity.
 *   focal_acc : [AtA00, AtA01, AtA11, Atb0, Atb1, 0] of cvInitIntrinsicParams2D's 2x2 system
 *   accumulate: [S (NI x NI) | rhs (NI) | err] — the extrinsic blocks eliminated (Schur complement),
 *               diagonals scaled by (1 + lambda) as CvLevMarq does
 *   step      : trial = current - delta (intrinsics, back-substituted extrinsics); returns the trial error
 */
void* orc_calib_begin(const double* obj, const double* img, int V, int P, int flags);
void orc_calib_end(void* h);
void orc_calib_focal_acc(void* h, int W, int H, double acc6[6]);
void orc_calib_set_intrinsics(void* h, const double intr[12], int init_extrinsics);
void orc_calib_accumulate(void* h, double lambda, double* acc);
double orc_calib_step(void* h, const double* delta);
void orc_calib_accept(void* h);
void orc_calib_get(void* h, double intr[12], double* rvecs, double* tvecs);

/* Camera.undistort (reference camera_sync/camera.py:245-255): cv.getOptimalNewCameraMatrix and cv.undistort
 * (oracle/undistort.cpp).  PARITY UNPINNED: the reference holds no output of either. */
int orc_optimal_new_camera_matrix(const double* K, const double* dist, int ndist, int W, int H, double alpha, int newW,
                                  int newH, double* newK, int32_t* roi);
int orc_undistort(const uint8_t* src, int H, int W, int C, const double* K, const double* dist, int ndist,
                  const double* newK, uint8_t* dst);

/* CPU twin of this repository's chessboard corner-candidate stage (oracle/chessboard.cpp; NOT a restatement of
 * cv.findChessboardCorners, see that file). */
int orc_chess_candidates(const uint8_t* gray, int H, int W, int scale, int max_cand, int32_t* xy, int32_t* strength);

/* Frame ingest (reference camera_sync/camera.py:104-118, 150-154: VideoCapture.read()): packed YUYV 4:2:2 as a UVC
 * camera delivers it -> BGR, as cv.cvtColor(COLOR_YUV2BGR_YUYV) (oracle/ingest.cpp).  PARITY UNPINNED. */
int orc_yuyv_to_bgr(const uint8_t* yuyv, int H, int W, uint8_t* bgr);

/* Frame ingest, compressed (same call sites: VideoCapture.read() on an MJPG camera; cv.imread of the reference's
 * pics/ *.jpg): baseline JPEG -> BGR as libjpeg-turbo decodes it with its defaults (oracle/jpeg.cpp).  Pinned against
 * Pillow's libjpeg-turbo on the reference's 21 JPEG photos, not against a reference-held output.
 * orc_jpeg_info: size of the image; orc_jpeg_decode: bgr u8 [H,W,3].  0, or < 0: -1 not a JPEG / truncated headers,
 * -2 outside the scope (progressive, arithmetic, 4 components, other samplings), -3 corrupt entropy data, -4 bgr too small. */
int orc_jpeg_info(const uint8_t* data, size_t n, int32_t* height, int32_t* width, int32_t* components);
int orc_jpeg_decode(const uint8_t* data, size_t n, uint8_t* bgr, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
