/*
 * oracle/detect.cpp — CPU restatement of cv.cvtColor + cv.aruco.ArucoDetector
 * .detectMarkers as called by the reference at camera_sync/aruco.py:246-264
 * (OpenCV 4.11.0.86: modules/objdetect/src/aruco/aruco_detector.cpp and the
 * imgproc routines it calls).  TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * Scalar, sequential, written for clarity; floating-point expression order
 * follows OpenCV's where the result is observable (cornerSubPix).
 * Build with -ffp-contract=off.
 */
#include "oracle.h"

#include <math.h>
#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {

struct Pt { int x, y; };
struct Pt2f { float x, y; };

/* ------------------------------------------------------------------ K1 */
/* imgproc color_rgb: RGB2Gray<uchar>, 15-bit coefficients B 3735, G 19235,
 * R 9798, descale by (x + 2^14) >> 15.  Call site: aruco.py:246. */
inline uint8_t luma_u8(int b, int g, int r) {
    return (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
}

/* ------------------------------------------------------------------ K2 */
/* adaptiveThreshold(ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY_INV): normalised
 * box mean with BORDER_REPLICATE, rounded to nearest (k*k is odd so no ties),
 * out = 255 where src - mean <= -floor(C).  Called from aruco_detector.cpp
 * _threshold() for windows 3,13,23 under aruco.py:249-264. */
void adaptive_threshold(const uint8_t* g, int h, int w, int win, double C, uint8_t* out) {
    if (win % 2 == 0) win++;
    const int r = win / 2;
    const int idelta = (int)floor(C); /* BINARY_INV uses cvFloor(delta) */
    std::vector<int32_t> hs((size_t)h * w);
    for (int y = 0; y < h; y++) {
        const uint8_t* row = g + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int d = -r; d <= r; d++) {
                int xx = std::min(std::max(x + d, 0), w - 1);
                s += row[xx];
            }
            hs[(size_t)y * w + x] = s;
        }
    }
    const double scale = 1.0 / ((double)win * win);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int d = -r; d <= r; d++) {
                int yy = std::min(std::max(y + d, 0), h - 1);
                s += hs[(size_t)yy * w + x];
            }
            int mean = (int)lrint(s * scale);
            int src = g[(size_t)y * w + x];
            out[(size_t)y * w + x] = (src - mean <= -idelta) ? 255 : 0;
        }
    }
}

/* ------------------------------------------------------------------ K3 */
/* findContours(RETR_LIST, CHAIN_APPROX_NONE): Suzuki–Abe border following as
 * OpenCV implements it.  8 directions, index grows counter-clockwise on
 * screen: 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE. */
const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

void fetch_contour(int8_t* img, int w, int x0, int y0, bool is_hole, std::vector<Pt>& out) {
    const int nbd = 2;
    int deltas[16];
    for (int i = 0; i < 8; i++) deltas[i] = deltas[i + 8] = DY[i] * w + DX[i];
    int8_t* i0 = img + (size_t)y0 * w + x0;
    int8_t *i1, *i3, *i4 = 0;
    int s, s_end;
    Pt pt = {x0, y0};
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
    } while (*i1 == 0 && s != s_end);

    if (s == s_end) { /* single pixel domain */
        *i0 = (int8_t)(nbd | -128);
        out.push_back(pt);
        return;
    }
    i3 = i0;
    for (;;) {
        s_end = s;
        while (s < 15) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        /* check "right" bound */
        if ((unsigned)(s - 1) < (unsigned)s_end)
            *i3 = (int8_t)(nbd | -128);
        else if (*i3 == 1)
            *i3 = nbd;
        out.push_back(pt);
        pt.x += DX[s];
        pt.y += DY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

/* Contours in OpenCV's output order (last discovered first). */
void find_contours(const uint8_t* bin, int h, int w, std::vector<std::vector<Pt> >& contours) {
    contours.clear();
    if (h < 3 || w < 3) return;
    std::vector<int8_t> img((size_t)h * w);
    for (size_t i = 0; i < img.size(); i++) img[i] = bin[i] ? 1 : 0;
    for (int x = 0; x < w; x++) img[x] = img[(size_t)(h - 1) * w + x] = 0;
    for (int y = 0; y < h; y++) img[(size_t)y * w] = img[(size_t)y * w + w - 1] = 0;

    std::vector<std::vector<Pt> > found;
    for (int y = 1; y < h - 1; y++) {
        int8_t* row = img.data() + (size_t)y * w;
        int prev = 0;
        for (int x = 1; x < w - 1; x++) {
            int p = row[x];
            if (p == prev) continue;
            bool is_hole = false;
            bool start = true;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) start = false;
                else is_hole = true;
            }
            if (start) {
                found.push_back(std::vector<Pt>());
                fetch_contour(img.data(), w, x - (is_hole ? 1 : 0), y, is_hole, found.back());
                p = row[x];
            }
            prev = p;
        }
    }
    contours.resize(found.size());
    for (size_t i = 0; i < found.size(); i++) contours[i].swap(found[found.size() - 1 - i]);
}

/* --------------------------------------------------------------- K4 bits */
/* imgproc approx.cpp approxPolyDP_<int>, closed curve. */
struct Range { int start, end; };

int approx_poly_dp(const Pt* src, int count0, double eps, std::vector<Pt>& dstv) {
    const int init_iters = 3;
    Range slice = {0, 0}, right_slice = {0, 0};
    Pt start_pt = {-1000000, -1000000}, end_pt = {0, 0}, pt = {0, 0};
    int i = 0, j, pos = 0, wpos, count = count0, new_count = 0;
    bool le_eps = false;
    std::vector<Range> stack;
    dstv.assign((size_t)std::max(count0, 1), Pt{0, 0});
    Pt* dst = dstv.data();
    if (count == 0) { dstv.clear(); return 0; }
    eps *= eps;

#define READ_PT(p_, pos_) do { p_ = src[pos_]; if (++pos_ >= count) pos_ = 0; } while (0)
#define READ_DST_PT(p_, pos_) do { p_ = dst[pos_]; if (++pos_ >= count) pos_ = 0; } while (0)

    /* 1. approximately the two farthest points of the contour */
    right_slice.start = 0;
    for (i = 0; i < init_iters; i++) {
        double dist, max_dist = 0;
        pos = (pos + right_slice.start) % count;
        READ_PT(start_pt, pos);
        for (j = 1; j < count; j++) {
            double dx, dy;
            READ_PT(pt, pos);
            dx = pt.x - start_pt.x;
            dy = pt.y - start_pt.y;
            dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_slice.start = j; }
        }
        le_eps = max_dist <= eps;
    }
    /* 2. initialise the stack */
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        stack.push_back(right_slice);
        stack.push_back(slice);
    } else {
        dst[new_count++] = start_pt;
    }
    /* 3. recursive split */
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        end_pt = src[slice.end];
        pos = slice.start;
        READ_PT(start_pt, pos);
        if (pos != slice.end) {
            double dx, dy, dist, max_dist = 0;
            dx = end_pt.x - start_pt.x;
            dy = end_pt.y - start_pt.y;
            while (pos != slice.end) {
                READ_PT(pt, pos);
                dist = fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) {
                    max_dist = dist;
                    right_slice.start = (pos + count - 1) % count;
                }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start_pt = src[slice.start];
        }
        if (le_eps) {
            dst[new_count++] = start_pt;
        } else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack.push_back(right_slice);
            stack.push_back(slice);
        }
    }
    /* 4. remove redundant points */
    count = new_count;
    pos = count - 1;
    READ_DST_PT(start_pt, pos);
    wpos = pos;
    READ_DST_PT(pt, pos);
    for (i = 0; i < count && new_count > 2; i++) {
        double dx, dy, dist, sip;
        READ_DST_PT(end_pt, pos);
        dx = end_pt.x - start_pt.x;
        dy = end_pt.y - start_pt.y;
        dist = fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        sip = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) +
              (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            new_count--;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= count) wpos = 0;
            READ_DST_PT(pt, pos);
            i++;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= count) wpos = 0;
        pt = end_pt;
    }
#undef READ_PT
#undef READ_DST_PT
    dstv.resize((size_t)new_count);
    return new_count;
}

/* imgproc convhull.cpp isContourConvex_<int>. */
bool is_contour_convex(const Pt* p, int n) {
    Pt prev_pt = p[(n - 2 + n) % n];
    Pt cur_pt = p[n - 1];
    int dx0 = cur_pt.x - prev_pt.x;
    int dy0 = cur_pt.y - prev_pt.y;
    int orientation = 0;
    for (int i = 0; i < n; i++) {
        prev_pt = cur_pt;
        cur_pt = p[i];
        int dx = cur_pt.x - prev_pt.x;
        int dy = cur_pt.y - prev_pt.y;
        int dxdy0 = dx * dy0;
        int dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return false;
        dx0 = dx;
        dy0 = dy;
    }
    return true;
}

struct Quad {
    Pt2f c[4];
    int start_key; /* y*w+x of the contour's first point */
    int clen;      /* contour length */
};

/* aruco_detector.cpp _findMarkerContours. */
void find_marker_contours(const uint8_t* bin, int h, int w, const orc_params& P, std::vector<Quad>& out) {
    const int maxdim = std::max(w, h);
    unsigned minPerimeterPixels = (unsigned)(P.minMarkerPerimeterRate * maxdim);
    unsigned maxPerimeterPixels = (unsigned)(P.maxMarkerPerimeterRate * maxdim);
    std::vector<std::vector<Pt> > contours;
    find_contours(bin, h, w, contours);
    std::vector<Pt> approx;
    for (size_t i = 0; i < contours.size(); i++) {
        const std::vector<Pt>& c = contours[i];
        if (c.size() < minPerimeterPixels || c.size() > maxPerimeterPixels) continue;
        approx_poly_dp(c.data(), (int)c.size(), (double)c.size() * P.polygonalApproxAccuracyRate, approx);
        if (approx.size() != 4 || !is_contour_convex(approx.data(), 4)) continue;
        double minDistSq = (double)maxdim * maxdim;
        for (int j = 0; j < 4; j++) {
            double dx = (double)(approx[j].x - approx[(j + 1) % 4].x);
            double dy = (double)(approx[j].y - approx[(j + 1) % 4].y);
            minDistSq = std::min(minDistSq, dx * dx + dy * dy);
        }
        double minCornerDistancePixels = (double)c.size() * P.minCornerDistanceRate;
        if (minDistSq < minCornerDistancePixels * minCornerDistancePixels) continue;
        bool tooNearBorder = false;
        for (int j = 0; j < 4; j++) {
            if (approx[j].x < P.minDistanceToBorder || approx[j].y < P.minDistanceToBorder ||
                approx[j].x > w - 1 - P.minDistanceToBorder || approx[j].y > h - 1 - P.minDistanceToBorder)
                tooNearBorder = true;
        }
        if (tooNearBorder) continue;
        Quad q;
        for (int j = 0; j < 4; j++) { q.c[j].x = (float)approx[j].x; q.c[j].y = (float)approx[j].y; }
        q.start_key = c[0].y * w + c[0].x;
        q.clen = (int)c.size();
        out.push_back(q);
    }
}

/* aruco_detector.cpp _reorderCandidatesCorners. */
void reorder_clockwise(Quad& q) {
    double dx1 = q.c[1].x - q.c[0].x, dy1 = q.c[1].y - q.c[0].y;
    double dx2 = q.c[2].x - q.c[0].x, dy2 = q.c[2].y - q.c[0].y;
    double cross = dx1 * dy2 - dy1 * dx2;
    if (cross < 0.0) std::swap(q.c[1], q.c[3]);
}

inline float norm_l2_sqr(Pt2f a, Pt2f b) {
    float dx = a.x - b.x, dy = a.y - b.y;
    return dx * dx + dy * dy;
}

float quad_perimeter(const Pt2f* c) {
    float p = 0.f;
    for (int i = 0; i < 4; i++) p += sqrtf(norm_l2_sqr(c[i], c[(i + 1) % 4]));
    return p;
}

/* aruco_utils getAverageModuleSize. */
float average_module_size(const Pt2f* c, int markerSize, int borderBits) {
    float s = 0.f;
    for (int i = 0; i < 4; i++) s += sqrtf(norm_l2_sqr(c[i], c[(i + 1) % 4]));
    int numModules = markerSize + borderBits * 2;
    s /= (4.f * numModules);
    return s;
}

/* aruco_detector.cpp getAverageDistance. */
float average_distance(const Pt2f* m1, const Pt2f* m2) {
    float minDistSq = FLT_MAX;
    for (int fc = 0; fc < 4; fc++) {
        float distSq = 0;
        for (int c = 0; c < 4; c++) {
            int modC = (c + fc) % 4;
            distSq += norm_l2_sqr(m1[modC], m2[c]);
        }
        distSq /= 4.f;
        minDistSq = std::min(minDistSq, distSq);
    }
    return sqrtf(minDistSq);
}

/* imgproc pointPolygonTest(measureDist=false), float contour path. */
int point_in_quad(const Pt2f* cnt, Pt2f pt) {
    int counter = 0;
    Pt2f v = cnt[3], v0;
    for (int i = 0; i < 4; i++) {
        v0 = v;
        v = cnt[i];
        if ((v0.y <= pt.y && v.y <= pt.y) || (v0.y > pt.y && v.y > pt.y) || (v0.x < pt.x && v.x < pt.x)) {
            if (pt.y == v.y && (pt.x == v.x || (pt.y == v0.y && ((v0.x <= pt.x && pt.x <= v.x) ||
                                                                (v.x <= pt.x && pt.x <= v0.x)))))
                return 0;
            continue;
        }
        double dist = (double)(pt.y - v0.y) * (v.x - v0.x) - (double)(pt.x - v0.x) * (v.y - v0.y);
        if (dist == 0) return 0;
        if (v.y < v0.y) dist = -dist;
        counter += dist > 0;
    }
    return counter % 2 == 0 ? -1 : 1;
}

bool marker1_in_marker2(const Pt2f* m1, const Pt2f* m2) {
    return point_in_quad(m2, m1[0]) > 0 && point_in_quad(m2, m1[1]) > 0 &&
           point_in_quad(m2, m1[2]) > 0 && point_in_quad(m2, m1[3]) > 0;
}

struct CandTree {
    Pt2f c[4];
    float perimeter;
    int parent, depth;
    std::vector<int> close; /* indices into the sorted candidate array */
};

/* ------------------------------------------------------------------ K6 */
/* imgproc getPerspectiveTransform: 8x8 solve, LU with partial pivoting. */
bool perspective_transform(const Pt2f* src, const Pt2f* dst, double M[9]) {
    double a[8][9];
    for (int i = 0; i < 4; i++) {
        double sx = src[i].x, sy = src[i].y, dx = dst[i].x, dy = dst[i].y;
        double r0[9] = {sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx};
        double r1[9] = {0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy};
        memcpy(a[i], r0, sizeof r0);
        memcpy(a[i + 4], r1, sizeof r1);
    }
    for (int i = 0; i < 8; i++) {
        int k = i;
        for (int j = i + 1; j < 8; j++)
            if (fabs(a[j][i]) > fabs(a[k][i])) k = j;
        if (fabs(a[k][i]) < DBL_EPSILON) return false;
        if (k != i)
            for (int j = i; j < 9; j++) std::swap(a[i][j], a[k][j]);
        double d = -1 / a[i][i];
        for (int j = i + 1; j < 8; j++) {
            double alpha = a[j][i] * d;
            for (int c = i + 1; c < 9; c++) a[j][c] += alpha * a[i][c];
        }
    }
    double x[8];
    for (int i = 7; i >= 0; i--) {
        double s = a[i][8];
        for (int k = i + 1; k < 8; k++) s -= a[i][k] * x[k];
        x[i] = s / a[i][i];
    }
    for (int i = 0; i < 8; i++) M[i] = x[i];
    M[8] = 1.0;
    return true;
}

bool invert3x3(const double* s, double* d) {
    double det = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) +
                 s[2] * (s[3] * s[7] - s[4] * s[6]);
    if (det == 0) return false;
    double id = 1. / det;
    double t[9];
    t[0] = (s[4] * s[8] - s[5] * s[7]) * id;
    t[1] = (s[2] * s[7] - s[1] * s[8]) * id;
    t[2] = (s[1] * s[5] - s[2] * s[4]) * id;
    t[3] = (s[5] * s[6] - s[3] * s[8]) * id;
    t[4] = (s[0] * s[8] - s[2] * s[6]) * id;
    t[5] = (s[2] * s[3] - s[0] * s[5]) * id;
    t[6] = (s[3] * s[7] - s[4] * s[6]) * id;
    t[7] = (s[1] * s[6] - s[0] * s[7]) * id;
    t[8] = (s[0] * s[4] - s[1] * s[3]) * id;
    memcpy(d, t, sizeof t);
    return true;
}

inline int round_half_even(double v) { return (int)lrint(v); }

/* imgproc thresh.cpp getThreshVal_Otsu_8u. */
int otsu_threshold(const uint8_t* px, int n) {
    int hist[256] = {0};
    for (int i = 0; i < n; i++) hist[px[i]]++;
    double mu = 0, scale = 1. / n;
    for (int i = 0; i < 256; i++) mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        double q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        double mu2 = (mu - q1 * mu1) / q2;
        double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return (int)max_val;
}

/* aruco_detector.cpp _extractBits: perspective-normalise to (n*cell)^2 with
 * INTER_NEAREST, Otsu, per-cell majority. */
void extract_bits(const uint8_t* gray, int h, int w, const Pt2f* corners, int markerSize,
                  const orc_params& P, uint8_t* bits /* (markerSize+2b)^2 */) {
    const int nb = markerSize + 2 * P.markerBorderBits;
    const int cell = P.perspectiveRemovePixelPerCell;
    const int margin = (int)(P.perspectiveRemoveIgnoredMarginPerCell * cell);
    const int S = nb * cell;
    Pt2f dstc[4] = {{0, 0}, {(float)S - 1, 0}, {(float)S - 1, (float)S - 1}, {0, (float)S - 1}};
    double M[9], Mi[9];
    std::vector<uint8_t> img((size_t)S * S, 0);
    bool ok = perspective_transform(corners, dstc, M) && invert3x3(M, Mi);
    if (ok) {
        /* warpPerspective, INTER_NEAREST, BORDER_CONSTANT(0): dst(x,y) =
         * src(round(X/W), round(Y/W)) with M inverted. */
        for (int y = 0; y < S; y++) {
            double X0 = Mi[0] * 0 + Mi[1] * y + Mi[2];
            double Y0 = Mi[3] * 0 + Mi[4] * y + Mi[5];
            double W0 = Mi[6] * 0 + Mi[7] * y + Mi[8];
            for (int x = 0; x < S; x++) {
                double W = W0 + Mi[6] * x;
                W = W ? 1. / W : 0;
                double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, (X0 + Mi[0] * x) * W));
                double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, (Y0 + Mi[3] * x) * W));
                int X = round_half_even(fX), Y = round_half_even(fY);
                /* remap stores coordinates as short */
                X = std::min(std::max(X, (int)SHRT_MIN), (int)SHRT_MAX);
                Y = std::min(std::max(Y, (int)SHRT_MIN), (int)SHRT_MAX);
                if (X >= 0 && X < w && Y >= 0 && Y < h) img[(size_t)y * S + x] = gray[(size_t)Y * w + X];
            }
        }
    }
    /* meanStdDev over the inner region (cell/2 margin) */
    const int m2 = cell / 2;
    long long sum = 0, sq = 0;
    int n = 0;
    for (int y = m2; y < S - m2; y++)
        for (int x = m2; x < S - m2; x++) {
            int v = img[(size_t)y * S + x];
            sum += v;
            sq += v * v;
            n++;
        }
    double mean = (double)sum / n;
    double var = std::max((double)sq / n - mean * mean, 0.);
    double stddev = sqrt(var);
    if (stddev < P.minOtsuStdDev) {
        memset(bits, mean > 127 ? 1 : 0, (size_t)nb * nb);
        return;
    }
    int thr = otsu_threshold(img.data(), S * S);
    for (int y = 0; y < nb; y++)
        for (int x = 0; x < nb; x++) {
            int xs = x * cell + margin, ys = y * cell + margin, side = cell - 2 * margin;
            int nz = 0;
            for (int yy = 0; yy < side; yy++)
                for (int xx = 0; xx < side; xx++) nz += img[(size_t)(ys + yy) * S + xs + xx] > thr;
            bits[y * nb + x] = (nz > side * side / 2) ? 1 : 0;
        }
}

/* Dictionary::getByteListFromBits restricted to <=16 inner bits: code of the
 * candidate as seen under rotation r. */
uint32_t rotated_code(const uint8_t* b, int n, int r) {
    uint32_t code = 0;
    for (int row = 0; row < n; row++)
        for (int col = 0; col < n; col++) {
            int v;
            switch (r) {
                case 0: v = b[row * n + col]; break;
                case 1: v = b[col * n + (n - 1 - row)]; break;
                case 2: v = b[(n - 1 - row) * n + (n - 1 - col)]; break;
                default: v = b[(n - 1 - col) * n + row]; break;
            }
            code = (code << 1) | (uint32_t)(v & 1);
        }
    return code;
}

inline int popcount32(uint32_t v) { return __builtin_popcount(v); }

/* aruco_detector.cpp _identifyOneCandidate + Dictionary::identify. */
bool identify_one(const uint8_t* gray, int h, int w, const Pt2f* corners, const orc_params& P,
                  const uint16_t* dict, int dict_n, int markerSize, int maxCorrectionBits,
                  int& id, int& rotation, uint8_t* bits_out) {
    const int nb = markerSize + 2 * P.markerBorderBits;
    std::vector<uint8_t> bits((size_t)nb * nb);
    extract_bits(gray, h, w, corners, markerSize, P, bits.data());
    if (bits_out) memcpy(bits_out, bits.data(), bits.size());
    int maximumErrorsInBorder = (int)(markerSize * markerSize * P.maxErroneousBitsInBorderRate);
    /* _getBorderErrors */
    int borderErrors = 0;
    const int bb = P.markerBorderBits;
    for (int y = 0; y < nb; y++)
        for (int k = 0; k < bb; k++) {
            if (bits[y * nb + k] != 0) borderErrors++;
            if (bits[y * nb + nb - 1 - k] != 0) borderErrors++;
        }
    for (int x = bb; x < nb - bb; x++)
        for (int k = 0; k < bb; k++) {
            if (bits[k * nb + x] != 0) borderErrors++;
            if (bits[(nb - 1 - k) * nb + x] != 0) borderErrors++;
        }
    if (borderErrors > maximumErrorsInBorder) return false;
    std::vector<uint8_t> inner((size_t)markerSize * markerSize);
    for (int y = 0; y < markerSize; y++)
        for (int x = 0; x < markerSize; x++) inner[y * markerSize + x] = bits[(y + bb) * nb + x + bb];
    /* Dictionary::identify: the dictionary stores each marker under 4
     * rotations; rotation r of marker m equals rotated_code(marker bits, r).
     * Equivalently compare the candidate against each rotation of m. */
    int maxCorr = (int)((double)maxCorrectionBits * P.errorCorrectionRate);
    uint32_t cand = rotated_code(inner.data(), markerSize, 0);
    id = -1;
    for (int m = 0; m < dict_n; m++) {
        /* expand marker m to bits, then its 4 stored rotations */
        uint8_t mb[16];
        for (int i = 0; i < markerSize * markerSize; i++)
            mb[i] = (dict[m] >> (markerSize * markerSize - 1 - i)) & 1;
        int best = markerSize * markerSize + 1, bestRot = -1;
        for (int r = 0; r < 4; r++) {
            int hd = popcount32(rotated_code(mb, markerSize, r) ^ cand);
            if (hd < best) { best = hd; bestRot = r; }
        }
        if (best <= maxCorr) { id = m; rotation = bestRot; break; }
    }
    return id != -1;
}

/* ------------------------------------------------------------------ K7 */
/* imgproc samplers.cpp getRectSubPix_8u32f (+ generic border path). */
void get_rect_subpix_8u32f(const uint8_t* src, int sw, int sh, float* dst, int ww, int wh, Pt2f center) {
    center.x -= (ww - 1) * 0.5f;
    center.y -= (wh - 1) * 0.5f;
    int ipx = (int)floorf(center.x), ipy = (int)floorf(center.y);
    if (0 <= ipx && ipx + ww < sw && 0 <= ipy && ipy + wh < sh) {
        float a = center.x - ipx;
        float b = center.y - ipy;
        a = std::max(a, 0.0001f);
        float a12 = a * (1.f - b);
        float a22 = a * b;
        float b1 = 1.f - b;
        float b2 = b;
        double s = (1. - a) / a;
        const uint8_t* p = src + (size_t)ipy * sw + ipx;
        for (int i = 0; i < wh; i++, p += sw, dst += ww) {
            float prev = (1 - a) * (b1 * p[0] + b2 * p[sw]);
            for (int j = 0; j < ww; j++) {
                float t = a12 * p[j + 1] + a22 * p[j + 1 + sw];
                dst[j] = prev + t;
                prev = (float)(t * s);
            }
        }
        return;
    }
    /* generic path: bilinear with replicated border (getRectSubPix_Cn_ +
     * adjustRect).  Columns that fall outside use the vertical 2-tap only. */
    float a = center.x - ipx, b = center.y - ipy;
    float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    float b1 = 1.f - b, b2 = b;
    int rx = ipx >= 0 ? 0 : std::min(-ipx, ww);
    int rwid = (ipx < sw - ww) ? ww : std::max(sw - ipx - 1, 0);
    for (int i = 0; i < wh; i++) {
        int y0 = std::min(std::max(ipy + i, 0), sh - 1);
        int y1 = std::min(std::max(ipy + i + 1, 0), sh - 1);
        const uint8_t* r0 = src + (size_t)y0 * sw;
        const uint8_t* r1 = src + (size_t)y1 * sw;
        for (int j = 0; j < ww; j++) {
            float v;
            if (j < rx) {
                v = r0[0] * b1 + r1[0] * b2;
            } else if (j >= rwid) {
                v = r0[sw - 1] * b1 + r1[sw - 1] * b2;
            } else {
                int x0 = ipx + j;
                v = r0[x0] * a11 + r0[x0 + 1] * a12 + r1[x0] * a21 + r1[x0 + 1] * a22;
            }
            dst[i * ww + j] = v;
        }
    }
}

/* imgproc cornersubpix.cpp cornerSubPix, zeroZone = (-1,-1). */
void corner_subpix(const uint8_t* src, int h, int w, Pt2f* corners, int count, int win, int max_iters_in, double eps_in) {
    const int MAX_ITERS = 100;
    const int win_w = win * 2 + 1, win_h = win * 2 + 1;
    int max_iters = std::min(std::max(max_iters_in, 1), MAX_ITERS);
    double eps = std::max(eps_in, 0.);
    eps *= eps;
    std::vector<float> mask((size_t)win_w * win_h), buf((size_t)(win_w + 2) * (win_h + 2));
    for (int i = 0; i < win_h; i++) {
        float y = (float)(i - win) / win;
        float vy = expf(-y * y);
        for (int j = 0; j < win_w; j++) {
            float x = (float)(j - win) / win;
            mask[i * win_w + j] = (float)(vy * expf(-x * x));
        }
    }
    for (int pt_i = 0; pt_i < count; pt_i++) {
        Pt2f cT = corners[pt_i], cI = cT;
        int iter = 0;
        double err = 0;
        do {
            Pt2f cI2;
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            get_rect_subpix_8u32f(src, w, h, buf.data(), win_w + 2, win_h + 2, cI);
            const float* subpix = buf.data() + (win_w + 2) + 1;
            for (int i = 0, k = 0; i < win_h; i++, subpix += win_w + 2) {
                double py = i - win;
                for (int j = 0; j < win_w; j++, k++) {
                    double m = mask[k];
                    double tgx = subpix[j + 1] - subpix[j - 1];
                    double tgy = subpix[j + win_w + 2] - subpix[j - win_w - 2];
                    double gxx = tgx * tgx * m;
                    double gxy = tgx * tgy * m;
                    double gyy = tgy * tgy * m;
                    double px = j - win;
                    a += gxx;
                    b += gxy;
                    c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            double det = a * c - b * b;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            double scale = 1.0 / det;
            cI2.x = (float)(cI.x + c * scale * bb1 - b * scale * bb2);
            cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
            err = (cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y);
            cI = cI2;
            if (cI.x < 0 || cI.x >= w || cI.y < 0 || cI.y >= h) break;
        } while (++iter < max_iters && err > eps);
        if (fabs(cI.x - cT.x) > win || fabs(cI.y - cT.y) > win) cI = cT;
        corners[pt_i] = cI;
    }
}

} /* namespace */

/* ================================================================= C API */
extern "C" {

void orc_default_params(orc_params* p) {
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

void orc_bgr2gray(const uint8_t* bgr, int h, int w, uint8_t* gray) {
    for (size_t i = 0; i < (size_t)h * w; i++) gray[i] = luma_u8(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2]);
}

void orc_adaptive_threshold(const uint8_t* gray, int h, int w, int win, double C, uint8_t* out) {
    adaptive_threshold(gray, h, w, win, C, out);
}

int orc_find_contours(const uint8_t* bin, int h, int w, int32_t* starts, int32_t* lens, int max_contours,
                      int32_t* pts, int max_pts, int32_t* total_pts) {
    std::vector<std::vector<Pt> > cs;
    find_contours(bin, h, w, cs);
    long long off = 0;
    for (size_t i = 0; i < cs.size(); i++) {
        if ((int)i < max_contours) {
            starts[i] = (int32_t)off;
            lens[i] = (int32_t)cs[i].size();
        }
        for (size_t k = 0; k < cs[i].size(); k++) {
            if (off + (long long)k < max_pts) {
                pts[2 * (off + k)] = cs[i][k].x;
                pts[2 * (off + k) + 1] = cs[i][k].y;
            }
        }
        off += (long long)cs[i].size();
    }
    if (total_pts) *total_pts = (int32_t)off;
    return (int)cs.size();
}

int orc_approx_poly_dp(const int32_t* pts, int n, double eps, int32_t* out, int max_out) {
    std::vector<Pt> src((size_t)n), dst;
    for (int i = 0; i < n; i++) { src[i].x = pts[2 * i]; src[i].y = pts[2 * i + 1]; }
    int m = approx_poly_dp(src.data(), n, eps, dst);
    for (int i = 0; i < m && i < max_out; i++) { out[2 * i] = dst[i].x; out[2 * i + 1] = dst[i].y; }
    return m;
}

int orc_quad_candidates(const uint8_t* bin, int h, int w, const orc_params* p, float* quads,
                        int32_t* start_key, int32_t* clen, int max_quads) {
    std::vector<Quad> q;
    find_marker_contours(bin, h, w, *p, q);
    for (size_t i = 0; i < q.size() && (int)i < max_quads; i++) {
        for (int j = 0; j < 4; j++) { quads[8 * i + 2 * j] = q[i].c[j].x; quads[8 * i + 2 * j + 1] = q[i].c[j].y; }
        if (start_key) start_key[i] = q[i].start_key;
        if (clen) clen[i] = q[i].clen;
    }
    return (int)q.size();
}

int orc_identify(const uint8_t* gray, int h, int w, const float* corners4, const orc_params* p,
                 const uint16_t* dict_codes, int dict_n, int32_t* id, int32_t* rotation, uint8_t* bits36) {
    Pt2f c[4];
    for (int i = 0; i < 4; i++) { c[i].x = corners4[2 * i]; c[i].y = corners4[2 * i + 1]; }
    int iid = -1, rot = 0;
    bool ok = identify_one(gray, h, w, c, *p, dict_codes, dict_n, 4, 1, iid, rot, bits36);
    *id = iid;
    *rotation = rot;
    return ok ? 1 : 0;
}

void orc_corner_subpix(const uint8_t* gray, int h, int w, float* pts, int n, int win, int max_iter, double eps) {
    corner_subpix(gray, h, w, (Pt2f*)pts, n, win, max_iter, eps);
}

/* ArucoDetector::detectMarkers, CORNER_REFINE_SUBPIX, no Aruco3, DICT_4X4_*
 * (markerSize 4, maxCorrectionBits 1).  aruco.py:256-264. */
int orc_detect(const uint8_t* img, int h, int w, int channels, const orc_params* pp,
               const uint16_t* dict_codes, int dict_n, int32_t* ids, float* corners, int max_markers,
               float* rejected, int max_rejected, int32_t* n_rejected) {
    const orc_params& P = *pp;
    const int markerSize = 4, maxCorrectionBits = 1;
    std::vector<uint8_t> gray((size_t)h * w);
    if (channels == 3) orc_bgr2gray(img, h, w, gray.data());
    else if (channels == 1) memcpy(gray.data(), img, gray.size());
    else return -1;

    /* STEP 1: candidates over the threshold scales (_detectInitialCandidates) */
    int nScales = (P.adaptiveThreshWinSizeMax - P.adaptiveThreshWinSizeMin) / P.adaptiveThreshWinSizeStep + 1;
    std::vector<Quad> cands;
    std::vector<uint8_t> bin((size_t)h * w);
    for (int i = 0; i < nScales; i++) {
        int win = P.adaptiveThreshWinSizeMin + i * P.adaptiveThreshWinSizeStep;
        adaptive_threshold(gray.data(), h, w, win, P.adaptiveThreshConstant, bin.data());
        find_marker_contours(bin.data(), h, w, P, cands);
    }
    for (size_t i = 0; i < cands.size(); i++) reorder_clockwise(cands[i]);

    /* STEP 2a: filterTooCloseCandidates */
    std::vector<CandTree> tree(cands.size());
    for (size_t i = 0; i < cands.size(); i++) {
        memcpy(tree[i].c, cands[i].c, sizeof tree[i].c);
        tree[i].perimeter = quad_perimeter(tree[i].c);
        tree[i].parent = -1;
        tree[i].depth = 0;
    }
    std::stable_sort(tree.begin(), tree.end(),
                     [](const CandTree& a, const CandTree& b) { return a.perimeter > b.perimeter; });
    const size_t n = tree.size();
    std::vector<int> groupId(n, -1);
    std::vector<std::vector<size_t> > grouped;
    std::vector<char> isSelected(n, 1);
    size_t countSelected = 0;
    for (size_t i = 0; i < n; i++) {
        for (size_t j = i + 1; j < n; j++) {
            float minDist = average_distance(tree[i].c, tree[j].c);
            if (minDist < tree[j].perimeter * (float)P.minMarkerDistanceRate) {
                isSelected[i] = 0;
                isSelected[j] = 0;
                if (groupId[i] < 0 && groupId[j] < 0) {
                    groupId[i] = groupId[j] = (int)grouped.size();
                    grouped.push_back(std::vector<size_t>{i, j});
                } else if (groupId[i] > -1 && groupId[j] == -1) {
                    int g = groupId[i];
                    groupId[j] = g;
                    grouped[g].push_back(j);
                } else if (groupId[j] > -1 && groupId[i] == -1) {
                    int g = groupId[j];
                    groupId[i] = g;
                    grouped[g].push_back(i);
                }
            }
        }
        countSelected += isSelected[i];
    }
    for (size_t g = 0; g < grouped.size(); g++) {
        std::vector<size_t>& gr = grouped[g];
        std::stable_sort(gr.begin(), gr.end());
        size_t currId = gr[0];
        isSelected[currId] = 1;
        for (size_t k = 1; k < gr.size(); k++) {
            size_t id = gr[k];
            float dist = average_distance(tree[id].c, tree[currId].c);
            float moduleSize = average_module_size(tree[id].c, markerSize, P.markerBorderBits);
            if (dist > P.minGroupDistance * moduleSize) {
                currId = id;
                tree[gr[0]].close.push_back((int)id);
            }
        }
    }
    std::vector<CandTree> sel;
    sel.reserve(grouped.size() + countSelected);
    std::vector<CandTree> closePool(tree); /* close-contour corners looked up here */
    for (size_t i = 0; i < n; i++)
        if (isSelected[i]) sel.push_back(tree[i]);
    for (int i = (int)sel.size() - 1; i >= 0; i--)
        for (int j = i - 1; j >= 0; j--)
            if (marker1_in_marker2(sel[i].c, sel[j].c)) {
                sel[i].parent = j;
                sel[j].depth = std::max(sel[j].depth, sel[i].depth + 1);
                break;
            }

    /* STEP 2b: identifyCandidates */
    const size_t nc = sel.size();
    std::vector<int> idsTmp(nc, -1), rotated(nc, 0);
    std::vector<char> valid(nc, 0), was(nc, 0);
    int maxDepth = 0;
    for (size_t i = 0; i < nc; i++) maxDepth = std::max(maxDepth, sel[i].depth);
    std::vector<std::vector<size_t> > depths((size_t)maxDepth + 1);
    for (size_t i = 0; i < nc; i++) depths[sel[i].depth].push_back(i);
    int depth = 0;
    size_t counter = 0;
    while (counter < nc && depth <= maxDepth) {
        for (size_t k = 0; k < depths[depth].size(); k++) {
            size_t v = depths[depth][k];
            was[v] = 1;
            valid[v] = identify_one(gray.data(), h, w, sel[v].c, P, dict_codes, dict_n, markerSize,
                                    maxCorrectionBits, idsTmp[v], rotated[v], 0);
            if (!valid[v]) {
                for (size_t q = 0; q < sel[v].close.size(); q++) {
                    const CandTree& cm = closePool[sel[v].close[q]];
                    valid[v] = identify_one(gray.data(), h, w, cm.c, P, dict_codes, dict_n, markerSize,
                                            maxCorrectionBits, idsTmp[v], rotated[v], 0);
                    if (valid[v]) {
                        memcpy(sel[v].c, cm.c, sizeof cm.c);
                        break;
                    }
                }
            }
        }
        for (size_t k = 0; k < depths[depth].size(); k++) {
            size_t v = depths[depth][k];
            if (valid[v]) {
                int parent = sel[v].parent;
                while (parent != -1) {
                    if (!was[parent]) { was[parent] = 1; counter++; }
                    parent = sel[parent].parent;
                }
            }
            counter++;
        }
        depth++;
    }

    int nacc = 0, nrej = 0;
    std::vector<Pt2f> acc;
    std::vector<int> accIds;
    for (size_t i = 0; i < nc; i++) {
        if (valid[i]) {
            Pt2f c[4];
            memcpy(c, sel[i].c, sizeof c);
            std::rotate(c, c + 4 - rotated[i], c + 4); /* correctCornerPosition */
            for (int k = 0; k < 4; k++) acc.push_back(c[k]);
            accIds.push_back(idsTmp[i]);
        } else {
            if (rejected && nrej < max_rejected)
                for (int k = 0; k < 4; k++) { rejected[8 * nrej + 2 * k] = sel[i].c[k].x; rejected[8 * nrej + 2 * k + 1] = sel[i].c[k].y; }
            nrej++;
        }
    }
    if (n_rejected) *n_rejected = nrej;

    /* STEP 3: cornerSubPix per accepted marker */
    nacc = (int)accIds.size();
    if (P.cornerRefinementMethod == 1) {
        for (int i = 0; i < nacc; i++) {
            int win = std::max(1, (int)lrint(P.relativeCornerRefinmentWinSize *
                                             average_module_size(&acc[4 * i], markerSize, P.markerBorderBits)));
            win = std::min(win, P.cornerRefinementWinSize);
            corner_subpix(gray.data(), h, w, &acc[4 * i], 4, win, P.cornerRefinementMaxIterations,
                          P.cornerRefinementMinAccuracy);
        }
    }
    for (int i = 0; i < nacc && i < max_markers; i++) {
        ids[i] = accIds[i];
        for (int k = 0; k < 4; k++) { corners[8 * i + 2 * k] = acc[4 * i + k].x; corners[8 * i + 2 * k + 1] = acc[4 * i + k].y; }
    }
    return nacc;
}

} /* extern "C" */
