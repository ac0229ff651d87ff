/*
 * oracle/jpeg.cpp — CPU restatement of baseline JPEG decoding as the reference's frame ingest gets it from OpenCV:
 * cv.VideoCapture(...).read() on a camera that delivers MJPG (reference camera_sync/camera.py:98-118, 150-154) and
 * cv.imread of its stored photos (pics/ *.jpg) both end in cv::imdecode -> libjpeg-turbo with its defaults
 * (JDCT_ISLOW, fancy upsampling, YCbCr -> RGB).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * libjpeg-turbo is an un-vendored dependency (it ships inside the opencv-contrib-python 4.11.0.86 wheel, uv.lock:925-943,
 * absent here); restated from its published algorithm: ITU T.81 Annex F sequential Huffman decoding (jdhuff.c),
 * the "accurate integer" inverse DCT (jidctint.c, 13-bit constants, two passes), the triangle-filter chroma
 * upsampling h2v1 / h2v2 (jdsample.c) with the context rows of jdmainct.c (the row above the first and below the last
 * real row is that row itself), and the table-driven colour conversion of jdcolor.c (16-bit fixed point).
 *
 * PIN: the reference holds 21 JPEG photos (pics/calib, pics/poses, pics/robot_calib; copies under tests/golden) but no
 * decoded pixels.  Pillow, importable in this image, decodes with the same library (libjpeg-turbo, same defaults); this
 * file equals its output bit for bit on all 21 photos and on re-encodings at 4:4:4 / 4:2:2 / 4:2:0 / gray, odd sizes,
 * optimised tables and restart intervals (tests/test_jpeg.py).  That pins the arithmetic to libjpeg-turbo, not to a
 * reference-held output.
 *
 * Arithmetic is 32-bit (as in libjpeg-turbo's SIMD code; its C code uses `long`): identical whenever no intermediate
 * of the inverse DCT exceeds 32 bits, which holds for the coefficients of any encoder's output.
 *
 * Scope: SOF0 / SOF1 (Huffman, 8-bit samples), one interleaved scan, 1 or 3 components (Y or YCbCr), luma sampling
 * 1x1 / 2x1 / 2x2 with 1x1 chroma, restart intervals, streams without DHT (MJPEG: the Annex K tables).
 */
#include <stdint.h>
#include <string.h>

#include <vector>

#include "oracle.h"

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ITU T.81 Annex K.3: the tables an MJPEG stream without DHT segments means */
const uint8_t kStdDcLumBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kStdDcChrBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kStdDcVal[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kStdAcLumBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kStdAcLumVal[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kStdAcChrBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kStdAcChrVal[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct Huff {
    bool set = false;
    int maxcode[18];    /* largest code of each length, -1 if none */
    int valoffset[17];  /* huffval index of the first code of the length minus that code */
    uint8_t huffval[256];
    bool build(const uint8_t* bits, const uint8_t* vals, int nvals) {
        int total = 0;
        for (int l = 0; l < 16; l++) total += bits[l];
        if (total > 256 || total > nvals) return false;
        memcpy(huffval, vals, total);
        int code = 0, p = 0;
        for (int l = 1; l <= 16; l++) {
            if (bits[l - 1]) {
                valoffset[l] = p - code;
                p += bits[l - 1];
                code += bits[l - 1];
                maxcode[l] = code - 1;
            } else {
                maxcode[l] = -1;
                valoffset[l] = 0;
            }
            if (code > (1 << l)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        set = true;
        return true;
    }
};

struct Comp { int id, h, v, tq, td, ta, bw, bh; /* blocks per row / column of the component's padded plane */ };

/* entropy-coded segment reader: byte stuffing, stops feeding at a marker (zero bits from there on, as jdhuff.c does) */
struct Bits {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int n = 0;
    bool hitMarker = false;
    void fill() {
        while (n <= 24) {
            uint32_t b = 0;
            if (!hitMarker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0) p += 2;
                    else { hitMarker = true; b = 0; }
                } else {
                    p++;
                }
            } else {
                hitMarker = true;
            }
            acc |= b << (24 - n);
            n += 8;
        }
    }
    int get(int k) {  /* 0 <= k <= 16 */
        if (!k) return 0;
        if (n < k) fill();
        const int v = (int)(acc >> (32 - k));
        acc <<= k;
        n -= k;
        return v;
    }
    void align() { acc = 0; n = 0; }  /* the bits left in the current byte are padding */
};

inline int decode_sym(Bits& b, const Huff& h, bool& ok) {
    int code = b.get(1), l = 1;
    while (code > h.maxcode[l]) {
        code = (code << 1) | b.get(1);
        l++;
    }
    if (l > 16) { ok = false; return 0; }
    return h.huffval[(code + h.valoffset[l]) & 255];
}

inline int extend(int r, int s) { return r < (1 << (s - 1)) ? r - (1 << s) + 1 : r; }

/* jidctint.c, jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2; arithmetic modulo 2^32 */
const int32_t F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
              F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
inline int32_t mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
inline int32_t add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t shl(int32_t a, int s) { return (int32_t)((uint32_t)a << s); }
inline int32_t descale(int32_t x, int n) { return add(x, 1 << (n - 1)) >> n; }

/* one 1-D pass over in[0..7] (stride 1 here; the caller gathers); shift = the pass's final descale */
inline void idct_1d(const int32_t* in, int32_t* out, int shift) {
    int32_t z2 = in[2], z3 = in[6];
    int32_t z1 = mul(add(z2, z3), F_0_541);
    int32_t tmp2 = add(z1, mul(z3, -F_1_847));
    int32_t tmp3 = add(z1, mul(z2, F_0_765));
    int32_t tmp0 = shl(add(in[0], in[4]), 13);
    int32_t tmp1 = shl(sub(in[0], in[4]), 13);
    const int32_t tmp10 = add(tmp0, tmp3), tmp13 = sub(tmp0, tmp3), tmp11 = add(tmp1, tmp2), tmp12 = sub(tmp1, tmp2);
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = add(tmp0, tmp3); z2 = add(tmp1, tmp2); z3 = add(tmp0, tmp2);
    int32_t z4 = add(tmp1, tmp3);
    const int32_t z5 = mul(add(z3, z4), F_1_175);
    tmp0 = mul(tmp0, F_0_298); tmp1 = mul(tmp1, F_2_053); tmp2 = mul(tmp2, F_3_072); tmp3 = mul(tmp3, F_1_501);
    z1 = mul(z1, -F_0_899); z2 = mul(z2, -F_2_562); z3 = mul(z3, -F_1_961); z4 = mul(z4, -F_0_390);
    z3 = add(z3, z5); z4 = add(z4, z5);
    tmp0 = add(tmp0, add(z1, z3)); tmp1 = add(tmp1, add(z2, z4)); tmp2 = add(tmp2, add(z2, z3)); tmp3 = add(tmp3, add(z1, z4));
    out[0] = descale(add(tmp10, tmp3), shift); out[7] = descale(sub(tmp10, tmp3), shift);
    out[1] = descale(add(tmp11, tmp2), shift); out[6] = descale(sub(tmp11, tmp2), shift);
    out[2] = descale(add(tmp12, tmp1), shift); out[5] = descale(sub(tmp12, tmp1), shift);
    out[3] = descale(add(tmp13, tmp0), shift); out[4] = descale(sub(tmp13, tmp0), shift);
}

/* range_limit[x & RANGE_MASK] of jdmaster.c's table, samples centred on 128 */
inline uint8_t range_limit_idct(int32_t x) {
    const int y = x & 1023;
    return (uint8_t)(y < 128 ? y + 128 : y < 512 ? 255 : y < 896 ? 0 : y - 896);
}

void idct_block(const int16_t* coef, const uint16_t* q, uint8_t* dst, int stride) {
    int32_t ws[64], col[8], res[8];
    for (int c = 0; c < 8; c++) {
        for (int r = 0; r < 8; r++) col[r] = mul(coef[8 * r + c], q[8 * r + c]);
        idct_1d(col, res, 13 - 2);
        for (int r = 0; r < 8; r++) ws[8 * r + c] = res[r];
    }
    for (int r = 0; r < 8; r++) {
        idct_1d(ws + 8 * r, res, 13 + 2 + 3);
        for (int c = 0; c < 8; c++) dst[(size_t)r * stride + c] = range_limit_idct(res[c]);
    }
}

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

}  // namespace

extern "C" {

/* Size of the image a JPEG holds, without decoding it.  0, or < 0 (not a JPEG this decoder takes). */
int orc_jpeg_info(const uint8_t* data, size_t n, int32_t* height, int32_t* width, int32_t* components) {
    if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
    size_t i = 2;
    while (i + 4 <= n) {
        if (data[i] != 0xFF) return -1;
        const int m = data[i + 1];
        if (m == 0xFF) { i++; continue; }
        const size_t L = ((size_t)data[i + 2] << 8) | data[i + 3];
        if (m == 0xC0 || m == 0xC1) {
            if (i + 2 + L > n || L < 8) return -1;
            if (height) *height = (data[i + 5] << 8) | data[i + 6];
            if (width) *width = (data[i + 7] << 8) | data[i + 8];
            if (components) *components = data[i + 9];
            return 0;
        }
        if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) return -2;  /* progressive, lossless, arithmetic */
        if (m == 0xDA) return -1;
        i += 2 + L;
    }
    return -1;
}

/* data[n] = one JPEG file / MJPEG frame -> bgr u8 [height, width, 3] (cap = bytes available there).
 * 0 ok; -1 not a JPEG / truncated headers; -2 a kind of JPEG outside the scope above; -3 corrupt entropy data;
 * -4 bgr too small. */
int orc_jpeg_decode(const uint8_t* data, size_t n, uint8_t* bgr, size_t cap) {
    if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
    uint16_t qt[4][64];
    bool qset[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    bool anyDht = false;
    Comp comp[3];
    int nc = 0, H = 0, W = 0, restart = 0;
    size_t i = 2, scan = 0;
    while (!scan) {
        if (i + 4 > n || data[i] != 0xFF) return -1;
        const int m = data[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { i += 2; continue; }
        const size_t L = ((size_t)data[i + 2] << 8) | data[i + 3];
        if (L < 2 || i + 2 + L > n) return -1;
        const uint8_t* s = data + i + 4;
        const size_t len = L - 2;
        if (m == 0xDB) {
            size_t k = 0;
            while (k < len) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                if (tq > 3 || pq > 1 || k + 1 + (pq ? 128 : 64) > len) return -1;
                for (int j = 0; j < 64; j++)
                    qt[tq][kZigzag[j]] = pq ? (uint16_t)((s[k + 1 + 2 * j] << 8) | s[k + 2 + 2 * j]) : s[k + 1 + j];
                qset[tq] = true;
                k += 1 + (pq ? 128 : 64);
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (len < 6 || s[0] != 8) return -2;
            H = (s[1] << 8) | s[2];
            W = (s[3] << 8) | s[4];
            nc = s[5];
            if (H <= 0 || W <= 0) return -2;
            if ((nc != 1 && nc != 3) || len < 6 + 3 * (size_t)nc) return -2;
            for (int c = 0; c < nc; c++) {
                comp[c].id = s[6 + 3 * c];
                comp[c].h = s[7 + 3 * c] >> 4;
                comp[c].v = s[7 + 3 * c] & 15;
                comp[c].tq = s[8 + 3 * c];
                if (comp[c].tq > 3) return -1;
            }
        } else if (m == 0xC4) {
            size_t k = 0;
            while (k < len) {
                if (k + 17 > len) return -1;
                const int tc = s[k] >> 4, th = s[k] & 15;
                int total = 0;
                for (int l = 0; l < 16; l++) total += s[k + 1 + l];
                if (tc > 1 || th > 3 || k + 17 + total > len) return -1;
                if (!(tc ? ac[th] : dc[th]).build(s + k + 1, s + k + 17, total)) return -1;
                anyDht = true;
                k += 17 + total;
            }
        } else if (m == 0xDD) {
            if (len < 2) return -1;
            restart = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!nc || len < 1 || s[0] != nc || len < 1 + 2 * (size_t)nc + 3) return -2;  /* one scan with every component */
            for (int c = 0; c < nc; c++) {
                if (s[1 + 2 * c] != comp[c].id) return -2;
                comp[c].td = s[2 + 2 * c] >> 4;
                comp[c].ta = s[2 + 2 * c] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) return -1;
            }
            if (s[1 + 2 * nc] != 0 || s[2 + 2 * nc] != 63 || s[3 + 2 * nc] != 0) return -2;
            scan = i + 2 + L;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC8)) {
            return -2;  /* progressive, lossless, differential, arithmetic coding */
        }
        i += 2 + L;
    }
    if (!anyDht) {  /* MJPEG frames leave the tables out */
        dc[0].build(kStdDcLumBits, kStdDcVal, 12);
        dc[1].build(kStdDcChrBits, kStdDcVal, 12);
        ac[0].build(kStdAcLumBits, kStdAcLumVal, 162);
        ac[1].build(kStdAcChrBits, kStdAcChrVal, 162);
    }
    if (nc == 1) { comp[0].h = comp[0].v = 1; }  /* a single component is never subsampled against itself */
    const int hmax = comp[0].h, vmax = comp[0].v;
    if (nc == 3) {
        if (comp[1].h != 1 || comp[1].v != 1 || comp[2].h != 1 || comp[2].v != 1) return -2;
        if (!((hmax == 1 && vmax == 1) || (hmax == 2 && vmax == 1) || (hmax == 2 && vmax == 2))) return -2;
    }
    for (int c = 0; c < nc; c++)
        if (!qset[comp[c].tq] || !dc[comp[c].td].set || !ac[comp[c].ta].set) return -1;
    if (cap < (size_t)H * W * 3 || !bgr) return -4;
    const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    std::vector<std::vector<uint8_t>> plane(nc);
    for (int c = 0; c < nc; c++) {
        comp[c].bw = mcux * comp[c].h;
        comp[c].bh = mcuy * comp[c].v;
        plane[c].assign((size_t)comp[c].bw * 8 * comp[c].bh * 8, 0);
    }

    /* ---- entropy decoding + inverse DCT, MCU by MCU (jdhuff.c decode_mcu, jdcoefct.c) ---- */
    Bits b;
    b.p = data + scan;
    b.end = data + n;
    int pred[3] = {0, 0, 0};
    int toRestart = restart, nextRst = 0;
    bool ok = true;
    int16_t coef[64];
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            if (restart && toRestart == 0) {
                b.align();
                /* (fill bytes 0xFF may precede the marker) */
                while (b.p + 1 < b.end && b.p[0] == 0xFF && b.p[1] == 0xFF) b.p++;
                if (!(b.p + 1 < b.end && b.p[0] == 0xFF && b.p[1] == 0xD0 + nextRst)) return -3;
                b.p += 2;
                b.hitMarker = false;
                nextRst = (nextRst + 1) & 7;
                pred[0] = pred[1] = pred[2] = 0;
                toRestart = restart;
            }
            for (int c = 0; c < nc; c++)
                for (int by = 0; by < comp[c].v; by++)
                    for (int bx = 0; bx < comp[c].h; bx++) {
                        memset(coef, 0, sizeof coef);
                        int s = decode_sym(b, dc[comp[c].td], ok);
                        if (!ok || s > 16) return -3;
                        if (s) pred[c] += extend(b.get(s), s);
                        coef[0] = (int16_t)pred[c];
                        for (int k = 1; k < 64; k++) {
                            const int rs = decode_sym(b, ac[comp[c].ta], ok);
                            if (!ok) return -3;
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s) {
                                k += r;
                                if (k > 63) return -3;
                                coef[kZigzag[k]] = (int16_t)extend(b.get(s), s);
                            } else {
                                if (r != 15) break;
                                k += 15;
                            }
                        }
                        const int stride = comp[c].bw * 8;
                        idct_block(coef, qt[comp[c].tq],
                                   plane[c].data() + (size_t)(my * comp[c].v + by) * 8 * stride + (size_t)(mx * comp[c].h + bx) * 8, stride);
                    }
            if (restart) toRestart--;
        }

    /* ---- upsampling (jdsample.c) and colour conversion (jdcolor.c) ---- */
    if (nc == 1) {
        const int stride = comp[0].bw * 8;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const uint8_t v = plane[0][(size_t)y * stride + x];
                uint8_t* o = bgr + ((size_t)y * W + x) * 3;
                o[0] = o[1] = o[2] = v;
            }
        return 0;
    }
    const int ys = comp[0].bw * 8, cs = comp[1].bw * 8;
    const int cw = (W + hmax - 1) / hmax, chh = (H + vmax - 1) / vmax;  /* downsampled_width / _height of the chroma planes */
    std::vector<uint8_t> up[2];
    for (int c = 0; c < 2; c++) {
        const uint8_t* src = plane[1 + c].data();
        up[c].assign((size_t)(chh * vmax) * (cw * hmax), 0);
        const int us = cw * hmax;
        if (hmax == 1) {
            for (int y = 0; y < chh; y++) memcpy(&up[c][(size_t)y * us], src + (size_t)y * cs, cw);
        } else if (vmax == 1) {  /* h2v1_fancy_upsample */
            for (int y = 0; y < chh; y++) {
                const uint8_t* in = src + (size_t)y * cs;
                uint8_t* out = &up[c][(size_t)y * us];
                if (cw <= 2) {  /* (h2v1_upsample: replication, for planes too narrow for the filter) */
                    for (int x = 0; x < cw; x++) out[2 * x] = out[2 * x + 1] = in[x];
                    continue;
                }
                out[0] = in[0];
                out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (int x = 1; x < cw - 1; x++) {
                    out[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
                    out[2 * x + 1] = (uint8_t)((in[x] * 3 + in[x + 1] + 2) >> 2);
                }
                out[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
                out[2 * cw - 1] = in[cw - 1];
            }
        } else {  /* h2v2_fancy_upsample */
            for (int y = 0; y < chh; y++)
                for (int v = 0; v < 2; v++) {
                    const int yn = v == 0 ? (y > 0 ? y - 1 : 0) : (y < chh - 1 ? y + 1 : chh - 1);  /* context rows: jdmainct.c */
                    const uint8_t* in0 = src + (size_t)y * cs;
                    const uint8_t* in1 = src + (size_t)yn * cs;
                    uint8_t* out = &up[c][(size_t)(2 * y + v) * us];
                    if (cw <= 2) {  /* (h2v2_upsample: replication) */
                        for (int x = 0; x < cw; x++) out[2 * x] = out[2 * x + 1] = in0[x];
                        continue;
                    }
                    int last, cur = in0[0] * 3 + in1[0], next = in0[1] * 3 + in1[1];
                    out[0] = (uint8_t)((cur * 4 + 8) >> 4);
                    out[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                    last = cur;
                    cur = next;
                    for (int x = 1; x < cw - 1; x++) {
                        next = in0[x + 1] * 3 + in1[x + 1];
                        out[2 * x] = (uint8_t)((cur * 3 + last + 8) >> 4);
                        out[2 * x + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                        last = cur;
                        cur = next;
                    }
                    out[2 * cw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4);
                    out[2 * cw - 1] = (uint8_t)((cur * 4 + 7) >> 4);
                }
        }
    }
    const int us = cw * hmax;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int yy = plane[0][(size_t)y * ys + x];
            const int cb = up[0][(size_t)y * us + x] - 128, cr = up[1][(size_t)y * us + x] - 128;
            uint8_t* o = bgr + ((size_t)y * W + x) * 3;
            o[2] = clamp8(yy + ((91881 * cr + 32768) >> 16));
            o[1] = clamp8(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
            o[0] = clamp8(yy + ((116130 * cb + 32768) >> 16));
        }
    return 0;
}

} /* extern "C" */
