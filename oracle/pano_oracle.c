/*
 * pano_oracle.c - see pano_oracle.h.  TEST INFRASTRUCTURE, PARITY UNPINNED.
 *
 * Every function names the OpenCV-3.4 routine it restates and the reference
 * call site (file:line under /root/reference) that reaches it.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC (see Makefile).
 * -ffp-contract=off matters: OpenCV's x86 builds do not fuse a*b+c, and the f32
 * expressions below are written in OpenCV's evaluation order.
 */
#define _GNU_SOURCE
#include "pano_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PO_PI_F ((float)3.1415926535897932384626433832795)

static int g_threads = 1;
void po_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int po_get_threads(void) { return g_threads; }

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* cvRound(float): SSE cvtss2si / cvtps2dq - round-half-even; "integer indefinite"
 * (INT_MIN) on overflow or NaN. */
static inline int cv_round_f(float v) {
    if (!(v >= -2147483648.f && v < 2147483648.f)) return INT_MIN;
    return (int)lrintf(v);
}
static inline int cv_round_d(double v) {
    if (!(v >= -2147483648.0 && v < 2147483648.0)) return INT_MIN;
    return (int)lrint(v);
}
static inline int16_t sat16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* cv::borderInterpolate (core/src/copy.cpp) for REFLECT / REFLECT_101 / CONSTANT(-1) */
static inline int border_interpolate(int p, int len, int border) {
    if ((unsigned)p < (unsigned)len) return p;
    if (border == PO_BORDER_CONSTANT) return -1;
    {
        int delta = border == PO_BORDER_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
    }
    return p;
}

/* ======================================================================== A1 */

/* cv::invert, n==3, CV_32F (core/src/lapack.cpp): closed form in double, rounded to float */
static void invert3x3_f(const float* S, float* D) {
#define Sf(r, c) S[(r) * 3 + (c)]
    double d = Sf(0, 0) * ((double)Sf(1, 1) * Sf(2, 2) - (double)Sf(1, 2) * Sf(2, 1)) -
               Sf(0, 1) * ((double)Sf(1, 0) * Sf(2, 2) - (double)Sf(1, 2) * Sf(2, 0)) +
               Sf(0, 2) * ((double)Sf(1, 0) * Sf(2, 1) - (double)Sf(1, 1) * Sf(2, 0));
    if (d != 0.) {
        double t[9];
        d = 1. / d;
        t[0] = (((double)Sf(1, 1) * Sf(2, 2) - (double)Sf(1, 2) * Sf(2, 1)) * d);
        t[1] = (((double)Sf(0, 2) * Sf(2, 1) - (double)Sf(0, 1) * Sf(2, 2)) * d);
        t[2] = (((double)Sf(0, 1) * Sf(1, 2) - (double)Sf(0, 2) * Sf(1, 1)) * d);
        t[3] = (((double)Sf(1, 2) * Sf(2, 0) - (double)Sf(1, 0) * Sf(2, 2)) * d);
        t[4] = (((double)Sf(0, 0) * Sf(2, 2) - (double)Sf(0, 2) * Sf(2, 0)) * d);
        t[5] = (((double)Sf(0, 2) * Sf(1, 0) - (double)Sf(0, 0) * Sf(1, 2)) * d);
        t[6] = (((double)Sf(1, 0) * Sf(2, 1) - (double)Sf(1, 1) * Sf(2, 0)) * d);
        t[7] = (((double)Sf(0, 1) * Sf(2, 0) - (double)Sf(0, 0) * Sf(2, 1)) * d);
        t[8] = (((double)Sf(0, 0) * Sf(1, 1) - (double)Sf(0, 1) * Sf(1, 0)) * d);
        for (int i = 0; i < 9; i++) D[i] = (float)t[i];
    } else {
        for (int i = 0; i < 9; i++) D[i] = 0.f;
    }
#undef Sf
}

/* cv::gemm small-matrix path, 3x3 CV_32F (core/src/matmul.cpp): f32 products summed left to right */
static void matmul3x3_f(const float* a, const float* b, float* d) {
    for (int i = 0; i < 3; i++) {
        float t0 = a[i * 3 + 0] * b[0] + a[i * 3 + 1] * b[3] + a[i * 3 + 2] * b[6];
        float t1 = a[i * 3 + 0] * b[1] + a[i * 3 + 1] * b[4] + a[i * 3 + 2] * b[7];
        float t2 = a[i * 3 + 0] * b[2] + a[i * 3 + 1] * b[5] + a[i * 3 + 2] * b[8];
        d[i * 3 + 0] = t0;
        d[i * 3 + 1] = t1;
        d[i * 3 + 2] = t2;
    }
}

/* ProjectorBase::setCameraParams (stitching/src/warpers.cpp); reached from
 * RotationWarperBase::warp / warpRoi, reference ocvstitcher.hpp:1171, :1057 */
void po_projector_set(po_projector* p, int kind, float scale, const float K[9], const float R[9]) {
    float kinv[9];
    p->kind = kind;
    p->scale = scale;
    for (int i = 0; i < 9; i++) p->k[i] = K[i];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) p->rinv[r * 3 + c] = R[c * 3 + r];
    invert3x3_f(K, kinv);
    matmul3x3_f(R, kinv, p->r_kinv);
    matmul3x3_f(K, p->rinv, p->k_rinv);
}

/* ---- variant switches (test infrastructure for the OpenCV pin, tests/test_oracle_variants.py) ---------------------------------
 * The transcendental functions of the projectors come from the platform's libm (glibc on x86-64 here, glibc on aarch64 on the
 * reference's Jetson): sinf / cosf / atan2f / acosf are not correctly rounded and may differ by an ulp between builds.  With a
 * perturbation set, every result of those four is moved by a whole number of ulps before it is used: 1: +1 ulp, 2: -1 ulp,
 * 3: -1 / 0 / +1 by a hash of the argument bits and the seed (a platform that disagrees "sometimes").  0 (default): libm as is. */
static int g_trig_mode = 0;
static unsigned g_trig_seed = 0;
void po_set_trig_perturbation(int mode, unsigned seed) { g_trig_mode = mode; g_trig_seed = seed; }
static inline float ulp_step(float v, int k) {  /* k in {-1, 0, +1}: the next representable float towards -inf / +inf */
    if (k == 0 || v != v || v == 0.f || isinf(v)) return v;
    return nextafterf(v, k > 0 ? INFINITY : -INFINITY);
}
static inline float trig_fix(float result, float arg, unsigned site) {
    if (!g_trig_mode) return result;
    int k;
    if (g_trig_mode == 1) k = 1;
    else if (g_trig_mode == 2) k = -1;
    else {
        union { float f; uint32_t u; } a;
        a.f = arg;
        uint32_t h = (a.u ^ (site * 0x9e3779b9u) ^ g_trig_seed) * 0x85ebca6bu;
        h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        k = (int)(h % 3u) - 1;
    }
    return ulp_step(result, k);
}
#define PO_SINF(x, site) trig_fix(sinf(x), (x), (site))
#define PO_COSF(x, site) trig_fix(cosf(x), (x), (site))
#define PO_ATAN2F(y, x, site) trig_fix(atan2f((y), (x)), (y) + (x), (site))
#define PO_ACOSF(x, site) trig_fix(acosf(x), (x), (site))

/* SphericalProjector::mapForward / CylindricalProjector::mapForward (warpers_inl.hpp) */
void po_map_forward(const po_projector* p, float x, float y, float* u, float* v) {
    const float* m = p->r_kinv;
    float x_ = m[0] * x + m[1] * y + m[2];
    float y_ = m[3] * x + m[4] * y + m[5];
    float z_ = m[6] * x + m[7] * y + m[8];
    if (p->kind == PO_SPHERICAL) {
        *u = p->scale * PO_ATAN2F(x_, z_, 1u);
        float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        *v = p->scale * (PO_PI_F - PO_ACOSF(w == w ? w : 0, 2u));
    } else {
        *u = p->scale * PO_ATAN2F(x_, z_, 1u);
        *v = p->scale * y_ / sqrtf(x_ * x_ + z_ * z_);
    }
}

/* SphericalProjector::mapBackward / CylindricalProjector::mapBackward (warpers_inl.hpp) */
void po_map_backward(const po_projector* p, float u, float v, float* x, float* y) {
    const float* m = p->k_rinv;
    float x_, y_, z_;
    u /= p->scale;
    v /= p->scale;
    if (p->kind == PO_SPHERICAL) {
        float sinv = PO_SINF(PO_PI_F - v, 3u);
        x_ = sinv * PO_SINF(u, 4u);
        y_ = PO_COSF(PO_PI_F - v, 5u);
        z_ = sinv * PO_COSF(u, 6u);
    } else {
        x_ = PO_SINF(u, 4u);
        y_ = v;
        z_ = PO_COSF(u, 6u);
    }
    float z;
    *x = m[0] * x_ + m[1] * y_ + m[2] * z_;
    *y = m[3] * x_ + m[4] * y_ + m[5] * z_;
    z = m[6] * x_ + m[7] * y_ + m[8] * z_;
    if (z > 0) {
        *x /= z;
        *y /= z;
    } else
        *x = *y = -1;
}

/* ======================================================================== A2 */

/* RotationWarperBase::detectResultRoiByBorder (warpers_inl.hpp) */
static void roi_by_border(const po_projector* p, int w, int h, float* tl_u, float* tl_v, float* br_u, float* br_v) {
    float tl_uf = 3.402823466e+38F, tl_vf = 3.402823466e+38F;
    float br_uf = -3.402823466e+38F, br_vf = -3.402823466e+38F;
    float u, v;
    for (int x = 0; x < w; ++x) {
        po_map_forward(p, (float)x, 0, &u, &v);
        tl_uf = fminf(tl_uf, u); tl_vf = fminf(tl_vf, v);
        br_uf = fmaxf(br_uf, u); br_vf = fmaxf(br_vf, v);
        po_map_forward(p, (float)x, (float)(h - 1), &u, &v);
        tl_uf = fminf(tl_uf, u); tl_vf = fminf(tl_vf, v);
        br_uf = fmaxf(br_uf, u); br_vf = fmaxf(br_vf, v);
    }
    for (int y = 0; y < h; ++y) {
        po_map_forward(p, 0, (float)y, &u, &v);
        tl_uf = fminf(tl_uf, u); tl_vf = fminf(tl_vf, v);
        br_uf = fmaxf(br_uf, u); br_vf = fmaxf(br_vf, v);
        po_map_forward(p, (float)(w - 1), (float)y, &u, &v);
        tl_uf = fminf(tl_uf, u); tl_vf = fminf(tl_vf, v);
        br_uf = fmaxf(br_uf, u); br_vf = fmaxf(br_vf, v);
    }
    *tl_u = tl_uf; *tl_v = tl_vf; *br_u = br_uf; *br_v = br_vf;
}

/* SphericalWarper::detectResultRoi (stitching/src/warpers.cpp) = by-border + pole fix-up;
 * CylindricalWarper::detectResultRoi = by-border only (stitching/detail/warpers.hpp) */
void po_detect_result_roi(const po_projector* p, int w, int h, int tl[2], int br[2]) {
    float tl_uf, tl_vf, br_uf, br_vf;
    roi_by_border(p, w, h, &tl_uf, &tl_vf, &br_uf, &br_vf);
    /* by-border truncates to int first, the spherical override then re-floats those ints */
    int tlx = (int)tl_uf, tly = (int)tl_vf, brx = (int)br_uf, bry = (int)br_vf;
    if (p->kind == PO_SPHERICAL) {
        tl_uf = (float)tlx; tl_vf = (float)tly; br_uf = (float)brx; br_vf = (float)bry;
        float x = p->rinv[1], y = p->rinv[4], z = p->rinv[7];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < w && y_ > 0.f && y_ < h) {
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, (float)(3.1415926535897932384626433832795 * p->scale));
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, (float)(3.1415926535897932384626433832795 * p->scale));
            }
        }
        x = p->rinv[1]; y = -p->rinv[4]; z = p->rinv[7];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < w && y_ > 0.f && y_ < h) {
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, 0.f);
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, 0.f);
            }
        }
        tlx = (int)tl_uf; tly = (int)tl_vf; brx = (int)br_uf; bry = (int)br_vf;
    }
    tl[0] = tlx; tl[1] = tly; br[0] = brx; br[1] = bry;
}

/* RotationWarperBase::warpRoi: Rect(tl, Point(br.x+1, br.y+1)); reference ocvstitcher.hpp:1057 */
void po_warp_roi(const po_projector* p, int w, int h, int rect[4]) {
    int tl[2], br[2];
    po_detect_result_roi(p, w, h, tl, br);
    rect[0] = tl[0]; rect[1] = tl[1];
    rect[2] = br[0] + 1 - tl[0]; rect[3] = br[1] + 1 - tl[1];
}

/* cv::detail::resultRoi(corners, sizes) (stitching/src/util.cpp); reference ocvstitcher.hpp:1110 */
void po_result_roi(int n, const int* corners, const int* sizes, int rect[4]) {
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; i++) {
        if (corners[2 * i] < tlx) tlx = corners[2 * i];
        if (corners[2 * i + 1] < tly) tly = corners[2 * i + 1];
        if (corners[2 * i] + sizes[2 * i] > brx) brx = corners[2 * i] + sizes[2 * i];
        if (corners[2 * i + 1] + sizes[2 * i + 1] > bry) bry = corners[2 * i + 1] + sizes[2 * i + 1];
    }
    rect[0] = tlx; rect[1] = tly; rect[2] = brx - tlx; rect[3] = bry - tly;
}

/* ======================================================================== A3 */

/* RotationWarperBase::buildMaps (warpers_inl.hpp): maps are (br-tl+1) sized, u,v integer grid */
void po_build_maps(const po_projector* p, int sw, int sh, float* xmap, float* ymap) {
    int tl[2], br[2];
    po_detect_result_roi(p, sw, sh, tl, br);
    int dw = br[0] - tl[0] + 1, dh = br[1] - tl[1] + 1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int v = 0; v < dh; ++v)
        for (int u = 0; u < dw; ++u)
            po_map_backward(p, (float)(u + tl[0]), (float)(v + tl[1]), &xmap[(size_t)v * dw + u], &ymap[(size_t)v * dw + u]);
}

/* cv::remap, CV_8U, map1/map2 CV_32FC1 (imgproc/src/imgwarp.cpp RemapInvoker + remapBilinear<FixedPtCast<int,uchar,15>>
 * / remapNearest).  INTER_BITS=5, INTER_REMAP_COEF_BITS=15.  cval = 0. */
void po_remap_8u(const uint8_t* src, int sw, int sh, size_t sstride, int cn, const float* xmap, const float* ymap,
                 int dw, int dh, int interp, int border, uint8_t* dst, size_t dstride) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        const float* sX = xmap + (size_t)dy * dw;
        const float* sY = ymap + (size_t)dy * dw;
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx) {
            if (interp == PO_INTER_NEAREST) {
                /* XY = saturate_cast<short>(float) = clamp(cvRound) */
                int sx = sat16(cv_round_f(sX[dx])), sy = sat16(cv_round_f(sY[dx]));
                if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
                    for (int c = 0; c < cn; c++) D[dx * cn + c] = src[(size_t)sy * sstride + sx * cn + c];
                } else if (border == PO_BORDER_CONSTANT) {
                    for (int c = 0; c < cn; c++) D[dx * cn + c] = 0;
                } else {
                    sx = border_interpolate(sx, sw, border);
                    sy = border_interpolate(sy, sh, border);
                    for (int c = 0; c < cn; c++) D[dx * cn + c] = src[(size_t)sy * sstride + sx * cn + c];
                }
                continue;
            }
            int isx = cv_round_f(sX[dx] * 32.f), isy = cv_round_f(sY[dx] * 32.f);
            int a = isx & 31, b = isy & 31;
            int sx = sat16(isx >> 5), sy = sat16(isy >> 5);
            /* BilinearTab_i: exact for 1/32 fractions, sums to 32768 */
            int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32, w11 = a * b * 32;
            int x0, x1, y0, y1;
            if (border == PO_BORDER_CONSTANT && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
                for (int c = 0; c < cn; c++) D[dx * cn + c] = 0;
                continue;
            }
            x0 = border_interpolate(sx, sw, border);
            x1 = border_interpolate(sx + 1, sw, border);
            y0 = border_interpolate(sy, sh, border);
            y1 = border_interpolate(sy + 1, sh, border);
            for (int c = 0; c < cn; c++) {
                int p00 = (x0 >= 0 && y0 >= 0) ? src[(size_t)y0 * sstride + x0 * cn + c] : 0;
                int p01 = (x1 >= 0 && y0 >= 0) ? src[(size_t)y0 * sstride + x1 * cn + c] : 0;
                int p10 = (x0 >= 0 && y1 >= 0) ? src[(size_t)y1 * sstride + x0 * cn + c] : 0;
                int p11 = (x1 >= 0 && y1 >= 0) ? src[(size_t)y1 * sstride + x1 * cn + c] : 0;
                D[dx * cn + c] = sat8((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
            }
        }
    }
}

/* RotationWarperBase::warp (warpers_inl.hpp); reference ocvstitcher.hpp:1171 (LINEAR, REFLECT),
 * :1085 (NEAREST, CONSTANT) */
void po_warp_8u(const po_projector* p, const uint8_t* src, int sw, int sh, size_t sstride, int cn, int interp,
                int border, uint8_t* dst, int corner[2]) {
    int r[4];
    po_warp_roi(p, sw, sh, r);
    float* xmap = (float*)malloc(sizeof(float) * (size_t)r[2] * r[3]);
    float* ymap = (float*)malloc(sizeof(float) * (size_t)r[2] * r[3]);
    po_build_maps(p, sw, sh, xmap, ymap);
    po_remap_8u(src, sw, sh, sstride, cn, xmap, ymap, r[2], r[3], interp, border, dst, (size_t)r[2] * cn);
    free(xmap);
    free(ymap);
    corner[0] = r[0];
    corner[1] = r[1];
}

/* ======================================================================== A4 */

/* cv::pyrDown, CV_16S (pyramids.cpp pyrDown_<FixPtCast<short,8>>): 5x5 [1 4 6 4 1]^2, REFLECT_101,
 * dst = ((w+1)/2, (h+1)/2), (v+128)>>8 */
void po_pyr_down_16s(const int16_t* src, int w, int h, int cn, int16_t* dst) {
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < dh; y++) {
        int* rows = (int*)malloc(sizeof(int) * 5 * (size_t)dw * cn);
        for (int k = 0; k < 5; k++) {
            int sy = border_interpolate(2 * y + k - 2, h, PO_BORDER_REFLECT_101);
            const int16_t* s = src + (size_t)sy * w * cn;
            int* row = rows + (size_t)k * dw * cn;
            for (int x = 0; x < dw; x++) {
                int x0 = border_interpolate(2 * x - 2, w, PO_BORDER_REFLECT_101) * cn;
                int x1 = border_interpolate(2 * x - 1, w, PO_BORDER_REFLECT_101) * cn;
                int x2 = 2 * x * cn;
                int x3 = border_interpolate(2 * x + 1, w, PO_BORDER_REFLECT_101) * cn;
                int x4 = border_interpolate(2 * x + 2, w, PO_BORDER_REFLECT_101) * cn;
                for (int c = 0; c < cn; c++)
                    row[x * cn + c] = s[x2 + c] * 6 + (s[x1 + c] + s[x3 + c]) * 4 + s[x0 + c] + s[x4 + c];
            }
        }
        int16_t* d = dst + (size_t)y * dw * cn;
        const int *r0 = rows, *r1 = rows + (size_t)dw * cn, *r2 = r1 + (size_t)dw * cn, *r3 = r2 + (size_t)dw * cn,
                  *r4 = r3 + (size_t)dw * cn;
        for (int x = 0; x < dw * cn; x++) d[x] = sat16((r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x] + 128) >> 8);
        free(rows);
    }
}

/* cv::pyrDown, CV_32F (pyrDown_<FltCast<float,8>>, imgproc/src/pyramids.cpp).  The sum 6 c + 4 (l1 + r1) + l2 + r2 is evaluated
 * in an order that depends on the OpenCV BUILD - in f32 the order decides the last ulp of the blend weights:
 *   scalar code (every build's borders and tails; the whole of a build without SIMD):
 *       row = s2*6 + (s1+s3)*4 + s0 + s4             dst = (r2*6 + (r1+r3)*4 + r0 + r4) * (1/256)
 *   vertical vector body (PyrDownVec_32f: SSE2 since 2.x, NEON since 3.0, universal intrinsics in late 3.4.x), x in
 *   [0, floor(width / n) * n), n = 8 (the hand-written SSE2 / NEON loops step 8 floats), 4 or 8 or 16 (v_float32::nlanes):
 *       SSE2 / universal   t = ((r0+r4) + (r2+r2)) + ((r1+r3)+r2)*4          dst = t * (1/256)
 *       NEON               t = ((r0+r4) + (r2+r2)) + ((r1+r2)+r3)*4
 *     (the multiply by 4 is exact, so a fused multiply-add gives the same bits)
 *   horizontal vector body (PyrDownVecH<float,float,1>, universal intrinsics, late 3.4.x and 4.x only), x in [1, 1 + floor((width0
 *   - 1) / n) * n):
 *       row = r2*6 + ((r1+r3)*4 + (r0+r4)), the outer multiply-add fused where the build dispatches to FMA3 / NEON-FMA (6 x is not
 *       exact in f32: fused and unfused differ)
 * Recalled from the OpenCV sources, NOT checkable here (no OpenCV in this container): which of these a given build runs is exactly
 * what tools/opencv_pin/pin.cpp's unit stages `pyrdown32f_*` settle.  Default = scalar everywhere (what the product's weight kernel
 * computes, pano_init.hip); po_set_pyrdown32f_variant selects the others for the pin loader and the robustness test. */
static int g_pd_vert = 0, g_pd_vbody = 8, g_pd_horz = 0, g_pd_hbody = 4;
void po_set_pyrdown32f_variant(int vertical, int vbody, int horizontal, int hbody) {
    g_pd_vert = vertical; g_pd_vbody = vbody > 0 ? vbody : 8;
    g_pd_horz = horizontal; g_pd_hbody = hbody > 0 ? hbody : 4;
}
void po_pyr_down_32f(const float* src, int w, int h, float* dst) {
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    const int vert = g_pd_vert, horz = g_pd_horz;
    const int vend = vert ? dw / g_pd_vbody * g_pd_vbody : 0;            /* the vertical vector body ends here */
    /* pyrDown_: width0 = min((ssize.width - PD_SZ/2 - 1) / 2 + 1, dsize.width): the columns whose five taps need no border table */
    int width0 = (w - 2 - 1) / 2 + 1;
    if (width0 > dw) width0 = dw;
    const int hend = horz && width0 > 1 ? 1 + (width0 - 1) / g_pd_hbody * g_pd_hbody : 0;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < dh; y++) {
        float* rows = (float*)malloc(sizeof(float) * 5 * (size_t)dw);
        for (int k = 0; k < 5; k++) {
            int sy = border_interpolate(2 * y + k - 2, h, PO_BORDER_REFLECT_101);
            const float* s = src + (size_t)sy * w;
            float* row = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                int x0 = border_interpolate(2 * x - 2, w, PO_BORDER_REFLECT_101);
                int x1 = border_interpolate(2 * x - 1, w, PO_BORDER_REFLECT_101);
                int x3 = border_interpolate(2 * x + 1, w, PO_BORDER_REFLECT_101);
                int x4 = border_interpolate(2 * x + 2, w, PO_BORDER_REFLECT_101);
                if (x >= 1 && x < hend) {
                    const float inner = (s[x1] + s[x3]) * 4 + (s[x0] + s[x4]);
                    row[x] = horz == 2 ? fmaf(s[2 * x], 6.f, inner) : s[2 * x] * 6 + inner;
                } else
                    row[x] = s[2 * x] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        const float *r0 = rows, *r1 = rows + dw, *r2 = r1 + dw, *r3 = r2 + dw, *r4 = r3 + dw;
        float* d = dst + (size_t)y * dw;
        for (int x = 0; x < dw; x++) {
            if (x < vend) {
                const float a = (r0[x] + r4[x]) + (r2[x] + r2[x]);
                const float b = vert == 2 ? (r1[x] + r2[x]) + r3[x] : (r1[x] + r3[x]) + r2[x];
                d[x] = (a + b * 4) * (1.f / 256);
            } else
                d[x] = (r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x]) * (1.f / 256);
        }
        free(rows);
    }
}

/* cv::pyrUp, CV_16S (pyrUp_<FixPtCast<short,6>>), dst exactly 2w x 2h (always true inside
 * MultiBandBlender because every tile/canvas is a multiple of 2^bands).
 * Horizontal: even 6s[x]+s[x-1]+s[x+1], odd 4(s[x]+s[x+1]); left edge reflect-101
 * (6s0+2s1), right edge replicate (s[n-2]+7s[n-1], 8s[n-1]); rows likewise; (v+32)>>6. */
void po_pyr_up_16s(const int16_t* src, int w, int h, int cn, int16_t* dst) {
    int dw = 2 * w;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < h; y++) {
        int* rows = (int*)malloc(sizeof(int) * 3 * (size_t)dw * cn);
        for (int k = 0; k < 3; k++) {
            int sy = y + k - 1;
            /* borderInterpolate(sy*2, h*2, REFLECT_101)/2 */
            sy = border_interpolate(sy * 2, h * 2, PO_BORDER_REFLECT_101) / 2;
            const int16_t* s = src + (size_t)sy * w * cn;
            int* row = rows + (size_t)k * dw * cn;
            for (int x = 0; x < w; x++) {
                int xm = x > 0 ? x - 1 : (w > 1 ? 1 : 0);
                int xp = x < w - 1 ? x + 1 : w - 1;
                for (int c = 0; c < cn; c++) {
                    row[(2 * x) * cn + c] = s[xm * cn + c] + s[x * cn + c] * 6 + s[xp * cn + c];
                    row[(2 * x + 1) * cn + c] = (s[x * cn + c] + s[xp * cn + c]) * 4;
                }
            }
        }
        const int *r0 = rows, *r1 = rows + (size_t)dw * cn, *r2 = r1 + (size_t)dw * cn;
        int16_t* d0 = dst + (size_t)(2 * y) * dw * cn;
        int16_t* d1 = d0 + (size_t)dw * cn;
        for (int x = 0; x < dw * cn; x++) {
            d1[x] = sat16(((r1[x] + r2[x]) * 4 + 32) >> 6);
            d0[x] = sat16((r0[x] + r1[x] * 6 + r2[x] + 32) >> 6);
        }
        free(rows);
    }
}

/* ======================================================================== misc imgproc */

/* cv::copyMakeBorder (core/src/copy.cpp) */
void po_copy_make_border_16s(const int16_t* src, int w, int h, int cn, int top, int bottom, int left, int right,
                             int border, int16_t* dst) {
    int dw = w + left + right, dh = h + top + bottom;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < dh; y++) {
        int sy = border_interpolate(y - top, h, border);
        for (int x = 0; x < dw; x++) {
            int sx = border_interpolate(x - left, w, border);
            for (int c = 0; c < cn; c++)
                dst[((size_t)y * dw + x) * cn + c] = (sx < 0 || sy < 0) ? 0 : src[((size_t)sy * w + sx) * cn + c];
        }
    }
}
void po_copy_make_border_32f(const float* src, int w, int h, int top, int bottom, int left, int right, float* dst) {
    int dw = w + left + right, dh = h + top + bottom;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int sx = x - left, sy = y - top;
            dst[(size_t)y * dw + x] = (sx < 0 || sy < 0 || sx >= w || sy >= h) ? 0.f : src[(size_t)sy * w + sx];
        }
}

/* cv::dilate(src, dst, Mat()) : 3x3 rect, centre anchor, BORDER_CONSTANT with the
 * morphology default border value (= ignored for max).  reference ocvstitcher.hpp:1097,1255 */
void po_dilate3x3_8u(const uint8_t* src, int w, int h, uint8_t* dst) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int m = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    if (src[(size_t)yy * w + xx] > m) m = src[(size_t)yy * w + xx];
                }
            dst[(size_t)y * w + x] = (uint8_t)m;
        }
}

/* cv::resize(..., INTER_LINEAR_EXACT), CV_8U (imgproc/src/resize.cpp resize_bitExact<uchar, interpolationLinear>):
 * ufixedpoint16 (8.8) coefficients from IEEE-double src = (dst+0.5)/inv_scale - 0.5, round-half-even;
 * horizontal pass keeps 8.8, vertical pass 16.16, final (v + 32768) >> 16.
 * reference ocvstitcher.hpp:1099,1256 (mask upscale), :988 (seam-size frames) */
typedef struct { int ofs; int c0, c1; } lin_coef;
/* inv_scale: what cv::resize hands to resize_bitExact.  With an explicit dsize cv::resize recomputes it as
 * dsize/ssize; with an EMPTY dsize (resize(src, dst, Size(), fx, fy)) it stays fx, while dsize = cvRound(ssize*fx)
 * - so the sampling grid is 1/fx, NOT ssize/dsize (found and fixed in round 2: ocvstitcher.hpp:988,1230 pass
 * Size(), seam_work_aspect).  inv_scale <= 0 here means "explicit dsize". */
static void linear_exact_coeffs(int ssize, int dsize, double inv_scale, lin_coef* co, int* pmin, int* pmax) {
    if (!(inv_scale > 0)) inv_scale = (double)dsize / ssize;
    double scale = 1.0 / inv_scale;
    int minofst = 0, maxofst = dsize;
    for (int val = 0; val < dsize; val++) {
        double fval = scale * ((double)val + 0.5) - 0.5;
        int ival = (int)floor(fval);
        co[val].ofs = 0; co[val].c0 = 256; co[val].c1 = 0;
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                co[val].ofs = ival;
                co[val].c1 = (int)lrint((fval - (double)ival) * 256.0); /* cvRound(softdouble*256) */
                co[val].c0 = 256 - co[val].c1;
            } else {
                co[val].ofs = ssize - 1;
                if (val < maxofst) maxofst = val;
            }
        } else {
            if (val + 1 > minofst) minofst = val + 1;
        }
    }
    *pmin = minofst;
    *pmax = maxofst;
}
void po_resize_linear_exact_8u(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh) {
    po_resize_linear_exact_8u_fxy(src, sw, sh, cn, dst, dw, dh, 0.0, 0.0);
}
/* resize(src, dst, Size(), fx, fy, INTER_LINEAR_EXACT): dw, dh must be cvRound(sw*fx), cvRound(sh*fy) (po_resize_dsize);
 * fx, fy <= 0: resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR_EXACT) */
int po_resize_dsize(int ssize, double f) { return cv_round_d(ssize * f); }
void po_resize_linear_exact_8u_fxy(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh, double fx,
                                   double fy) {
    lin_coef* cx = (lin_coef*)malloc(sizeof(lin_coef) * dw);
    lin_coef* cy = (lin_coef*)malloc(sizeof(lin_coef) * dh);
    int minx, maxx, miny, maxy;
    linear_exact_coeffs(sw, dw, fx, cx, &minx, &maxx);
    linear_exact_coeffs(sh, dh, fy, cy, &miny, &maxy);
    if (maxx < minx) maxx = minx;
    if (maxy < miny) maxy = miny;
    /* horizontal pass of every source row into 8.8 */
    uint16_t* hbuf = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)sh * dw * cn);
    for (int y = 0; y < sh; y++) {
        const uint8_t* s = src + (size_t)y * sw * cn;
        uint16_t* d = hbuf + (size_t)y * dw * cn;
        for (int x = 0; x < dw; x++)
            for (int c = 0; c < cn; c++) {
                unsigned v;
                if (x < minx) v = (unsigned)s[c] << 8;
                else if (x >= maxx) v = (unsigned)s[(sw - 1) * cn + c] << 8;
                else {
                    unsigned a = s[cx[x].ofs * cn + c] * (unsigned)cx[x].c0;
                    unsigned b = s[(cx[x].ofs + 1) * cn + c] * (unsigned)cx[x].c1;
                    v = a + b;
                    if (v > 65535u) v = 65535u;
                }
                d[x * cn + c] = (uint16_t)v;
            }
    }
    for (int y = 0; y < dh; y++) {
        uint8_t* d = dst + (size_t)y * dw * cn;
        if (y < miny || y >= maxy) {
            const uint16_t* r = hbuf + (size_t)(y < miny ? 0 : sh - 1) * dw * cn;
            for (int x = 0; x < dw * cn; x++) d[x] = sat8((r[x] + 128) >> 8);
        } else {
            const uint16_t* r0 = hbuf + (size_t)cy[y].ofs * dw * cn;
            const uint16_t* r1 = r0 + (size_t)dw * cn;
            for (int x = 0; x < dw * cn; x++) {
                uint64_t v = (uint64_t)r0[x] * (unsigned)cy[y].c0 + (uint64_t)r1[x] * (unsigned)cy[y].c1;
                if (v > 0xffffffffull) v = 0xffffffffull;
                d[x] = sat8((int)((v + 32768) >> 16));
            }
        }
    }
    free(hbuf);
    free(cx);
    free(cy);
}

/* cv::distanceTransform(src, dst, DIST_L1, 3) (imgproc/src/distransform.cpp distanceTransform_3x3 with
 * mask {1,2}): exact city-block distance to the nearest zero pixel; image border = infinitely far. */
void po_distance_l1(const uint8_t* src, int w, int h, float* dst) {
    const int BIG = INT_MAX >> 2 >> 16 << 0; /* DIST_MAX >> DIST_SHIFT magnitude (~8191) */
    int* t = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v;
            if (!src[(size_t)y * w + x]) v = 0;
            else {
                v = BIG * 4;
                if (x > 0 && t[(size_t)y * w + x - 1] + 1 < v) v = t[(size_t)y * w + x - 1] + 1;
                if (y > 0 && t[(size_t)(y - 1) * w + x] + 1 < v) v = t[(size_t)(y - 1) * w + x] + 1;
            }
            t[(size_t)y * w + x] = v;
        }
    for (int y = h - 1; y >= 0; y--)
        for (int x = w - 1; x >= 0; x--) {
            int v = t[(size_t)y * w + x];
            if (x < w - 1 && t[(size_t)y * w + x + 1] + 1 < v) v = t[(size_t)y * w + x + 1] + 1;
            if (y < h - 1 && t[(size_t)(y + 1) * w + x] + 1 < v) v = t[(size_t)(y + 1) * w + x] + 1;
            t[(size_t)y * w + x] = v;
            dst[(size_t)y * w + x] = (float)v;
        }
    free(t);
}

/* ======================================================================== A5 */

/* ocvstitcher.hpp:1188-1195 / stitching_detailed.cpp:855-864 */
int po_bands_from_strength(int dst_w, int dst_h, float strength) {
    float blend_width = sqrtf((float)(dst_w * dst_h)) * strength / 100.f;
    if (blend_width < 1.f) return -1;
    return (int)(ceil((double)logf(blend_width) / log(2.)) - 1.);
}

#define PO_MAX_LEVELS 16
struct po_blender {
    int requested_bands; /* actual_num_bands_, or -1 for Blender::NO */
    int num_bands;
    int dst_roi[4], dst_roi_final[4];
    int lw[PO_MAX_LEVELS], lh[PO_MAX_LEVELS];
    int16_t* lap[PO_MAX_LEVELS];
    float* wgt[PO_MAX_LEVELS];
    uint8_t* dst_mask; /* Blender::NO */
    int last_tile[4], last_tblr[4];
};

po_blender* po_blender_create(int num_bands) {
    po_blender* b = (po_blender*)calloc(1, sizeof(po_blender));
    b->requested_bands = num_bands;
    return b;
}
static void blender_free_levels(po_blender* b) {
    for (int i = 0; i < PO_MAX_LEVELS; i++) {
        free(b->lap[i]); b->lap[i] = NULL;
        free(b->wgt[i]); b->wgt[i] = NULL;
    }
    free(b->dst_mask); b->dst_mask = NULL;
}
void po_blender_destroy(po_blender* b) {
    if (!b) return;
    blender_free_levels(b);
    free(b);
}

/* MultiBandBlender::prepare(Rect) / Blender::prepare (blenders.cpp); reference ocvstitcher.hpp:1198 */
void po_blender_prepare(po_blender* b, int n, const int* corners, const int* sizes) {
    int roi[4];
    po_result_roi(n, corners, sizes, roi);
    blender_free_levels(b);
    memcpy(b->dst_roi_final, roi, sizeof(roi));
    if (b->requested_bands < 0) { /* Blender::NO */
        b->num_bands = 0;
        memcpy(b->dst_roi, roi, sizeof(roi));
        b->lw[0] = roi[2]; b->lh[0] = roi[3];
        b->lap[0] = (int16_t*)calloc((size_t)roi[2] * roi[3] * 3, sizeof(int16_t));
        b->dst_mask = (uint8_t*)calloc((size_t)roi[2] * roi[3], 1);
        return;
    }
    double max_len = (double)(roi[2] > roi[3] ? roi[2] : roi[3]);
    int crop = (int)ceil(log(max_len) / log(2.0));
    b->num_bands = b->requested_bands < crop ? b->requested_bands : crop;
    int m = 1 << b->num_bands;
    roi[2] += (m - roi[2] % m) % m;
    roi[3] += (m - roi[3] % m) % m;
    memcpy(b->dst_roi, roi, sizeof(roi));
    b->lw[0] = roi[2]; b->lh[0] = roi[3];
    for (int i = 1; i <= b->num_bands; i++) {
        b->lw[i] = (b->lw[i - 1] + 1) / 2;
        b->lh[i] = (b->lh[i - 1] + 1) / 2;
    }
    for (int i = 0; i <= b->num_bands; i++) {
        b->lap[i] = (int16_t*)calloc((size_t)b->lw[i] * b->lh[i] * 3, sizeof(int16_t));
        b->wgt[i] = (float*)calloc((size_t)b->lw[i] * b->lh[i], sizeof(float));
    }
}

/* MultiBandBlender::feed / Blender::feed (blenders.cpp); reference ocvstitcher.hpp:1202 */
void po_blender_feed(po_blender* b, const int16_t* img, const uint8_t* mask, int w, int h, int tlx, int tly) {
    if (b->requested_bands < 0) { /* Blender::feed: masked copy, mask OR */
        int dx = tlx - b->dst_roi[0], dy = tly - b->dst_roi[1], W = b->dst_roi[2];
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                size_t d = (size_t)(dy + y) * W + dx + x;
                if (mask[(size_t)y * w + x])
                    for (int c = 0; c < 3; c++) b->lap[0][d * 3 + c] = img[((size_t)y * w + x) * 3 + c];
                b->dst_mask[d] |= mask[(size_t)y * w + x];
            }
        return;
    }
    const int nb = b->num_bands;
    const int* R = b->dst_roi;
    int gap = 3 * (1 << nb);
    int tl_new_x = R[0] > tlx - gap ? R[0] : tlx - gap;
    int tl_new_y = R[1] > tly - gap ? R[1] : tly - gap;
    int br_new_x = (R[0] + R[2]) < (tlx + w + gap) ? (R[0] + R[2]) : (tlx + w + gap);
    int br_new_y = (R[1] + R[3]) < (tly + h + gap) ? (R[1] + R[3]) : (tly + h + gap);
    tl_new_x = R[0] + (((tl_new_x - R[0]) >> nb) << nb);
    tl_new_y = R[1] + (((tl_new_y - R[1]) >> nb) << nb);
    int width = br_new_x - tl_new_x, height = br_new_y - tl_new_y;
    width += ((1 << nb) - width % (1 << nb)) % (1 << nb);
    height += ((1 << nb) - height % (1 << nb)) % (1 << nb);
    br_new_x = tl_new_x + width;
    br_new_y = tl_new_y + height;
    int dy = br_new_y - (R[1] + R[3]); if (dy < 0) dy = 0;
    int dx = br_new_x - (R[0] + R[2]); if (dx < 0) dx = 0;
    tl_new_x -= dx; br_new_x -= dx;
    tl_new_y -= dy; br_new_y -= dy;
    int top = tly - tl_new_y, left = tlx - tl_new_x;
    int bottom = br_new_y - tly - h, right = br_new_x - tlx - w;
    b->last_tile[0] = tl_new_x - R[0]; b->last_tile[1] = tl_new_y - R[1];
    b->last_tile[2] = width; b->last_tile[3] = height;
    b->last_tblr[0] = top; b->last_tblr[1] = bottom; b->last_tblr[2] = left; b->last_tblr[3] = right;

    /* source Laplacian pyramid (createLaplacePyr, 16S branch) */
    int16_t* pyr[PO_MAX_LEVELS];
    float* wp[PO_MAX_LEVELS];
    int pw[PO_MAX_LEVELS], ph[PO_MAX_LEVELS];
    pw[0] = width; ph[0] = height;
    pyr[0] = (int16_t*)malloc(sizeof(int16_t) * (size_t)width * height * 3);
    po_copy_make_border_16s(img, w, h, 3, top, bottom, left, right, PO_BORDER_REFLECT, pyr[0]);
    for (int i = 0; i < nb; i++) {
        pw[i + 1] = (pw[i] + 1) / 2; ph[i + 1] = (ph[i] + 1) / 2;
        pyr[i + 1] = (int16_t*)malloc(sizeof(int16_t) * (size_t)pw[i + 1] * ph[i + 1] * 3);
        po_pyr_down_16s(pyr[i], pw[i], ph[i], 3, pyr[i + 1]);
    }
    for (int i = 0; i < nb; i++) {
        int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * (size_t)pw[i] * ph[i] * 3);
        po_pyr_up_16s(pyr[i + 1], pw[i + 1], ph[i + 1], 3, tmp); /* sizes are exact doubles here */
        long cnt = (long)pw[i] * ph[i] * 3;
        int16_t* pi = pyr[i];
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long k = 0; k < cnt; k++) pi[k] = sat16((int)pi[k] - (int)tmp[k]);
        free(tmp);
    }
    /* weight Gaussian pyramid: mask * (1/255.) as f32, CONSTANT border, pyrDown */
    {
        float* wm = (float*)malloc(sizeof(float) * (size_t)w * h);
        const float s = (float)(1. / 255.);
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long k = 0; k < (long)w * h; k++) wm[k] = (float)mask[k] * s;
        wp[0] = (float*)malloc(sizeof(float) * (size_t)width * height);
        po_copy_make_border_32f(wm, w, h, top, bottom, left, right, wp[0]);
        free(wm);
        for (int i = 0; i < nb; i++) {
            wp[i + 1] = (float*)malloc(sizeof(float) * (size_t)pw[i + 1] * ph[i + 1]);
            po_pyr_down_32f(wp[i], pw[i], ph[i], wp[i + 1]);
        }
    }
    int y_tl = tl_new_y - R[1], y_br = br_new_y - R[1], x_tl = tl_new_x - R[0], x_br = br_new_x - R[0];
    for (int i = 0; i <= nb; i++) {
        int rw = x_br - x_tl, rh = y_br - y_tl;
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (int y = 0; y < rh; y++) {
            const int16_t* srow = pyr[i] + (size_t)y * pw[i] * 3;
            const float* wrow = wp[i] + (size_t)y * pw[i];
            int16_t* drow = b->lap[i] + ((size_t)(y_tl + y) * b->lw[i] + x_tl) * 3;
            float* dwrow = b->wgt[i] + (size_t)(y_tl + y) * b->lw[i] + x_tl;
            for (int x = 0; x < rw; x++) {
                float wv = wrow[x];
                drow[x * 3 + 0] = (int16_t)(drow[x * 3 + 0] + (int16_t)(srow[x * 3 + 0] * wv));
                drow[x * 3 + 1] = (int16_t)(drow[x * 3 + 1] + (int16_t)(srow[x * 3 + 1] * wv));
                drow[x * 3 + 2] = (int16_t)(drow[x * 3 + 2] + (int16_t)(srow[x * 3 + 2] * wv));
                dwrow[x] += wv;
            }
        }
        x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
    }
    for (int i = 0; i <= nb; i++) { free(pyr[i]); free(wp[i]); }
}

/* MultiBandBlender::blend + normalizeUsingWeightMap + restoreImageFromLaplacePyr + Blender::blend
 * (blenders.cpp); reference ocvstitcher.hpp:1207 */
void po_blender_blend(po_blender* b, int16_t* dst, uint8_t* dst_mask) {
    const float WEIGHT_EPS = 1e-5f;
    int fw = b->dst_roi_final[2], fh = b->dst_roi_final[3];
    if (b->requested_bands < 0) {
        for (int y = 0; y < fh; y++)
            for (int x = 0; x < fw; x++) {
                size_t k = (size_t)y * fw + x;
                uint8_t m = b->dst_mask[k];
                for (int c = 0; c < 3; c++) dst[k * 3 + c] = m ? b->lap[0][k * 3 + c] : 0;
                dst_mask[k] = m;
            }
        return;
    }
    int nb = b->num_bands;
    for (int i = 0; i <= nb; i++) {
        long cnt = (long)b->lw[i] * b->lh[i];
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long k = 0; k < cnt; k++) {
            float wv = b->wgt[i][k] + WEIGHT_EPS;
            b->lap[i][k * 3 + 0] = (int16_t)(b->lap[i][k * 3 + 0] / wv);
            b->lap[i][k * 3 + 1] = (int16_t)(b->lap[i][k * 3 + 1] / wv);
            b->lap[i][k * 3 + 2] = (int16_t)(b->lap[i][k * 3 + 2] / wv);
        }
    }
    for (int i = nb; i > 0; --i) {
        long cnt = (long)b->lw[i - 1] * b->lh[i - 1] * 3;
        int16_t* tmp = (int16_t*)malloc(sizeof(int16_t) * cnt);
        po_pyr_up_16s(b->lap[i], b->lw[i], b->lh[i], 3, tmp);
        int16_t* lo = b->lap[i - 1];
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long k = 0; k < cnt; k++) lo[k] = sat16((int)tmp[k] + (int)lo[k]);
        free(tmp);
    }
    int W = b->lw[0];
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < fh; y++)
        for (int x = 0; x < fw; x++) {
            size_t s = (size_t)y * W + x, d = (size_t)y * fw + x;
            uint8_t m = b->wgt[0][s] > WEIGHT_EPS ? 255 : 0;
            dst_mask[d] = m;
            for (int c = 0; c < 3; c++) dst[d * 3 + c] = m ? b->lap[0][s * 3 + c] : 0;
        }
}

int po_blender_num_bands(const po_blender* b) { return b->num_bands; }
void po_blender_dst_roi(const po_blender* b, int r[4]) { memcpy(r, b->dst_roi, 4 * sizeof(int)); }
void po_blender_dst_roi_final(const po_blender* b, int r[4]) { memcpy(r, b->dst_roi_final, 4 * sizeof(int)); }
void po_blender_level_size(const po_blender* b, int level, int wh[2]) { wh[0] = b->lw[level]; wh[1] = b->lh[level]; }
const int16_t* po_blender_level_laplace(const po_blender* b, int level) { return b->lap[level]; }
const float* po_blender_level_weights(const po_blender* b, int level) { return b->wgt[level]; }
void po_blender_last_tile(const po_blender* b, int rect[4], int tblr[4]) {
    memcpy(rect, b->last_tile, 4 * sizeof(int));
    memcpy(tblr, b->last_tblr, 4 * sizeof(int));
}

/* ======================================================================== A7 */

/* overlapRoi (stitching/src/util.cpp) */
static int overlap_roi(const int* tl1, const int* tl2, const int* sz1, const int* sz2, int roi[4]) {
    int x_tl = tl1[0] > tl2[0] ? tl1[0] : tl2[0];
    int y_tl = tl1[1] > tl2[1] ? tl1[1] : tl2[1];
    int x_br = (tl1[0] + sz1[0]) < (tl2[0] + sz2[0]) ? (tl1[0] + sz1[0]) : (tl2[0] + sz2[0]);
    int y_br = (tl1[1] + sz1[1]) < (tl2[1] + sz2[1]) ? (tl1[1] + sz1[1]) : (tl2[1] + sz2[1]);
    if (x_tl < x_br && y_tl < y_br) {
        roi[0] = x_tl; roi[1] = y_tl; roi[2] = x_br - x_tl; roi[3] = y_br - y_tl;
        return 1;
    }
    return 0;
}

/* VoronoiSeamFinder::findInPair (stitching/src/seam_finders.cpp) */
static void voronoi_in_pair(const int* tl1, const int* tl2, const int* sz1, const int* sz2, uint8_t* mask1,
                            uint8_t* mask2, const int roi[4]) {
    const int gap = 10;
    int W = roi[2] + 2 * gap, H = roi[3] + 2 * gap;
    uint8_t* sub1 = (uint8_t*)malloc((size_t)W * H);
    uint8_t* sub2 = (uint8_t*)malloc((size_t)W * H);
    for (int y = -gap; y < roi[3] + gap; ++y)
        for (int x = -gap; x < roi[2] + gap; ++x) {
            int y1 = roi[1] - tl1[1] + y, x1 = roi[0] - tl1[0] + x;
            sub1[(size_t)(y + gap) * W + x + gap] =
                (y1 >= 0 && x1 >= 0 && y1 < sz1[1] && x1 < sz1[0]) ? mask1[(size_t)y1 * sz1[0] + x1] : 0;
            int y2 = roi[1] - tl2[1] + y, x2 = roi[0] - tl2[0] + x;
            sub2[(size_t)(y + gap) * W + x + gap] =
                (y2 >= 0 && x2 >= 0 && y2 < sz2[1] && x2 < sz2[0]) ? mask2[(size_t)y2 * sz2[0] + x2] : 0;
        }
    /* unique = submask with collisions zeroed; distanceTransform(unique == 0) */
    uint8_t* in1 = (uint8_t*)malloc((size_t)W * H);
    uint8_t* in2 = (uint8_t*)malloc((size_t)W * H);
    for (size_t k = 0; k < (size_t)W * H; k++) {
        int coll = sub1[k] != 0 && sub2[k] != 0;
        uint8_t u1 = coll ? 0 : sub1[k], u2 = coll ? 0 : sub2[k];
        in1[k] = u1 == 0 ? 255 : 0;
        in2[k] = u2 == 0 ? 255 : 0;
    }
    float* d1 = (float*)malloc(sizeof(float) * (size_t)W * H);
    float* d2 = (float*)malloc(sizeof(float) * (size_t)W * H);
    po_distance_l1(in1, W, H, d1);
    po_distance_l1(in2, W, H, d2);
    for (int y = 0; y < roi[3]; ++y)
        for (int x = 0; x < roi[2]; ++x) {
            size_t k = (size_t)(y + gap) * W + x + gap;
            if (d1[k] < d2[k])
                mask2[(size_t)(roi[1] - tl2[1] + y) * sz2[0] + (roi[0] - tl2[0] + x)] = 0;
            else
                mask1[(size_t)(roi[1] - tl1[1] + y) * sz1[0] + (roi[0] - tl1[0] + x)] = 0;
        }
    free(sub1); free(sub2); free(in1); free(in2); free(d1); free(d2);
}

/* PairwiseSeamFinder::run + VoronoiSeamFinder::find; reference stitching_detailed.cpp:728-729,758 */
void po_voronoi_find(int n, const int* corners, const int* sizes, uint8_t** masks) {
    for (int i = 0; i < n - 1; ++i)
        for (int j = i + 1; j < n; ++j) {
            int roi[4];
            if (overlap_roi(corners + 2 * i, corners + 2 * j, sizes + 2 * i, sizes + 2 * j, roi))
                voronoi_in_pair(corners + 2 * i, corners + 2 * j, sizes + 2 * i, sizes + 2 * j, masks[i], masks[j], roi);
        }
}

/* ---- GraphCutSeamFinder(COST_COLOR), the reference's seam finder (ocvstitcher.hpp:1033-1035, :1244) */

/* GCGraph<float> (imgproc/src/gcgraph.hpp): Boykov-Kolmogorov max-flow with OpenCV's bookkeeping (active list threaded
 * through `next`, time stamps + distances for the orphan adoption).  RECALLED from the OpenCV sources; the labelling of
 * vertices that end in neither tree depends on these details */
typedef struct { int next; int parent; int first; int ts; int dist; float weight; uint8_t t; } gc_vtx;   /* next: index + 1, 0 = none, -1 = nil */
typedef struct { int dst; int next; float weight; } gc_edge;
typedef struct { gc_vtx* v; int nv; gc_edge* e; int ne; float flow; } gc_graph;

static void gc_init(gc_graph* g, int nv, int ne) {
    g->v = (gc_vtx*)calloc((size_t)nv, sizeof(gc_vtx));
    g->nv = nv;
    g->e = (gc_edge*)calloc((size_t)2 * ne + 2, sizeof(gc_edge));
    g->ne = 2;   /* edges 0 and 1 are unused so that "first == 0" means no edge and e ^ 1 is the reverse edge */
    g->flow = 0;
}
static void gc_free(gc_graph* g) { free(g->v); free(g->e); }
static void gc_add_edges(gc_graph* g, int i, int j, float w, float revw) {
    gc_edge* a = &g->e[g->ne];
    a->dst = j; a->next = g->v[i].first; a->weight = w; g->v[i].first = g->ne++;
    gc_edge* b = &g->e[g->ne];
    b->dst = i; b->next = g->v[j].first; b->weight = revw; g->v[j].first = g->ne++;
}
static void gc_add_term_weights(gc_graph* g, int i, float source_w, float sink_w) {
    float dw = g->v[i].weight;
    if (dw > 0) source_w += dw; else sink_w -= dw;
    g->flow += source_w < sink_w ? source_w : sink_w;
    g->v[i].weight = source_w - sink_w;
}
#define GC_NIL (-1)
static void gc_max_flow(gc_graph* g) {
    const int TERMINAL = -1, ORPHAN = -2;
    gc_vtx* V = g->v;
    gc_edge* E = g->e;
    /* the active list: first/last are vertex indices, GC_NIL terminates it; V[i].next: 0 = not in the list,
     * otherwise index + 1 of the successor or GC_NIL for the tail */
    int first = GC_NIL, last = GC_NIL;
    int curr_ts = 0;
    int* orphans = (int*)malloc(sizeof(int) * (size_t)(g->nv > 16 ? g->nv : 16));
    size_t n_orph = 0, cap_orph = (size_t)(g->nv > 16 ? g->nv : 16);
#define PUSH_ORPHAN(ix) do { if (n_orph == cap_orph) { cap_orph *= 2; orphans = (int*)realloc(orphans, sizeof(int) * cap_orph); } orphans[n_orph++] = (ix); } while (0)
#define APPEND_ACTIVE(ix) do { V[ix].next = GC_NIL; if (last == GC_NIL) first = (ix); else V[last].next = (ix) + 1; last = (ix); } while (0)
    for (int i = 0; i < g->nv; i++) {
        gc_vtx* v = &V[i];
        v->ts = 0;
        if (v->weight != 0) {
            APPEND_ACTIVE(i);
            v->dist = 1;
            v->parent = TERMINAL;
            v->t = v->weight < 0;
        } else
            v->parent = 0;
    }
    for (;;) {
        int e0 = -1, ei = 0, ej = 0;
        float min_weight, weight;
        uint8_t vt;
        /* grow the S and T trees, find an edge that connects them */
        while (first != GC_NIL) {
            const int vi = first;
            gc_vtx* v = &V[vi];
            if (v->parent) {
                vt = v->t;
                for (ei = v->first; ei != 0; ei = E[ei].next) {
                    if (E[ei ^ vt].weight == 0) continue;
                    const int ui = E[ei].dst;
                    gc_vtx* u = &V[ui];
                    if (!u->parent) {
                        u->t = vt;
                        u->parent = ei ^ 1;
                        u->ts = v->ts;
                        u->dist = v->dist + 1;
                        if (!u->next) APPEND_ACTIVE(ui);
                        continue;
                    }
                    if (u->t != vt) { e0 = ei ^ vt; break; }
                    if (u->dist > v->dist + 1 && u->ts <= v->ts) {
                        u->parent = ei ^ 1;
                        u->ts = v->ts;
                        u->dist = v->dist + 1;
                    }
                }
                if (e0 > 0) break;
            }
            /* exclude the vertex from the active list */
            const int nx = v->next;
            first = nx == GC_NIL ? GC_NIL : nx - 1;
            if (first == GC_NIL) last = GC_NIL;
            v->next = 0;
        }
        if (e0 <= 0) break;
        /* bottleneck of the path: k = 1 source tree, k = 0 sink tree */
        min_weight = E[e0].weight;
        for (int k = 1; k >= 0; k--) {
            int vi;
            for (vi = E[e0 ^ k].dst;; vi = E[ei].dst) {
                if ((ei = V[vi].parent) < 0) break;
                weight = E[ei ^ k].weight;
                min_weight = min_weight < weight ? min_weight : weight;
            }
            weight = fabsf(V[vi].weight);
            min_weight = min_weight < weight ? min_weight : weight;
        }
        /* push it, collect orphans */
        E[e0].weight -= min_weight;
        E[e0 ^ 1].weight += min_weight;
        g->flow += min_weight;
        for (int k = 1; k >= 0; k--) {
            int vi;
            for (vi = E[e0 ^ k].dst;; vi = E[ei].dst) {
                if ((ei = V[vi].parent) < 0) break;
                E[ei ^ (k ^ 1)].weight += min_weight;
                if ((E[ei ^ k].weight -= min_weight) == 0) {
                    PUSH_ORPHAN(vi);
                    V[vi].parent = ORPHAN;
                }
            }
            V[vi].weight = V[vi].weight + min_weight * (float)(1 - k * 2);
            if (V[vi].weight == 0) {
                PUSH_ORPHAN(vi);
                V[vi].parent = ORPHAN;
            }
        }
        /* adopt the orphans */
        curr_ts++;
        while (n_orph) {
            const int v2i = orphans[--n_orph];
            gc_vtx* v2 = &V[v2i];
            int d, min_dist = INT_MAX;
            e0 = 0;
            vt = v2->t;
            for (ei = v2->first; ei != 0; ei = E[ei].next) {
                if (E[ei ^ (vt ^ 1)].weight == 0) continue;
                gc_vtx* u = &V[E[ei].dst];
                if (u->t != vt || u->parent == 0) continue;
                for (d = 0;;) {   /* distance to the tree root */
                    if (u->ts == curr_ts) { d += u->dist; break; }
                    ej = u->parent;
                    d++;
                    if (ej < 0) {
                        if (ej == ORPHAN) d = INT_MAX - 1;
                        else { u->ts = curr_ts; u->dist = 1; }
                        break;
                    }
                    u = &V[E[ej].dst];
                }
                if (++d < INT_MAX) {
                    if (d < min_dist) { min_dist = d; e0 = ei; }
                    for (u = &V[E[ei].dst]; u->ts != curr_ts; u = &V[E[u->parent].dst]) {
                        u->ts = curr_ts;
                        u->dist = --d;
                    }
                }
            }
            if ((v2->parent = e0) > 0) {
                v2->ts = curr_ts;
                v2->dist = min_dist;
                continue;
            }
            /* no parent: the vertex becomes free, its children orphans, its neighbours active */
            v2->ts = 0;
            for (ei = v2->first; ei != 0; ei = E[ei].next) {
                const int ui = E[ei].dst;
                gc_vtx* u = &V[ui];
                ej = u->parent;
                if (u->t != vt || !ej) continue;
                if (E[ei ^ (vt ^ 1)].weight != 0 && !u->next) APPEND_ACTIVE(ui);
                if (ej > 0 && E[ej].dst == v2i) {
                    PUSH_ORPHAN(ui);
                    u->parent = ORPHAN;
                }
            }
        }
    }
    free(orphans);
#undef PUSH_ORPHAN
#undef APPEND_ACTIVE
}

/* GraphCutSeamFinder::Impl::findInPair + setGraphWeightsColor (stitching/src/seam_finders.cpp): images CV_32FC3,
 * terminal_cost 10000, bad_region_penalty 1000, gap 10; normL2 there is the SQUARED distance */
static float gc_norm2(const float* a, const float* b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
}
static void graphcut_in_pair(const float* img1, const float* img2, const int* tl1, const int* tl2, const int* sz1,
                             const int* sz2, uint8_t* mask1, uint8_t* mask2, const int roi[4]) {
    const int gap = 10;
    const float terminal_cost = 10000.f, bad_region_penalty = 1000.f;
    const int W = roi[2] + 2 * gap, H = roi[3] + 2 * gap;
    float* s1 = (float*)calloc((size_t)W * H * 3, sizeof(float));
    float* s2 = (float*)calloc((size_t)W * H * 3, sizeof(float));
    uint8_t* m1 = (uint8_t*)calloc((size_t)W * H, 1);
    uint8_t* m2 = (uint8_t*)calloc((size_t)W * H, 1);
    for (int y = -gap; y < roi[3] + gap; ++y)
        for (int x = -gap; x < roi[2] + gap; ++x) {
            const size_t k = (size_t)(y + gap) * W + x + gap;
            int y1 = roi[1] - tl1[1] + y, x1 = roi[0] - tl1[0] + x;
            if (y1 >= 0 && x1 >= 0 && y1 < sz1[1] && x1 < sz1[0]) {
                memcpy(s1 + 3 * k, img1 + ((size_t)y1 * sz1[0] + x1) * 3, 3 * sizeof(float));
                m1[k] = mask1[(size_t)y1 * sz1[0] + x1];
            }
            int y2 = roi[1] - tl2[1] + y, x2 = roi[0] - tl2[0] + x;
            if (y2 >= 0 && x2 >= 0 && y2 < sz2[1] && x2 < sz2[0]) {
                memcpy(s2 + 3 * k, img2 + ((size_t)y2 * sz2[0] + x2) * 3, 3 * sizeof(float));
                m2[k] = mask2[(size_t)y2 * sz2[0] + x2];
            }
        }
    gc_graph g;
    gc_init(&g, W * H, (H - 1) * W + (W - 1) * H);
    for (int k = 0; k < W * H; k++) gc_add_term_weights(&g, k, m1[k] ? terminal_cost : 0.f, m2[k] ? terminal_cost : 0.f);
    const float weight_eps = 1.f;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int v = y * W + x;
            if (x < W - 1) {
                float weight = gc_norm2(s1 + 3 * v, s2 + 3 * v) + gc_norm2(s1 + 3 * (v + 1), s2 + 3 * (v + 1)) + weight_eps;
                if (!m1[v] || !m1[v + 1] || !m2[v] || !m2[v + 1]) weight += bad_region_penalty;
                gc_add_edges(&g, v, v + 1, weight, weight);
            }
            if (y < H - 1) {
                float weight = gc_norm2(s1 + 3 * v, s2 + 3 * v) + gc_norm2(s1 + 3 * (v + W), s2 + 3 * (v + W)) + weight_eps;
                if (!m1[v] || !m1[v + W] || !m2[v] || !m2[v + W]) weight += bad_region_penalty;
                gc_add_edges(&g, v, v + W, weight, weight);
            }
        }
    gc_max_flow(&g);
    if (getenv("PO_GC_DUMP")) {
        /* test hook: the pair's graph and labels in the product's pano_debug_graphcut_dump record format (i, j unknown here: -1) */
        FILE* df = fopen(getenv("PO_GC_DUMP"), "ab");
        if (df) {
            const int hdr[4] = {-1, -1, W, H};
            fwrite(hdr, sizeof(int), 4, df);
            for (int pass = 0; pass < 3; pass++)
                for (int k = 0; k < W * H; k++) {
                    float v = 0.f;
                    const int x = k % W, y = k / W;
                    if (pass == 0) v = (m1[k] ? terminal_cost : 0.f) - (m2[k] ? terminal_cost : 0.f);
                    else {
                        const int u = pass == 1 ? k + 1 : k + W;
                        if ((pass == 1 && x < W - 1) || (pass == 2 && y < H - 1)) {
                            v = gc_norm2(s1 + 3 * k, s2 + 3 * k) + gc_norm2(s1 + 3 * u, s2 + 3 * u) + weight_eps;
                            if (!m1[k] || !m1[u] || !m2[k] || !m2[u]) v += bad_region_penalty;
                        }
                    }
                    fwrite(&v, sizeof(float), 1, df);
                }
            for (int k = 0; k < W * H; k++) { const uint8_t l = g.v[k].t == 0; fwrite(&l, 1, 1, df); }
            fclose(df);
        }
    }
    for (int y = 0; y < roi[3]; ++y)
        for (int x = 0; x < roi[2]; ++x) {
            const size_t k1 = (size_t)(roi[1] - tl1[1] + y) * sz1[0] + (roi[0] - tl1[0] + x);
            const size_t k2 = (size_t)(roi[1] - tl2[1] + y) * sz2[0] + (roi[0] - tl2[0] + x);
            if (g.v[(y + gap) * W + x + gap].t == 0) {   /* inSourceSegment */
                if (mask1[k1]) mask2[k2] = 0;
            } else {
                if (mask2[k2]) mask1[k1] = 0;
            }
        }
    gc_free(&g);
    free(s1); free(s2); free(m1); free(m2);
}

/* test hook: the max-flow on a W x H 4-connected grid given as arrays: term[k] = source - sink weight of vertex k
 * (added as addTermWeights(k, max(t,0), max(-t,0))), wh[k] = weight of edge (k, k+1), wv[k] = weight of (k, k+W), both
 * directions equal; labels[k] = inSourceSegment.  Returns the flow */
float po_gc_grid_max_flow(int W, int H, const float* term, const float* wh, const float* wv, uint8_t* labels) {
    gc_graph g;
    gc_init(&g, W * H, (H - 1) * W + (W - 1) * H);
    for (int k = 0; k < W * H; k++) gc_add_term_weights(&g, k, term[k] > 0 ? term[k] : 0.f, term[k] < 0 ? -term[k] : 0.f);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int v = y * W + x;
            if (x < W - 1) gc_add_edges(&g, v, v + 1, wh[v], wh[v]);
            if (y < H - 1) gc_add_edges(&g, v, v + W, wv[v], wv[v]);
        }
    gc_max_flow(&g);
    for (int k = 0; k < W * H; k++) labels[k] = g.v[k].t == 0;
    float f = g.flow;
    gc_free(&g);
    return f;
}

/* PairwiseSeamFinder::run + GraphCutSeamFinder(COST_COLOR)::find: images are the 8UC3 warps (converted to f32 here,
 * images_warped[i].convertTo(images_warped_f[i], CV_32F), ocvstitcher.hpp:1028-1029) */
void po_graphcut_find(int n, const int* corners, const int* sizes, const uint8_t* const* images, uint8_t** masks) {
    float** f = (float**)calloc((size_t)n + 1, sizeof(float*));
    for (int i = 0; i < n; i++) {
        const size_t px = (size_t)sizes[2 * i] * sizes[2 * i + 1] * 3;
        f[i] = (float*)malloc(sizeof(float) * px);
        for (size_t k = 0; k < px; k++) f[i][k] = (float)images[i][k];
    }
    for (int i = 0; i < n - 1; ++i)
        for (int j = i + 1; j < n; ++j) {
            int roi[4];
            if (overlap_roi(corners + 2 * i, corners + 2 * j, sizes + 2 * i, sizes + 2 * j, roi))
                graphcut_in_pair(f[i], f[j], corners + 2 * i, corners + 2 * j, sizes + 2 * i, sizes + 2 * j, masks[i], masks[j], roi);
        }
    for (int i = 0; i < n; i++) free(f[i]);
    free(f);
}

/* ======================================================================== A8 */

/* cv::resize CV_32FC1 INTER_LINEAR (resize.cpp resizeGeneric_ with HResizeLinear<float,float,float,1> and
 * VResizeLinear<float,float,float,Cast>) */
void po_resize_linear_32f(const float* src, int sw, int sh, float* dst, int dw, int dh) {
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int* xo = (int*)malloc(sizeof(int) * dw);
    float* xa = (float*)malloc(sizeof(float) * 2 * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xo[dx] = sx;
        xa[2 * dx] = 1.f - fx;
        xa[2 * dx + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        float b0 = 1.f - fy, b1 = fy;
        const float* S0 = src + (size_t)sy0 * sw;
        const float* S1 = src + (size_t)sy1 * sw;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xo[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx;
            float h0 = S0[sx] * xa[2 * dx] + S0[sx1] * xa[2 * dx + 1];
            float h1 = S1[sx] * xa[2 * dx] + S1[sx1] * xa[2 * dx + 1];
            dst[(size_t)dy * dw + dx] = h0 * b0 + h1 * b1;
        }
    }
    free(xo);
    free(xa);
}

/* BlocksGainCompensator::apply inner loop (exposure_compensate.cpp): saturate_cast<uchar>(px * gain);
 * reference stitching_detailed.cpp:841 */
void po_gain_apply_8uc3(uint8_t* img, int w, int h, const float* g) {
    for (size_t k = 0; k < (size_t)w * h; k++)
        for (int c = 0; c < 3; c++) img[k * 3 + c] = sat8(cv_round_f(img[k * 3 + c] * g[k]));
}

/* ======================================================================== caller-side assembly */

/* cv::resize INTER_LINEAR CV_8U (resize.cpp): fx = (float)((dx+0.5)*scale - 0.5), coefficients
 * saturate_cast<short>(c * 2048); HResizeLinear int accumulation; VResizeLinear 8u special:
 * ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2 >> 2.  Equal sizes are a plain copy. */
void po_resize_linear_8u(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh) {
    if (sw == dw && sh == dh) {
        memcpy(dst, src, (size_t)sw * sh * cn);
        return;
    }
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat16(cv_round_f((1.f - fx) * 2048));
        ialpha[2 * dx + 1] = sat16(cv_round_f(fx * 2048));
    }
    int* r0 = (int*)malloc(sizeof(int) * (size_t)dw * cn);
    int* r1 = (int*)malloc(sizeof(int) * (size_t)dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        short b0 = sat16(cv_round_f((1.f - fy) * 2048)), b1 = sat16(cv_round_f(fy * 2048));
        int y0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy), y1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        const uint8_t* S0 = src + (size_t)y0 * sw * cn;
        const uint8_t* S1 = src + (size_t)y1 * sw * cn;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            for (int c = 0; c < cn; c++) {
                r0[dx * cn + c] = S0[sx * cn + c] * ialpha[2 * dx] + S0[sx1 * cn + c] * ialpha[2 * dx + 1];
                r1[dx * cn + c] = S1[sx * cn + c] * ialpha[2 * dx] + S1[sx1 * cn + c] * ialpha[2 * dx + 1];
            }
        }
        uint8_t* D = dst + (size_t)dy * dw * cn;
        for (int x = 0; x < dw * cn; x++) D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(r0); free(r1);
}

/* src/master.cpp:321-326 */
void po_stack_master(const uint8_t* up, int uw, int uh, const uint8_t* down, int dw, int dh, uint8_t* out) {
    po_resize_linear_8u(up, uw, uh, 3, out, dw, dh);
    memcpy(out + (size_t)dw * dh * 3, down, (size_t)dw * dh * 3);
    int rows = 2 * dh;
    for (int y = rows / 2 - 5; y < rows / 2 + 5; y++)  /* cv::rectangle(Rect(0, rows/2-5, cols, 10), 0, filled) */
        if (y >= 0 && y < rows) memset(out + (size_t)y * dw * 3, 0, (size_t)dw * 3);
}

/* src/panocamimpl.cpp:354-360 */
void po_stack_finalcut(const uint8_t* up, int uw, int uh, const uint8_t* down, int dw, int dh, int finalcut, uint8_t* out) {
    int width = uw < dw ? uw : dw, height = (uh < dh ? uh : dh) - finalcut * 2;
    for (int y = 0; y < height; y++) {
        memcpy(out + (size_t)y * width * 3, up + ((size_t)(y + finalcut) * uw) * 3, (size_t)width * 3);
        memcpy(out + (size_t)(y + height) * width * 3, down + ((size_t)(y + finalcut) * dw) * 3, (size_t)width * 3);
    }
    for (int y = height - 2; y < height + 2; y++)  /* cv::rectangle(Rect(0, height-2, width, 4), 0, filled) */
        if (y >= 0 && y < 2 * height) memset(out + (size_t)y * width * 3, 0, (size_t)width * 3);
}

/* ======================================================================== mask preparation */

/* ocvStitcher::initSeam (ocvstitcher.hpp:975-1101) with VoronoiSeamFinder in place of GraphCut
 * (stitching_detailed.cpp:728-729 is the reference's own Voronoi option) */
void po_prepare_masks_voronoi(int n, int kind, int sw, int sh, const float* Ks, const float* Rs, float wscale,
                              uint8_t** masks_out) {
    double swa = sqrt(1e5 / ((double)sh * sw));
    if (swa > 1.0) swa = 1.0;
    int ssw = cv_round_d(sw * swa), ssh = cv_round_d(sh * swa);
    float seam_scale = (float)(wscale * swa);
    float swa_f = (float)swa;
    int* corners = (int*)calloc((size_t)2 * n, sizeof(int));
    int* sizes = (int*)calloc((size_t)2 * n, sizeof(int));
    uint8_t** mw = (uint8_t**)malloc(sizeof(uint8_t*) * n);
    uint8_t* ones = (uint8_t*)malloc((size_t)(ssw > sw ? ssw : sw) * (ssh > sh ? ssh : sh));
    memset(ones, 255, (size_t)(ssw > sw ? ssw : sw) * (ssh > sh ? ssh : sh));
    for (int i = 0; i < n; i++) {
        float K[9];
        memcpy(K, Ks + 9 * i, sizeof(K));
        K[0] *= swa_f; K[2] *= swa_f; K[4] *= swa_f; K[5] *= swa_f;
        po_projector p;
        po_projector_set(&p, kind, seam_scale, K, Rs + 9 * i);
        int r[4];
        po_warp_roi(&p, ssw, ssh, r);
        corners[2 * i] = r[0]; corners[2 * i + 1] = r[1];
        sizes[2 * i] = r[2]; sizes[2 * i + 1] = r[3];
        mw[i] = (uint8_t*)malloc((size_t)r[2] * r[3]);
        int c[2];
        po_warp_8u(&p, ones, ssw, ssh, (size_t)ssw, 1, PO_INTER_NEAREST, PO_BORDER_CONSTANT, mw[i], c);
    }
    po_voronoi_find(n, corners, sizes, mw);
    for (int i = 0; i < n; i++) {
        po_projector p;
        po_projector_set(&p, kind, wscale, Ks + 9 * i, Rs + 9 * i);
        int r[4], c[2];
        po_warp_roi(&p, sw, sh, r);
        uint8_t* full = (uint8_t*)malloc((size_t)r[2] * r[3]);
        po_warp_8u(&p, ones, sw, sh, (size_t)sw, 1, PO_INTER_NEAREST, PO_BORDER_CONSTANT, full, c);
        uint8_t* dil = (uint8_t*)malloc((size_t)sizes[2 * i] * sizes[2 * i + 1]);
        po_dilate3x3_8u(mw[i], sizes[2 * i], sizes[2 * i + 1], dil);
        uint8_t* seam = (uint8_t*)malloc((size_t)r[2] * r[3]);
        po_resize_linear_exact_8u(dil, sizes[2 * i], sizes[2 * i + 1], 1, seam, r[2], r[3]);
        for (size_t k = 0; k < (size_t)r[2] * r[3]; k++) masks_out[i][k] = seam[k] & full[k];
        free(full); free(dil); free(seam); free(mw[i]);
    }
    free(ones); free(mw); free(corners); free(sizes);
}

/* ======================================================================== exposure: gain estimation */

/* cv::solve(A, b, x, DECOMP_LU) on CV_64F without LAPACK: hal::LU64f -> LUImpl<double> (core/src/matrix_decomp.cpp),
 * eps = DBL_EPSILON * 100.  Partial pivoting, row elimination with d = -1/pivot, back substitution.  A (m x m, row
 * major) and b (m) are overwritten; b holds the solution.  Returns 0 when a pivot is below eps (cv::solve -> false) */
static int lu_solve_64f(double* A, int m, double* b) {
    const double eps = 2.220446049250313e-16 * 100;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++)
            if (fabs(A[(size_t)j * m + i]) > fabs(A[(size_t)k * m + i])) k = j;
        if (fabs(A[(size_t)k * m + i]) < eps) return 0;
        if (k != i) {
            for (int j = i; j < m; j++) { double t = A[(size_t)i * m + j]; A[(size_t)i * m + j] = A[(size_t)k * m + j]; A[(size_t)k * m + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / A[(size_t)i * m + i];
        for (int j = i + 1; j < m; j++) {
            double alpha = A[(size_t)j * m + i] * d;
            for (k = i + 1; k < m; k++) A[(size_t)j * m + k] += alpha * A[(size_t)i * m + k];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < m; k++) s -= A[(size_t)i * m + k] * b[k];
        b[i] = s / A[(size_t)i * m + i];
    }
    return 1;
}

/* detail::GainCompensator::feed (stitching/src/exposure_compensate.cpp, 3.4.0): for every overlapping pair i <= j
 * (a sub-image overlaps itself) N = max(1, #pixels both masks mark), I = mean of sqrt(b^2+g^2+r^2) over them,
 * accumulated in double in row-major order; then the normal equations with alpha = 0.01, beta = 100 and cv::solve.
 * Sub-images are given as pointer + stride (BlocksGainCompensator feeds views into the warped images) */
int po_gain_feed(int n, const int* corners, const int* sizes, const uint8_t* const* imgs, const size_t* istride,
                 const uint8_t* const* masks, const size_t* mstride, double* gains) {
    int* N = (int*)calloc((size_t)n * n, sizeof(int));
    double* I = (double*)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            int roi[4];
            if (!overlap_roi(corners + 2 * i, corners + 2 * j, sizes + 2 * i, sizes + 2 * j, roi)) continue;
            const int x1 = roi[0] - corners[2 * i], y1 = roi[1] - corners[2 * i + 1];
            const int x2 = roi[0] - corners[2 * j], y2 = roi[1] - corners[2 * j + 1];
            int cnt = 0;
            double Isum1 = 0, Isum2 = 0;
            for (int y = 0; y < roi[3]; ++y) {
                const uint8_t* r1 = imgs[i] + (size_t)(y1 + y) * istride[i] + 3 * (size_t)x1;
                const uint8_t* r2 = imgs[j] + (size_t)(y2 + y) * istride[j] + 3 * (size_t)x2;
                const uint8_t* m1 = masks[i] + (size_t)(y1 + y) * mstride[i] + x1;
                const uint8_t* m2 = masks[j] + (size_t)(y2 + y) * mstride[j] + x2;
                for (int x = 0; x < roi[2]; ++x)
                    if (m1[x] == 255 && m2[x] == 255) {
                        cnt++;
                        Isum1 += sqrt((double)(r1[3 * x] * r1[3 * x] + r1[3 * x + 1] * r1[3 * x + 1] + r1[3 * x + 2] * r1[3 * x + 2]));
                        Isum2 += sqrt((double)(r2[3 * x] * r2[3 * x] + r2[3 * x + 1] * r2[3 * x + 1] + r2[3 * x + 2] * r2[3 * x + 2]));
                    }
            }
            N[(size_t)i * n + j] = N[(size_t)j * n + i] = cnt > 1 ? cnt : 1;
            I[(size_t)i * n + j] = Isum1 / N[(size_t)i * n + j];
            I[(size_t)j * n + i] = Isum2 / N[(size_t)i * n + j];
        }
    const double alpha = 0.01, beta = 100;
    double* A = (double*)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n; ++i) gains[i] = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            gains[i] += beta * N[(size_t)i * n + j];
            A[(size_t)i * n + i] += beta * N[(size_t)i * n + j];
            if (j == i) continue;
            A[(size_t)i * n + i] += 2 * alpha * I[(size_t)i * n + j] * I[(size_t)i * n + j] * N[(size_t)i * n + j];
            A[(size_t)i * n + j] -= 2 * alpha * I[(size_t)i * n + j] * I[(size_t)j * n + i] * N[(size_t)i * n + j];
        }
    int ok = lu_solve_64f(A, n, gains);
    free(N); free(I); free(A);
    return ok;
}

/* cv::sepFilter2D(m, m, CV_32F, [.25 .5 .25], [.25 .5 .25]) (BORDER_REFLECT_101): SymmRowSmallFilter then
 * SymmColumnSmallFilter, both  S[0]*k0 + (S[-1] + S[1])*k1  in f32 (imgproc/src/filter.cpp) */
static void sep_filter_121(float* m, int w, int h) {
    float* t = (float*)malloc(sizeof(float) * (size_t)w * h);
    const float k0 = 0.5f, k1 = 0.25f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int xl = border_interpolate(x - 1, w, PO_BORDER_REFLECT_101), xr = border_interpolate(x + 1, w, PO_BORDER_REFLECT_101);
            t[(size_t)y * w + x] = m[(size_t)y * w + x] * k0 + (m[(size_t)y * w + xl] + m[(size_t)y * w + xr]) * k1;
        }
    for (int y = 0; y < h; y++) {
        int yu = border_interpolate(y - 1, h, PO_BORDER_REFLECT_101), yd = border_interpolate(y + 1, h, PO_BORDER_REFLECT_101);
        for (int x = 0; x < w; x++)
            m[(size_t)y * w + x] = (t[(size_t)yu * w + x] + t[(size_t)yd * w + x]) * k1 + t[(size_t)y * w + x] * k0;
    }
    free(t);
}

/* detail::BlocksGainCompensator::feed (exposure_compensate.cpp, 3.4.0; the reference's compensator,
 * ocvstitcher.hpp:1031-1032, stitching_detailed.cpp:722-723): cut every image into ceil(w/bl_w) x ceil(h/bl_h) equal
 * blocks, one gain per block from GainCompensator::feed on the blocks, then two [1 2 1]/4 smoothing passes per map.
 * maps[i] must hold po_gain_blocks_map_size(i) floats */
void po_gain_blocks_map_size(int w, int h, int bl_w, int bl_h, int wh[2]) {
    wh[0] = (w + bl_w - 1) / bl_w;
    wh[1] = (h + bl_h - 1) / bl_h;
}
int po_gain_blocks_feed(int n, const int* corners, const int* sizes, const uint8_t* const* imgs,
                        const uint8_t* const* masks, int bl_w_, int bl_h_, float** maps) {
    int nb = 0;
    for (int i = 0; i < n; i++) {
        int wh[2];
        po_gain_blocks_map_size(sizes[2 * i], sizes[2 * i + 1], bl_w_, bl_h_, wh);
        nb += wh[0] * wh[1];
    }
    int* bc = (int*)calloc((size_t)2 * nb + 1, sizeof(int));
    int* bs = (int*)calloc((size_t)2 * nb + 1, sizeof(int));
    const uint8_t** bi = (const uint8_t**)calloc((size_t)nb + 1, sizeof(void*));
    const uint8_t** bm = (const uint8_t**)calloc((size_t)nb + 1, sizeof(void*));
    size_t* bis = (size_t*)calloc((size_t)nb + 1, sizeof(size_t));
    size_t* bms = (size_t*)calloc((size_t)nb + 1, sizeof(size_t));
    double* gains = (double*)calloc((size_t)nb + 1, sizeof(double));
    int k = 0;
    for (int i = 0; i < n; i++) {
        const int cols = sizes[2 * i], rows = sizes[2 * i + 1];
        int per[2];
        po_gain_blocks_map_size(cols, rows, bl_w_, bl_h_, per);
        const int bl_width = (cols + per[0] - 1) / per[0], bl_height = (rows + per[1] - 1) / per[1];
        for (int by = 0; by < per[1]; ++by)
            for (int bx = 0; bx < per[0]; ++bx, ++k) {
                const int tlx = bx * bl_width, tly = by * bl_height;
                const int brx = tlx + bl_width < cols ? tlx + bl_width : cols, bry = tly + bl_height < rows ? tly + bl_height : rows;
                bc[2 * k] = corners[2 * i] + tlx; bc[2 * k + 1] = corners[2 * i + 1] + tly;
                bs[2 * k] = brx - tlx; bs[2 * k + 1] = bry - tly;
                bi[k] = imgs[i] + ((size_t)tly * cols + tlx) * 3; bis[k] = (size_t)cols * 3;
                bm[k] = masks[i] + (size_t)tly * cols + tlx; bms[k] = (size_t)cols;
            }
    }
    int ok = po_gain_feed(nb, bc, bs, bi, bis, bm, bms, gains);
    k = 0;
    for (int i = 0; i < n; i++) {
        int per[2];
        po_gain_blocks_map_size(sizes[2 * i], sizes[2 * i + 1], bl_w_, bl_h_, per);
        for (int b = 0; b < per[0] * per[1]; b++, k++) maps[i][b] = (float)gains[k];
        sep_filter_121(maps[i], per[0], per[1]);
        sep_filter_121(maps[i], per[0], per[1]);
    }
    free(bc); free(bs); free(bi); free(bm); free(bis); free(bms); free(gains);
    return ok;
}

/* the reference's feed of the compensator, ocvStitcher::initSeam (ocvstitcher.hpp:981-1032): frames resized by
 * seam_work_aspect (INTER_LINEAR_EXACT), warped at the seam scale (INTER_LINEAR, BORDER_REFLECT) beside the
 * INTER_NEAREST-warped all-255 masks, then BlocksGainCompensator(32, 32)::feed.  seam_sizes (2n) and maps are outputs */
int po_estimate_gains(int n, int kind, int sw, int sh, const uint8_t* const* frames, const float* Ks, const float* Rs,
                      float wscale, int bl_w, int bl_h, int* seam_sizes, float** maps) {
    double swa = sqrt(1e5 / ((double)sh * sw));
    if (swa > 1.0) swa = 1.0;
    int ssw = cv_round_d(sw * swa), ssh = cv_round_d(sh * swa);
    float seam_scale = (float)(wscale * swa);
    float swa_f = (float)swa;
    int* corners = (int*)calloc((size_t)2 * n, sizeof(int));
    uint8_t** iw = (uint8_t**)calloc((size_t)n + 1, sizeof(uint8_t*));
    uint8_t** mw = (uint8_t**)calloc((size_t)n + 1, sizeof(uint8_t*));
    uint8_t* ones = (uint8_t*)malloc((size_t)ssw * ssh);
    uint8_t* small = (uint8_t*)malloc((size_t)ssw * ssh * 3);
    memset(ones, 255, (size_t)ssw * ssh);
    for (int i = 0; i < n; i++) {
        float K[9];
        memcpy(K, Ks + 9 * i, sizeof(K));
        K[0] *= swa_f; K[2] *= swa_f; K[4] *= swa_f; K[5] *= swa_f;
        po_projector p;
        po_projector_set(&p, kind, seam_scale, K, Rs + 9 * i);
        int r[4], c[2];
        po_warp_roi(&p, ssw, ssh, r);
        corners[2 * i] = r[0]; corners[2 * i + 1] = r[1];
        seam_sizes[2 * i] = r[2]; seam_sizes[2 * i + 1] = r[3];
        if (ssw == sw && ssh == sh) memcpy(small, frames[i], (size_t)sw * sh * 3);
        else po_resize_linear_exact_8u_fxy(frames[i], sw, sh, 3, small, ssw, ssh, swa, swa);
        iw[i] = (uint8_t*)malloc((size_t)r[2] * r[3] * 3);
        mw[i] = (uint8_t*)malloc((size_t)r[2] * r[3]);
        po_warp_8u(&p, small, ssw, ssh, (size_t)ssw * 3, 3, PO_INTER_LINEAR, PO_BORDER_REFLECT, iw[i], c);
        po_warp_8u(&p, ones, ssw, ssh, (size_t)ssw, 1, PO_INTER_NEAREST, PO_BORDER_CONSTANT, mw[i], c);
    }
    int ok = po_gain_blocks_feed(n, corners, seam_sizes, (const uint8_t* const*)iw, (const uint8_t* const*)mw, bl_w, bl_h, maps);
    for (int i = 0; i < n; i++) { free(iw[i]); free(mw[i]); }
    free(iw); free(mw); free(ones); free(small); free(corners);
    return ok;
}

/* ocvStitcher::updateMask (ocvstitcher.hpp:1218-1261; the same steps inside initSeam, :981-1101) as the reference
 * runs it: frames resized by seam_work_aspect (INTER_LINEAR_EXACT), seam-scale LINEAR/REFLECT and NEAREST warps,
 * GraphCutSeamFinder(COST_COLOR) on the f32 images, then dilate 3x3, resize INTER_LINEAR_EXACT to the ROI, AND with
 * the full-scale NEAREST mask */
void po_prepare_masks_graphcut(int n, int kind, int sw, int sh, const uint8_t* const* frames, const float* Ks,
                               const float* Rs, float wscale, uint8_t** masks_out) {
    double swa = sqrt(1e5 / ((double)sh * sw));
    if (swa > 1.0) swa = 1.0;
    int ssw = cv_round_d(sw * swa), ssh = cv_round_d(sh * swa);
    float seam_scale = (float)(wscale * swa);
    float swa_f = (float)swa;
    int* corners = (int*)calloc((size_t)2 * n + 1, sizeof(int));
    int* sizes = (int*)calloc((size_t)2 * n + 1, sizeof(int));
    uint8_t** iw = (uint8_t**)calloc((size_t)n + 1, sizeof(uint8_t*));
    uint8_t** mw = (uint8_t**)calloc((size_t)n + 1, sizeof(uint8_t*));
    uint8_t* ones = (uint8_t*)malloc((size_t)sw * sh);
    uint8_t* small = (uint8_t*)malloc((size_t)ssw * ssh * 3);
    memset(ones, 255, (size_t)sw * sh);
    for (int i = 0; i < n; i++) {
        float K[9];
        memcpy(K, Ks + 9 * i, sizeof(K));
        K[0] *= swa_f; K[2] *= swa_f; K[4] *= swa_f; K[5] *= swa_f;
        po_projector p;
        po_projector_set(&p, kind, seam_scale, K, Rs + 9 * i);
        int r[4], c[2];
        po_warp_roi(&p, ssw, ssh, r);
        corners[2 * i] = r[0]; corners[2 * i + 1] = r[1];
        sizes[2 * i] = r[2]; sizes[2 * i + 1] = r[3];
        if (ssw == sw && ssh == sh) memcpy(small, frames[i], (size_t)sw * sh * 3);
        else po_resize_linear_exact_8u_fxy(frames[i], sw, sh, 3, small, ssw, ssh, swa, swa);
        iw[i] = (uint8_t*)malloc((size_t)r[2] * r[3] * 3);
        mw[i] = (uint8_t*)malloc((size_t)r[2] * r[3]);
        po_warp_8u(&p, small, ssw, ssh, (size_t)ssw * 3, 3, PO_INTER_LINEAR, PO_BORDER_REFLECT, iw[i], c);
        po_warp_8u(&p, ones, ssw, ssh, (size_t)ssw, 1, PO_INTER_NEAREST, PO_BORDER_CONSTANT, mw[i], c);
    }
    po_graphcut_find(n, corners, sizes, (const uint8_t* const*)iw, mw);
    for (int i = 0; i < n; i++) {
        po_projector p;
        po_projector_set(&p, kind, wscale, Ks + 9 * i, Rs + 9 * i);
        int r[4], c[2];
        po_warp_roi(&p, sw, sh, r);
        uint8_t* full = (uint8_t*)malloc((size_t)r[2] * r[3]);
        po_warp_8u(&p, ones, sw, sh, (size_t)sw, 1, PO_INTER_NEAREST, PO_BORDER_CONSTANT, full, c);
        uint8_t* dil = (uint8_t*)malloc((size_t)sizes[2 * i] * sizes[2 * i + 1]);
        po_dilate3x3_8u(mw[i], sizes[2 * i], sizes[2 * i + 1], dil);
        uint8_t* seam = (uint8_t*)malloc((size_t)r[2] * r[3]);
        po_resize_linear_exact_8u(dil, sizes[2 * i], sizes[2 * i + 1], 1, seam, r[2], r[3]);
        for (size_t k = 0; k < (size_t)r[2] * r[3]; k++) masks_out[i][k] = seam[k] & full[k];
        free(full); free(dil); free(seam); free(mw[i]); free(iw[i]);
    }
    free(ones); free(small); free(mw); free(iw); free(corners); free(sizes);
}

/* ======================================================================== fused undistort front end */

/* cvUndistortPoints without R / P (imgproc/src/undistort.cpp cvUndistortPointsInternal): 5 fixed-point iterations */
static void undistort_point(const double K[9], const double d[4], double u, double v, double* ox, double* oy) {
    double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    double x = (u - cx) * ifx, y = (v - cy) * ify, x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((0 * r2 + d[1]) * r2 + d[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        double deltaX = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x);
        double deltaY = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    *ox = x; *oy = y;
}

/* cvGetOptimalNewCameraMatrix, alpha = 1, newImgSize = imgSize, centerPrincipalPoint = 0: icvGetRectangles on a 9x9
 * grid of float points, then the projection that maps the OUTER rectangle to the viewport */
void po_optimal_new_camera_matrix(const double K[9], const double dist[4], int w, int h, double newK[9]) {
    const int N = 9;
    float oX0 = 3.402823466e+38F, oX1 = -3.402823466e+38F, oY0 = 3.402823466e+38F, oY1 = -3.402823466e+38F;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            float px = (float)x * w / (N - 1), py = (float)y * h / (N - 1);
            double ux, uy;
            undistort_point(K, dist, (double)px, (double)py, &ux, &uy);
            float fx_ = (float)ux, fy_ = (float)uy;
            if (fx_ < oX0) oX0 = fx_;
            if (fx_ > oX1) oX1 = fx_;
            if (fy_ < oY0) oY0 = fy_;
            if (fy_ > oY1) oY1 = fy_;
        }
    float ow = oX1 - oX0, oh = oY1 - oY0;
    double fx1 = (w - 1) / ow, fy1 = (h - 1) / oh;
    double cx1 = -fx1 * oX0, cy1 = -fy1 * oY0;
    for (int i = 0; i < 9; i++) newK[i] = 0;
    /* alpha = 1: M = fx0*(1-alpha) + fx1*alpha with (1-alpha) == 0 */
    newK[0] = fx1; newK[4] = fy1; newK[2] = cx1; newK[5] = cy1; newK[8] = 1;
}

/* inverse of: resize(undist->out), resize(crop->undist), crop, remap(undistort maps), resize(raw->undist).
 * cv::resize maps dst centre (d+0.5)*scale-0.5; the undistort map is initUndistortRectifyMap's formula with R = I
 * (imgproc/src/undistort.cpp), evaluated at the fractional position.  double throughout, one final cast. */
void po_front_end_map(const po_front_end* fe, const double newK[9], float xo, float yo, float* xr, float* yr) {
    double x = ((double)xo + 0.5) * ((double)fe->undist_w / fe->out_w) - 0.5;
    double y = ((double)yo + 0.5) * ((double)fe->undist_h / fe->out_h) - 0.5;
    x = (x + 0.5) * ((double)fe->rect[2] / fe->undist_w) - 0.5 + fe->rect[0];
    y = (y + 0.5) * ((double)fe->rect[3] / fe->undist_h) - 0.5 + fe->rect[1];
    double nx = (x - newK[2]) / newK[0], ny = (y - newK[5]) / newK[4];
    double x2 = nx * nx, y2 = ny * ny, r2 = x2 + y2, _2xy = 2 * nx * ny;
    double kr = 1 + ((0 * r2 + fe->dist[1]) * r2 + fe->dist[0]) * r2;
    double xd = nx * kr + fe->dist[2] * _2xy + fe->dist[3] * (r2 + 2 * x2);
    double yd = ny * kr + fe->dist[2] * (r2 + 2 * y2) + fe->dist[3] * _2xy;
    double u = fe->K[0] * xd + fe->K[2], v = fe->K[4] * yd + fe->K[5];
    u = (u + 0.5) * ((double)fe->raw_w / fe->undist_w) - 0.5;
    v = (v + 0.5) * ((double)fe->raw_h / fe->undist_h) - 0.5;
    *xr = (float)u;
    *yr = (float)v;
}

/* ======================================================================== whole frame */

static __thread double g_ms[3];
void po_last_timings(double ms[3]) { ms[0] = g_ms[0]; ms[1] = g_ms[1]; ms[2] = g_ms[2]; }

/* ocvStitcher::process (ocvstitcher.hpp:1141-1216); with gain_maps != NULL the exposure apply of
 * stitching_detailed.cpp:841 sits between warp and the 16S conversion */
int po_compose(const po_compose_args* a, uint8_t* out, int out_wh[2]) {
    int n = a->n;
    int* corners = (int*)malloc(sizeof(int) * 2 * n);
    int* sizes = (int*)malloc(sizeof(int) * 2 * n);
    po_projector* P = (po_projector*)malloc(sizeof(po_projector) * n);
    g_ms[0] = g_ms[1] = g_ms[2] = 0;
    for (int i = 0; i < n; i++) {
        int r[4];
        po_projector_set(&P[i], a->kind, a->scale, a->K9s + 9 * i, a->R9s + 9 * i);
        po_warp_roi(&P[i], a->src_w, a->src_h, r);
        corners[2 * i] = r[0]; corners[2 * i + 1] = r[1];
        sizes[2 * i] = r[2]; sizes[2 * i + 1] = r[3];
    }
    po_blender* bl = po_blender_create(a->num_bands);
    po_blender_prepare(bl, n, corners, sizes);
    for (int i = 0; i < n; i++) {
        int w = sizes[2 * i], h = sizes[2 * i + 1], c[2];
        double t0 = now_ms();
        uint8_t* warped = (uint8_t*)malloc((size_t)w * h * 3);
        if (a->front) {
            /* fused front end: the spherical map of the stitcher frame, pushed through the five inverse steps,
             * then cv::remap's fixed-point bilinear on the RAW frame */
            const po_front_end* fe = &a->front[i];
            double newK[9];
            po_optimal_new_camera_matrix(fe->K, fe->dist, fe->undist_w, fe->undist_h, newK);
            float* xm = (float*)malloc(sizeof(float) * (size_t)w * h);
            float* ym = (float*)malloc(sizeof(float) * (size_t)w * h);
            po_build_maps(&P[i], a->src_w, a->src_h, xm, ym);
            for (size_t k = 0; k < (size_t)w * h; k++) po_front_end_map(fe, newK, xm[k], ym[k], &xm[k], &ym[k]);
            po_remap_8u(a->frames[i], fe->raw_w, fe->raw_h, (size_t)fe->raw_w * 3, 3, xm, ym, w, h, PO_INTER_LINEAR,
                        PO_BORDER_REFLECT, warped, (size_t)w * 3);
            free(xm); free(ym);
            c[0] = corners[2 * i]; c[1] = corners[2 * i + 1];
        } else
        po_warp_8u(&P[i], a->frames[i], a->src_w, a->src_h, (size_t)a->src_w * 3, 3, PO_INTER_LINEAR, PO_BORDER_REFLECT,
                   warped, c);
        if (a->gain_maps && a->gain_maps[i]) po_gain_apply_8uc3(warped, w, h, a->gain_maps[i]);
        int16_t* ws = (int16_t*)malloc(sizeof(int16_t) * (size_t)w * h * 3);
        for (size_t k = 0; k < (size_t)w * h * 3; k++) ws[k] = warped[k];
        double t1 = now_ms();
        po_blender_feed(bl, ws, a->masks[i], w, h, corners[2 * i], corners[2 * i + 1]);
        double t2 = now_ms();
        g_ms[0] += t1 - t0;
        g_ms[1] += t2 - t1;
        free(warped);
        free(ws);
    }
    double t0 = now_ms();
    int fr[4];
    po_blender_dst_roi_final(bl, fr);
    int16_t* res = (int16_t*)malloc(sizeof(int16_t) * (size_t)fr[2] * fr[3] * 3);
    uint8_t* rmask = (uint8_t*)malloc((size_t)fr[2] * fr[3]);
    po_blender_blend(bl, res, rmask);
    int cx = a->cut[0], cy = a->cut[1], cw = a->cut[2], ch = a->cut[3];
    if (cw == 0 || ch == 0) { cx = cy = 0; cw = fr[2]; ch = fr[3]; }
    int rc = 0;
    if (cx < 0 || cy < 0 || cx + cw > fr[2] || cy + ch > fr[3]) rc = -1;
    else
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw * 3; x++) out[(size_t)y * cw * 3 + x] = sat8(res[((size_t)(y + cy) * fr[2] + cx) * 3 + x]);
    out_wh[0] = cw; out_wh[1] = ch;
    g_ms[2] = now_ms() - t0;
    free(res); free(rmask); free(corners); free(sizes); free(P);
    po_blender_destroy(bl);
    return rc;
}
