// pano_kernels.hip - gfx950 (MI355X, CDNA4) kernels of the panorama compose path.
//
// Integer / fixed-point arithmetic follows the OpenCV-3.4 CPU routines the reference reaches through
// ocvStitcher::process (reference include/ocvstitcher.hpp:1141-1216):
//   K1 warp      : RotationWarper::warp = buildMaps + cv::remap(INTER_LINEAR fixed point, BORDER_REFLECT)
//                  (:1171) + convertTo(CV_16S) (:1180) + copyMakeBorder(BORDER_REFLECT) of
//                  MultiBandBlender::feed (:1202), fused
//   K2 pyr_down  : cv::pyrDown CV_16S of createLaplacePyr (feed, :1202)
//   K3 blend     : pyrUp + subtract (Laplacian), weight multiply + accumulate (feed), normalise,
//                  pyrUp + add (restoreImageFromLaplacePyr), mask, convertTo(CV_8U), cut
//                  (blend :1207, :1208-1210) - one launch per level, coarse to fine
// Compile with -ffp-contract=off: f32 expressions are evaluated in OpenCV's order, unfused.
// Wave = 64 lanes; HBM-bound byte work - no MFMA on this path.

#include "pano_kernels.hpp"

#include <limits.h>

namespace pano {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int cv_round_dev(float v) {
    // cvRound: round-half-even; x86 "integer indefinite" on overflow / NaN
    int r = __float2int_rn(v);
    return (v >= -2147483648.f && v < 2147483648.f) ? r : INT_MIN;
}
__device__ __forceinline__ int sat16i(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int sat8i(int v) { return min(max(v, 0), 255); }

// cv::borderInterpolate BORDER_REFLECT, closed form (period 2n)
__device__ __forceinline__ int reflect_idx(int p, int n) {
    if ((unsigned)p < (unsigned)n) return p;
    if (n == 1) return 0;
    int period = 2 * n;
    int q = p % period;
    if (q < 0) q += period;
    return q < n ? q : period - 1 - q;
}
// BORDER_REFLECT_101, closed form (period 2n-2)
__device__ __forceinline__ int reflect101_idx(int p, int n) {
    if ((unsigned)p < (unsigned)n) return p;
    if (n == 1) return 0;
    int period = 2 * n - 2;
    int q = p % period;
    if (q < 0) q += period;
    return q < n ? q : period - q;
}

// Spherical/CylindricalProjector::mapBackward from the separable factors, then the 1/32-pixel
// quantisation of cv::remap (INTER_BITS = 5)
__device__ __forceinline__ void map_backward(const float* __restrict__ m, float2 A, float2 B, float& x, float& y) {
    float x_ = B.x * A.x;
    float y_ = B.y;
    float z_ = B.x * A.y;
    x = m[0] * x_ + m[1] * y_ + m[2] * z_;
    y = m[3] * x_ + m[4] * y_ + m[5] * z_;
    float z = m[6] * x_ + m[7] * y_ + m[8] * z_;
    if (z > 0) {
        x /= z;
        y /= z;
    } else {
        x = y = -1.f;
    }
}

// remapBilinear<FixedPtCast<int,uchar,15>>: sum(p*w)+16384 >> 15 with w = (32-a|a)(32-b|b)*32
// == ((32-b)*(p00*(32-a)+p01*a) + b*(p10*(32-a)+p11*a) + 512) >> 10, exact in integers.
__device__ __forceinline__ void sample_bilinear_reflect(const uint8_t* __restrict__ src, int sw, int sh, int stride,
                                                        float fx, float fy, int out[3]) {
    int isx = cv_round_dev(fx * 32.f), isy = cv_round_dev(fy * 32.f);
    int a = isx & 31, b = isy & 31;
    int ix = sat16i(isx >> 5), iy = sat16i(isy >> 5);
    int wa0 = 32 - a, wb0 = 32 - b;
    if (ix >= 0 && ix <= sw - 3 && iy >= 0 && iy <= sh - 2) {
        // interior: the two taps of a row are 6 consecutive bytes; one unaligned 8-byte load per row
        const uint8_t* p = src + (size_t)iy * stride + 3 * ix;
        uint2 t, u;
        __builtin_memcpy(&t, p, 8);
        __builtin_memcpy(&u, p + stride, 8);
        int t0 = t.x & 0xff, t1 = (t.x >> 8) & 0xff, t2 = (t.x >> 16) & 0xff;
        int t3 = t.x >> 24, t4 = t.y & 0xff, t5 = (t.y >> 8) & 0xff;
        int u0 = u.x & 0xff, u1 = (u.x >> 8) & 0xff, u2 = (u.x >> 16) & 0xff;
        int u3 = u.x >> 24, u4 = u.y & 0xff, u5 = (u.y >> 8) & 0xff;
        out[0] = (wb0 * (t0 * wa0 + t3 * a) + b * (u0 * wa0 + u3 * a) + 512) >> 10;
        out[1] = (wb0 * (t1 * wa0 + t4 * a) + b * (u1 * wa0 + u4 * a) + 512) >> 10;
        out[2] = (wb0 * (t2 * wa0 + t5 * a) + b * (u2 * wa0 + u5 * a) + 512) >> 10;
    } else {
        int x0 = reflect_idx(ix, sw), x1 = reflect_idx(ix + 1, sw);
        int y0 = reflect_idx(iy, sh), y1 = reflect_idx(iy + 1, sh);
        const uint8_t* r0 = src + (size_t)y0 * stride;
        const uint8_t* r1 = src + (size_t)y1 * stride;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int p00 = r0[3 * x0 + c], p01 = r0[3 * x1 + c], p10 = r1[3 * x0 + c], p11 = r1[3 * x1 + c];
            out[c] = (wb0 * (p00 * wa0 + p01 * a) + b * (p10 * wa0 + p11 * a) + 512) >> 10;
        }
    }
}

// BlocksGainCompensator::apply: gain = bilinear (cv::resize INTER_LINEAR, f32) of the block map,
// px = saturate_cast<uchar>(px * gain)
__device__ __forceinline__ void apply_gain(const WarpCam& c, int x, int y, int v[3]) {
    int2 gx = c.gcol[x];
    float2 ax = c.gcolw[x];
    int2 gy = c.grow[y];
    float2 by = c.groww[y];
    const float* S0 = c.gain + (size_t)gy.x * c.gw;
    const float* S1 = c.gain + (size_t)gy.y * c.gw;
    float h0 = S0[gx.x] * ax.x + S0[gx.y] * ax.y;
    float h1 = S1[gx.x] * ax.x + S1[gx.y] * ax.y;
    float g = h0 * by.x + h1 * by.y;
#pragma unroll
    for (int k = 0; k < 3; k++) v[k] = sat8i(cv_round_dev((float)v[k] * g));
}

// ------------------------------------------------------------------------------------------------
// K1: fused warp of every camera's bordered feed() tile.  grid = (ceil(tw/128), ceil(th/4), ncam),
// block = (64,4): one wave per tile row segment, 2 adjacent pixels per lane -> each lane stores
// 12 contiguous bytes (3 dwords), a wave stores 768 contiguous bytes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_tiles_kernel(WarpParams P) {
    const WarpCam& c = P.cam[blockIdx.z];
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 2;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= c.tw || y >= c.th) return;
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; i++) m[i] = c.m[i];
    const float2 B = c.rowB[y];
    int v[2][3];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        int x = min(x0 + j, c.tw - 1);
        float fx, fy;
        map_backward(m, c.colA[x], B, fx, fy);
        sample_bilinear_reflect(c.src, c.src_w, c.src_h, c.src_stride, fx, fy, v[j]);
        if (c.gain) apply_gain(c, x, y, v[j]);
    }
    int16_t* row = (int16_t*)c.dst + ((size_t)y * c.dst_pitch + x0) * 3;
    if (x0 + 1 < c.tw) {
        uint3 pk;
        pk.x = (unsigned)v[0][0] | ((unsigned)v[0][1] << 16);
        pk.y = (unsigned)v[0][2] | ((unsigned)v[1][0] << 16);
        pk.z = (unsigned)v[1][1] | ((unsigned)v[1][2] << 16);
        *reinterpret_cast<uint3*>(row) = pk;
    } else {
        row[0] = (int16_t)v[0][0];
        row[1] = (int16_t)v[0][1];
        row[2] = (int16_t)v[0][2];
    }
}

void launch_warp_tiles(const WarpParams& p, int ncam, int max_tw, int max_th, hipStream_t s) {
    dim3 block(64, 4, 1);
    dim3 grid((max_tw + 127) / 128, (max_th + 3) / 4, ncam);
    hipLaunchKernelGGL(warp_tiles_kernel, grid, block, 0, s, p);
}

// stage entry: RotationWarper::warp to an 8UC3 image (no border, byte pitch)
__global__ __launch_bounds__(256) void warp_image_kernel(WarpCam c) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= c.tw || y >= c.th) return;
    float fx, fy;
    int v[3];
    map_backward(c.m, c.colA[x], c.rowB[y], fx, fy);
    sample_bilinear_reflect(c.src, c.src_w, c.src_h, c.src_stride, fx, fy, v);
    if (c.gain) apply_gain(c, x, y, v);
    uint8_t* d = (uint8_t*)c.dst + (size_t)y * c.dst_pitch + 3 * x;
    d[0] = (uint8_t)v[0];
    d[1] = (uint8_t)v[1];
    d[2] = (uint8_t)v[2];
}
void launch_warp_image(const WarpCam& c, hipStream_t s) {
    dim3 block(64, 4, 1), grid((c.tw + 63) / 64, (c.th + 3) / 4, 1);
    hipLaunchKernelGGL(warp_image_kernel, grid, block, 0, s, c);
}

// remapNearest of an all-255 mask with BORDER_CONSTANT: 255 where the rounded source position is
// inside the frame
__global__ __launch_bounds__(256) void warp_mask_kernel(WarpCam c, uint8_t* dst, int dst_stride) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= c.tw || y >= c.th) return;
    float fx, fy;
    map_backward(c.m, c.colA[x], c.rowB[y], fx, fy);
    int sx = sat16i(cv_round_dev(fx)), sy = sat16i(cv_round_dev(fy));
    dst[(size_t)y * dst_stride + x] = ((unsigned)sx < (unsigned)c.src_w && (unsigned)sy < (unsigned)c.src_h) ? 255 : 0;
}
void launch_warp_mask(const WarpCam& c, uint8_t* dst, int dst_stride, hipStream_t s) {
    dim3 block(64, 4, 1), grid((c.tw + 63) / 64, (c.th + 3) / 4, 1);
    hipLaunchKernelGGL(warp_mask_kernel, grid, block, 0, s, c, dst, dst_stride);
}

// ------------------------------------------------------------------------------------------------
// K2: pyrDown CV_16S x3, REFLECT_101, (v+128)>>8.  One thread per output pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrParams P, unsigned cam_bits, int l) {
    if (!((cam_bits >> blockIdx.z) & 1u)) return;
    const PyrCam& c = P.cam[blockIdx.z];
    const int sw = c.w0 >> l, sh = c.h0 >> l;
    const int dw = sw >> 1, dh = sh >> 1;
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int16_t* __restrict__ src = c.lvl[l];
    const int sp = c.pitch[l] * 3;
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = reflect101_idx(2 * x + k - 2, sw) * 3;
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int16_t* r = src + (size_t)reflect101_idx(2 * y + k - 2, sh) * sp;
        const int wy = (k == 2) ? 6 : ((k == 1 || k == 3) ? 4 : 1);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            int h = r[xs[2] + ch] * 6 + (r[xs[1] + ch] + r[xs[3] + ch]) * 4 + r[xs[0] + ch] + r[xs[4] + ch];
            acc[ch] += h * wy;
        }
    }
    int16_t* d = c.lvl[l + 1] + ((size_t)y * c.pitch[l + 1] + x) * 3;
    d[0] = (int16_t)sat16i((acc[0] + 128) >> 8);
    d[1] = (int16_t)sat16i((acc[1] + 128) >> 8);
    d[2] = (int16_t)sat16i((acc[2] + 128) >> 8);
}

void launch_pyr_down(const PyrParams& p, unsigned cam_bits, int l, hipStream_t s) {
    int mw = 0, mh = 0;
    for (int i = 0; i < p.ncam; i++)
        if ((cam_bits >> i) & 1u) {
            mw = max(mw, p.cam[i].w0 >> (l + 1));
            mh = max(mh, p.cam[i].h0 >> (l + 1));
        }
    if (mw == 0 || mh == 0) return;
    dim3 block(64, 4, 1), grid((mw + 63) / 64, (mh + 3) / 4, p.ncam);
    hipLaunchKernelGGL(pyr_down_kernel, grid, block, 0, s, p, cam_bits, l);
}

// ------------------------------------------------------------------------------------------------
// pyrUp CV_16S x3 sampled at one destination pixel (X, Y) of an exactly-2x image:
// even: s[x-1] + 6 s[x] + s[x+1], odd: 4 (s[x] + s[x+1]); left/top reflect-101, right/bottom
// replicate; (v + 32) >> 6, saturate.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pyr_up_px(const int16_t* __restrict__ S, int n, int m, int pitch, int X, int Y,
                                          int out[3]) {
    const int x = X >> 1, y = Y >> 1;
    int xi[3], wx[3], yi[3], wy[3];
    if (!(X & 1)) {
        xi[0] = x > 0 ? x - 1 : (n > 1 ? 1 : 0); xi[1] = x; xi[2] = min(x + 1, n - 1);
        wx[0] = 1; wx[1] = 6; wx[2] = 1;
    } else {
        xi[0] = x; xi[1] = min(x + 1, n - 1); xi[2] = x;
        wx[0] = 4; wx[1] = 4; wx[2] = 0;
    }
    if (!(Y & 1)) {
        yi[0] = y > 0 ? y - 1 : (m > 1 ? 1 : 0); yi[1] = y; yi[2] = min(y + 1, m - 1);
        wy[0] = 1; wy[1] = 6; wy[2] = 1;
    } else {
        yi[0] = y; yi[1] = min(y + 1, m - 1); yi[2] = y;
        wy[0] = 4; wy[1] = 4; wy[2] = 0;
    }
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int16_t* r = S + (size_t)yi[j] * pitch * 3;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            int h = r[xi[0] * 3 + ch] * wx[0] + r[xi[1] * 3 + ch] * wx[1] + r[xi[2] * 3 + ch] * wx[2];
            acc[ch] += h * wy[j];
        }
    }
    out[0] = sat16i((acc[0] + 32) >> 6);
    out[1] = sat16i((acc[1] + 32) >> 6);
    out[2] = sat16i((acc[2] + 32) >> 6);
}

// ------------------------------------------------------------------------------------------------
// K3: one level of the blend, one thread per canvas pixel.
//   acc  = sum over cameras in feed order of (short)(lap * w)          (wrapping short add)
//   lap  = sat16(G_l - pyrUp(G_{l+1}))  (top level: G_l)
//   norm = (short)(acc / (W + 1e-5f))
//   out  = sat16(norm + pyrUp(out_{l+1}))                                (top level: norm)
// level 0 applies dst_mask (W0 > eps), convertTo(CV_8U) and the cut, and writes the panorama.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void blend_level_kernel(PyrParams P, CanvasParams C, int l) {
    const int cw = C.w0 >> l, ch = C.h0 >> l;
    int X = blockIdx.x * 64 + threadIdx.x;
    int Y = blockIdx.y * 4 + threadIdx.y;
    if (l == 0) {
        if (X >= C.cut_w || Y >= C.cut_h) return;
        X += C.cut_x;
        Y += C.cut_y;
    } else if (X >= cw || Y >= ch) {
        return;
    }
    int acc[3] = {0, 0, 0};
    for (int i = 0; i < P.ncam; i++) {
        const PyrCam& c = P.cam[i];
        const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
        const int tw = c.w0 >> l, th = c.h0 >> l;
        if ((unsigned)x >= (unsigned)tw || (unsigned)y >= (unsigned)th) continue;
        const float w = c.wgt[l][(size_t)y * c.pitch[l] + x];
        if (w == 0.f) continue;  // (short)(lap * 0) == 0
        const int16_t* g = c.lvl[l] + ((size_t)y * c.pitch[l] + x) * 3;
        int lap[3] = {g[0], g[1], g[2]};
        if (l < C.bands) {
            int up[3];
            pyr_up_px(c.lvl[l + 1], tw >> 1, th >> 1, c.pitch[l + 1], x, y, up);
            lap[0] = sat16i(lap[0] - up[0]);
            lap[1] = sat16i(lap[1] - up[1]);
            lap[2] = sat16i(lap[2] - up[2]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = (int16_t)(acc[k] + (int16_t)(int)((float)lap[k] * w));
    }
    const float W = C.wsum[l][(size_t)Y * cw + X];
    const float den = W + 1e-5f;
    int v[3];
#pragma unroll
    for (int k = 0; k < 3; k++) v[k] = (int16_t)(int)((float)acc[k] / den);
    if (l < C.bands) {
        int up[3];
        pyr_up_px(C.img[l + 1], cw >> 1, ch >> 1, cw >> 1, X, Y, up);
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = sat16i(v[k] + up[k]);
    }
    if (l > 0) {
        int16_t* d = C.img[l] + ((size_t)Y * cw + X) * 3;
        d[0] = (int16_t)v[0];
        d[1] = (int16_t)v[1];
        d[2] = (int16_t)v[2];
    } else {
        const bool on = W > 1e-5f;
        uint8_t* d = C.out + (size_t)(Y - C.cut_y) * C.out_stride + 3 * (X - C.cut_x);
        d[0] = on ? (uint8_t)sat8i(v[0]) : 0;
        d[1] = on ? (uint8_t)sat8i(v[1]) : 0;
        d[2] = on ? (uint8_t)sat8i(v[2]) : 0;
    }
}

void launch_blend_level(const PyrParams& p, const CanvasParams& c, int l, hipStream_t s) {
    int w = l == 0 ? c.cut_w : (c.w0 >> l), h = l == 0 ? c.cut_h : (c.h0 >> l);
    dim3 block(64, 4, 1), grid((w + 63) / 64, (h + 3) / 4, 1);
    hipLaunchKernelGGL(blend_level_kernel, grid, block, 0, s, p, c, l);
}

// Blender::NO: Blender::feed masked copy in feed order, Blender::blend zeroing, convertTo(8U), cut
struct NoBlendArgs {
    const uint8_t* mask[kCams];
    int mpitch[kCams];
    int rx[kCams], ry[kCams], rw[kCams], rh[kCams];
};
__global__ __launch_bounds__(256) void no_blend_kernel(PyrParams P, NoBlendArgs A, CanvasParams C) {
    int X = blockIdx.x * 64 + threadIdx.x;
    int Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= C.cut_w || Y >= C.cut_h) return;
    X += C.cut_x;
    Y += C.cut_y;
    int v[3] = {0, 0, 0};
    for (int i = 0; i < P.ncam; i++) {
        const int x = X - A.rx[i], y = Y - A.ry[i];
        if ((unsigned)x >= (unsigned)A.rw[i] || (unsigned)y >= (unsigned)A.rh[i]) continue;
        if (!A.mask[i][(size_t)y * A.mpitch[i] + x]) continue;
        const int16_t* g = P.cam[i].lvl[0] + ((size_t)y * P.cam[i].pitch[0] + x) * 3;
        v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
    }
    uint8_t* d = C.out + (size_t)(Y - C.cut_y) * C.out_stride + 3 * (X - C.cut_x);
    d[0] = (uint8_t)sat8i(v[0]);
    d[1] = (uint8_t)sat8i(v[1]);
    d[2] = (uint8_t)sat8i(v[2]);
}
void launch_no_blend(const PyrParams& p, const uint8_t* const* masks, const int* mask_pitch, const int* roi_x,
                     const int* roi_y, const int* roi_w, const int* roi_h, const CanvasParams& c, hipStream_t s) {
    NoBlendArgs a;
    for (int i = 0; i < p.ncam; i++) {
        a.mask[i] = masks[i]; a.mpitch[i] = mask_pitch[i];
        a.rx[i] = roi_x[i]; a.ry[i] = roi_y[i]; a.rw[i] = roi_w[i]; a.rh[i] = roi_h[i];
    }
    dim3 block(64, 4, 1), grid((c.cut_w + 63) / 64, (c.cut_h + 3) / 4, 1);
    hipLaunchKernelGGL(no_blend_kernel, grid, block, 0, s, p, a, c);
}

// ------------------------------------------------------------------------------------------------
// weights: mask * (1/255.f) with copyMakeBorder(CONSTANT 0); pyrDown CV_32F (scalar evaluation order);
// canvas sum in feed order
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_to_weight_kernel(const uint8_t* mask, int mw, int mh, int mpitch, int left,
                                                             int top, float* w0, int tw, int th, int pitch) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= tw || y >= th) return;
    const int sx = x - left, sy = y - top;
    float v = 0.f;
    if ((unsigned)sx < (unsigned)mw && (unsigned)sy < (unsigned)mh)
        v = (float)mask[(size_t)sy * mpitch + sx] * (float)(1. / 255.);
    w0[(size_t)y * pitch + x] = v;
}
void launch_mask_to_weight(const uint8_t* mask, int mw, int mh, int mpitch, int left, int top, float* w0, int tw,
                           int th, int pitch, hipStream_t s) {
    dim3 block(64, 4, 1), grid((tw + 63) / 64, (th + 3) / 4, 1);
    hipLaunchKernelGGL(mask_to_weight_kernel, grid, block, 0, s, mask, mw, mh, mpitch, left, top, w0, tw, th, pitch);
}

__global__ __launch_bounds__(256) void pyr_down_f32_kernel(const float* __restrict__ src, int sw, int sh, int spitch,
                                                           float* __restrict__ dst, int dpitch) {
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = reflect101_idx(2 * x + k - 2, sw);
    float row[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float* r = src + (size_t)reflect101_idx(2 * y + k - 2, sh) * spitch;
        row[k] = r[xs[2]] * 6 + (r[xs[1]] + r[xs[3]]) * 4 + r[xs[0]] + r[xs[4]];
    }
    dst[(size_t)y * dpitch + x] = (row[2] * 6 + (row[1] + row[3]) * 4 + row[0] + row[4]) * (1.f / 256);
}
void launch_pyr_down_f32(const float* src, int sw, int sh, int spitch, float* dst, int dpitch, hipStream_t s) {
    int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dim3 block(64, 4, 1), grid((dw + 63) / 64, (dh + 3) / 4, 1);
    hipLaunchKernelGGL(pyr_down_f32_kernel, grid, block, 0, s, src, sw, sh, spitch, dst, dpitch);
}

__global__ __launch_bounds__(256) void sum_weights_kernel(PyrParams P, int l, float* wsum, int cw, int ch) {
    const int X = blockIdx.x * 64 + threadIdx.x, Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= cw || Y >= ch) return;
    float W = 0.f;
    for (int i = 0; i < P.ncam; i++) {
        const PyrCam& c = P.cam[i];
        const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
        if ((unsigned)x >= (unsigned)(c.w0 >> l) || (unsigned)y >= (unsigned)(c.h0 >> l)) continue;
        W += c.wgt[l][(size_t)y * c.pitch[l] + x];
    }
    wsum[(size_t)Y * cw + X] = W;
}
void launch_sum_weights(const PyrParams& p, int l, float* wsum, int cw, int ch, hipStream_t s) {
    dim3 block(64, 4, 1), grid((cw + 63) / 64, (ch + 3) / 4, 1);
    hipLaunchKernelGGL(sum_weights_kernel, grid, block, 0, s, p, l, wsum, cw, ch);
}

// ------------------------------------------------------------------------------------------------
// mask preparation
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dilate3x3_kernel(const uint8_t* src, uint8_t* dst, int w, int h) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    int m = 0;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            int xx = x + dx, yy = y + dy;
            if ((unsigned)xx < (unsigned)w && (unsigned)yy < (unsigned)h) m = max(m, (int)src[(size_t)yy * w + xx]);
        }
    dst[(size_t)y * w + x] = (uint8_t)m;
}
void launch_dilate3x3(const uint8_t* src, uint8_t* dst, int w, int h, hipStream_t s) {
    dim3 block(64, 4, 1), grid((w + 63) / 64, (h + 3) / 4, 1);
    hipLaunchKernelGGL(dilate3x3_kernel, grid, block, 0, s, src, dst, w, h);
}

// cv::resize INTER_LINEAR_EXACT CV_8UC1: 8.8 horizontal, 16.16 vertical; coefficient tables from the host
__global__ __launch_bounds__(256) void resize_linear_exact_kernel(const uint8_t* src, int sw, int sh, uint8_t* dst,
                                                                  int dw, int dh, const int* xofs, const int* xc1,
                                                                  const int* yofs, const int* yc1, int minx, int maxx,
                                                                  int miny, int maxy) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    auto hval = [&](int row) -> unsigned {
        const uint8_t* s = src + (size_t)row * sw;
        if (x < minx) return (unsigned)s[0] << 8;
        if (x >= maxx) return (unsigned)s[sw - 1] << 8;
        int o = xofs[x], c1 = xc1[x];
        unsigned v = s[o] * (unsigned)(256 - c1) + s[o + 1] * (unsigned)c1;
        return v > 65535u ? 65535u : v;
    };
    int out;
    if (y < miny) out = (int)((hval(0) + 128) >> 8);
    else if (y >= maxy) out = (int)((hval(sh - 1) + 128) >> 8);
    else {
        int o = yofs[y], c1 = yc1[y];
        unsigned long long v = (unsigned long long)hval(o) * (unsigned)(256 - c1) + (unsigned long long)hval(o + 1) * (unsigned)c1;
        if (v > 0xffffffffull) v = 0xffffffffull;
        out = (int)((v + 32768) >> 16);
    }
    dst[(size_t)y * dw + x] = (uint8_t)sat8i(out);
}
void launch_resize_linear_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, const int* xofs,
                                const int* xc1, const int* yofs, const int* yc1, int minx, int maxx, int miny, int maxy,
                                hipStream_t s) {
    dim3 block(64, 4, 1), grid((dw + 63) / 64, (dh + 3) / 4, 1);
    hipLaunchKernelGGL(resize_linear_exact_kernel, grid, block, 0, s, src, sw, sh, dst, dw, dh, xofs, xc1, yofs, yc1,
                       minx, maxx, miny, maxy);
}

__global__ __launch_bounds__(256) void and_kernel(const uint8_t* a, const uint8_t* b, uint8_t* d, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = a[i] & b[i];
}
void launch_and(const uint8_t* a, const uint8_t* b, uint8_t* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(and_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, dst, n);
}

// VoronoiSeamFinder::findInPair.  The L1 distance transform (cv::distanceTransform DIST_L1 mask 3 =
// exact city-block distance) is two 1-D min-plus scans: along columns, then along rows.
constexpr int kVorGap = 10;
constexpr int kVorInf = 1 << 28;
struct VorArgs {
    uint8_t *m1, *m2;
    int w1, h1, tlx1, tly1, w2, h2, tlx2, tly2;
    int rx, ry, rw, rh;
    int* d1;
    int* d2;
};
__global__ void voronoi_init_kernel(VorArgs a) {
    const int W = a.rw + 2 * kVorGap, H = a.rh + 2 * kVorGap;
    const int gx = blockIdx.x * 64 + threadIdx.x, gy = blockIdx.y * 4 + threadIdx.y;
    if (gx >= W || gy >= H) return;
    const int x = gx - kVorGap, y = gy - kVorGap;
    const int y1 = a.ry - a.tly1 + y, x1 = a.rx - a.tlx1 + x;
    const int y2 = a.ry - a.tly2 + y, x2 = a.rx - a.tlx2 + x;
    int s1 = (y1 >= 0 && x1 >= 0 && y1 < a.h1 && x1 < a.w1) ? a.m1[(size_t)y1 * a.w1 + x1] : 0;
    int s2 = (y2 >= 0 && x2 >= 0 && y2 < a.h2 && x2 < a.w2) ? a.m2[(size_t)y2 * a.w2 + x2] : 0;
    const bool coll = s1 != 0 && s2 != 0;
    if (coll) s1 = s2 = 0;
    a.d1[(size_t)gy * W + gx] = s1 != 0 ? 0 : kVorInf;
    a.d2[(size_t)gy * W + gx] = s2 != 0 ? 0 : kVorInf;
}
__global__ void voronoi_cols_kernel(int* d1, int* d2, int W, int H) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= W) return;
    int* d = blockIdx.y == 0 ? d1 : d2;
    int run = kVorInf;
    for (int y = 0; y < H; y++) {
        run = min(d[(size_t)y * W + x], run + 1);
        d[(size_t)y * W + x] = run;
    }
    run = kVorInf;
    for (int y = H - 1; y >= 0; y--) {
        run = min(d[(size_t)y * W + x], run + 1);
        d[(size_t)y * W + x] = run;
    }
}
__global__ void voronoi_rows_kernel(int* d1, int* d2, int W, int H) {
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= H) return;
    int* d = (blockIdx.y == 0 ? d1 : d2) + (size_t)y * W;
    int run = kVorInf;
    for (int x = 0; x < W; x++) {
        run = min(d[x], run + 1);
        d[x] = run;
    }
    run = kVorInf;
    for (int x = W - 1; x >= 0; x--) {
        run = min(d[x], run + 1);
        d[x] = run;
    }
}
__global__ void voronoi_apply_kernel(VorArgs a) {
    const int W = a.rw + 2 * kVorGap;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= a.rw || y >= a.rh) return;
    const size_t k = (size_t)(y + kVorGap) * W + x + kVorGap;
    // clamp like a saturated "far" value so that two unreachable distances compare equal
    const int e1 = min(a.d1[k], kVorInf), e2 = min(a.d2[k], kVorInf);
    if (e1 < e2)
        a.m2[(size_t)(a.ry - a.tly2 + y) * a.w2 + (a.rx - a.tlx2 + x)] = 0;
    else
        a.m1[(size_t)(a.ry - a.tly1 + y) * a.w1 + (a.rx - a.tlx1 + x)] = 0;
}
size_t voronoi_scratch_ints(int rw, int rh) { return 2 * (size_t)(rw + 2 * kVorGap) * (rh + 2 * kVorGap); }
void launch_voronoi_pair(uint8_t* mask1, int w1, int h1, int tlx1, int tly1, uint8_t* mask2, int w2, int h2, int tlx2,
                         int tly2, int rx, int ry, int rw, int rh, int* scratch, hipStream_t s) {
    const int W = rw + 2 * kVorGap, H = rh + 2 * kVorGap;
    VorArgs a{mask1, mask2, w1, h1, tlx1, tly1, w2, h2, tlx2, tly2, rx, ry, rw, rh, scratch, scratch + (size_t)W * H};
    hipLaunchKernelGGL(voronoi_init_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, s, a);
    hipLaunchKernelGGL(voronoi_cols_kernel, dim3((W + 63) / 64, 2), dim3(64), 0, s, a.d1, a.d2, W, H);
    hipLaunchKernelGGL(voronoi_rows_kernel, dim3((H + 63) / 64, 2), dim3(64), 0, s, a.d1, a.d2, W, H);
    hipLaunchKernelGGL(voronoi_apply_kernel, dim3((rw + 63) / 64, (rh + 3) / 4), dim3(64, 4), 0, s, a);
}

}  // namespace pano
