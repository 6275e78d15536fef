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

#include <hip/hip_ext.h>

#include <limits.h>
#include <stdlib.h>

namespace pano {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ int cv_round_dev(float v) {
    // cvRound: round-half-even; x86 "integer indefinite" on overflow / NaN
    int r = __float2int_rn(v);
    return __builtin_fabsf(v) < 2147483648.f ? r : INT_MIN;  // NaN compares false
}
__device__ __forceinline__ int sat16i(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int sat8i(int v) { return min(max(v, 0), 255); }

// cv::borderInterpolate BORDER_REFLECT.  One fold covers -n <= p < 2n (every tap the warp can ask for
// near a frame); the closed form (period 2n) with its integer division is kept for anything farther out.
__device__ __forceinline__ int reflect_idx(int p, int n) {
    if ((unsigned)p < (unsigned)n) return p;
    int q = p < 0 ? -p - 1 : 2 * n - 1 - p;
    if ((unsigned)q < (unsigned)n) return q;
    if (n == 1) return 0;
    int period = 2 * n;
    q = p % period;
    if (q < 0) q += period;
    return q < n ? q : period - 1 - q;
}
// BORDER_REFLECT_101 for the pyramid stencils: p is never farther than 2 outside [0, n).  Three folds cover every
// n >= 2 (n == 2: -2 -> 2 -> 0, 3 -> -1 -> 1), exactly like cv::borderInterpolate's loop.
__device__ __forceinline__ int reflect101_idx(int p, int n) {
    if ((unsigned)p < (unsigned)n) return p;
    if (n == 1) return 0;
    p = p < 0 ? -p : p;
    p = p >= n ? 2 * n - 2 - p : p;
    p = p < 0 ? -p : p;
    return min(p, n - 1);  // only reached by the out-of-image outputs of a partial last group (values unused)
}

// n - sign(n): the normalisation where the summed weight is exactly 1.0f (see blend_level_vec_kernel).  Spelled as
// v_med3_i32 + v_sub: the compiler turns every C spelling of sign() back into two compares and two selects.
__device__ __forceinline__ int toward_zero_by_one(int n) {
    int sgn;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sgn) : "v"(n));
    return n - sgn;
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

// stitcher-frame coordinates -> raw-frame coordinates through the inverse of the reference's undistort chain
// (nvcam.hpp:898-921, :1094): resize(undist->out), resize(crop->undist), crop, initUndistortRectifyMap's formula
// with R = I at the fractional position, resize(raw->undist).  double, same expression order as the oracle.
__device__ __forceinline__ void front_end_map(const FrontEndDev& fe, float xo, float yo, float& xr, float& yr) {
    double x = ((double)xo + 0.5) * ((double)fe.undist_w / fe.out_w) - 0.5;
    double y = ((double)yo + 0.5) * ((double)fe.undist_h / fe.out_h) - 0.5;
    x = (x + 0.5) * ((double)fe.rect[2] / fe.undist_w) - 0.5 + fe.rect[0];
    y = (y + 0.5) * ((double)fe.rect[3] / fe.undist_h) - 0.5 + fe.rect[1];
    const double nx = (x - fe.newK[2]) / fe.newK[0], ny = (y - fe.newK[5]) / fe.newK[4];
    const double x2 = nx * nx, y2 = ny * ny, r2 = x2 + y2, _2xy = 2 * nx * ny;
    const double kr = 1 + ((0 * r2 + fe.dist[1]) * r2 + fe.dist[0]) * r2;
    const double xd = nx * kr + fe.dist[2] * _2xy + fe.dist[3] * (r2 + 2 * x2);
    const double yd = ny * kr + fe.dist[2] * (r2 + 2 * y2) + fe.dist[3] * _2xy;
    double u = fe.K[0] * xd + fe.K[2], v = fe.K[4] * yd + fe.K[5];
    u = (u + 0.5) * ((double)fe.raw_w / fe.undist_w) - 0.5;
    v = (v + 0.5) * ((double)fe.raw_h / fe.undist_h) - 0.5;
    xr = (float)u;
    yr = (float)v;
}
// mapBackward (+ front end)
__device__ __forceinline__ void map_source(const WarpCam& c, const float* __restrict__ m, float2 A, float2 B, float& x,
                                           float& y) {
    map_backward(m, A, B, x, y);
    if (c.fe) front_end_map(*c.fe, x, y, x, y);
}

// remapBilinear<FixedPtCast<int,uchar,15>>: sum(p*w)+16384 >> 15 with w = (32-a|a)(32-b|b)*32
// == ((32-b)*(p00*(32-a)+p01*a) + b*(p10*(32-a)+p11*a) + 512) >> 10, exact in integers.
__device__ __forceinline__ void sample_bilinear_reflect(const uint8_t* __restrict__ src, int sw, int sh, int stride,
                                                        float fx, float fy, int out[3]) {
    int isx = cv_round_dev(fx * 32.f), isy = cv_round_dev(fy * 32.f);
    int a = isx & 31, b = isy & 31;
    int ix = sat16i(isx >> 5), iy = sat16i(isy >> 5);

    int wa0 = 32 - a, wb0 = 32 - b;
    if (ix >= 0 && ix <= sw - 3) {
        // the two taps of a row are 6 consecutive bytes: one unaligned 8-byte load per row (ix <= sw-3 keeps
        // the 2 spare bytes inside the row); rows reflect independently
        int y0 = iy, y1 = iy + 1;
        if (iy < 0 || iy > sh - 2) {
            y0 = reflect_idx(iy, sh);
            y1 = reflect_idx(iy + 1, sh);
        }
        const uint8_t* p = src + 3 * ix;
        uint2 t, u;
        __builtin_memcpy(&t, p + (size_t)y0 * stride, 8);
        __builtin_memcpy(&u, p + (size_t)y1 * stride, 8);
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
    const int2 gy = c.grow[y];
    const float2 by = c.groww[y];
    const float h0 = c.ghrow[(unsigned)(gy.x * c.ghrow_pitch + x)];
    const float h1 = c.ghrow[(unsigned)(gy.y * c.ghrow_pitch + x)];
    const float g = h0 * by.x + h1 * by.y;
#pragma unroll
    for (int k = 0; k < 3; k++) v[k] = sat8i(cv_round_dev((float)v[k] * g));
}

// ------------------------------------------------------------------------------------------------
// K1: fused warp of every camera's bordered feed() tile.  grid = (ceil(tw/256), ceil(th/4), ncam),
// block = (64,4): one wave per tile-row segment, 4 adjacent pixels per lane.  Output is planar u8
// (B, G, R planes): each lane stores one dword per plane, a wave stores 3 x 256 contiguous bytes.
// Four independent tap fetches per lane are in flight together (latency hiding by ILP).
// ------------------------------------------------------------------------------------------------
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));

// The 6 tap bytes B0 G0 R0 B1 G1 R1 at byte offset o of the frame, returned in the low 6 bytes of a uint2.
// An unaligned 8-byte load costs the texture-address unit roughly twice an aligned one (measured: the
// table-form K1 runs 30 us with unaligned taps, 22 us aligned), so fetch the enclosing 4-byte-aligned
// 12 bytes (global_load_dwordx3) and realign in registers (2 x v_alignbyte_b32).
__device__ __forceinline__ uint2 load_taps6(const uint8_t* __restrict__ src, unsigned lo, unsigned o) {
    const unsigned k = (o + lo) & 3u;
    // signed: with an unaligned frame pointer the dword under the first pixels starts up to 3 bytes BEFORE src
    const uint3 d = *reinterpret_cast<const uint3*>(src + (int)(o - k));
    return make_uint2(__builtin_amdgcn_alignbyte(d.y, d.x, k), __builtin_amdgcn_alignbyte(d.z, d.y, k));
}

// The same 6 bytes one at a time, never past byte `last` of the frame: for the few pixels whose aligned 12-byte fetch would
// end beyond the frame (taps on the last pixels of the last rows).  A clamped byte only ever meets weight 0.
template <typename P>
__device__ __forceinline__ uint2 taps6_bytes(P src, unsigned o, unsigned last) {
    unsigned b[6];
#pragma unroll
    for (int i = 0; i < 6; i++) b[i] = src[min(o + i, last)];
    return make_uint2(b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24), b[4] | (b[5] << 8));
}

// x / z and y / z, correctly rounded (IEEE-754 round-to-nearest-even), for a shared denominator.
// This is the AMDGPU f32 division expansion (v_rcp + Newton refinement + two residual corrections) with
// the reciprocal refinement shared by both quotients and without the exponent pre-scaling, which is a
// no-op when 2^-40 <= z <= 2^40 (checked by the caller; anything else takes the generic path).
__device__ __forceinline__ void div2_shared(float x, float y, float z, float& qx, float& qy) {
    float r = __builtin_amdgcn_rcpf(z);
    const float e = __builtin_fmaf(-z, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = x * r;
    float rem = __builtin_fmaf(-z, q, x);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-z, q, x);
    qx = __builtin_fmaf(rem, r, q);
    q = y * r;
    rem = __builtin_fmaf(-z, q, y);
    q = __builtin_fmaf(rem, r, q);
    rem = __builtin_fmaf(-z, q, y);
    qy = __builtin_fmaf(rem, r, q);
}

// one interior pixel: two unaligned 8-byte tap loads already issued (t = row iy, u = row iy+1, each holding
// the 6 bytes B0 G0 R0 B1 G1 R1); bilinear in packed 16-bit lanes:
//   h = p_left * (32-a) + p_right * a   for the top (low half) and bottom (high half) rows at once
//   out = (h_top * (32-b) + h_bot * b + 512) >> 10          (v_dot2_u32_u16)
__device__ __forceinline__ void bilinear_packed(uint2 t, uint2 u, int a, int b, int out[3]) {
    const unsigned wa0 = (unsigned)(32 - a) * 0x00010001u, wa1 = (unsigned)a * 0x00010001u;
    const us2_t W0 = __builtin_bit_cast(us2_t, wa0), W1 = __builtin_bit_cast(us2_t, wa1);
    const us2_t WB = __builtin_bit_cast(us2_t, (unsigned)(32 - b) | ((unsigned)b << 16));
    // v_perm_b32: bytes 0-3 come from the 2nd operand, 4-7 from the 1st, 0x0c = zero
    const unsigned l0 = __builtin_amdgcn_perm(u.x, t.x, 0x0c040c00u), r0 = __builtin_amdgcn_perm(u.x, t.x, 0x0c070c03u);
    const unsigned l1 = __builtin_amdgcn_perm(u.x, t.x, 0x0c050c01u), r1 = __builtin_amdgcn_perm(u.y, t.y, 0x0c040c00u);
    const unsigned l2 = __builtin_amdgcn_perm(u.x, t.x, 0x0c060c02u), r2 = __builtin_amdgcn_perm(u.y, t.y, 0x0c050c01u);
    const us2_t h0 = __builtin_bit_cast(us2_t, l0) * W0 + __builtin_bit_cast(us2_t, r0) * W1;
    const us2_t h1 = __builtin_bit_cast(us2_t, l1) * W0 + __builtin_bit_cast(us2_t, r1) * W1;
    const us2_t h2 = __builtin_bit_cast(us2_t, l2) * W0 + __builtin_bit_cast(us2_t, r2) * W1;
    out[0] = (int)(__builtin_amdgcn_udot2(h0, WB, 512u, false) >> 10);
    out[1] = (int)(__builtin_amdgcn_udot2(h1, WB, 512u, false) >> 10);
    out[2] = (int)(__builtin_amdgcn_udot2(h2, WB, 512u, false) >> 10);
}

// ------------------------------------------------------------------------------------------------
// K1: fused warp of every camera's bordered feed() tile.  grid = (ceil(tw/256), ceil(th/4), ncam),
// block = (64,4): one wave per tile-row segment, 4 adjacent pixels per lane.  Output is planar u8
// (B, G, R planes): each lane stores one dword per plane, a wave stores 3 x 256 contiguous bytes.
// The common case - all four pixels project inside the frame - is straight-line code: four maps, eight
// tap loads in flight together, packed bilinear.  Anything else (reflected taps, z <= 0, extreme
// exponents) takes the per-pixel generic path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_tiles_kernel(WarpParams P) {
    const WarpCam& c = P.cam[blockIdx.z];
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= c.tw || y >= c.th) return;
    {
        const unsigned lg = (unsigned)c.live_by0_gap;
        const int gap0 = (int)((lg >> 12) & 0x3ffu), by0 = (int)(lg & 0xfffu);
        if ((x0 >> 6) < c.live_bx0 || (x0 >> 6) > c.live_bx1 || (y >> 4) < by0 || (y >> 4) > c.live_by1) return;
        if ((x0 >> 6) >= gap0 && (x0 >> 6) < gap0 + (int)(lg >> 22)) return;   // the dead middle of a +-pi straddler
    }
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; i++) m[i] = c.m[i];
    const float2 B = c.rowB[y];
    // colA is padded to a multiple of 4 entries: two 16-byte loads
    const float4 a01 = *reinterpret_cast<const float4*>(c.colA + x0);
    const float4 a23 = *reinterpret_cast<const float4*>(c.colA + x0 + 2);
    const float2 A[4] = {make_float2(a01.x, a01.y), make_float2(a01.z, a01.w), make_float2(a23.x, a23.y),
                         make_float2(a23.z, a23.w)};
    const int sw = c.src_w, sh = c.src_h, stride = c.src_stride;
    const unsigned src_lo = (unsigned)(size_t)c.src & 3u;
    // mapBackward, in OpenCV's evaluation order: (m0*x_ + m1*y_) + m2*z_ ; the m1*y_ products are per row
    const float y_ = B.y, t1x = m[1] * y_, t1y = m[4] * y_, t1z = m[7] * y_;
    float X[4], Y[4], Z[4];
    bool fast = c.fe == nullptr;  // the straight-line path projects into the stitcher frame only
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float x_ = B.x * A[j].x, z_ = B.x * A[j].y;
        X[j] = (m[0] * x_ + t1x) + m[2] * z_;
        Y[j] = (m[3] * x_ + t1y) + m[5] * z_;
        Z[j] = (m[6] * x_ + t1z) + m[8] * z_;
        fast &= Z[j] >= 0x1p-40f && Z[j] <= 0x1p40f;
    }
    int ix[4], iy[4], fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float qx, qy;
        div2_shared(X[j], Y[j], Z[j], qx, qy);
        const int isx = cv_round_dev(qx * 32.f), isy = cv_round_dev(qy * 32.f);
        fa[j] = isx & 31; fb[j] = isy & 31;
        ix[j] = isx >> 5; iy[j] = isy >> 5;  // |value| < 2^26: the saturate_cast<short> is decided by the range test below
        // the 12-byte aligned fetch of row iy+1 must end inside the frame: ix <= sw-4
        fast &= ix[j] >= 0 && ix[j] <= sw - 4 && iy[j] >= 0 && iy[j] <= sh - 2;
    }
    int v[4][3];
    if (fast) {
        uint2 t[4], u[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned o = (unsigned)(iy[j] * stride + 3 * ix[j]);
            t[j] = load_taps6(c.src, src_lo, o);
            u[j] = load_taps6(c.src, src_lo, o + stride);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) bilinear_packed(t[j], u[j], fa[j], fb[j], v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float fx, fy;
            map_source(c, m, A[j], B, fx, fy);
            sample_bilinear_reflect(c.src, sw, sh, stride, fx, fy, v[j]);
        }
    }
    if (c.gain) {
#pragma unroll
        for (int j = 0; j < 4; j++) apply_gain(c, min(x0 + j, c.tw - 1), y, v[j]);
    }
    uint8_t* d = (uint8_t*)c.dst + (size_t)y * c.dst_pitch + x0;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        unsigned pk = (unsigned)v[0][ch] | ((unsigned)v[1][ch] << 8) | ((unsigned)v[2][ch] << 16) | ((unsigned)v[3][ch] << 24);
        *reinterpret_cast<unsigned*>(d + (size_t)ch * c.dst_plane) = pk;  // rows are padded to 16 bytes
    }
}

// ------------------------------------------------------------------------------------------------
// K1, table form.  K, R and the tile geometry are fixed after pano_prepare, so everything in front of the
// tap fetch - mapBackward, the 1/32-pixel quantisation of cv::remap, saturate_cast<short> and the
// BORDER_REFLECT resolution of both taps on both axes - is a per-pixel constant.  It is folded into one
// dword per tile pixel, with the same arithmetic as the on-the-fly kernel:
//   bits  0..15  xs * 32 + a'     bits 16..31  ys * 32 + b'
// (xs, ys) = the smaller reflected tap index, a'/b' = weight of tap xs+1 / row ys+1 in 1/32:
//   taps increasing (x1 == x0 + 1): a' = a;  mirrored (x1 == x0 - 1): a' = 32 - a;  same pixel: a' = 0;
//   a' == 32 is stored as (xs + 1, 0), which weighs the same pixel.
// xs, ys are relative to the origin of the source box of the pixel's 64 x 16 workgroup (build_warp_table_kernel), so 11
// bits per axis serve frames of any size.  Every pixel has a code; 0xffffffff does not occur.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lut_axis(int i0, int n, int frac, int& base, int& w1) {
    // BORDER_REFLECT maps neighbouring indices to neighbouring or equal ones: r1 - r0 is -1, 0 or +1
    const int r0 = reflect_idx(i0, n), r1 = reflect_idx(i0 + 1, n);
    if (r1 == r0 + 1) { base = r0; w1 = frac; }
    else if (r1 == r0 - 1) { base = r1; w1 = 32 - frac; }
    else { base = r0; w1 = 0; }
    if (w1 == 32) { base += 1; w1 = 0; }
}
// Source boxes.  The table is static, so the set of frame pixels a 64 x 16 pixel workgroup of K1 taps is static too.
// Its bounding box does two jobs:
//   * the codes of the table are stored RELATIVE to the box origin (xs - xmin, ys - ymin): 11 bits per axis are then
//     enough for any frame size (a 64 x 16 patch never spans 2048 source pixels), so 4K frames take the table path;
//   * K1 copies the box into LDS and reads the taps there (see warp_tiles_lut_kernel).
// Box entry: {xmin, ymin, rows << 8 | 16-byte chunks per row, ceil(2^16 / chunks)}.  Boxes that do not fit kBoxBytes (far
// outside the frame, where BORDER_REFLECT folds pile up) or that would read past the last bytes of the frame get
// rows == 0 and tap global memory instead; the origin is valid either way.
constexpr int kBoxBytes = 16 * 1024;             // LDS per workgroup, one spare row included
constexpr int kBoxIters = kBoxBytes / 16 / 256;  // 16-byte chunk loads per lane, at most
__global__ __launch_bounds__(256) void build_warp_table_kernel(WarpCam c, uint32_t* lut, int lut_pitch, int4* boxes, int gx,
                                                               unsigned* counters /* [0] boxes without LDS, [1] spans too wide */) {
    __shared__ int lim[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid == 0) { lim[0] = INT_MAX; lim[1] = -1; lim[2] = INT_MAX; lim[3] = -1; }
    __syncthreads();
    const int x0 = (blockIdx.x * 16 + (threadIdx.x & 15)) * 4;
    const int y = blockIdx.y * 16 + threadIdx.y * 4 + (threadIdx.x >> 4);
    const bool in = x0 < lut_pitch && y < c.th;  // the pad columns (x >= tw, never read downstream) repeat the last pixel
    int xs[4], a1[4], ys[4], b1[4];
    if (in) {
        int xa = INT_MAX, xb = -1, ya = INT_MAX, yb = -1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float fx, fy;
            map_source(c, c.m, c.colA[min(x0 + j, c.tw - 1)], c.rowB[y], fx, fy);
            const int isx = cv_round_dev(fx * 32.f), isy = cv_round_dev(fy * 32.f);
            const int ix = sat16i(isx >> 5), iy = sat16i(isy >> 5);
            lut_axis(ix, c.src_w, isx & 31, xs[j], a1[j]);
            lut_axis(iy, c.src_h, isy & 31, ys[j], b1[j]);
            xa = min(xa, xs[j]); xb = max(xb, xs[j]);
            ya = min(ya, ys[j]); yb = max(yb, ys[j]);
        }
        atomicMin(&lim[0], xa); atomicMax(&lim[1], xb);
        atomicMin(&lim[2], ya); atomicMax(&lim[3], yb);
    }
    __syncthreads();
    const int xmin = lim[0], ymin = lim[2];
    if (in) {
        // Every pixel has a code (0xffffffff cannot occur: a base on the last column or row of the frame carries weight 0).
        // What the table does NOT promise is that an aligned 12-byte fetch at (xs, ys) stays inside the frame: the
        // kernels check that themselves for the last bytes of the frame.
        unsigned code[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            code[j] = (uint32_t)((xs[j] - xmin) * 32 + a1[j]) | ((uint32_t)((ys[j] - ymin) * 32 + b1[j]) << 16);
        *reinterpret_cast<uint4*>(lut + (size_t)y * lut_pitch + x0) = make_uint4(code[0], code[1], code[2], code[3]);
    }
    if (tid != 0 || lim[1] < 0) return;
    const int sw = c.src_w, sh = c.src_h;
    if (lim[1] - xmin >= 2048 || lim[3] - ymin >= 2048) atomicAdd(&counters[1], 1u);  // 11 bits per axis do not hold this patch
    // rows ymin .. min(ymax + 1, sh - 1) are loaded; the taps of row ymax + 1 == sh (weight 0) read the spare row
    const int h = min(lim[3] + 1, sh - 1) - ymin + 1;
    // bytes 3*xmin .. 3*xmax+5 of each row, fetched from the enclosing 16-byte boundary (phase <= 15); the
    // realigning tap read touches up to 6 bytes more (the first bytes of the next row)
    const int cpr = (3 * (lim[1] - xmin) + 21 + 15) / 16;
    // LDS rows are packed at cpr * 16 bytes: a tap read may run a few bytes into the next row, the taps of the spare
    // row read whatever follows the box, and the last wave's copy rounds the box up to 64 chunks
    bool ok = (h + 1) * cpr * 16 + 16 <= kBoxBytes && cpr <= 63 && (h * cpr + 63) / 64 * 64 * 16 <= kBoxBytes;
    // the chunks of the last frame row must end inside the frame
    ok &= !(ymin + h - 1 == sh - 1 && 3 * xmin + cpr * 16 > 3 * sw);
    if (!ok) atomicAdd(&counters[0], 1u);
    boxes[blockIdx.y * gx + blockIdx.x] = ok ? make_int4(xmin, ymin, (h << 8) | cpr, (65536 + cpr - 1) / cpr) : make_int4(xmin, ymin, 0, 0);
}
void launch_build_warp_table(const WarpCam& c, uint32_t* lut, int lut_pitch, int4* boxes, unsigned* counters, hipStream_t s) {
    dim3 block(64, 4, 1), grid((lut_pitch + 63) / 64, (c.th + 15) / 16, 1);
    hipLaunchKernelGGL(build_warp_table_kernel, grid, block, 0, s, c, lut, lut_pitch, boxes, (c.tw + 63) / 64, counters);
}

// Packed table.  The map is smooth, so inside a 4-pixel group the steps between neighbouring codes are a group
// constant plus a rounding wobble: 8 bytes per group (2 per pixel) instead of 16.
//   word 0         the code of pixel 0 (format above)
//   word 1  0.. 7  Db  signed 8   x step base (negative in mirrored BORDER_REFLECT regions)
//           8..13  Eb  signed 6   y step base
//          14..31  three fields {cx signed 3, cy signed 3} for pixels 1..3:
//                  X[j] = X[j-1] + Db + cx[j],  Y[j] = Y[j-1] + Eb + cy[j]     (X = xs*32+a', Y = ys*32+b')
// Exact or not at all: a group whose steps do not fit (a reflect fold inside the group, > 4x
// magnification) stores word 0 = 0xffffffff and K1 reads its four codes from the dense table instead (on the 1080p rig
// about one group in 300; `flags` marks the 64 x 16 pixel workgroups that hold one, for the statistics).
__global__ __launch_bounds__(256) void pack_warp_lut_kernel(const uint32_t* lut, int lut_pitch, int tw, int th, uint2* lutc,
                                                            int lutc_pitch, uint32_t* flags, int gx) {
    const int g = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (g >= lutc_pitch || y >= th) return;
    const uint4 m = *reinterpret_cast<const uint4*>(lut + (size_t)y * lut_pitch + 4 * g);
    const unsigned code[4] = {m.x, m.y, m.z, m.w};
    bool ok = true;
    int dx[3], dy[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        dx[j] = (int)(code[j + 1] & 0xffffu) - (int)(code[j] & 0xffffu);
        dy[j] = (int)(code[j + 1] >> 16) - (int)(code[j] >> 16);
    }
    const int Db = min(dx[0], min(dx[1], dx[2])) + 4, Eb = min(dy[0], min(dy[1], dy[2])) + 4;
    ok &= Db >= -128 && Db <= 127 && Eb >= -32 && Eb <= 31;
    unsigned w1 = ((unsigned)Db & 0xffu) | (((unsigned)Eb & 0x3fu) << 8);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int cx = dx[j] - Db, cy = dy[j] - Eb;
        ok &= cx >= -4 && cx <= 3 && cy >= -4 && cy <= 3;
        w1 |= (((unsigned)cx & 7u) << (14 + 6 * j)) | (((unsigned)cy & 7u) << (17 + 6 * j));
    }
    lutc[(size_t)y * lutc_pitch + g] = ok ? make_uint2(code[0], w1) : make_uint2(0xffffffffu, 0u);
    if (!ok && 4 * g < tw) flags[(y >> 4) * gx + (g >> 4)] = 1u;  // same value from every writer
}
void launch_pack_warp_lut(const uint32_t* lut, int lut_pitch, int tw, int th, uint2* lutc, int lutc_pitch, uint32_t* flags,
                          hipStream_t s) {
    dim3 block(64, 4, 1), grid((lutc_pitch + 63) / 64, (th + 3) / 4, 1);
    hipLaunchKernelGGL(pack_warp_lut_kernel, grid, block, 0, s, lut, lut_pitch, tw, th, lutc, lutc_pitch, flags,
                       (tw + 63) / 64);
}
__device__ __forceinline__ int sbits(unsigned w, int off, int n) { return (int)(w << (32 - off - n)) >> (32 - n); }

// v_pk_mul_lo_u16 / v_pk_mad_u16 with the SAME half of the weight register feeding both 16-bit lanes (op_sel), so a
// weight pair (32-a) | a << 16 serves both products without being splatted first.
__device__ __forceinline__ unsigned pk_mul_whi(unsigned x, unsigned w) {
    unsigned d;
    asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(x), "v"(w));
    return d;
}
__device__ __forceinline__ unsigned pk_mad_wlo(unsigned x, unsigned w, unsigned acc) {
    unsigned d;
    asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(x), "v"(w), "v"(acc));
    return d;
}
// The bilinear of one pixel, result left in byte 2 of r[c] (bits 16..23): the vertical weights carry a factor 64, so
//   r = ((32-b)*64*h_top + b*64*h_bot + 512*64)  =  ((sum + 512) >> 10) << 16  +  (low 16 bits of no interest)
// and four pixels are packed into a plane dword with byte permutes instead of shifts.
__device__ __forceinline__ void bilinear_b2(uint2 t, uint2 u, unsigned a, unsigned b, unsigned r[3]) {
    const unsigned WA = a * 0xffffu + 32u;        // (32-a) | a << 16
    const unsigned WB = b * 0x3fffc0u + 2048u;    // (32-b)*64 | b*64 << 16
    const unsigned l0 = __builtin_amdgcn_perm(u.x, t.x, 0x0c040c00u), r0 = __builtin_amdgcn_perm(u.x, t.x, 0x0c070c03u);
    const unsigned l1 = __builtin_amdgcn_perm(u.x, t.x, 0x0c050c01u), r1 = __builtin_amdgcn_perm(u.y, t.y, 0x0c040c00u);
    const unsigned l2 = __builtin_amdgcn_perm(u.x, t.x, 0x0c060c02u), r2 = __builtin_amdgcn_perm(u.y, t.y, 0x0c050c01u);
    const unsigned h0 = pk_mad_wlo(l0, WA, pk_mul_whi(r0, WA));
    const unsigned h1 = pk_mad_wlo(l1, WA, pk_mul_whi(r1, WA));
    const unsigned h2 = pk_mad_wlo(l2, WA, pk_mul_whi(r2, WA));
    const us2_t wb = __builtin_bit_cast(us2_t, WB);
    r[0] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, h0), wb, 32768u, false);
    r[1] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, h1), wb, 32768u, false);
    r[2] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, h2), wb, 32768u, false);
}

// The per-lane body of the general kernel: frames or strides of any alignment, dense table, global taps.
__device__ __forceinline__ void warp_lane_checked(const WarpCam& c, int x0, int y, uint4 mm, int ox, int oy, int v[4][3]) {
    const int stride = c.src_stride, sh1 = c.src_h - 1;
    const unsigned src_lo = (unsigned)(size_t)c.src & 3u;
    const unsigned last = (unsigned)(sh1 * stride + 3 * c.src_w - 1);  // offset of the last byte of the frame
    const unsigned code[4] = {mm.x, mm.y, mm.z, mm.w};
    uint2 t[4], u[4];
    int fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const unsigned mx = code[j] & 0xffffu, my = code[j] >> 16;
        fa[j] = mx & 31; fb[j] = my & 31;
        const int xs = ox + (int)(mx >> 5), ys = oy + (int)(my >> 5), ys1 = min(ys + 1, sh1);  // codes are relative to the box origin
        const unsigned ot = (unsigned)(ys * stride) + 3 * xs, ou = (unsigned)(ys1 * stride) + 3 * xs;
        // the aligned 12-byte fetch starts up to 3 bytes before the tap and must end inside the frame
        t[j] = ot + 12 <= last + 1 ? load_taps6(c.src, src_lo, ot) : taps6_bytes(c.src, ot, last);
        u[j] = ou + 12 <= last + 1 ? load_taps6(c.src, src_lo, ou) : taps6_bytes(c.src, ou, last);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) bilinear_packed(t[j], u[j], fa[j], fb[j], v[j]);
}

// K1, table form, the general kernel: any frame alignment, exposure gains, dense table, global taps.
__global__ __launch_bounds__(256) void warp_tiles_lut_checked_kernel(WarpParams P) {
    const WarpCam& c = P.cam[blockIdx.x];
    const unsigned lg = (unsigned)c.live_by0_gap;
    int bx = c.live_bx0 + (int)blockIdx.y;  // the grid is laid over the live blocks ...
    const int by = (int)(lg & 0xfffu) + (int)blockIdx.z;
    if (bx >= (int)((lg >> 12) & 0x3ffu)) bx += (int)(lg >> 22);  // ... minus the dead middle of a +-pi straddler
    if (bx > c.live_bx1 || by > c.live_by1) return;
    const int x0 = (bx * 16 + (threadIdx.x & 15)) * 4;
    const int y = by * 16 + threadIdx.y * 4 + (threadIdx.x >> 4);
    if (x0 >= c.tw || y >= c.th) return;
    const uint4 mm = *reinterpret_cast<const uint4*>(c.lut + (size_t)y * c.lut_pitch + x0);
    const int4 bb = c.box[by * ((c.tw + 63) >> 6) + bx];
    int v[4][3];
    warp_lane_checked(c, x0, y, mm, bb.x, bb.y, v);
    if (c.gain) {
#pragma unroll
        for (int j = 0; j < 4; j++) apply_gain(c, min(x0 + j, c.tw - 1), y, v[j]);
    }
    uint8_t* d = (uint8_t*)c.dst + (size_t)y * c.dst_pitch + x0;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const unsigned pk = (unsigned)v[0][ch] | ((unsigned)v[1][ch] << 8) | ((unsigned)v[2][ch] << 16) | ((unsigned)v[3][ch] << 24);
        *reinterpret_cast<unsigned*>(d + (size_t)ch * c.dst_plane) = pk;  // rows are padded to 16 bytes
    }
}

// K1, table form: one lane = 4 pixels of one tile row, one workgroup = a 64 x 16 pixel patch (4 waves of 64 x 4).
// What bounds it (measured on the 8-camera launch, SQ counters + ablations):
//   * not bytes: halving the table (packed form) or serving every tap from one L1-resident corner of the frame
//     changes nothing;
//   * the texture-address unit: a 64-lane gather costs the same ~19 cycles per CU whether it fetches 1, 2, 3 or
//     4 dwords per lane, and eight tap gathers per wave (2 rows x 4 pixels) were 16 of the kernel's 33 us;
//   * VALU issue: 223 instructions per wave kept the vector ALUs 70 % busy.
// Tried on top of this and rejected: walking 4 patches per workgroup with the next box loaded straight into a second
// LDS buffer (global_load_lds_dwordx4) while the current one is computed - exact, 35 us instead of 31 (two buffers
// and 89 VGPRs leave 5 waves per SIMD); 8-byte ds reads at odd addresses instead of 3 dwords + v_alignbyte - 55 us.
// So the taps do not come from global memory: the static table fixes the set of frame bytes a workgroup touches (its
// source box, see build_warp_boxes_kernel); the workgroup copies that box into LDS with coalesced 16-byte
// direct-to-LDS loads (global_load_lds_dwordx4: 1 to 2 gathers per wave instead of 8, no staging registers), issued
// together with the table load, and reads the taps from LDS.
// Workgroups whose box does not fit keep the global taps.  Needs 4-byte aligned frames and strides % 16 == 0
// (checked by the launcher; anything else runs warp_tiles_lut_checked_kernel).
// GAIN: an instantiation of its own that also applies the exposure gain maps (BlocksGainCompensator::apply) of the
// cameras that carry one; the plain instantiation stays at 37 VGPRs.
template <bool GAIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void warp_tiles_lut_kernel(WarpParams P) {
    __shared__ uint4 sbox[kBoxBytes / 16];
    // grid = (ncam, ceil(tw/64), ceil(th/16)): the camera is the FASTEST workgroup coordinate.  Linear workgroup ids
    // are dealt round-robin over the 8 XCDs (each with its own L2), so with 8 (or 4, 2) cameras an XCD's L2 only ever
    // holds one camera's frame, and every camera still advances top to bottom with all XCDs busy.  Measured on the
    // 8-camera launch against camera-major dispatch order: 33.1 vs 36.0 us, FETCH_SIZE 50.8 vs 77.8 MB.
    // Tried and rejected: XCD-aware order of the blocks WITHIN a camera (fewer fetched bytes, 15 % slower).
    // Prologue: an empty body over this grid (53 K workgroups) costs 9.4 us when the camera block is read field by
    // field behind branches - a chain of dependent scalar loads per wave, which the compiler is free to build by
    // sinking kernarg loads below the bounds test.  Fetch the 64-byte hot part with ONE s_load_dwordx16.
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
    struct Hot {
        const uint8_t* src; uint8_t* dst; const uint2* lutc; const int4* box;
        int tw, th, live_bx0, live_by0_gap, src_stride, dst_pitch, dst_plane, lutc_pitch;
    };
    static_assert(sizeof(Hot) == 64 && offsetof(WarpCam, lutc_pitch) == 60 && offsetof(WarpCam, src) == 0 &&
                      offsetof(WarpCam, box) == 24 && offsetof(WarpCam, live_bx0) == 40, "hot part layout");
    union { i32x16 v; Hot h; } hot;
    // WarpParams is the only kernel argument: P.cam[i] sits at kernarg + i * sizeof(WarpCam)
    const char __attribute__((address_space(4)))* ka =
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.x * sizeof(WarpCam);
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hot.v) : "s"(ka) : "memory");
    // the camera block as a plain pointer for the rare out-of-line paths (a reference to the by-value kernel argument
    // would make the compiler copy all of WarpParams to scratch)
    const WarpCam* const cg = (const WarpCam*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.x * sizeof(WarpCam));
    // the pointers come out of the asm block without an address space: say "global", or every access is a flat_load
#define PANO_GLOBAL __attribute__((address_space(1)))
    const uint8_t PANO_GLOBAL* const src = (const uint8_t PANO_GLOBAL*)hot.h.src;
    uint8_t PANO_GLOBAL* const dst = (uint8_t PANO_GLOBAL*)hot.h.dst;
    const u32x2 PANO_GLOBAL* const lutc = (const u32x2 PANO_GLOBAL*)hot.h.lutc;
    const int tw = hot.h.tw, th = hot.h.th;
    // the grid is laid over the live blocks of the camera: workgroup (0, 0) is block (live_bx0, live_by0); the dead
    // block columns in the middle of a +-pi straddler (gap_len from gap_bx0 on, else 0) are stepped over
    const unsigned lg = (unsigned)hot.h.live_by0_gap;
    int bx = hot.h.live_bx0 + (int)blockIdx.y;
    const int by = (int)(lg & 0xfffu) + (int)blockIdx.z;
    if (bx >= (int)((lg >> 12) & 0x3ffu)) bx += (int)(lg >> 22);
    const unsigned stride = (unsigned)hot.h.src_stride;
    const unsigned dst_pitch = (unsigned)hot.h.dst_pitch, dst_plane = (unsigned)hot.h.dst_plane, lutc_pitch = (unsigned)hot.h.lutc_pitch;
    const int gxc = (tw + 63) >> 6;
    if (bx >= gxc || by * 16 >= th) return;  // the whole workgroup leaves: nobody waits at the barrier below
    // the blocks anything downstream reads (4 ints behind the hot part) and this workgroup's source box: two more
    // scalar loads, issued together
    i32x4 live;  // {live_bx1, live_by1, src_w, src_h}
    asm volatile("s_load_dwordx4 %0, %1, 0x40" : "=s"(live) : "s"(ka) : "memory");
    static_assert(offsetof(WarpCam, live_bx1) == 64 && offsetof(WarpCam, src_h) == 76, "second scalar load layout");
    i32x4 bb;
    {
        const char __attribute__((address_space(4)))* bp =
            (const char __attribute__((address_space(4)))*)hot.h.box + (unsigned)(by * gxc + bx) * 16u;
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(bb) : "s"(bp) : "memory");
    }
    // Wave shape: 16 lanes x 4 rows = a 64 x 4 pixel patch, not a 256-pixel strip.  Where the projection tilts rows
    // (towards the tile edges) a long strip drags in dozens of source rows; compact patches keep the box small.
    // Measured per 4-camera launch (global taps): 256x1 21.7 us, 128x2 19.2, 64x4 19.2, 32x8 21.4, 16x16 36.6.
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int x0 = (bx * 16 + (threadIdx.x & 15)) * 4;
    const int y = by * 16 + threadIdx.y * 4 + (threadIdx.x >> 4);
    const bool active = x0 < tw && y < th;
    // plain (cached) loads and stores: non-temporal ones for the streamed table and tile measured 15 % slower
    u32x2 e = u32x2{0u, 0u};
    if (active)  // 32-bit byte offset: scalar base + vector offset addressing, no 64-bit multiply
        e = *reinterpret_cast<const u32x2 PANO_GLOBAL*>((const uint8_t PANO_GLOBAL*)lutc + (((unsigned)y * lutc_pitch + (unsigned)(x0 >> 2)) << 3));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(bb), "+s"(live) : : "memory");  // bb, live are only valid past this point
    if (bx > live.x || by > live.y) return;  // beyond this camera's live blocks (workgroup-uniform)
    const int src_w = live.z, src_h = live.w;
    const int bh = bb.z >> 8, cpr = bb.z & 255;
    const unsigned lpitch = (unsigned)cpr * 16u;  // rows packed: chunk k = r * cpr + ci lands at LDS byte 16 * k
    // byte phase of the box origin inside its first 16-byte chunk; the same for every row because stride % 16 == 0
    const unsigned lo16 = (unsigned)(size_t)hot.h.src & 15u;
    const unsigned og = (unsigned)bb.y * stride + 3u * (unsigned)bb.x + lo16;  // from the 16-byte boundary below src
    const unsigned ph = og & 15u;
    if (bh) {
        // chunk k = tid + 256 * it -> (row r = k / cpr, column ci = k % cpr), copied by global_load_lds_dwordx4: the 64
        // lanes of a wave write 64 consecutive 16-byte chunks at M0 - no staging registers, no ds_write.  Whole
        // iterations a wave does not reach are skipped by a scalar branch; the surplus lanes of its last one re-read the
        // last chunk and land behind the box, inside the buffer.
        const int total = bh * cpr;
        const uint8_t PANO_GLOBAL* const srca = (src - lo16) + (og - ph);  // 16-byte aligned
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
#pragma unroll
        for (int it = 0; it < kBoxIters; it++) {
            if (wv * 64 + 256 * it < total) {
                const unsigned k = (unsigned)min(tid + 256 * it, total - 1);
                const unsigned r = __umul24(k, (unsigned)bb.w) >> 16, ci = k - __umul24(r, (unsigned)cpr);
                __builtin_amdgcn_global_load_lds(
                    srca + (__umul24(r, stride) + ci * 16u),
                    (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) uint8_t*)&sbox[0] + (wv * 64 + 256 * it) * 16),
                    16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!active) return;
    uint8_t PANO_GLOBAL* d = dst + ((unsigned)y * dst_pitch + (unsigned)x0);  // 32-bit offsets: a tile is far below 4 GB
    unsigned X[4], Y[4];
    {
        const int Db = sbits(e.y, 0, 8), Eb = sbits(e.y, 8, 6);
        X[0] = e.x & 0xffffu; Y[0] = e.x >> 16;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            X[j + 1] = X[j] + (unsigned)(Db + sbits(e.y, 14 + 6 * j, 3));
            Y[j + 1] = Y[j] + (unsigned)(Eb + sbits(e.y, 17 + 6 * j, 3));
        }
    }
    // an escaped group (a BORDER_REFLECT fold inside it) reads its four codes from the dense table: one more dependent
    // load for the waves that hold one (about one in eight)
    if (e.x == 0xffffffffu) {
        const uint4 mm = *reinterpret_cast<const uint4*>(cg->lut + ((unsigned)y * (4u * lutc_pitch) + (unsigned)x0));
        const unsigned code[4] = {mm.x, mm.y, mm.z, mm.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            X[j] = code[j] & 0xffffu;
            Y[j] = code[j] >> 16;
        }
    }
    uint2 t[4], u[4];
    if (bh) {
        // LDS byte offset of a pixel: (ys - ymin) * lpitch + 3 * (xs - xmin) + ph.  Three aligned dwords and
        // v_alignbyte, like the global taps: 8-byte ds reads at odd addresses work but run the kernel at half speed.
        const unsigned* sb = reinterpret_cast<const unsigned*>(sbox);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned off = __umul24(Y[j] >> 5, lpitch) + __umul24(X[j] >> 5, 3u) + ph;  // codes are box relative
            const unsigned k = off & 3u;
            const unsigned* wt = sb + (off >> 2);
            const unsigned* wu = wt + (lpitch >> 2);  // row ys + 1; for ys == sh - 1 (weight 0) the spare row
            t[j] = make_uint2(__builtin_amdgcn_alignbyte(wt[1], wt[0], k), __builtin_amdgcn_alignbyte(wt[2], wt[1], k));
            u[j] = make_uint2(__builtin_amdgcn_alignbyte(wu[1], wu[0], k), __builtin_amdgcn_alignbyte(wu[2], wu[1], k));
        }
    } else {
        // Global taps (the box of this patch does not fit LDS or touches the end of the frame).  `o_last` is the last
        // aligned offset a 12-byte fetch may start at; a pixel whose row ys + 1 fetch would start beyond it - the last
        // pixels of the last rows, or row ys + 1 == src_h, which carries weight 0 - reads its taps byte by byte.
        const unsigned last = (unsigned)(src_h - 1) * stride + 3u * (unsigned)src_w - 1u;
        const unsigned o_last = last >= 11u ? (last + 1u - 12u) & ~3u : 0u;  // a frame of < 12 bytes has no such offset
        const unsigned sh1 = (unsigned)(src_h - 1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned ys = (Y[j] >> 5) + (unsigned)bb.y;
            const unsigned ot = __umul24(ys, stride) + __umul24((X[j] >> 5) + (unsigned)bb.x, 3u);
            const unsigned k = ot & 3u, oa = ot & ~3u;
            if (o_last != 0u && oa + stride <= o_last) {
                const u32x3 dt = *reinterpret_cast<const u32x3 PANO_GLOBAL*>(src + oa);
                const u32x3 du = *reinterpret_cast<const u32x3 PANO_GLOBAL*>(src + oa + stride);
                t[j] = make_uint2(__builtin_amdgcn_alignbyte(dt.y, dt.x, k), __builtin_amdgcn_alignbyte(dt.z, dt.y, k));
                u[j] = make_uint2(__builtin_amdgcn_alignbyte(du.y, du.x, k), __builtin_amdgcn_alignbyte(du.z, du.y, k));
            } else {
                t[j] = taps6_bytes(src, ot, last);
                u[j] = taps6_bytes(src, ot + (ys < sh1 ? stride : 0u), last);
            }
        }
    }
    unsigned r[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) bilinear_b2(t[j], u[j], X[j] & 31u, Y[j] & 31u, r[j]);
    unsigned pk[3];
    if (GAIN && cg->gain != nullptr) {  // per camera: workgroup-uniform
        // the gains of the lane's four pixels: two 16-byte reads of the horizontally resized map rows, then the vertical pass
        const int2 gy = cg->grow[y];
        const float2 by = cg->groww[y];
        const int gp = cg->ghrow_pitch;
        const float4 h0 = *reinterpret_cast<const float4*>(cg->ghrow + (unsigned)(gy.x * gp + x0));
        const float4 h1 = *reinterpret_cast<const float4*>(cg->ghrow + (unsigned)(gy.y * gp + x0));
        const float g[4] = {h0.x * by.x + h1.x * by.y, h0.y * by.x + h1.y * by.y, h0.z * by.x + h1.z * by.y, h0.w * by.x + h1.w * by.y};
        // saturate_cast<uchar>(px * gain) = round-half-even + clamp is what v_cvt_pk_u8_f32 does, and it drops the byte
        // where the plane dword wants it: three instructions per value (v_cvt_f32_ubyte2, v_mul_f32, v_cvt_pk_u8_f32)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            pk[ch] = 0u;
#pragma unroll
            for (int j = 0; j < 4; j++)
                pk[ch] = __builtin_amdgcn_cvt_pk_u8_f32((float)((r[j][ch] >> 16) & 0xffu) * g[j], (unsigned)j, pk[ch]);
        }
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            // byte 2 of r[0..3][ch] -> bytes 0..3
            const unsigned lo = __builtin_amdgcn_perm(r[1][ch], r[0][ch], 0x0c0c0602u);
            const unsigned hi = __builtin_amdgcn_perm(r[3][ch], r[2][ch], 0x06020c0cu);
            pk[ch] = lo | hi;
        }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        *reinterpret_cast<unsigned PANO_GLOBAL*>(d + (unsigned)ch * dst_plane) = pk[ch];  // rows are padded to 16 bytes
    }
}

void launch_warp_tiles(const WarpParams& p, int ncam, int max_tw, int max_th, hipStream_t s, hipEvent_t ev_start,
                       hipEvent_t ev_stop) {
    dim3 block(64, 4, 1);
    dim3 grid((max_tw + 255) / 256, (max_th + 3) / 4, ncam);      // projecting kernel: 256 x 4 pixel blocks
    // table kernels: 64 x 16 pixel workgroups laid over the live blocks of each camera, camera fastest
    int lbx = 0, lby = 0;
    for (int i = 0; i < ncam; i++) {
        const WarpCam& c = p.cam[i];
        const int nbx = (c.tw + 63) / 64, nby = (c.th + 15) / 16;
        lbx = max(lbx, min(c.live_bx1, nbx - 1) - c.live_bx0 + 1 - (int)((unsigned)c.live_by0_gap >> 22));
        lby = max(lby, min(c.live_by1, nby - 1) - (c.live_by0_gap & 0xfff) + 1);
    }
    lbx = max(lbx, 1); lby = max(lby, 1);  // nothing live (empty masks): one workgroup that leaves at once keeps the events valid
    const dim3 grid_lut(ncam, lbx, lby);
    // Tried and rejected on the blocks WITHIN a camera (A/B in one process, same outputs): (1) an XCD-aware block order
    // and (2) padding the column blocks to a multiple of 8 so that each XCD owns a 256-pixel column stripe.  Both cut the
    // fetched bytes to the minimum and both ran slower (23.8 / 27.1 us vs 21.8 us per 4-camera launch): concentrating
    // an XCD on a narrow address range loses more in channel spread than the L2 reuse gains.  What ships is one camera
    // per XCD (the camera is the fastest grid coordinate), which gets the same minimum without that cost.
    // the table form needs every camera of the launch to carry a table
    bool all_lut = true;
    for (int i = 0; i < ncam; i++) all_lut &= p.cam[i].lut != nullptr;
#define PANO_LAUNCH_K1(K, G)                                                                   \
    do {                                                                                       \
        if (ev_start && ev_stop) hipExtLaunchKernelGGL(K, G, block, 0, s, ev_start, ev_stop, 0, p); \
        else hipLaunchKernelGGL(K, G, block, 0, s, p);                                        \
    } while (0)
    if (all_lut) {
        // the LDS kernel wants 4-byte aligned frames and strides % 16 == 0; anything else takes the general kernel (same
        // table, global taps, per-pixel checked body)
        bool fast = true;
        for (int i = 0; i < ncam; i++)
            fast &= ((size_t)p.cam[i].src & 3u) == 0 && (p.cam[i].src_stride & 15) == 0 && p.cam[i].lutc != nullptr &&
                    p.cam[i].box != nullptr;
        bool gains = false;
        for (int i = 0; i < ncam; i++) gains |= p.cam[i].gain != nullptr;
        if (fast && gains) PANO_LAUNCH_K1(warp_tiles_lut_kernel<true>, grid_lut);
        else if (fast) PANO_LAUNCH_K1(warp_tiles_lut_kernel<false>, grid_lut);
        else PANO_LAUNCH_K1(warp_tiles_lut_checked_kernel, grid_lut);
    }
    else {
        PANO_LAUNCH_K1(warp_tiles_kernel, grid);
    }
#undef PANO_LAUNCH_K1
}

// stage entry: RotationWarper::warp to an 8UC3 image (no border, byte pitch)
__global__ __launch_bounds__(256) void warp_image_kernel(WarpCam c) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= c.tw || y >= c.th) return;
    float fx, fy;
    int v[3];
    map_source(c, c.m, c.colA[x], c.rowB[y], fx, fy);
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
    dst[(size_t)y * dst_stride + x] = ((unsigned)sx < (unsigned)c.out_w && (unsigned)sy < (unsigned)c.out_h) ? 255 : 0;
}
void launch_warp_mask(const WarpCam& c, uint8_t* dst, int dst_stride, hipStream_t s) {
    dim3 block(64, 4, 1), grid((c.tw + 63) / 64, (c.th + 3) / 4, 1);
    hipLaunchKernelGGL(warp_mask_kernel, grid, block, 0, s, c, dst, dst_stride);
}

// ------------------------------------------------------------------------------------------------
// K2: pyrDown (cv::pyrDown CV_16S semantics: 5x5 [1 4 6 4 1]^2, REFLECT_101, (v+128)>>8) on planar u8.
// One thread = 4 x 2 output pixels of one plane: 7 input rows x one 16-byte load, the horizontal 5-tap as
// v_dot4_u32_u8 on byte windows, the vertical pass in registers.  grid.z = camera * 3 + plane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pyr_down_hrow(const uint4 q, int h[4]) {
    // q = 16 bytes starting at column 8t-4; the four outputs are centred on columns 8t, 8t+2, 8t+4, 8t+6, i.e. on
    // bytes 4, 6, 8, 10 of the window.  Each 5-tap window (1 4 6 4 1) straddles two dwords: two chained dot products
    // with the taps placed on the right bytes - no realignment, no byte extraction.
    h[0] = (int)__builtin_amdgcn_udot4(q.y, 0x00010406u, __builtin_amdgcn_udot4(q.x, 0x04010000u, 0u, false), false);  // bytes 2..6
    h[1] = (int)__builtin_amdgcn_udot4(q.z, 0x00000001u, __builtin_amdgcn_udot4(q.y, 0x04060401u, 0u, false), false);  // bytes 4..8
    h[2] = (int)__builtin_amdgcn_udot4(q.z, 0x00010406u, __builtin_amdgcn_udot4(q.y, 0x04010000u, 0u, false), false);  // bytes 6..10
    h[3] = (int)__builtin_amdgcn_udot4(q.w, 0x00000001u, __builtin_amdgcn_udot4(q.z, 0x04060401u, 0u, false), false);  // bytes 8..12
}
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrParams P, unsigned cam_bits, int l) {
    const int ci = blockIdx.z / 3, pl = blockIdx.z - ci * 3;
    if (!((cam_bits >> ci) & 1u)) return;
    const PyrCam& c = P.cam[ci];
    const int sw = c.w0 >> l, sh = c.h0 >> l;
    const int dw = sw >> 1, dh = sh >> 1;
    // Outputs of level l + 1 that nothing downstream reads are not produced (their inputs may not exist either): the
    // grid is laid over the live rect, so that whole waves - not lanes - fall off its far side.
    const int lx0 = c.live[l + 1][0], ly0 = c.live[l + 1][1], lx1 = c.live[l + 1][2], ly1 = c.live[l + 1][3];
    const int t = (lx0 >> 2) + blockIdx.x * 64 + threadIdx.x;  // group of 4 output columns
    // a wave is one threadIdx.y: tell the compiler, and the row indices, the REFLECT_101 of the seven source rows and
    // their addresses are scalar work (a quarter of this kernel's vector instructions otherwise)
    const int y0 = ((ly0 >> 1) + blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * 2;  // pair of output rows
    if (y0 >= dh || y0 > ly1) return;
    if (t * 4 >= dw || t * 4 > lx1) return;
    if (t * 4 >= c.gap[l + 1][0] && t * 4 + 3 <= c.gap[l + 1][1]) return;  // the dead middle of a +-pi straddler's tile
    const uint8_t* __restrict__ src = c.lvl[l] + (size_t)pl * c.plane[l];
    const int sp = c.pitch[l];
    // Every lane does seven 16-byte loads (columns 8t-4 .. 8t+11; lane 0 loads columns 0..15 and shifts).
    // REFLECT_101 at the two row ends touches at most three bytes, patched in registers:
    //   left  (t == 0): columns -2, -1 are columns 2, 1
    //   right (8t+8 == sw, the last group): column sw is column sw-2
    int acc0[4] = {128, 128, 128, 128}, acc1[4] = {128, 128, 128, 128};  // the rounding term of (v + 128) >> 8
    const int off = t == 0 ? 0 : 8 * t - 4;
    uint4 q[7];
#pragma unroll
    for (int r = 0; r < 7; r++)
        q[r] = *reinterpret_cast<const uint4*>(src + ((unsigned)(reflect101_idx(2 * y0 - 2 + r, sh) * sp) + (unsigned)off));  // scalar row + lane offset
    if (t == 0) {
#pragma unroll
        for (int r = 0; r < 7; r++) {
            const unsigned a = q[r].x;
            q[r].w = q[r].z;
            q[r].z = q[r].y;
            q[r].y = a;
            // byte2 = column -2 = column 2 (column 0 when the row has only two), byte3 = column -1 = column 1
            q[r].x = (sw > 2 ? (a & 0x00ff0000u) : ((a & 0xffu) << 16)) | ((a & 0x0000ff00u) << 16);
        }
    }
    const int ksw = sw - (8 * t - 4);  // byte position of column sw in the 16-byte window
    if (ksw <= 12) {
        // ksw is even (sw and 8t-4 are even) and >= 6: the source byte ksw-2 sits in the same or the previous dword
#pragma unroll
        for (int r = 0; r < 7; r++) {
            unsigned d[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
            const int ks = ksw - 2;
            unsigned v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((ks >> 2) == j) v = (d[j] >> (8 * (ks & 3))) & 0xffu;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((ksw >> 2) == j) d[j] = (d[j] & ~(0xffu << (8 * (ksw & 3)))) | (v << (8 * (ksw & 3)));
            q[r] = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 7; r++) {
        int h[4];
        pyr_down_hrow(q[r], h);
        const int wa = r == 0 ? 1 : (r == 1 ? 4 : (r == 2 ? 6 : (r == 3 ? 4 : (r == 4 ? 1 : 0))));
        const int wb = r == 2 ? 1 : (r == 3 ? 4 : (r == 4 ? 6 : (r == 5 ? 4 : (r == 6 ? 1 : 0))));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc0[j] += h[j] * wa;
            acc1[j] += h[j] * wb;
        }
    }
    uint8_t* d = c.lvl[l + 1] + (size_t)pl * c.plane[l + 1] + (size_t)y0 * c.pitch[l + 1] + 4 * t;
    unsigned p0 = 0, p1 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // no saturate_cast: the taps sum to 256, so (256 * 255 + 128) >> 8 = 255 is the largest value there is
        p0 |= (unsigned)(acc0[j] >> 8) << (8 * j);
        p1 |= (unsigned)(acc1[j] >> 8) << (8 * j);
    }
    *reinterpret_cast<unsigned*>(d) = p0;  // rows are padded to 16 bytes
    if (y0 + 1 < dh) *reinterpret_cast<unsigned*>(d + c.pitch[l + 1]) = p1;
}

void launch_pyr_down(const PyrParams& p, unsigned cam_bits, int l, hipStream_t s) {
    int mw = 0, mh = 0;
    for (int i = 0; i < p.ncam; i++)
        if ((cam_bits >> i) & 1u) {
            mw = max(mw, p.cam[i].w0 >> (l + 1));
            mh = max(mh, p.cam[i].h0 >> (l + 1));
        }
    if (mw == 0 || mh == 0) return;
    dim3 block(64, 4, 1), grid((mw + 255) / 256, (mh + 7) / 8, p.ncam * 3);
    hipLaunchKernelGGL(pyr_down_kernel, grid, block, 0, s, p, cam_bits, l);
}

// ------------------------------------------------------------------------------------------------
// pyrUp (cv::pyrUp CV_16S semantics) sampled at one destination pixel (X, Y) of an exactly-2x plane:
// even: s[x-1] + 6 s[x] + s[x+1], odd: 4 (s[x] + s[x+1]); left/top reflect-101, right/bottom
// replicate; (v + 32) >> 6, saturate.  T = uint8_t (camera Gaussian planes) or int16_t (canvas planes).
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ int pyr_up_px(const T* __restrict__ S, int n, int m, int pitch, int X, int Y) {
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
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const T* r = S + (size_t)yi[j] * pitch;
        acc += ((int)r[xi[0]] * wx[0] + (int)r[xi[1]] * wx[1] + (int)r[xi[2]] * wx[2]) * wy[j];
    }
    return sat16i((acc + 32) >> 6);
}

// ------------------------------------------------------------------------------------------------
// K3 (generic form, any alignment): one level of the blend, one thread per canvas pixel.
//   acc  = sum over cameras in feed order of (short)(lap * w)          (wrapping short add)
//   lap  = sat16(G_l - pyrUp(G_{l+1}))  (top level: G_l)
//   W    = sum over cameras in feed order of w                          (dst_band_weights)
//   norm = (short)(acc / (W + 1e-5f))
//   out  = sat16(norm + pyrUp(out_{l+1}))                                (top level: norm)
// level 0 applies dst_mask (W0 > eps), convertTo(CV_8U) and the cut, and writes the panorama.
// Cameras whose weight is exactly 0 at the pixel add (short)(lap*0) = 0 and +0.f: they are skipped.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cam_weight(const PyrCam& c, int l, int x, int y) {
    if (l == 0) return (float)c.mask0[(size_t)y * c.pitch[0] + x] * (float)(1. / 255.);
    return c.wgt[l][(size_t)y * c.wpitch[l] + x];
}

__global__ __launch_bounds__(256) void blend_level_kernel(PyrParams P, CanvasSet CS, int l) {
    const CanvasParams& C = CS.c[blockIdx.z];
    const int cam_lo = C.cam_lo, cam_n = C.cam_n;
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
    // phase A: every load that does not depend on another load - all cameras' weights and the coarser
    // canvas level - is issued together (these levels are latency bound, not bandwidth bound)
    float wv[kCams];
#pragma unroll
    for (int i = 0; i < kCams; i++) {
        wv[i] = 0.f;
        if (i < cam_n) {
            const PyrCam& c = P.cam[cam_lo + i];
            const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
            if ((unsigned)x < (unsigned)(c.w0 >> l) && (unsigned)y < (unsigned)(c.h0 >> l)) wv[i] = cam_weight(c, l, x, y);
        }
    }
    int cup[3] = {0, 0, 0};
    if (l < C.bands) {
#pragma unroll
        for (int k = 0; k < 3; k++)
            cup[k] = pyr_up_px<int16_t>(C.img[l + 1] + (size_t)k * C.cplane[l + 1], cw >> 1, ch >> 1, C.cpitch[l + 1], X, Y);
    }
    // phase B: cameras with a non-zero weight (one in the interior, two or three on a seam)
    int acc[3] = {0, 0, 0};
    float W = 0.f;
#pragma unroll
    for (int i = 0; i < kCams; i++) {
        const float w = wv[i];
        if (w == 0.f) continue;
        const PyrCam& c = P.cam[cam_lo + i];
        const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
        const int tw = c.w0 >> l, th = c.h0 >> l;
        W += w;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint8_t* g = c.lvl[l] + (size_t)k * c.plane[l];
            int lap = g[(size_t)y * c.pitch[l] + x];
            if (l < C.bands)
                lap = sat16i(lap - pyr_up_px<uint8_t>(c.lvl[l + 1] + (size_t)k * c.plane[l + 1], tw >> 1, th >> 1,
                                                      c.pitch[l + 1], x, y));
            acc[k] = (int16_t)(acc[k] + (int16_t)(int)((float)lap * w));
        }
    }
    const float den = W + 1e-5f;
    int v[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // W == 1.0f: (short)(n / 1.00001f) == n - sign(n), see the vector kernel
        if (W == 1.0f) v[k] = toward_zero_by_one(acc[k]);
        else v[k] = (int16_t)(int)((float)acc[k] / den);
        if (l < C.bands) v[k] = sat16i(v[k] + cup[k]);
    }
    if (l > 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) C.img[l][(size_t)k * C.cplane[l] + (size_t)Y * C.cpitch[l] + X] = (int16_t)v[k];
    } else {
        const bool on = W > 1e-5f;
        uint8_t* d = C.out + (size_t)(Y - C.cut_y) * C.out_stride + 3 * (X - C.cut_x);
        d[0] = on ? (uint8_t)sat8i(v[0]) : 0;
        d[1] = on ? (uint8_t)sat8i(v[1]) : 0;
        d[2] = on ? (uint8_t)sat8i(v[2]) : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// K3 (vector form): one thread = 4 x 2 canvas pixels (X0 multiple of 4, Y0 even).  Used for levels where
// every tile origin/size is a multiple of 4 x 2 at that level (C.fast[l]), i.e. all but the two coarsest.
// Per camera and plane: 2 dword loads of G_l, 3 unaligned dword loads of G_{l+1} (the 4 x 3 coarse
// neighbourhood serves all eight pyrUp samples).  Same arithmetic as the generic form.
// ------------------------------------------------------------------------------------------------
// horizontally upsample 4 coarse samples p[0..3] (columns x-1 .. x+2) to fine X0..X0+3 (X0 = 2x)
__device__ __forceinline__ void up_h4(const int p[4], int o[4]) {
    o[0] = p[0] + 6 * p[1] + p[2];
    o[1] = 4 * (p[1] + p[2]);
    o[2] = p[1] + 6 * p[2] + p[3];
    o[3] = 4 * (p[2] + p[3]);
}
// the 4 x 3 coarse neighbourhood of a 4 x 2 fine block: rows y-1, y, y+1 (y = Y0/2), columns x-1 .. x+2, with
// pyrUp's border rule (left/top reflect-101, right/bottom replicate) applied.  Branch-free, so that a caller's
// loads can all be issued before the first is consumed: each row is ONE aligned fetch of a window that is
// clamped into the row (8 bytes of u8 / 12 bytes of int16); the border rule is a byte permutation (v_perm_b32)
// of that window.  The four samples of a row stay PACKED: q[r][0] = 4 bytes (u8) or q[r][0..1] = 2 x 2 shorts.
typedef short s2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ void load_coarse(const T* __restrict__ S, int n, int m, int pitch, int x, int y,
                                            unsigned q[3][2]) {
    const int yi[3] = {y > 0 ? y - 1 : (m > 1 ? 1 : 0), y, min(y + 1, m - 1)};
    const int xi[4] = {x > 0 ? x - 1 : (n > 1 ? 1 : 0), x, min(x + 1, n - 1), min(x + 2, n - 1)};
    const int base = min(max(x - 1, 0), max(n - 4, 0));
    // sample k is element xi[k] - base (0..3) of the 4-element window starting at `base`
    const unsigned sh0 = xi[0] - base, sh1 = xi[1] - base, sh2 = xi[2] - base, sh3 = xi[3] - base;
    if (sizeof(T) == 1) {
        const int ab = base & ~3;  // 4-byte aligned fetch of 8 bytes; the window starts at byte base - ab (0..3)
        const unsigned sel = sh0 | (sh1 << 8) | (sh2 << 16) | (sh3 << 24);
        uint2 d[3];
#pragma unroll
        for (int r = 0; r < 3; r++) d[r] = *reinterpret_cast<const uint2*>(S + (unsigned)(__mul24(yi[r], pitch) + ab));  // v_mul_lo_u32 is quarter rate
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const unsigned win = __builtin_amdgcn_alignbyte(d[r].y, d[r].x, (unsigned)(base - ab));
            q[r][0] = __builtin_amdgcn_perm(0u, win, sel);
            q[r][1] = 0;
        }
    } else {
        const int ab = base & ~1;  // even element index = 4-byte aligned fetch of 12 bytes; window at element base - ab (0..1)
        const unsigned bs = (unsigned)(base - ab) * 2u;  // 0 or 2 bytes
        // short k of the result = short sh[k] of the 4-short window {w1:w0}: byte selectors 2*sh, 2*sh+1
        const unsigned selA = (2 * sh0) | ((2 * sh0 + 1) << 8) | ((2 * sh1) << 16) | ((2 * sh1 + 1) << 24);
        const unsigned selB = (2 * sh2) | ((2 * sh2 + 1) << 8) | ((2 * sh3) << 16) | ((2 * sh3 + 1) << 24);
        uint3 d[3];
#pragma unroll
        for (int r = 0; r < 3; r++) d[r] = *reinterpret_cast<const uint3*>(S + (unsigned)(__mul24(yi[r], pitch) + ab));
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const unsigned w0 = __builtin_amdgcn_alignbyte(d[r].y, d[r].x, bs);
            const unsigned w1 = __builtin_amdgcn_alignbyte(d[r].z, d[r].y, bs);
            q[r][0] = __builtin_amdgcn_perm(w1, w0, selA);
            q[r][1] = __builtin_amdgcn_perm(w1, w0, selB);
        }
    }
}
// a . w for two packed int16 pairs, v_dot2_i32_i16 with an inline-constant 0 accumulator (w is a compile-time constant)
__device__ __forceinline__ int sdot2_from_zero(unsigned a, unsigned w) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "s"(w));
    return d;
}
// pyrUp of a 4 x 2 block from the packed 4 x 3 neighbourhood: up[0][..] = fine row Y0 (even), up[1][..] = row Y0+1.
// Horizontal pass per coarse row: (p0 + 6 p1 + p2, 4 (p1 + p2), p1 + 6 p2 + p3, 4 (p2 + p3)) as dot products
// (v_dot4_u32_u8 on the byte window / v_dot2_i32_i16 on the short pairs); vertical pass in 32-bit ints.
template <typename T>
__device__ __forceinline__ void up_block(const unsigned q[3][2], int up[2][4]) {
    int h[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        if (sizeof(T) == 1) {
            const unsigned w = q[r][0];
            h[r][0] = (int)__builtin_amdgcn_udot4(w, 0x00010601u, 0u, false);
            h[r][1] = (int)__builtin_amdgcn_udot4(w, 0x00040400u, 0u, false);
            h[r][2] = (int)__builtin_amdgcn_udot4(w, 0x01060100u, 0u, false);
            h[r][3] = (int)__builtin_amdgcn_udot4(w, 0x04040000u, 0u, false);
        } else {
            const s2_t A = __builtin_bit_cast(s2_t, q[r][0]), B = __builtin_bit_cast(s2_t, q[r][1]);
            const s2_t c16 = {1, 6}, c04 = {0, 4}, c01 = {0, 1};
            // the products that start a sum use the three-operand form with an inline 0: the builtin becomes v_dot2c,
            // whose accumulator is the destination, and costs a v_mov to zero it first
            h[r][0] = __builtin_amdgcn_sdot2(A, c16, (int)B.x, false);
            h[r][1] = __builtin_amdgcn_sdot2(A, c04, sdot2_from_zero(q[r][1], 0x00000004u), false);  // B . (4, 0)
            h[r][2] = __builtin_amdgcn_sdot2(A, c01, sdot2_from_zero(q[r][1], 0x00010006u), false);  // B . (6, 1)
            h[r][3] = sdot2_from_zero(q[r][1], 0x00040004u);                                         // B . (4, 4)
        }
    }
    // No saturate_cast here: it cannot trigger.  Camera planes are 8-bit (h <= 8*255), and a collapsed canvas level
    // is bounded by 255 per remaining level (|norm_l| <= 255, pyrUp is a convex combination + rounding), i.e.
    // |out_l| <= 9*255 + 9 for the maximum of 8 bands - far inside int16.
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int h1 = h[1][k];
        up[0][k] = (h[0][k] + h[2][k] + (h1 << 2) + (h1 << 1) + 32) >> 6;
        up[1][k] = (h1 + h[2][k] + 8) >> 4;
    }
}

// store a finished 4 x 2 block: canvas level (planar int16) or, at level 0, dst_mask + convertTo(8U) + cut
// ALLON: every pixel of the block carries weight (dst_mask set) - the caller's guarantee, no per-pixel select
template <bool L0, bool ALLON = false, int NPL = 3>
__device__ __forceinline__ void store_block(const CanvasParams& C, int l, int X0, int Y0, const int v[3][2][4], bool o00,
                                            bool o01, bool o02, bool o03, bool o10, bool o11, bool o12, bool o13, int pb = 0) {
    if (!L0) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                uint2 pk;
                pk.x = ((unsigned)v[pl][r][0] & 0xffffu) | ((unsigned)v[pl][r][1] << 16);
                pk.y = ((unsigned)v[pl][r][2] & 0xffffu) | ((unsigned)v[pl][r][3] << 16);
                *reinterpret_cast<uint2*>(C.img[l] + (size_t)(pb + pl) * C.cplane[l] + (unsigned)(__mul24(Y0 + r, C.cpitch[l]) + X0)) = pk;
            }
    } else {
        const bool on[2][4] = {{o00, o01, o02, o03}, {o10, o11, o12, o13}};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int Y = Y0 + r;
            if (Y < C.cut_y || Y >= C.cut_y + C.cut_h) continue;
            unsigned b[12];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) b[3 * k + pl] = (ALLON || on[r][k]) ? (unsigned)sat8i(v[pl][r][k]) : 0u;
            // signed: a block that starts left of the cut has a negative column offset (its bytes are masked below)
            uint8_t* d = C.out + (int)(__mul24(Y - C.cut_y, C.out_stride) + 3 * (X0 - C.cut_x));
            const bool whole = X0 >= C.cut_x && X0 + 4 <= C.cut_x + C.cut_w;
            if (whole && (((size_t)d) & 3) == 0) {
                uint3 pk;
                pk.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                pk.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                pk.z = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
                *reinterpret_cast<uint3*>(d) = pk;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (X0 + k >= C.cut_x && X0 + k < C.cut_x + C.cut_w) {
                        d[3 * k] = (uint8_t)b[3 * k];
                        d[3 * k + 1] = (uint8_t)b[3 * k + 1];
                        d[3 * k + 2] = (uint8_t)b[3 * k + 2];
                    }
            }
        }
    }
}

// One 4 x 2 block of a vector level (X0 multiple of 4, Y0 even; the caller has checked that it lies inside the level /
// the cut hull): the wave-uniform single-owner path when every lane of the wave sits on the same owner, else the general
// path.  pb: first plane of this lane (NPL == 1: one plane per lane)
template <bool L0, int NPL>
__device__ __forceinline__ void blend_block(const PyrParams& P, const CanvasParams& C, const int l, const int X0, const int Y0,
                                            const int pb) {
    const int cam_lo = C.cam_lo;
    const int cw = C.w0 >> l, ch = C.h0 >> l;
    // Away from the seams a block belongs to exactly one camera with weight 1.0f everywhere (or to none):
    // the static owner map says so in one byte, and the block needs no weights, no float math and no division:
    //   acc = lap, W = 1  =>  norm = lap - sign(lap)  (see below)
    // one 16-bit entry per block: low byte = owner code, high byte = the cameras that carry weight anywhere on the block
    const unsigned entry = C.owner[l][(unsigned)(__mul24(Y0 >> 1, C.opitch[l]) + (X0 >> 2))];
    const unsigned code = entry & 0xffu;
    const unsigned ucode = __builtin_amdgcn_readfirstlane(code);
    if (ucode != 0xffu && __builtin_amdgcn_ballot_w64(code != ucode) == 0) {
        // the whole wave (a 256 x 2 strip) has one owner: its parameters are scalar, the code is straight-line
        // and every load is in flight before the first use
        int v[3][2][4];
        unsigned cp[3][3][2];
        if (l < C.bands) {
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
                load_coarse<int16_t>(C.img[l + 1] + (size_t)(pb + pl) * C.cplane[l + 1], cw >> 1, ch >> 1, C.cpitch[l + 1],
                                     X0 >> 1, Y0 >> 1, cp[pl]);
            }
        }
        if (ucode < 8u) {
            const PyrCam& c = P.cam[cam_lo + ucode];
            const int x = X0 - (c.tx >> l), y = Y0 - (c.ty >> l);
            const int tw = c.w0 >> l, th = c.h0 >> l;
            unsigned g0[3], g1[3];
            unsigned p[3][3][2];
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
                const uint8_t* g = c.lvl[l] + (size_t)(pb + pl) * c.plane[l] + (unsigned)(__mul24(y, c.pitch[l]) + x);
                g0[pl] = *reinterpret_cast<const unsigned*>(g);
                g1[pl] = *reinterpret_cast<const unsigned*>(g + c.pitch[l]);
                if (l < C.bands)
                    load_coarse<uint8_t>(c.lvl[l + 1] + (size_t)(pb + pl) * c.plane[l + 1], tw >> 1, th >> 1, c.pitch[l + 1],
                                         x >> 1, y >> 1, p[pl]);
            }
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
                int up[2][4];
                if (l < C.bands) {
                    up_block<uint8_t>(p[pl], up);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) up[0][k] = up[1][k] = 0;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int l0 = (int)((g0[pl] >> (8 * k)) & 0xffu) - up[0][k];  // |lap| <= 255: no saturation possible
                    const int l1 = (int)((g1[pl] >> (8 * k)) & 0xffu) - up[1][k];
                    v[pl][0][k] = toward_zero_by_one(l0);
                    v[pl][1][k] = toward_zero_by_one(l1);
                }
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < NPL; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 4; k++) v[pl][r][k] = 0;
        }
        if (l < C.bands) {
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
                int up[2][4];
                up_block<int16_t>(cp[pl], up);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 4; k++) v[pl][r][k] += up[r][k];  // bounded by 9*255+9: see up_block
            }
        }
        if (L0 && ucode >= 8u) {
            // an unowned block has W == 0: dst_mask is clear and the pixel is black whatever the coarser levels hold
#pragma unroll
            for (int pl = 0; pl < NPL; pl++)
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int k = 0; k < 4; k++) v[pl][r][k] = 0;
        }
        store_block<L0, true, NPL>(C, l, X0, Y0, v, true, true, true, true, true, true, true, true, pb);
        return;
    }
    // which cameras carry weight on this 4 x 2 block is static (it follows the masks): the high byte of the owner entry.
    // A seam wave used to spend its first round trip loading every covering camera's weights only to find that out
    const unsigned live = entry >> 8;
    // the coarser canvas level: early on the latency-bound small levels, late (fewer live registers) on level 0
    unsigned cp[3][3][2];
    if (!L0 && l < C.bands) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
            load_coarse<int16_t>(C.img[l + 1] + (size_t)(pb + pl) * C.cplane[l + 1], cw >> 1, ch >> 1, C.cpitch[l + 1], X0 >> 1,
                                 Y0 >> 1, cp[pl]);
    }
    int acc[3][2][4];
    float W[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            W[r][k] = 0.f;
            acc[0][r][k] = acc[1][r][k] = acc[2][r][k] = 0;
        }
    // phase B: cameras with weight, in feed order
#pragma unroll
    for (int i = 0; i < kCams; i++) {
        if (!((live >> i) & 1u)) continue;
        const PyrCam& c = P.cam[cam_lo + i];
        const int x = X0 - (c.tx >> l), y = Y0 - (c.ty >> l);
        const int tw = c.w0 >> l, th = c.h0 >> l;
        float w[2][4];
        if (L0) {
            const unsigned mk[2] = {*reinterpret_cast<const unsigned*>(c.mask0 + (unsigned)(__mul24(y, c.pitch[0]) + x)),
                                    *reinterpret_cast<const unsigned*>(c.mask0 + (unsigned)(__mul24(y + 1, c.pitch[0]) + x))};
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) w[r][k] = (float)((mk[r] >> (8 * k)) & 0xffu) * (float)(1. / 255.);
        } else {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const float4 f = *reinterpret_cast<const float4*>(c.wgt[l] + (size_t)(y + r) * c.wpitch[l] + x);
                w[r][0] = f.x; w[r][1] = f.y; w[r][2] = f.z; w[r][3] = f.w;
            }
        }
        // away from the seams the weight is exactly 1.0f on the whole block: no float path
        bool unit = true;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) unit &= w[r][k] == 1.0f;
        // issue every load of this camera before using any
        unsigned g0[3], g1[3];
        unsigned p[3][3][2];
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            const uint8_t* g = c.lvl[l] + (size_t)(pb + pl) * c.plane[l] + (size_t)y * c.pitch[l] + x;
            g0[pl] = *reinterpret_cast<const unsigned*>(g);
            g1[pl] = *reinterpret_cast<const unsigned*>(g + c.pitch[l]);
            if (l < C.bands)
                load_coarse<uint8_t>(c.lvl[l + 1] + (size_t)(pb + pl) * c.plane[l + 1], tw >> 1, th >> 1, c.pitch[l + 1], x >> 1,
                                     y >> 1, p[pl]);
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) W[r][k] += w[r][k];  // + 0.f is exact
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            int up[2][4];
            if (l < C.bands) {
                up_block<uint8_t>(p[pl], up);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) up[0][k] = up[1][k] = 0;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int l0 = sat16i((int)((g0[pl] >> (8 * k)) & 0xffu) - up[0][k]);
                const int l1 = sat16i((int)((g1[pl] >> (8 * k)) & 0xffu) - up[1][k]);
                if (unit) {  // (short)(lap * 1.0f) == lap
                    acc[pl][0][k] = (int16_t)(acc[pl][0][k] + l0);
                    acc[pl][1][k] = (int16_t)(acc[pl][1][k] + l1);
                } else {
                    acc[pl][0][k] = (int16_t)(acc[pl][0][k] + (int16_t)(int)((float)l0 * w[0][k]));
                    acc[pl][1][k] = (int16_t)(acc[pl][1][k] + (int16_t)(int)((float)l1 * w[1][k]));
                }
            }
        }
    }
    if (L0 && l < C.bands) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
            load_coarse<int16_t>(C.img[l + 1] + (size_t)(pb + pl) * C.cplane[l + 1], cw >> 1, ch >> 1, C.cpitch[l + 1], X0 >> 1,
                                 Y0 >> 1, cp[pl]);
    }
    // (short)(n / (1.0f + 1e-5f)) == n - sign(n) for every int16 n: the quotient lies strictly between
    // |n|-1 and |n| (|n| * 1e-5 < 1, and far more than an ulp of n), and the cast truncates toward zero.
    // So where the summed weight is exactly 1.0f (everywhere but the seams) no division is needed.
    bool unitW = true;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) unitW &= W[r][k] == 1.0f;
    int v[3][2][4];
#pragma unroll
    for (int pl = 0; pl < NPL; pl++) {
        int up[2][4];
        if (l < C.bands) {
            up_block<int16_t>(cp[pl], up);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) up[0][k] = up[1][k] = 0;
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int a = acc[pl][r][k];
                int nrm;
                if (unitW) nrm = toward_zero_by_one(a);
                else nrm = (int16_t)(int)((float)a / (W[r][k] + 1e-5f));
                v[pl][r][k] = l < C.bands ? sat16i(nrm + up[r][k]) : nrm;
            }
    }
    bool on[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) on[r][k] = W[r][k] > 1e-5f;
    store_block<L0, false, NPL>(C, l, X0, Y0, v, on[0][0], on[0][1], on[0][2], on[0][3], on[1][0], on[1][1], on[1][2], on[1][3], pb);
}

// NPL = 3: a lane does the three colour planes of its block.  NPL = 1 (canvas levels >= 1 only, where planes are stored
// apart): grid.z = canvas * 3 + plane and a lane does one plane - a third of the serial work per wave, three times the
// waves: these levels are one round of waves whose seam waves set the kernel's duration.
template <bool L0, int NPL = 3>
__global__ __launch_bounds__(256) void blend_level_vec_kernel(PyrParams P, CanvasSet CS, int lvl) {
    static_assert(NPL == 3 || !L0, "level 0 writes interleaved BGR");
    unsigned bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    if (L0 && NPL == 3 && ((lvl >> 8) & 15) == 4) {
        // XCD bands with the seam tiles first (shape 4, level 0): the band of XCD k is walked in the order of the static table
        // CanvasParams::order0 - tiles that hold a wave without a single owner (the general path: four times the instructions,
        // two dependent rounds of loads) come first, so their long chains run under the bulk instead of behind it
        const unsigned maxper = ((unsigned)lvl >> 12) & 0xfffffu;
        const unsigned k = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned cvi = j / maxper, jj = j - cvi * maxper;
        if (cvi >= (unsigned)CS.n) return;
        const CanvasParams& Cq = CS.c[cvi];
        if (jj >= (unsigned)Cq.order_per) return;
        const unsigned tile = Cq.order0[k * Cq.order_per + jj];
        if (tile == 0xffffu) return;
        bxi = tile % (unsigned)Cq.order_gx; byi = tile / (unsigned)Cq.order_gx; bzi = cvi;
    } else if (((lvl >> 8) & 15) == 3) {
        // XCD bands (shape 3, the default): a 1-D grid of 8 * per workgroups; the hardware deals consecutive ids round-robin
        // over the 8 XCDs, so XCD k is given the logical workgroups [k * per, (k + 1) * per) - a contiguous band of canvas
        // rows, whose neighbouring workgroups share their cache lines and pyrUp halos in ONE L2
        const unsigned gx = ((unsigned)lvl >> 12) & 0x3ffu, gy = ((unsigned)lvl >> 22) & 0x3ffu;
        const unsigned total = gx * gy * (unsigned)(NPL == 3 ? CS.n : CS.n * 3);
        const unsigned per = (total + 7u) / 8u;
        const unsigned logical = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per || logical >= total) return;
        bxi = logical % gx; byi = (logical / gx) % gy; bzi = logical / (gx * gy);
    }
    const int pb = NPL == 3 ? 0 : (int)(bzi % 3);  // first plane of this lane
    const CanvasParams& C = CS.c[NPL == 3 ? bzi : bzi / 3];
    const int l = L0 ? 0 : (lvl & 0xff);
    const int cw = C.w0 >> l, ch = C.h0 >> l;
    // level 0 covers only the block-aligned hull of the cut rectangle
    const int bx0 = L0 ? (C.cut_x & ~3) : 0, by0 = L0 ? (C.cut_y & ~1) : 0;
    // a wave is 16 x 4 blocks = 64 x 8 pixels (not a 256-pixel strip): four times fewer waves straddle a seam,
    // and a wave that does not straddle one takes the single-owner fast path below.
    // The four waves of a workgroup form a 2 x 2 patch (128 x 16 pixels).  Consecutive workgroups go to different XCDs,
    // each with its own L2, so what a workgroup reads of a row should be whole 128-byte lines: stacked (64 x 32 pixels,
    // shape 1) the 64 bytes a wave reads of a u8 plane row are half a line and the level-0 launch fetched 152 MB for the
    // ~45 MB it uses; side by side (256 x 8, shape 0) it fetches 74 MB.  Measured per frame (levels 0-2, C2), stacked /
    // side by side / 2 x 2: HBM bytes of these launches 247 / 145 / 182 MB, panoramas/s one frame at a time 7.18 / 7.35 /
    // 7.27 k, with four frames in flight 11.77 / 11.64 / 11.83 k (same box, alternating): bytes are not what bounds the
    // pipeline, and 2 x 2 is the fastest of the three.  On top of 2 x 2, XCD bands (above): blend stage 76.6 -> 72.8 us,
    // 7.24 -> 7.46 k one frame at a time, 11.71 -> 11.92 k in flight.  PANO_K3_SHAPE=0|1|2|3.
    int X0, Y0;
    if (((lvl >> 8) & 15) == 1) {         // stacked
        const int tid = threadIdx.y * 64 + threadIdx.x;
        X0 = bx0 + (bxi * 16 + (tid & 15)) * 4;
        Y0 = by0 + (byi * 16 + (tid >> 4)) * 2;
    } else if (((lvl >> 8) & 15) == 0) {  // side by side
        X0 = bx0 + ((bxi * 4 + threadIdx.y) * 16 + (threadIdx.x & 15)) * 4;
        Y0 = by0 + (byi * 4 + (threadIdx.x >> 4)) * 2;
    } else {                              // 2 x 2
        X0 = bx0 + ((bxi * 2 + (threadIdx.y & 1)) * 16 + (threadIdx.x & 15)) * 4;
        Y0 = by0 + ((byi * 2 + (threadIdx.y >> 1)) * 4 + (threadIdx.x >> 4)) * 2;
    }
    if (L0) {
        if (X0 >= C.cut_x + C.cut_w || Y0 >= C.cut_y + C.cut_h) return;
    } else if (X0 >= cw || Y0 >= ch) {
        return;
    }
    blend_block<L0, NPL>(P, C, l, X0, Y0, pb);
}

// ------------------------------------------------------------------------------------------------
// K3 level 0, strip form.  blend_level_vec_kernel spends 700 vector instructions on a single-owner 4 x 2 block, most of them
// on the two pyrUps (camera level 1 and canvas level 1): every block redoes the horizontal pass of THREE coarse rows of each,
// although vertically adjacent blocks share two of them, and the whole chain runs one value per instruction.
// Here a lane owns a strip of S blocks stacked vertically (4 x 2S pixels) and walks down it:
//   * one new coarse row per step and source - the horizontal pass of a coarse row is done once, a sliding window of three
//     rows feeds the vertical pass (S + 2 rows per S blocks instead of 3 S);
//   * the camera side runs two values per instruction: horizontal sums <= 8 * 255 and vertical sums <= 64 * 255 + 32 fit
//     uint16 (v_pk_add_u16, v_pk_mad_u16, v_pk_lshrrev_b16), with the rounding terms (+32 on even rows, +8 on odd rows) riding
//     as +4 in every horizontal sum (1 + 6 + 1 = 8 = 32 / 4, 1 + 1 = 2 = 8 / 4);
//   * Laplacian, n - sign(n), + canvas, saturate and byte packing on int16 pairs (v_pk_sub_i16, v_pk_min/max_i16, v_perm_b32).
//   The canvas side stays 32-bit in the vertical pass (|out_1| <= 9 * 255 + 9 makes 64-fold sums that do not fit 16 bits).
// A wave is 16 x 4 strips (64 x 8S pixels): as narrow as before, so no more waves straddle a seam than before.  Waves that
// do (or that lie on no single owner) run blend_block on each of their blocks.
// ------------------------------------------------------------------------------------------------
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));

// horizontal position of a lane's 4-sample coarse window (columns x-1 .. x+2 with pyrUp's border rule): what load_coarse
// derives per block, derived once per strip
struct CoarseX {
    unsigned fetch;  // element offset of the aligned fetch inside a row
    unsigned shift;  // byte shift of the window inside the fetched bytes
    unsigned selA, selB;
};
template <typename T>
__device__ __forceinline__ CoarseX coarse_x(int n, int x) {
    const int xi[4] = {x > 0 ? x - 1 : (n > 1 ? 1 : 0), x, min(x + 1, n - 1), min(x + 2, n - 1)};
    const int base = min(max(x - 1, 0), max(n - 4, 0));
    const unsigned sh0 = xi[0] - base, sh1 = xi[1] - base, sh2 = xi[2] - base, sh3 = xi[3] - base;
    CoarseX cx;
    if (sizeof(T) == 1) {
        const int ab = base & ~3;
        cx.fetch = (unsigned)ab;
        cx.shift = (unsigned)(base - ab);
        cx.selA = sh0 | (sh1 << 8) | (sh2 << 16) | (sh3 << 24);
        cx.selB = 0;
    } else {
        const int ab = base & ~1;
        cx.fetch = (unsigned)ab;
        cx.shift = (unsigned)(base - ab) * 2u;
        cx.selA = (2 * sh0) | ((2 * sh0 + 1) << 8) | ((2 * sh1) << 16) | ((2 * sh1 + 1) << 24);
        cx.selB = (2 * sh2) | ((2 * sh2 + 1) << 8) | ((2 * sh3) << 16) | ((2 * sh3 + 1) << 24);
    }
    return cx;
}
// coarse row j of the sliding window (j = -1 .. m): pyrUp's vertical border rule (top reflect-101, bottom replicate)
__device__ __forceinline__ int coarse_row(int j, int m) { return j < 0 ? (m > 1 ? 1 : 0) : min(j, m - 1); }

// camera level-1 row (u8): fetched 8 bytes -> the four horizontal sums as two uint16 pairs, each carrying +4
__device__ __forceinline__ void cam_hrow(uint2 d, const CoarseX& cx, us2_t& h01, us2_t& h23) {
    const unsigned w = __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(d.y, d.x, cx.shift), cx.selA);
    const unsigned h0 = __builtin_amdgcn_udot4(w, 0x00010601u, 4u, false), h1 = __builtin_amdgcn_udot4(w, 0x00040400u, 4u, false);
    const unsigned h2 = __builtin_amdgcn_udot4(w, 0x01060100u, 4u, false), h3 = __builtin_amdgcn_udot4(w, 0x04040000u, 4u, false);
    h01 = __builtin_bit_cast(us2_t, h0 | (h1 << 16));
    h23 = __builtin_bit_cast(us2_t, h2 | (h3 << 16));
}
// canvas level-1 row (int16): fetched 12 bytes -> the four horizontal sums, 32-bit
__device__ __forceinline__ void cv_hrow(uint3 d, const CoarseX& cx, int h[4]) {
    const unsigned w0 = __builtin_amdgcn_alignbyte(d.y, d.x, cx.shift), w1 = __builtin_amdgcn_alignbyte(d.z, d.y, cx.shift);
    const unsigned q0 = __builtin_amdgcn_perm(w1, w0, cx.selA), q1 = __builtin_amdgcn_perm(w1, w0, cx.selB);
    const s2_t A = __builtin_bit_cast(s2_t, q0), B = __builtin_bit_cast(s2_t, q1);
    const s2_t c16 = {1, 6}, c04 = {0, 4}, c01 = {0, 1};
    h[0] = __builtin_amdgcn_sdot2(A, c16, (int)B.x, false);
    h[1] = __builtin_amdgcn_sdot2(A, c04, sdot2_from_zero(q1, 0x00000004u), false);
    h[2] = __builtin_amdgcn_sdot2(A, c01, sdot2_from_zero(q1, 0x00010006u), false);
    h[3] = sdot2_from_zero(q1, 0x00040004u);
}
// one output row of level 0 (4 pixels) from the three planes' bytes: dst_mask is all set here (single owner), so
// convertTo(8U) + the cut is all that is left
__device__ __forceinline__ void store_row_l0(const CanvasParams& C, int X0, int Y, uint3 pk) {
    if (Y < C.cut_y || Y >= C.cut_y + C.cut_h) return;
    uint8_t* d = C.out + (int)(__mul24(Y - C.cut_y, C.out_stride) + 3 * (X0 - C.cut_x));
    const bool whole = X0 >= C.cut_x && X0 + 4 <= C.cut_x + C.cut_w;
    if (whole && (((size_t)d) & 3) == 0) {
        *reinterpret_cast<uint3*>(d) = pk;
    } else {
        const unsigned w[3] = {pk.x, pk.y, pk.z};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (X0 + k >= C.cut_x && X0 + k < C.cut_x + C.cut_w) {
#pragma unroll
                for (int q = 0; q < 3; q++) d[3 * k + q] = (uint8_t)(w[(3 * k + q) >> 2] >> (8 * ((3 * k + q) & 3)));
            }
    }
}

template <int S>
__global__ __launch_bounds__(256) void blend_level0_strip_kernel(PyrParams P, CanvasSet CS, int lvl) {
    // XCD bands over workgroups of 2 x 2 waves (see blend_level_vec_kernel): gx, gy = logical grid, packed in lvl
    const unsigned gx = ((unsigned)lvl >> 12) & 0x3ffu, gy = ((unsigned)lvl >> 22) & 0x3ffu;
    const unsigned total = gx * gy * (unsigned)CS.n;
    const unsigned per = (total + 7u) / 8u;
    const unsigned logical = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per || logical >= total) return;
    const unsigned bxi = logical % gx, byi = (logical / gx) % gy, bzi = logical / (gx * gy);
    const CanvasParams& C = CS.c[bzi];
    const int cw = C.w0, ch = C.h0;
    const int bx0 = C.cut_x & ~3, by0 = C.cut_y & ~1;
    const int X0 = bx0 + ((bxi * 2 + (threadIdx.y & 1)) * 16 + (threadIdx.x & 15)) * 4;
    const int Yb = by0 + ((byi * 2 + (threadIdx.y >> 1)) * 4 + (threadIdx.x >> 4)) * (2 * S);
    const int y_end = min(C.cut_y + C.cut_h, ch);
    if (X0 >= C.cut_x + C.cut_w || Yb >= y_end) return;
    // the strip's owner entries: single owner for the whole wave?
    unsigned entry[S];
#pragma unroll
    for (int s = 0; s < S; s++)
        entry[s] = C.owner[0][(unsigned)(__mul24(min(Yb + 2 * s, ch - 2) >> 1, C.opitch[0]) + (X0 >> 2))];
    const unsigned ucode = __builtin_amdgcn_readfirstlane(entry[0] & 0xffu);
    bool same = true;
#pragma unroll
    for (int s = 0; s < S; s++) same &= (entry[s] & 0xffu) == ucode;
    if (ucode == 0xffu || __builtin_amdgcn_ballot_w64(!same) != 0) {
        for (int s = 0; s < S; s++) {
            const int Y0 = Yb + 2 * s;
            if (Y0 < y_end) blend_block<true, 3>(P, C, 0, X0, Y0, 0);
        }
        return;
    }
    if (ucode >= 8u) {  // no camera carries weight here: dst_mask clear, black
#pragma unroll
        for (int r = 0; r < 2 * S; r++)
            if (Yb + r < y_end) store_row_l0(C, X0, Yb + r, make_uint3(0u, 0u, 0u));
        return;
    }
    // ---- the whole wave lies on camera `ucode` with weight 1: acc = lap, W = 1, norm = lap - sign(lap)
    const PyrCam& c = P.cam[C.cam_lo + ucode];
    const int x = X0 - c.tx, y = Yb - c.ty;      // tile coordinates (y even: tile origins are multiples of 2^bands)
    const int n1 = c.w0 >> 1, m1 = c.h0 >> 1;    // camera level 1
    const int cn = cw >> 1, cm = ch >> 1;        // canvas level 1
    const CoarseX kx = coarse_x<uint8_t>(n1, x >> 1), vx = coarse_x<int16_t>(cn, X0 >> 1);
    const int kr0 = (y >> 1) - 1, vr0 = (Yb >> 1) - 1;  // first row of the two sliding windows
    // plane by plane: the loads of a plane - 2S fine rows, S + 2 coarse rows of the camera and of the canvas - are issued
    // together and consumed before the next plane's are issued (all three planes at once: 183 VGPRs at S = 4, two waves
    // per SIMD)
    // The plane loop is a real loop: unrolled, the compiler overlaps the planes and allocates 179 VGPRs at S = 4 (two waves per
    // SIMD).  A plane's four bytes per row are inserted into the row's three interleaved BGR dwords (B0 G0 R0 B1 | G1 R1 B2 G2
    // | R2 B3 G3 R3) by v_perm_b32 with per-plane selectors (0-3 = old bytes, 4-7 = the plane's pixels 0-3)
    uint3 out[2 * S];
#pragma unroll
    for (int r = 0; r < 2 * S; r++) out[r] = make_uint3(0u, 0u, 0u);
#pragma unroll 1
    for (int pl = 0; pl < 3; pl++) {
        const unsigned ins0 = pl == 0 ? 0x05020104u : (pl == 1 ? 0x03020400u : 0x03040100u);
        const unsigned ins1 = pl == 0 ? 0x03060100u : (pl == 1 ? 0x06020105u : 0x03020500u);
        const unsigned ins2 = pl == 0 ? 0x03020700u : (pl == 1 ? 0x03070100u : 0x07020106u);
        unsigned g[2 * S];
        uint2 kq[S + 2];
        uint3 vq[S + 2];
        {
            const uint8_t* g0 = c.lvl[0] + (size_t)pl * c.plane[0] + (unsigned)(__mul24(y, c.pitch[0]) + x);
#pragma unroll
            for (int r = 0; r < 2 * S; r++) g[r] = *reinterpret_cast<const unsigned*>(g0 + __mul24(min(r, c.h0 - 1 - y), c.pitch[0]));
            const uint8_t* k1 = c.lvl[1] + (size_t)pl * c.plane[1] + kx.fetch;
#pragma unroll
            for (int r = 0; r < S + 2; r++) kq[r] = *reinterpret_cast<const uint2*>(k1 + (unsigned)__mul24(coarse_row(kr0 + r, m1), c.pitch[1]));
            const int16_t* v1 = C.img[1] + (size_t)pl * C.cplane[1] + vx.fetch;
#pragma unroll
            for (int r = 0; r < S + 2; r++) vq[r] = *reinterpret_cast<const uint3*>(v1 + (unsigned)__mul24(coarse_row(vr0 + r, cm), C.cpitch[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        us2_t ka01, ka23, kb01, kb23, kc01, kc23;
        int va[4], vb[4], vc[4];
        cam_hrow(kq[0], kx, ka01, ka23);
        cam_hrow(kq[1], kx, kb01, kb23);
        cv_hrow(vq[0], vx, va);
        cv_hrow(vq[1], vx, vb);
#pragma unroll
        for (int s = 0; s < S; s++) {
            cam_hrow(kq[s + 2], kx, kc01, kc23);
            cv_hrow(vq[s + 2], vx, vc);
            // pyrUp of the camera's level 1: rows 2k (1 6 1) and 2k + 1 (4 4); the +4 in every sum is the rounding term
            const us2_t e01 = (us2_t)(ka01 + kc01 + kb01 * (us2_t)6) >> (us2_t)6, e23 = (us2_t)(ka23 + kc23 + kb23 * (us2_t)6) >> (us2_t)6;
            const us2_t o01 = (us2_t)(kb01 + kc01) >> (us2_t)4, o23 = (us2_t)(kb23 + kc23) >> (us2_t)4;
            // pyrUp of the canvas' level 1, packed to int16 pairs (|out_1| <= 9 * 255 + 9)
            int ve[4], vo[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                ve[k] = (va[k] + vc[k] + 32 + vb[k] * 6) >> 6;
                vo[k] = (vb[k] + vc[k] + 8) >> 4;
            }
            const s2_t ce01 = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm((unsigned)ve[1], (unsigned)ve[0], 0x05040100u));
            const s2_t ce23 = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm((unsigned)ve[3], (unsigned)ve[2], 0x05040100u));
            const s2_t co01 = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm((unsigned)vo[1], (unsigned)vo[0], 0x05040100u));
            const s2_t co23 = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm((unsigned)vo[3], (unsigned)vo[2], 0x05040100u));
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const unsigned gw = g[2 * s + r];
                const s2_t g01 = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm(0u, gw, 0x0c010c00u));
                const s2_t g23 = __builtin_bit_cast(s2_t, __builtin_amdgcn_perm(0u, gw, 0x0c030c02u));
                const s2_t u01 = __builtin_bit_cast(s2_t, r == 0 ? e01 : o01), u23 = __builtin_bit_cast(s2_t, r == 0 ? e23 : o23);
                const s2_t one = {1, 1}, mone = {-1, -1}, zero = {0, 0}, top = {255, 255};
                s2_t l01 = g01 - u01, l23 = g23 - u23;                                // |lap| <= 255
                l01 -= __builtin_elementwise_max(__builtin_elementwise_min(l01, one), mone);   // n - sign(n)
                l23 -= __builtin_elementwise_max(__builtin_elementwise_min(l23, one), mone);
                l01 += r == 0 ? ce01 : co01;
                l23 += r == 0 ? ce23 : co23;
                l01 = __builtin_elementwise_min(__builtin_elementwise_max(l01, zero), top);    // convertTo(CV_8U)
                l23 = __builtin_elementwise_min(__builtin_elementwise_max(l23, zero), top);
                const unsigned px = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, l23), __builtin_bit_cast(unsigned, l01), 0x06040200u);
                out[2 * s + r].x = __builtin_amdgcn_perm(px, out[2 * s + r].x, ins0);
                out[2 * s + r].y = __builtin_amdgcn_perm(px, out[2 * s + r].y, ins1);
                out[2 * s + r].z = __builtin_amdgcn_perm(px, out[2 * s + r].z, ins2);
            }
            ka01 = kb01; ka23 = kb23; kb01 = kc01; kb23 = kc23;
#pragma unroll
            for (int k = 0; k < 4; k++) { va[k] = vb[k]; vb[k] = vc[k]; }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 2 * S; r++)
        if (Yb + r < y_end) store_row_l0(C, X0, Yb + r, out[r]);
}


// owner map of a vector level: one byte per 4 x 2 block (see CanvasParams::owner)
__global__ __launch_bounds__(256) void build_owner_kernel(PyrParams P, CanvasParams C, int l, uint16_t* owner) {
    const int cw = C.w0 >> l, ch = C.h0 >> l;
    const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
    if (bx * 4 >= cw || by * 2 >= ch) return;
    int holders = 0, unit_cam = -1;
    unsigned mask = 0;
    bool all_unit = true;
    for (int i = 0; i < P.ncam; i++) {
        const PyrCam& c = P.cam[i];
        const int x = bx * 4 - (c.tx >> l), y = by * 2 - (c.ty >> l);
        if ((unsigned)x >= (unsigned)(c.w0 >> l) || (unsigned)y >= (unsigned)(c.h0 >> l)) continue;
        bool any = false, unit = true;
        for (int r = 0; r < 2; r++)
            for (int k = 0; k < 4; k++) {
                const float w = cam_weight(c, l, x + k, y + r);
                any |= w != 0.f;
                unit &= w == 1.0f;
            }
        if (any) {
            holders++;
            unit_cam = i;
            mask |= 1u << i;
            all_unit &= unit;
        }
    }
    uint8_t code = 0xff;
    if (holders == 0) code = 0xfe;
    else if (holders == 1 && all_unit) code = (uint8_t)unit_cam;
    owner[(size_t)by * C.opitch[l] + bx] = (uint16_t)(code | (mask << 8));
}
void launch_build_owner(const PyrParams& p, const CanvasParams& c, int l, uint16_t* owner, hipStream_t s) {
    const int bw = (c.w0 >> l) / 4, bh = (c.h0 >> l) / 2;
    dim3 block(64, 4, 1), grid((bw + 63) / 64, (bh + 3) / 4, 1);
    hipLaunchKernelGGL(build_owner_kernel, grid, block, 0, s, p, c, l, owner);
}

// which 128 x 16-pixel workgroup tiles of level 0 hold a wave that takes the general path (no single owner)?  One flag per tile,
// the same lane -> block mapping as blend_level_vec_kernel's 2 x 2 shape over the hull of the cut
__global__ __launch_bounds__(256) void tile_mixed_kernel(CanvasParams C, int gx, uint8_t* flags) {
    const int bx0 = C.cut_x & ~3, by0 = C.cut_y & ~1;
    const int X0 = bx0 + ((blockIdx.x * 2 + (threadIdx.y & 1)) * 16 + (threadIdx.x & 15)) * 4;
    const int Y0 = by0 + ((blockIdx.y * 2 + (threadIdx.y >> 1)) * 4 + (threadIdx.x >> 4)) * 2;
    const bool valid = X0 < C.cut_x + C.cut_w && Y0 < C.cut_y + C.cut_h;
    unsigned code = 0;
    if (valid) code = C.owner[0][(unsigned)(__mul24(Y0 >> 1, C.opitch[0]) + (X0 >> 2))] & 0xffu;
    const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
    int mixed = 0;
    if (vm) {
        const unsigned c0 = (unsigned)__shfl((int)code, __ffsll((long long)vm) - 1);
        mixed = c0 == 0xffu || __builtin_amdgcn_ballot_w64(valid && code != c0) != 0;
    }
    mixed = __syncthreads_or(mixed);
    if (threadIdx.x == 0 && threadIdx.y == 0) flags[blockIdx.y * gx + blockIdx.x] = (uint8_t)(mixed != 0);
}
void launch_tile_mixed(const CanvasParams& c, int gx, int gy, uint8_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(tile_mixed_kernel, dim3(gx, gy, 1), dim3(64, 4, 1), 0, s, c, gx, flags);
}

// ------------------------------------------------------------------------------------------------
// Small levels (small_base .. bands): every one of them is latency bound on its own (a handful of waves
// walking a chain of dependent loads), and the per-level launches used to cost more than the two large
// levels together.  They are split differently:
//   1. norm_small_kernel - ONE launch over all small levels: the camera half of the blend,
//        norm_l = (short)(sum_cams (short)(lap_l * w_l) / (sum_cams w_l + 1e-5f)),
//      which does not depend on any other canvas level, written to the canvas level buffers;
//   2. collapse_small_kernel - ONE launch: each workgroup owns a 64 x 16 tile of level small_base and
//      rebuilds the collapse chain out_l = sat16(norm_l + pyrUp(out_{l+1})) for the footprint of its tile
//      through LDS, coarse to fine (the halo is recomputed per workgroup: a few hundred pixels).
// Only out_{small_base} is needed by the next (vector) level; canvas levels above it keep norm_l.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_small_kernel(PyrParams P, CanvasSet CS) {
    const int nsmall = CS.c[0].bands - CS.c[0].small_base + 1;
    const CanvasParams& C = CS.c[blockIdx.z / nsmall];
    const int cam_lo = C.cam_lo, cam_n = C.cam_n;
    const int l = C.small_base + blockIdx.z % nsmall;
    const int cw = C.w0 >> l, ch = C.h0 >> l;
    const int X = blockIdx.x * 64 + threadIdx.x, Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= cw || Y >= ch) return;
    float wv[kCams];
#pragma unroll
    for (int i = 0; i < kCams; i++) {
        wv[i] = 0.f;
        if (i < cam_n) {
            const PyrCam& c = P.cam[cam_lo + i];
            const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
            if ((unsigned)x < (unsigned)(c.w0 >> l) && (unsigned)y < (unsigned)(c.h0 >> l)) wv[i] = cam_weight(c, l, x, y);
        }
    }
    int acc[3] = {0, 0, 0};
    float W = 0.f;
#pragma unroll
    for (int i = 0; i < kCams; i++) {
        const float w = wv[i];
        if (w == 0.f) continue;
        const PyrCam& c = P.cam[cam_lo + i];
        const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
        const int tw = c.w0 >> l, th = c.h0 >> l;
        W += w;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int lap = c.lvl[l][(size_t)k * c.plane[l] + (size_t)y * c.pitch[l] + x];
            if (l < C.bands)
                lap = sat16i(lap - pyr_up_px<uint8_t>(c.lvl[l + 1] + (size_t)k * c.plane[l + 1], tw >> 1, th >> 1,
                                                      c.pitch[l + 1], x, y));
            acc[k] = (int16_t)(acc[k] + (int16_t)(int)((float)lap * w));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int v;
        if (W == 1.0f) v = toward_zero_by_one(acc[k]);
        else v = (int16_t)(int)((float)acc[k] / (W + 1e-5f));
        C.img[l][(size_t)k * C.cplane[l] + (size_t)Y * C.cpitch[l] + X] = (int16_t)v;
    }
}

constexpr int kSmallTileW = 32, kSmallTileH = 8;
// footprint of a level-`small_base` tile at the coarser levels: R_{l+1} = [R_l.lo/2 - 1, R_l.hi/2 + 1] clamped
struct SmallRegion {
    int x0, y0, w, h;
};
constexpr int kSmallLdsElems = 512;  // int16 per plane for all coarser regions of a tile (< 400 needed); 3 KB total

// One workgroup = a 32 x 8 tile of level small_base; its footprint at every coarser level fits the same 32 x 8
// thread grid (18 x 6, 11 x 5, 8 x 5, ...), so each lane owns at most ONE pixel per level: short serial code per
// lane, 4 workgroups per CU.
// NORM: the workgroup also computes norm_l over its footprints itself (norm_small_kernel's per-pixel code) instead of reading
// it from the canvas buffers: the small levels are then ONE launch, not two.  The footprints overlap between neighbouring tiles
// (419 pixels per tile for 336 of its own with three levels), so a quarter more norm work than the separate launch does, for
// one launch (~4 us of latency) less.  P is only read when NORM.
template <bool NORM>
__global__ __launch_bounds__(256) void collapse_small_kernel(PyrParams P, CanvasSet CS) {
    const CanvasParams& C = CS.c[blockIdx.z];
    __shared__ int16_t lds[3 * kSmallLdsElems];
    const int k0 = C.small_base, nb = C.bands;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int tx = tid & 31, ty = tid >> 5;
    // footprints of this workgroup's tile at every level of the chain (block-uniform: scalar registers)
    int rx0[kLevels], ry0[kLevels], rw[kLevels], rh[kLevels], ro[kLevels];
    {
        const int cw = C.w0 >> k0, ch = C.h0 >> k0;
        int x0 = blockIdx.x * kSmallTileW, y0 = blockIdx.y * kSmallTileH;
        int x1 = min(x0 + kSmallTileW, cw) - 1, y1 = min(y0 + kSmallTileH, ch) - 1;
        int o = 0;
#pragma unroll
        for (int j = 0; j < kLevels; j++) {
            const int l = k0 + j;
            rx0[j] = x0; ry0[j] = y0; rw[j] = x1 - x0 + 1; rh[j] = y1 - y0 + 1;
            ro[j] = o;
            if (j > 0) o += rw[j] * rh[j];  // level k0 goes straight to global memory
            const int nw = C.w0 >> (l + 1), nh = C.h0 >> (l + 1);
            x0 = max((x0 >> 1) - 1, 0); y0 = max((y0 >> 1) - 1, 0);
            x1 = min((x1 >> 1) + 1, max(nw - 1, 0)); y1 = min((y1 >> 1) + 1, max(nh - 1, 0));
        }
    }
    // phase 1: every norm_l value this lane will need, all loads in flight together
    int16_t nv[kLevels][3];
#pragma unroll
    for (int j = 0; j < kLevels; j++) {
        const int l = k0 + j;
#pragma unroll
        for (int pl = 0; pl < 3; pl++) nv[j][pl] = 0;
        if (!(l <= nb && tx < rw[j] && ty < rh[j])) continue;
        if (!NORM) {
#pragma unroll
            for (int pl = 0; pl < 3; pl++)
                nv[j][pl] = C.img[l][(size_t)pl * C.cplane[l] + (size_t)(ry0[j] + ty) * C.cpitch[l] + rx0[j] + tx];
        } else {
            // norm_l = (short)(sum_cams (short)(lap_l * w_l) / (sum_cams w_l + 1e-5f)) at canvas pixel (X, Y) of level l
            const int X = rx0[j] + tx, Y = ry0[j] + ty;
            const int cam_lo = C.cam_lo, cam_n = C.cam_n;
            // Every camera whose tile holds the pixel is read, weight or not: (short)(lap * 0.f) == 0 and W + 0.f == W, so the
            // result is the same as skipping the weightless ones - and the pixel loads do not wait for the weight loads
            int acc[3] = {0, 0, 0};
            float W = 0.f;
#pragma unroll
            for (int i = 0; i < kCams; i++) {
                if (i >= cam_n) continue;
                const PyrCam& c = P.cam[cam_lo + i];
                const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
                const int tw = c.w0 >> l, th = c.h0 >> l;
                if (!((unsigned)x < (unsigned)tw && (unsigned)y < (unsigned)th)) continue;
                const float w = cam_weight(c, l, x, y);
                W += w;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    int lap = c.lvl[l][(size_t)k * c.plane[l] + (size_t)y * c.pitch[l] + x];
                    if (l < nb)
                        lap = sat16i(lap - pyr_up_px<uint8_t>(c.lvl[l + 1] + (size_t)k * c.plane[l + 1], tw >> 1, th >> 1,
                                                              c.pitch[l + 1], x, y));
                    acc[k] = (int16_t)(acc[k] + (int16_t)(int)((float)lap * w));
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                int v;
                if (W == 1.0f) v = toward_zero_by_one(acc[k]);
                else v = (int16_t)(int)((float)acc[k] / (W + 1e-5f));
                nv[j][k] = (int16_t)v;
            }
        }
    }
    // phase 2: collapse coarse -> fine through LDS
#pragma unroll
    for (int j = kLevels - 1; j >= 0; j--) {
        const int l = k0 + j;
        if (l > nb) continue;  // block-uniform
        const int cw = C.w0 >> l, ch = C.h0 >> l;
        const int nx = cw >> 1, ny = ch >> 1;
        if (tx < rw[j] && ty < rh[j]) {
            const int X = rx0[j] + tx, Y = ry0[j] + ty;
            const int x = X >> 1, y = Y >> 1;
            int xi[3], wx[3], yi[3], wy[3];
            if (!(X & 1)) {
                xi[0] = x > 0 ? x - 1 : (nx > 1 ? 1 : 0); xi[1] = x; xi[2] = min(x + 1, nx - 1);
                wx[0] = 1; wx[1] = 6; wx[2] = 1;
            } else {
                xi[0] = x; xi[1] = min(x + 1, nx - 1); xi[2] = x;
                wx[0] = 4; wx[1] = 4; wx[2] = 0;
            }
            if (!(Y & 1)) {
                yi[0] = y > 0 ? y - 1 : (ny > 1 ? 1 : 0); yi[1] = y; yi[2] = min(y + 1, ny - 1);
                wy[0] = 1; wy[1] = 6; wy[2] = 1;
            } else {
                yi[0] = y; yi[1] = min(y + 1, ny - 1); yi[2] = y;
                wy[0] = 4; wy[1] = 4; wy[2] = 0;
            }
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                int v = nv[j][pl];
                if (l < nb && j + 1 < kLevels) {
                    // pyrUp of out_{l+1} from its LDS region (every index it needs lies inside that region)
                    const int jc = j + 1 < kLevels ? j + 1 : j;
                    const int16_t* S = lds + pl * kSmallLdsElems + ro[jc];
                    int acc = 0;
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        const int16_t* row = S + (yi[t] - ry0[jc]) * rw[jc] - rx0[jc];
                        acc += ((int)row[xi[0]] * wx[0] + (int)row[xi[1]] * wx[1] + (int)row[xi[2]] * wx[2]) * wy[t];
                    }
                    v = sat16i(v + sat16i((acc + 32) >> 6));
                }
                if (j > 0) lds[pl * kSmallLdsElems + ro[j] + ty * rw[j] + tx] = (int16_t)v;
                else C.img[l][(size_t)pl * C.cplane[l] + (size_t)Y * C.cpitch[l] + X] = (int16_t)v;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The small levels in ONE launch (CanvasParams::small_fused): camera pyramid levels small_base+1 .. bands, the camera half
// of the blend (norm_l) and the collapse chain, per 64 x 16 tile and colour plane of canvas level small_base, all through LDS.
// Replaces, per frame, the pyrDown launches above level small_base, norm_small_kernel, collapse_small_kernel - and, because
// small_base may then sit one level lower, a vector blend level: launches of 3 - 10 us each with almost no work (a dependent
// launch costs ~4 us before it does anything).  The price is recomputation: a tile needs G_{k0+1} over its pyrUp footprint,
// which needs G_{k0} over the pyrDown footprint of that, and so on (117 x 69 pixels of G_{k0} for a 64 x 16 tile with four
// fused levels); the pyrDown of the recomputed levels runs four outputs per lane on LDS dwords (the v_dot4 rows of pyr_down_kernel).
// Same arithmetic as pyr_down_kernel / norm_small_kernel / collapse_small_kernel, bit for bit; camera levels above
// small_base are not written to memory at all (pano_debug_get_level builds them on demand).
//
// LDS boxes per live camera: level j (= k0 + j) is held over the REAL pixel range need_j that anything consumes, padded by
// 2 on every side in VIRTUAL coordinates: cell v holds G(reflect101(v)), so the pyrDown of the next level reads 5 x 5
// windows with no border logic.  Box x origins are multiples of 4, so a group of four outputs reads the 16 bytes around it
// as four LDS dwords.
constexpr int kFuseMaxJ = 3;             // at most four fused levels (k0 .. k0 + 3)
constexpr int kFuseTileW = 64, kFuseTileH = 16;
constexpr int kFuseLdsBytes = 13 << 10;  // >= the worst-case boxes of one plane: 124 x 73 + 64 x 37 + 36 x 19 + 20 x 10 bytes + slack
constexpr int kFuseCollapseElems = 640;  // int16: the footprints 34 x 10 + 19 x 7 + 12 x 6 of the coarser canvas levels
// pixels a lane owns per level: level j's footprint (<= 64 x 16, 34 x 10, 19 x 7, 12 x 6) is walked by 64 x 4 lanes in
// kFuseQ[j] passes of 4 rows (levels 0, 1) or by 32 x 8 lanes in one pass (levels 2, 3)
__device__ __forceinline__ constexpr int fuse_q(int j) { return j == 0 ? 4 : (j == 1 ? 3 : 1); }
struct FuseBox {
    int x0, y0, x1, y1;   // need: real pixel range, inclusive (x1 < x0: empty)
    int bx0, by0, bw, bh; // LDS box: virtual origin, row pitch in bytes (multiple of 4), rows
    int off;              // byte offset in LDS
};
// lane -> pixel q of level j's footprint (rw x rh): false when the lane has no such pixel
__device__ __forceinline__ bool fuse_pixel(int j, int q, int tid, int rw, int rh, int& px, int& py) {
    if (j <= 1) { px = tid & 63; py = (tid >> 6) + 4 * q; }
    else { px = tid & 31; py = tid >> 5; }
    return px < rw && py < rh;
}
__device__ __forceinline__ void fuse_footprints(const CanvasParams& C, int bx, int by, int rx0[], int ry0[], int rw[], int rh[], int ro[]) {
    const int k0 = C.small_base;
    const int cw = C.w0 >> k0, ch = C.h0 >> k0;
    int x0 = bx * kFuseTileW, y0 = by * kFuseTileH;
    int x1 = min(x0 + kFuseTileW, cw) - 1, y1 = min(y0 + kFuseTileH, ch) - 1;
    int o = 0;
#pragma unroll
    for (int j = 0; j <= kFuseMaxJ; j++) {
        const int l = k0 + j;
        rx0[j] = x0; ry0[j] = y0; rw[j] = x1 - x0 + 1; rh[j] = y1 - y0 + 1;
        ro[j] = o;
        if (j > 0) o += rw[j] * rh[j];
        const int nw = C.w0 >> (l + 1), nh = C.h0 >> (l + 1);
        x0 = max((x0 >> 1) - 1, 0); y0 = max((y0 >> 1) - 1, 0);
        x1 = min((x1 >> 1) + 1, max(nw - 1, 0)); y1 = min((y1 >> 1) + 1, max(nh - 1, 0));
    }
}

// one byte per (tile, canvas): the cameras that carry weight anywhere on the tile's footprint at any fused level.  Static
// (it follows the masks), so the fused kernel knows at once whose pixels to fetch - no weight round trip in front of the loads
__global__ __launch_bounds__(256) void small_live_kernel(PyrParams P, CanvasParams C, uint8_t* table) {
    const int k0 = C.small_base, J = C.bands - k0;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    int rx0[kFuseMaxJ + 1], ry0[kFuseMaxJ + 1], rw[kFuseMaxJ + 1], rh[kFuseMaxJ + 1], ro[kFuseMaxJ + 1];
    fuse_footprints(C, blockIdx.x, blockIdx.y, rx0, ry0, rw, rh, ro);
    int bits = 0;
#pragma unroll
    for (int j = 0; j <= kFuseMaxJ; j++) {
        if (j > J) continue;
        const int l = k0 + j;
#pragma unroll
        for (int q = 0; q < fuse_q(j); q++) {
            int px, py;
            if (!fuse_pixel(j, q, tid, rw[j], rh[j], px, py)) continue;
            for (int i = 0; i < C.cam_n; i++) {
                const PyrCam& c = P.cam[C.cam_lo + i];
                const int x = rx0[j] + px - (c.tx >> l), y = ry0[j] + py - (c.ty >> l);
                if ((unsigned)x < (unsigned)(c.w0 >> l) && (unsigned)y < (unsigned)(c.h0 >> l) && cam_weight(c, l, x, y) != 0.f) bits |= 1 << i;
            }
        }
    }
    // (__syncthreads_or answers "any lane non-zero", not the OR of the values)
    __shared__ int all_bits;
    if (tid == 0) all_bits = 0;
    __syncthreads();
    if (bits) atomicOr(&all_bits, bits);
    __syncthreads();
    if (tid == 0) table[blockIdx.y * gridDim.x + blockIdx.x] = (uint8_t)all_bits;
}
void launch_small_live(const PyrParams& p, const CanvasParams& c, uint8_t* table, hipStream_t s) {
    const int cw = c.w0 >> c.small_base, ch = c.h0 >> c.small_base;
    dim3 block(64, 4, 1), grid((cw + kFuseTileW - 1) / kFuseTileW, (ch + kFuseTileH - 1) / kFuseTileH, 1);
    hipLaunchKernelGGL(small_live_kernel, grid, block, 0, s, p, c, table);
}

// grid.z = canvas * 3 + plane: a workgroup does ONE colour plane of its tile (three times the workgroups, a third of the
// serial work in each: the kernel's duration is the length of one workgroup's chain of barrier-separated phases)
__global__ __launch_bounds__(256) void small_fused_kernel(PyrParams P, CanvasSet CS) {
    const CanvasParams& C = CS.c[blockIdx.z / 3];
    const int pl = blockIdx.z % 3;
    __shared__ __attribute__((aligned(16))) uint8_t box_lds[kFuseLdsBytes];
    __shared__ int16_t lds[kFuseCollapseElems];
    const int k0 = C.small_base, nb = C.bands, J = nb - k0;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const unsigned livebits = C.small_live[blockIdx.y * gridDim.x + blockIdx.x];
    // canvas footprints of this workgroup's tile at every fused level (block-uniform: scalar registers)
    int rx0[kFuseMaxJ + 1], ry0[kFuseMaxJ + 1], rw[kFuseMaxJ + 1], rh[kFuseMaxJ + 1], ro[kFuseMaxJ + 1];
    fuse_footprints(C, blockIdx.x, blockIdx.y, rx0, ry0, rw, rh, ro);
    int acc[kFuseMaxJ + 1][4];
    float W[kFuseMaxJ + 1][4];
#pragma unroll
    for (int j = 0; j <= kFuseMaxJ; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            W[j][q] = 0.f;
            acc[j][q] = 0;
        }
    bool first = true;
    // A real loop over the cameras, ONE copy of the body: unrolled eight times the kernel is 60 KB of code and workgroups on
    // different cameras evict each other from the instruction cache.  The camera block is read through the kernarg segment
    // (PyrParams is the first kernel argument) - indexing the by-value argument with a runtime index would copy it to scratch
    static_assert(offsetof(PyrParams, cam) == 0, "PyrParams::cam first");
#pragma unroll 1
    for (int i = 0; i < C.cam_n; i++) {  // feed order
        if (!((livebits >> i) & 1u)) continue;  // block-uniform
        typedef const PyrCam __attribute__((address_space(4))) KCam;
        KCam* kc = (KCam*)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + (size_t)(C.cam_lo + i) * sizeof(PyrCam));
        struct {  // what this kernel needs of the camera, in scalar registers
            const uint8_t* g0; int pitch0, plane0, tx, ty, w0, h0;
            const float* wgt[kFuseMaxJ + 1]; int wpitch[kFuseMaxJ + 1];
        } c;
        c.g0 = kc->lvl[k0]; c.pitch0 = kc->pitch[k0]; c.plane0 = kc->plane[k0];
        c.tx = kc->tx; c.ty = kc->ty; c.w0 = kc->w0; c.h0 = kc->h0;
#pragma unroll
        for (int j = 0; j <= kFuseMaxJ; j++) {
            c.wgt[j] = kc->wgt[min(k0 + j, kLevels - 1)];
            c.wpitch[j] = kc->wpitch[min(k0 + j, kLevels - 1)];
        }
        // this lane's weights at its pixels of every level (0 outside the camera's tile): in flight with the pixel loads below
        float wv[kFuseMaxJ + 1][4];
#pragma unroll
        for (int j = 0; j <= kFuseMaxJ; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                wv[j][q] = 0.f;
                int px, py;
                if (j <= J && q < fuse_q(j) && fuse_pixel(j, q, tid, rw[j], rh[j], px, py)) {
                    const int l = k0 + j;
                    const int x = rx0[j] + px - (c.tx >> l), y = ry0[j] + py - (c.ty >> l);
                    if ((unsigned)x < (unsigned)(c.w0 >> l) && (unsigned)y < (unsigned)(c.h0 >> l)) wv[j][q] = c.wgt[j][(size_t)y * c.wpitch[j] + x];  // k0 >= 1: f32 levels
                }
            }
        if (!first) __syncthreads();  // the previous camera's boxes stay until every lane has read its taps
        first = false;
        // what is needed of every camera level, coarse to fine (block-uniform)
        FuseBox B[kFuseMaxJ + 1];
        {
            int off = 0;
#pragma unroll
            for (int j = kFuseMaxJ; j >= 0; j--) {
                if (j > J) continue;
                const int l = k0 + j;
                const int dw = c.w0 >> l, dh = c.h0 >> l;
                FuseBox b;
                b.x0 = max(rx0[j] - (c.tx >> l), 0); b.y0 = max(ry0[j] - (c.ty >> l), 0);
                b.x1 = min(rx0[j] + rw[j] - 1 - (c.tx >> l), dw - 1); b.y1 = min(ry0[j] + rh[j] - 1 - (c.ty >> l), dh - 1);
                if (b.x1 < b.x0 || b.y1 < b.y0) { b.x0 = b.y0 = 0; b.x1 = b.y1 = -1; }
                const int jn = j + 1 <= kFuseMaxJ ? j + 1 : j;
                if (j < J && B[jn].x1 >= B[jn].x0) {
                    const FuseBox& n = B[jn];
                    const int fx0 = max(2 * n.x0 - 2, 0), fy0 = max(2 * n.y0 - 2, 0);
                    const int fx1 = min(2 * n.x1 + 2, dw - 1), fy1 = min(2 * n.y1 + 2, dh - 1);
                    if (b.x1 < b.x0) { b.x0 = fx0; b.y0 = fy0; b.x1 = fx1; b.y1 = fy1; }
                    else { b.x0 = min(b.x0, fx0); b.y0 = min(b.y0, fy0); b.x1 = max(b.x1, fx1); b.y1 = max(b.y1, fy1); }
                }
                b.bx0 = (b.x0 - 2) & ~3; b.by0 = b.y0 - 2;
                b.bw = b.x1 >= b.x0 ? ((b.x1 + 2 - b.bx0 + 1 + 3) & ~3) : 0;
                b.bh = b.x1 >= b.x0 ? b.y1 + 2 - b.by0 + 1 : 0;
                b.off = off;
                off += b.bw * b.bh;
                B[j] = b;
            }
        }
        // level k0: global -> LDS, four virtual columns per lane and step (lanes = 32 column groups x 8 rows: no divisions);
        // the loads of a lane (at most 10 rows) are all issued before the first is stored
        {
            const FuseBox& b = B[0];
            const int dw = c.w0 >> k0, dh = c.h0 >> k0;
            const int gpr = b.bw >> 2;    // <= 31 column groups
            constexpr int kRowSteps = 10; // box rows <= 73
            const int gx = tid & 31, gy = tid >> 5;
            const uint8_t* plane = c.g0 + (size_t)pl * c.plane0;
            if (gx < gpr) {
                const int vx = b.bx0 + 4 * gx;
                const bool inside = vx >= 0 && vx + 3 < dw;
                unsigned d[kRowSteps];
                if (inside) {
#pragma unroll
                    for (int k = 0; k < kRowSteps; k++) {
                        const int cy = gy + 8 * k;
                        const unsigned rowoff = (unsigned)reflect101_idx(b.by0 + min(cy, b.bh - 1), dh) * (unsigned)c.pitch0 + (unsigned)vx;
                        d[k] = *reinterpret_cast<const unsigned*>(plane + rowoff);
                    }
                } else {
                    int xr[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) xr[q] = reflect101_idx(vx + q, dw);
#pragma unroll
                    for (int k = 0; k < kRowSteps; k++) {
                        const int cy = gy + 8 * k;
                        const uint8_t* row = plane + (unsigned)reflect101_idx(b.by0 + min(cy, b.bh - 1), dh) * (unsigned)c.pitch0;
                        d[k] = (unsigned)row[xr[0]] | ((unsigned)row[xr[1]] << 8) | ((unsigned)row[xr[2]] << 16) | ((unsigned)row[xr[3]] << 24);
                    }
                }
#pragma unroll
                for (int k = 0; k < kRowSteps; k++) {
                    const int cy = gy + 8 * k;
                    if (cy < b.bh) *reinterpret_cast<unsigned*>(box_lds + b.off + cy * b.bw + 4 * gx) = d[k];
                }
            }
        }
        __syncthreads();
        // levels k0+1 .. nb: pyrDown LDS -> LDS, a group of four cells (pads included) per lane and step (lanes = 16 groups x
        // 16 rows).  Interior groups read 16 bytes per source row as four dwords and take pyr_down_hrow's two-dot4 windows;
        // groups that touch a border or leave the needed range go cell by cell through the reflected coordinates
#pragma unroll
        for (int j = 1; j <= kFuseMaxJ; j++) {
            if (j > J) continue;
            const FuseBox& b = B[j];
            const FuseBox& a = B[j - 1];
            const int l = k0 + j;
            const int dw = c.w0 >> l, dh = c.h0 >> l;            // this level
            const int aw = c.w0 >> (l - 1), ah = c.h0 >> (l - 1);  // the finer level it is made from
            const int gx = tid & 15;  // box rows are at most 64 bytes = 16 groups
            const uint8_t* A0 = box_lds + a.off;
            if (4 * gx < b.bw) {
                const int vx = b.bx0 + 4 * gx;
                const bool xfast = vx >= b.x0 && vx + 3 <= b.x1 && 2 * vx - 2 >= 0 && 2 * (vx + 3) + 2 <= aw - 1;
                for (int cy = tid >> 4; cy < b.bh; cy += 16) {
                    const int vy = b.by0 + cy;
                    unsigned packed = 0;
                    if (xfast && vy >= b.y0 && vy <= b.y1 && 2 * vy - 2 >= 0 && 2 * vy + 2 <= ah - 1) {
                        const uint8_t* S = A0 + (2 * vy - 2 - a.by0) * a.bw + (2 * vx - 4 - a.bx0);  // dword aligned
                        int v[4] = {128, 128, 128, 128};
#pragma unroll
                        for (int t = 0; t < 5; t++) {
                            const unsigned* r32 = reinterpret_cast<const unsigned*>(S + t * a.bw);
                            int h[4];
                            pyr_down_hrow(make_uint4(r32[0], r32[1], r32[2], r32[3]), h);
                            const int wt = t == 0 || t == 4 ? 1 : (t == 2 ? 6 : 4);
#pragma unroll
                            for (int k = 0; k < 4; k++) v[k] += h[k] * wt;
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) packed |= (unsigned)(v[k] >> 8) << (8 * k);  // no saturate: the taps sum to 256
                    } else {
                        const int qy = reflect101_idx(vy, dh);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int qx = reflect101_idx(vx + k, dw);
                            int out = 0;
                            if (qx >= b.x0 && qx <= b.x1 && qy >= b.y0 && qy <= b.y1) {
                                const int col = 2 * qx - 2 - a.bx0;  // even; the window is bytes col .. col + 4 of the row
                                const unsigned sh = (unsigned)col & 3u;
                                const uint8_t* S = A0 + (2 * qy - 2 - a.by0) * a.bw + (col & ~3);
                                int v = 128;
#pragma unroll
                                for (int t = 0; t < 5; t++) {
                                    const unsigned d0 = *reinterpret_cast<const unsigned*>(S + t * a.bw);
                                    const unsigned d1 = *reinterpret_cast<const unsigned*>(S + t * a.bw + 4);
                                    const unsigned lo4 = __builtin_amdgcn_alignbyte(d1, d0, sh);
                                    const unsigned t5 = (d1 >> (8 * sh)) & 0xffu;
                                    v += (int)__builtin_amdgcn_udot4(lo4, 0x04060401u, t5, false) * (t == 0 || t == 4 ? 1 : (t == 2 ? 6 : 4));
                                }
                                out = v >> 8;
                            }
                            packed |= (unsigned)out << (8 * k);
                        }
                    }
                    *reinterpret_cast<unsigned*>(box_lds + b.off + cy * b.bw + 4 * gx) = packed;
                }
            }
            __syncthreads();
        }
        // this camera's weighted Laplacian at the lane's pixels of every level
#pragma unroll
        for (int j = 0; j <= kFuseMaxJ; j++) {
            if (j > J) continue;
            const int l = k0 + j;
            const FuseBox& b = B[j];
            const FuseBox& u = B[j + 1 <= kFuseMaxJ ? j + 1 : j];
#pragma unroll
            for (int q = 0; q < fuse_q(j); q++) {
                if (wv[j][q] == 0.f) continue;
                int px, py;
                fuse_pixel(j, q, tid, rw[j], rh[j], px, py);
                const int x = rx0[j] + px - (c.tx >> l), y = ry0[j] + py - (c.ty >> l);
                W[j][q] += wv[j][q];
                int lap = box_lds[b.off + (y - b.by0) * b.bw + (x - b.bx0)];
                if (j < J) {
                    const uint8_t* S = box_lds + u.off - u.by0 * u.bw - u.bx0;  // (0, 0) of the real plane
                    lap = sat16i(lap - pyr_up_px<uint8_t>(S, (c.w0 >> l) >> 1, (c.h0 >> l) >> 1, u.bw, x, y));
                }
                acc[j][q] = (int16_t)(acc[j][q] + (int16_t)(int)((float)lap * wv[j][q]));
            }
        }
    }
    __syncthreads();
    // norm_l, then the collapse chain coarse -> fine through LDS (collapse_small_kernel's phase 2)
#pragma unroll
    for (int j = kFuseMaxJ; j >= 0; j--) {
        if (j > J) continue;  // block-uniform
        const int l = k0 + j;
        const int cw = C.w0 >> l, ch = C.h0 >> l;
        const int jc = j + 1 <= kFuseMaxJ ? j + 1 : j;
#pragma unroll
        for (int q = 0; q < fuse_q(j); q++) {
            int px, py;
            if (!fuse_pixel(j, q, tid, rw[j], rh[j], px, py)) continue;
            const int X = rx0[j] + px, Y = ry0[j] + py;
            int v;
            if (W[j][q] == 1.0f) v = toward_zero_by_one(acc[j][q]);
            else v = (int16_t)(int)((float)acc[j][q] / (W[j][q] + 1e-5f));
            if (j < J) {
                // out_{l+1} over its footprint, as a plane whose (0, 0) is canvas pixel (0, 0) of that level
                const int16_t* S = lds + ro[jc] - ry0[jc] * rw[jc] - rx0[jc];
                v = sat16i(v + pyr_up_px<int16_t>(S, cw >> 1, ch >> 1, rw[jc], X, Y));
            }
            if (j > 0) lds[ro[j] + py * rw[j] + px] = (int16_t)v;
            else C.img[l][(size_t)pl * C.cplane[l] + (size_t)Y * C.cpitch[l] + X] = (int16_t)v;
        }
        __syncthreads();
    }
}

void launch_blend_small(const PyrParams& p, const CanvasSet& cs, hipStream_t s) {
    const CanvasParams& c = cs.c[0];
    const int k0 = c.small_base;
    int cw = 0, ch = 0;
    for (int g = 0; g < cs.n; g++) {
        cw = max(cw, cs.c[g].w0 >> k0);
        ch = max(ch, cs.c[g].h0 >> k0);
    }
    dim3 block(64, 4, 1);
    if (c.small_fused) {
        dim3 gf((cw + kFuseTileW - 1) / kFuseTileW, (ch + kFuseTileH - 1) / kFuseTileH, cs.n * 3);
        hipLaunchKernelGGL(small_fused_kernel, gf, block, 0, s, p, cs);
        return;
    }
    if (c.small_merged) {  // normalise + collapse in one launch
        dim3 gm((cw + kSmallTileW - 1) / kSmallTileW, (ch + kSmallTileH - 1) / kSmallTileH, cs.n);
        hipLaunchKernelGGL(collapse_small_kernel<true>, gm, block, 0, s, p, cs);
        return;
    }
    dim3 g1((cw + 63) / 64, (ch + 3) / 4, (c.bands - k0 + 1) * cs.n);
    hipLaunchKernelGGL(norm_small_kernel, g1, block, 0, s, p, cs);
    dim3 g2((cw + kSmallTileW - 1) / kSmallTileW, (ch + kSmallTileH - 1) / kSmallTileH, cs.n);
    hipLaunchKernelGGL(collapse_small_kernel<false>, g2, block, 0, s, p, cs);
}

void launch_blend_level(const PyrParams& p, const CanvasSet& cs, int l, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
#define PANO_LAUNCH_L0(K, G)                                                                         \
    do {                                                                                             \
        if (ev_start && ev_stop) hipExtLaunchKernelGGL(K, G, block, 0, s, ev_start, ev_stop, 0, p, cs, karg); \
        else hipLaunchKernelGGL(K, G, block, 0, s, p, cs, karg);                                    \
    } while (0)
    const CanvasParams& c = cs.c[0];
    if (c.fast[l]) {
        int w = 0, h = 0;
        for (int g = 0; g < cs.n; g++) {
            const CanvasParams& cg = cs.c[g];
            if (l == 0) {
                w = max(w, cg.cut_x + cg.cut_w - (cg.cut_x & ~3));
                h = max(h, cg.cut_y + cg.cut_h - (cg.cut_y & ~1));
            } else {
                w = max(w, cg.w0 >> l);
                h = max(h, cg.h0 >> l);
            }
        }
        // workgroup shape (see the kernel): 3 = 2 x 2 waves in XCD bands (default), 2 = 2 x 2 waves, 0 = side by side, 1 = stacked
        const int shape_env = c.k3_shape & 3;
        // one plane per lane on the canvas levels >= 1 (measured: levels 1 + 2 29 -> 23 us, in flight no worse);
        // PANO_BLEND_PLANES=0 keeps three planes per lane
        const bool split = c.blend_split != 0;
        int shape = shape_env;
        if (shape == 3 && ((w + 127) / 128 > 1023 || (h + 15) / 16 > 1023)) shape = 2;  // the band form packs the extents in 10 bits each
        dim3 block(64, 4, 1), grid((w + 127) / 128, (h + 15) / 16, cs.n);
        if (shape == 0) grid = dim3((w + 255) / 256, (h + 7) / 8, cs.n);
        if (shape == 1) grid = dim3((w + 63) / 64, (h + 31) / 32, cs.n);
        int larg = l | (shape << 8);
        const dim3 grid3 = grid;  // the logical extents
        if (shape == 3) {       // XCD bands over the 2 x 2 shape: a 1-D grid of 8 * ceil(workgroups / 8)
            larg |= (int)((grid3.x & 0x3ffu) << 12) | (int)((grid3.y & 0x3ffu) << 22);
            const unsigned zext = (l == 0 || !split) ? cs.n : cs.n * 3;
            grid = dim3(8u * ((grid3.x * grid3.y * zext + 7u) / 8u), 1, 1);
        }
        // level 0 in strips of S blocks per lane (blend_level0_strip_kernel): opt-in, PANO_L0_STRIPS=2 / 4 / 8 pick S.  Measured on
        // config 2: half the vector instructions per pixel, yet 36.7 / 49.4 / 72.5 us against 31.4 us
        // for one block per lane alone and the same panoramas/s with frames in flight
        const int strips = c.l0_strips;
        if (l == 0 && c.bands >= 1 && (strips == 2 || strips == 4 || strips == 8)) {
            const unsigned sgx = (w + 127) / 128, sgy = (h + 16 * strips - 1) / (16 * strips);
            if (sgx <= 1023 && sgy <= 1023) {
                const int karg = (int)(sgx << 12) | (int)(sgy << 22);
                const dim3 sgrid(8u * ((sgx * sgy * cs.n + 7u) / 8u), 1, 1);
                if (strips == 2) PANO_LAUNCH_L0(blend_level0_strip_kernel<2>, sgrid);
                else if (strips == 4) PANO_LAUNCH_L0(blend_level0_strip_kernel<4>, sgrid);
                else PANO_LAUNCH_L0(blend_level0_strip_kernel<8>, sgrid);
                return;
            }
        }
        if (l == 0) {
            bool ordered = shape == 3;
            unsigned maxper = 0;
            for (int g = 0; g < cs.n; g++) {
                ordered = ordered && cs.c[g].order0 != nullptr;
                maxper = max(maxper, (unsigned)cs.c[g].order_per);
            }
            if (ordered && maxper > 0 && maxper < (1u << 20)) {  // XCD bands, seam tiles first
                const int karg = (4 << 8) | (int)(maxper << 12);
                PANO_LAUNCH_L0((blend_level_vec_kernel<true, 3>), dim3(8u * maxper * cs.n, 1, 1));
            } else {
                const int karg = larg;
                PANO_LAUNCH_L0((blend_level_vec_kernel<true, 3>), grid);
            }
        } else {
            if (split) hipLaunchKernelGGL((blend_level_vec_kernel<false, 1>), shape == 3 ? grid : dim3(grid.x, grid.y, cs.n * 3), block, 0, s, p, cs, larg);
            else hipLaunchKernelGGL((blend_level_vec_kernel<false, 3>), grid, block, 0, s, p, cs, larg);
        }
        return;
    }
    int w = 0, h = 0;
    for (int g = 0; g < cs.n; g++) {
        w = max(w, l == 0 ? cs.c[g].cut_w : (cs.c[g].w0 >> l));
        h = max(h, l == 0 ? cs.c[g].cut_h : (cs.c[g].h0 >> l));
    }
    dim3 block(64, 4, 1), grid((w + 63) / 64, (h + 3) / 4, cs.n);
    hipLaunchKernelGGL(blend_level_kernel, grid, block, 0, s, p, cs, l);
}

// Blender::NO: Blender::feed masked copy in feed order, Blender::blend zeroing, convertTo(8U), cut.
// The level-0 tile is the ROI itself (no border) and mask0 the blend mask.
__global__ __launch_bounds__(256) void no_blend_kernel(PyrParams P, CanvasSet CS) {
    const CanvasParams& C = CS.c[blockIdx.z];
    const int cam_lo = C.cam_lo, cam_n = C.cam_n;
    int X = blockIdx.x * 64 + threadIdx.x;
    int Y = blockIdx.y * 4 + threadIdx.y;
    if (X >= C.cut_w || Y >= C.cut_h) return;
    X += C.cut_x;
    Y += C.cut_y;
    int v[3] = {0, 0, 0};
    for (int i = 0; i < cam_n; i++) {
        const PyrCam& c = P.cam[cam_lo + i];
        const int x = X - c.tx, y = Y - c.ty;
        if ((unsigned)x >= (unsigned)c.w0 || (unsigned)y >= (unsigned)c.h0) continue;
        if (!c.mask0[(size_t)y * c.pitch[0] + x]) continue;
#pragma unroll
        for (int k = 0; k < 3; k++) v[k] = c.lvl[0][(size_t)k * c.plane[0] + (size_t)y * c.pitch[0] + x];
    }
    uint8_t* d = C.out + (size_t)(Y - C.cut_y) * C.out_stride + 3 * (X - C.cut_x);
    d[0] = (uint8_t)v[0];
    d[1] = (uint8_t)v[1];
    d[2] = (uint8_t)v[2];
}
void launch_no_blend(const PyrParams& p, const CanvasSet& cs, hipStream_t s) {
    int w = 0, h = 0;
    for (int g = 0; g < cs.n; g++) {
        w = max(w, cs.c[g].cut_w);
        h = max(h, cs.c[g].cut_h);
    }
    dim3 block(64, 4, 1), grid((w + 63) / 64, (h + 3) / 4, cs.n);
    hipLaunchKernelGGL(no_blend_kernel, grid, block, 0, s, p, cs);
}

// ------------------------------------------------------------------------------------------------
// weights: mask * (1/255.f) with copyMakeBorder(CONSTANT 0); pyrDown CV_32F (scalar evaluation order);
// canvas sum in feed order
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_to_weight_kernel(const uint8_t* mask, int mw, int mh, int mpitch, int left,
                                                             int top, float* w0, int wpitch, uint8_t* m0, int mpitch0,
                                                             int tw, int th) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= tw || y >= th) return;
    const int sx = x - left, sy = y - top;
    uint8_t mv = 0;
    if ((unsigned)sx < (unsigned)mw && (unsigned)sy < (unsigned)mh) mv = mask[(size_t)sy * mpitch + sx];
    m0[(size_t)y * mpitch0 + x] = mv;
    w0[(size_t)y * wpitch + x] = (float)mv * (float)(1. / 255.);
}
void launch_mask_to_weight(const uint8_t* mask, int mw, int mh, int mpitch, int left, int top, float* w0, int wpitch,
                           uint8_t* m0, int mpitch0, int tw, int th, hipStream_t s) {
    dim3 block(64, 4, 1), grid((tw + 63) / 64, (th + 3) / 4, 1);
    hipLaunchKernelGGL(mask_to_weight_kernel, grid, block, 0, s, mask, mw, mh, mpitch, left, top, w0, wpitch, m0, mpitch0,
                       tw, th);
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
        W += c.wgt[l][(size_t)y * c.wpitch[l] + x];
    }
    wsum[(size_t)Y * cw + X] = W;
}
void launch_sum_weights(const PyrParams& p, int l, float* wsum, int cw, int ch, hipStream_t s) {
    dim3 block(64, 4, 1), grid((cw + 63) / 64, (ch + 3) / 4, 1);
    hipLaunchKernelGGL(sum_weights_kernel, grid, block, 0, s, p, l, wsum, cw, ch);
}

// ------------------------------------------------------------------------------------------------
// caller-side assembly (src/master.cpp:321-326, src/panocamimpl.cpp:354-360): optional cv::resize INTER_LINEAR of
// the upper half (CV_8U: short coefficients x2048 from fx = (float)((dx+0.5)*scale-0.5), int horizontal pass,
// ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2 >> 2 vertical pass), vconcat, black divider
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void linear_coef_8u(int d, int ssize, int dsize, bool clamp_edge, int& s0, int& s1, int& a0,
                                               int& a1) {
    const double scale = (double)ssize / dsize;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_edge) {  // horizontal: fx is reset at the borders
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    }
    a0 = sat16i(cv_round_dev((1.f - f) * 2048.f));
    a1 = sat16i(cv_round_dev(f * 2048.f));
    s0 = min(max(s, 0), ssize - 1);
    s1 = min(max(s + 1, 0), ssize - 1);
}
__global__ __launch_bounds__(256) void stack_kernel(const uint8_t* up, int up_w, int up_h, int up_stride, int up_y0,
                                                    int resize_up, const uint8_t* down, int down_stride, int down_y0,
                                                    uint8_t* out, int out_w, int top_h, int out_stride, int bar_y,
                                                    int bar_h) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= out_w || y >= 2 * top_h) return;
    int v[3] = {0, 0, 0};
    if (y < bar_y || y >= bar_y + bar_h) {
        if (y >= top_h) {
            const uint8_t* p = down + (size_t)(y - top_h + down_y0) * down_stride + 3 * x;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        } else if (!resize_up) {
            const uint8_t* p = up + (size_t)(y + up_y0) * up_stride + 3 * x;
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
        } else {
            int x0, x1, a0, a1, y0, y1, b0, b1;
            linear_coef_8u(x, up_w, out_w, true, x0, x1, a0, a1);
            linear_coef_8u(y, up_h, top_h, false, y0, y1, b0, b1);
            const uint8_t* r0 = up + (size_t)y0 * up_stride;
            const uint8_t* r1 = up + (size_t)y1 * up_stride;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int h0 = r0[3 * x0 + c] * a0 + r0[3 * x1 + c] * a1;
                const int h1 = r1[3 * x0 + c] * a0 + r1[3 * x1 + c] * a1;
                v[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            }
        }
    }
    uint8_t* d = out + (size_t)y * out_stride + 3 * x;
    d[0] = (uint8_t)v[0];
    d[1] = (uint8_t)v[1];
    d[2] = (uint8_t)v[2];
}
void launch_stack(const uint8_t* up, int up_w, int up_h, int up_stride, int up_y0, bool resize_up, const uint8_t* down,
                  int down_stride, int down_y0, uint8_t* out, int out_w, int top_h, int out_stride, int bar_y, int bar_h,
                  hipStream_t s) {
    dim3 block(64, 4, 1), grid((out_w + 63) / 64, (2 * top_h + 3) / 4, 1);
    hipLaunchKernelGGL(stack_kernel, grid, block, 0, s, up, up_w, up_h, up_stride, up_y0, resize_up ? 1 : 0, down,
                       down_stride, down_y0, out, out_w, top_h, out_stride, bar_y, bar_h);
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

// cv::resize INTER_LINEAR_EXACT CV_8UC1 / CV_8UC3: 8.8 horizontal, 16.16 vertical; coefficient tables from the host
template <int CN>
__global__ __launch_bounds__(256) void resize_linear_exact_kernel(const uint8_t* src, int sw, int sh, uint8_t* dst,
                                                                  int dw, int dh, const int* xofs, const int* xc1,
                                                                  const int* yofs, const int* yc1, int minx, int maxx,
                                                                  int miny, int maxy) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
#pragma unroll
    for (int ch = 0; ch < CN; ch++) {
        auto hval = [&](int row) -> unsigned {
            const uint8_t* s = src + (size_t)row * sw * CN + ch;
            if (x < minx) return (unsigned)s[0] << 8;
            if (x >= maxx) return (unsigned)s[(sw - 1) * CN] << 8;
            int o = xofs[x], c1 = xc1[x];
            unsigned v = s[o * CN] * (unsigned)(256 - c1) + s[(o + 1) * CN] * (unsigned)c1;
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
        dst[((size_t)y * dw + x) * CN + ch] = (uint8_t)sat8i(out);
    }
}
void launch_resize_linear_exact(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh, const int* xofs,
                                const int* xc1, const int* yofs, const int* yc1, int minx, int maxx, int miny, int maxy,
                                hipStream_t s) {
    dim3 block(64, 4, 1), grid((dw + 63) / 64, (dh + 3) / 4, 1);
    if (cn == 3)
        hipLaunchKernelGGL(resize_linear_exact_kernel<3>, grid, block, 0, s, src, sw, sh, dst, dw, dh, xofs, xc1, yofs, yc1,
                           minx, maxx, miny, maxy);
    else
        hipLaunchKernelGGL(resize_linear_exact_kernel<1>, grid, block, 0, s, src, sw, sh, dst, dw, dh, xofs, xc1, yofs, yc1,
                           minx, maxx, miny, maxy);
}

// GraphCutSeamFinder::Impl::findInPair, the pixel work in front of the max-flow (seam_finders.cpp, COST_COLOR): the
// overlap ROI of images a and b padded by gap = 10 on every side is a W x H grid graph.  Per vertex the terminal weight
// (10000 towards the source where mask a is set, towards the sink where mask b is set), per horizontal / vertical
// neighbour pair the capacity |a - b|^2(v) + |a - b|^2(v') + 1, plus 1000 where any of the four mask samples is clear.
// Images are the 8UC3 seam-scale warps (the reference converts them to f32 first; the values are the same integers).
struct GcSample {
    float nd;  // squared colour distance of the two images at the vertex (0 outside either image)
    bool ma, mb;
};
__device__ __forceinline__ GcSample gc_sample(const GainImages& g, const GcPair& q, int x, int y) {
    GcSample r;
    const int xa = q.ax + x, ya = q.ay + y, xb = q.bx + x, yb = q.by + y;
    float pa[3] = {0.f, 0.f, 0.f}, pb[3] = {0.f, 0.f, 0.f};
    r.ma = r.mb = false;
    if (xa >= 0 && ya >= 0 && xa < q.wa && ya < q.ha) {
        const uint8_t* p = g.img[q.a] + ((size_t)ya * q.wa + xa) * 3;
        pa[0] = p[0]; pa[1] = p[1]; pa[2] = p[2];
        r.ma = g.mask[q.a][(size_t)ya * q.wa + xa] != 0;
    }
    if (xb >= 0 && yb >= 0 && xb < q.wb && yb < q.hb) {
        const uint8_t* p = g.img[q.b] + ((size_t)yb * q.wb + xb) * 3;
        pb[0] = p[0]; pb[1] = p[1]; pb[2] = p[2];
        r.mb = g.mask[q.b][(size_t)yb * q.wb + xb] != 0;
    }
    const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
    r.nd = dx * dx + dy * dy + dz * dz;
    return r;
}
__global__ __launch_bounds__(256) void graphcut_weights_kernel(GainImages g, GcPair q, float* __restrict__ term,
                                                               float* __restrict__ wh, float* __restrict__ wv) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= q.W || y >= q.H) return;
    const GcSample c = gc_sample(g, q, x, y);
    const int v = y * q.W + x;
    term[v] = (c.ma ? 10000.f : 0.f) - (c.mb ? 10000.f : 0.f);
    float h = 0.f, d = 0.f;
    if (x < q.W - 1) {
        const GcSample n = gc_sample(g, q, x + 1, y);
        h = c.nd + n.nd + 1.f;
        if (!c.ma || !n.ma || !c.mb || !n.mb) h += 1000.f;
    }
    if (y < q.H - 1) {
        const GcSample n = gc_sample(g, q, x, y + 1);
        d = c.nd + n.nd + 1.f;
        if (!c.ma || !n.ma || !c.mb || !n.mb) d += 1000.f;
    }
    wh[v] = h;
    wv[v] = d;
}
void launch_graphcut_weights(const GainImages& g, const GcPair& q, float* term, float* wh, float* wv, hipStream_t s) {
    dim3 block(64, 4, 1), grid((q.W + 63) / 64, (q.H + 3) / 4, 1);
    hipLaunchKernelGGL(graphcut_weights_kernel, grid, block, 0, s, g, q, term, wh, wv);
}
// ... and behind it: inside the ROI a vertex of the source segment keeps image a (mask b is cleared where mask a is
// set), a vertex of the sink segment keeps image b
__global__ __launch_bounds__(256) void graphcut_apply_kernel(GcPair q, uint8_t* mask_a, uint8_t* mask_b,
                                                             const uint8_t* __restrict__ in_source, int gap) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= q.W - 2 * gap || y >= q.H - 2 * gap) return;
    // ROI pixel (x, y) is padded-grid vertex (x + gap, y + gap) and image pixel (ax + gap + x, ay + gap + y)
    const size_t ka = (size_t)(q.ay + gap + y) * q.wa + (q.ax + gap + x), kb = (size_t)(q.by + gap + y) * q.wb + (q.bx + gap + x);
    if (in_source[(y + gap) * q.W + x + gap]) {
        if (mask_a[ka]) mask_b[kb] = 0;
    } else {
        if (mask_b[kb]) mask_a[ka] = 0;
    }
}
void launch_graphcut_apply(const GcPair& q, uint8_t* mask_a, uint8_t* mask_b, const uint8_t* in_source, int gap, hipStream_t s) {
    dim3 block(64, 4, 1), grid((q.W - 2 * gap + 63) / 64, (q.H - 2 * gap + 3) / 4, 1);
    hipLaunchKernelGGL(graphcut_apply_kernel, grid, block, 0, s, q, mask_a, mask_b, in_source, gap);
}

// detail::GainCompensator::feed, the pixel loop of one overlapping pair of sub-images (exposure_compensate.cpp):
// count of pixels both masks mark and the two sums of sqrt(b^2 + g^2 + r^2) over them.  The sums are f64 and
// f64 addition does not reassociate, so one lane walks one pair in the reference's row-major order; the pairs (a few
// thousand 32 x 32 blocks, once per mask refresh) are the parallel axis.  sqrt(f64) is correctly rounded on gfx950.
__global__ __launch_bounds__(64) void gain_pair_kernel(GainImages g, const GainPair* __restrict__ pairs, int npairs,
                                                       int* __restrict__ count, double* __restrict__ sum_a,
                                                       double* __restrict__ sum_b) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= npairs) return;
    const GainPair q = pairs[p];
    const int wa = g.w[q.a], wb = g.w[q.b];
    int n = 0;
    double sa = 0.0, sb = 0.0;
    for (int y = 0; y < q.h; y++) {
        const uint8_t* ra = g.img[q.a] + ((size_t)(q.ay + y) * wa + q.ax) * 3;
        const uint8_t* rb = g.img[q.b] + ((size_t)(q.by + y) * wb + q.bx) * 3;
        const uint8_t* ma = g.mask[q.a] + (size_t)(q.ay + y) * wa + q.ax;
        const uint8_t* mb = g.mask[q.b] + (size_t)(q.by + y) * wb + q.bx;
        for (int x = 0; x < q.w; x++) {
            if (ma[x] != 255 || mb[x] != 255) continue;
            n++;
            const int a0 = ra[3 * x], a1 = ra[3 * x + 1], a2 = ra[3 * x + 2];
            const int b0 = rb[3 * x], b1 = rb[3 * x + 1], b2 = rb[3 * x + 2];
            sa += __builtin_sqrt((double)(a0 * a0 + a1 * a1 + a2 * a2));
            sb += __builtin_sqrt((double)(b0 * b0 + b1 * b1 + b2 * b2));
        }
    }
    count[p] = n;
    sum_a[p] = sa;
    sum_b[p] = sb;
}
void launch_gain_pairs(const GainImages& g, const GainPair* pairs, int npairs, int* count, double* sum_a, double* sum_b,
                       hipStream_t s) {
    if (npairs < 1) return;
    hipLaunchKernelGGL(gain_pair_kernel, dim3((npairs + 63) / 64), dim3(64), 0, s, g, pairs, npairs, count, sum_a, sum_b);
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
