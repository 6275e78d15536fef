// pano_warp.hip - K1: the fused warp (remap table build / pack, table and on-the-fly kernels, stage warps)
// Device helpers: pano_dev.hpp; launch interface: pano_kernels.hpp.  Compile with -ffp-contract=off.

#include "pano_dev.hpp"

namespace pano {

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
// rows == 0 (and their extent in the fourth word) and tap global memory instead; the origin is valid either way.
constexpr int kBoxBytes = 16 * 1024;             // LDS per workgroup, one spare row included
constexpr int kBoxIters = kBoxBytes / 16 / 256;  // 16-byte chunk loads per lane, at most
static_assert(kBoxBytes % (16 * 256) == 0, "a box of kBoxBytes is copied in whole iterations of 256 lanes x 16 bytes");
// (round 5, timing: 8 / 10 / 12 KB push more patches to global taps - 19.5 / 18.7 / 18.15 us against 17.9 at 16 KB; 20 KB: no change)
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
    // a box without LDS carries its extent instead (x span << 16 | y span): what the host needs to know which frame bytes
    // the patch's global taps can touch (live_source_rects in pano_api.cpp)
    boxes[blockIdx.y * gx + blockIdx.x] = ok ? make_int4(xmin, ymin, (h << 8) | cpr, (65536 + cpr - 1) / cpr)
                                             : make_int4(xmin, ymin, 0, (min(lim[1] - xmin, 32767) << 16) | min(lim[3] - ymin, 65535));
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
// The deal (WarpDeal) travels as nine scalar arguments IN FRONT of the camera blocks: gfx950 preloads the first kernel arguments into
// SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=9), so finding the block costs no memory round trip.
constexpr unsigned kWarpParamsAt = 64;  // offset of P in the kernarg segment: 9 dwords, then alignof(WarpParams)
template <bool GAIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void warp_tiles_lut_kernel(
    unsigned de0, unsigned de1, unsigned de2, unsigned de3, unsigned de4, unsigned de5, unsigned de6, unsigned de7, unsigned dper, WarpParams P) {
    static_assert(alignof(WarpParams) == 64, "kWarpParamsAt");
    __shared__ uint4 sbox[kBoxBytes / 16];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // GAIN: the workgroup's 64 columns of up to kGainRows rows of ghrow, then its 16 row entries {sy0, sy1, 1 - fy, fy} (as bits)
    __shared__ f32x4 sgain[GAIN ? kGainRows * 16 + 16 : 1];
    // grid = (8, blocks per XCD): linear workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2), so
    // blockIdx.x IS the XCD, and XCD k works through the k-th eighth of the list of live blocks (WarpDeal, pano_kernels.hpp):
    // with 8 (or 4, 2) cameras an XCD's L2 holds one camera's frame (FETCH_SIZE 50.8 against 77.8 MB for camera-major
    // dispatch, 33.1 against 36.0 us in round 1), every camera advances top to bottom with all XCDs busy, and no XCD has
    // more blocks than another.
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
    // Which block is this?  (WarpDeal)  blockIdx.x = XCD, blockIdx.y = position in the XCD's share of the list of live blocks
    const unsigned dend[kCams] = {de0, de1, de2, de3, de4, de5, de6, de7};
    static_assert(kCams == 8, "eight end arguments");
    const unsigned lin = blockIdx.x * dper + blockIdx.y;
    if (lin >= dend[kCams - 1]) return;  // past the end of the list (the whole workgroup leaves)
    unsigned ci = 0, dstart = 0;
#pragma unroll
    for (int c = 0; c < kCams - 1; c++) {
        if (lin >= dend[c]) {
            ci = c + 1;
            dstart = dend[c];
        }
    }
    const char __attribute__((address_space(4)))* ka =
        (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + (kWarpParamsAt + ci * sizeof(WarpCam));
    // GAIN: the camera's gain block {gain, grow_base, ghrow, grow4 | ghrow_pitch, gh} rides along (two more scalar loads in
    // flight with the first), so that the workgroup's gain rows can be requested together with its source box
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    struct GainBlk {
        const float* gain; const int* grow_base; const float* ghrow; const void* grow4;
    };
    union { i32x8 v; GainBlk g; } gq;
    i32x2 gq2 = {0, 0};  // {ghrow_pitch, gh}
    gq.v = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
    u32x2 dcm;  // {deal_cols, deal_mcols}
    static_assert(offsetof(WarpCam, deal_cols) == 0x78 && offsetof(WarpCam, deal_mcols) == 0x7c, "deal columns layout");
    if (GAIN) {
        static_assert(sizeof(GainBlk) == 32 && offsetof(WarpCam, gain) == 80 && offsetof(WarpCam, grow_base) == 88 && offsetof(WarpCam, ghrow) == 96 &&
                          offsetof(WarpCam, grow4) == 104 && offsetof(WarpCam, ghrow_pitch) == 112 && offsetof(WarpCam, gh) == 116, "gain block layout");
        // every batch of scalar loads and its wait are ONE asm statement with early-clobber outputs: between two statements the
        // compiler may copy a register whose load has not landed yet
        asm volatile("s_load_dwordx8 %0, %4, 0x50\n\ts_load_dwordx2 %1, %4, 0x70\n\ts_load_dwordx2 %2, %4, 0x78\n\ts_load_dwordx16 %3, %4, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&s"(gq.v), "=&s"(gq2), "=&s"(dcm), "=&s"(hot.v) : "s"(ka) : "memory");
    } else {
        asm volatile("s_load_dwordx2 %0, %2, 0x78\n\ts_load_dwordx16 %1, %2, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(dcm), "=&s"(hot.v) : "s"(ka) : "memory");
    }
    // the camera block as a plain pointer for the rare out-of-line paths (a reference to the by-value kernel argument
    // would make the compiler copy all of WarpParams to scratch)
    const WarpCam* const cg = (const WarpCam*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + (kWarpParamsAt + ci * sizeof(WarpCam)));
    // the pointers come out of the asm block without an address space: say "global", or every access is a flat_load
#define PANO_GLOBAL __attribute__((address_space(1)))
    const uint8_t PANO_GLOBAL* const src = (const uint8_t PANO_GLOBAL*)hot.h.src;
    uint8_t PANO_GLOBAL* const dst = (uint8_t PANO_GLOBAL*)hot.h.dst;
    const u32x2 PANO_GLOBAL* const lutc = (const u32x2 PANO_GLOBAL*)hot.h.lutc;
    const int tw = hot.h.tw, th = hot.h.th;
    // the grid is laid over the live blocks of the camera: workgroup (0, 0) is block (live_bx0, live_by0); the dead
    // block columns in the middle of a +-pi straddler (gap_len from gap_bx0 on, else 0) are stepped over
    const unsigned lg = (unsigned)hot.h.live_by0_gap;
    const unsigned dl = lin - dstart;                                   // position among this camera's live blocks, row after row
    const unsigned drow = dcm.y ? __umulhi(dl, dcm.y) : dl;             // dl / deal_cols
    int bx = hot.h.live_bx0 + (int)(dl - drow * dcm.x);
    const int by = (int)(lg & 0xfffu) + (int)drow;
    if (bx >= (int)((lg >> 12) & 0x3ffu)) bx += (int)(lg >> 22);
    const unsigned stride = (unsigned)hot.h.src_stride;
    const unsigned dst_pitch = (unsigned)hot.h.dst_pitch, dst_plane = (unsigned)hot.h.dst_plane, lutc_pitch = (unsigned)hot.h.lutc_pitch;
    const int gxc = (tw + 63) >> 6;
    if (bx >= gxc || by * 16 >= th) return;  // the whole workgroup leaves: nobody waits at the barrier below
    // the blocks anything downstream reads (4 ints behind the hot part) and this workgroup's source box: two more
    // scalar loads, issued together
    i32x4 live;  // {live_bx1, live_by1, src_w, src_h}
    static_assert(offsetof(WarpCam, live_bx1) == 64 && offsetof(WarpCam, src_h) == 76, "second scalar load layout");
    i32x4 bb;
    const char __attribute__((address_space(4)))* const bp =
        (const char __attribute__((address_space(4)))*)hot.h.box + (unsigned)(by * gxc + bx) * 16u;
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
    // Exposure gains (GAIN instantiation; per camera, workgroup-uniform).  BlocksGainCompensator::apply resizes the block map to
    // the tile (cv::resize INTER_LINEAR, f32): the horizontal pass is a table (ghrow: map rows x tile columns), the vertical pass
    // g = h0 * (1 - fy) + h1 * fy is done here.  Fetched per lane that was four more vector loads per wave (two per-row entries,
    // two 16-byte map reads: 8 bytes of L2 traffic per pixel for 3 bytes of output) in a kernel whose waves issue six - config
    // 4 ran 76 us with gains against 59 without.  A 16-row patch reads the same two or three rows of ghrow in every one of its
    // rows (a block is 32 pixels high), so the workgroup copies its 64 columns of those rows (1 KB) and its 16 row entries
    // (256 B) into LDS with ONE vector load in each of two waves, and every lane reads its row entry and four gains from there.
    const bool gains = GAIN && gq.g.gain != nullptr;
    int gbase = -1;
    // `live`, the source box and (GAIN) the patch's first gain row - workgroup-uniform scalar loads, spelled out (left to the compiler
    // the last became a vector load with a wait behind it) - go out together BEHIND the request of the lane's table entry above, and
    // are waited for in the same asm statement (early-clobber outputs: nothing can touch the registers before the loads have landed)
    if (GAIN) {
        // a camera without gains reads a dword of its own kernarg block instead (dropped below): the statement stays unconditional
        const char __attribute__((address_space(4)))* const gbp =
            gains ? (const char __attribute__((address_space(4)))*)gq.g.grow_base + (unsigned)by * 4u : ka;
        asm volatile("s_load_dwordx4 %0, %3, 0x40\n\ts_load_dwordx4 %1, %4, 0x0\n\ts_load_dword %2, %5, 0x0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(live), "=&s"(bb), "=&s"(gbase) : "s"(ka), "s"(bp), "s"(gbp) : "memory");
        if (!gains) gbase = -1;
    } else {
        asm volatile("s_load_dwordx4 %0, %2, 0x40\n\ts_load_dwordx4 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(live), "=&s"(bb) : "s"(ka), "s"(bp) : "memory");
    }
    if (bx > live.x || by > live.y) return;  // beyond this camera's live blocks (workgroup-uniform)
    const int src_w = live.z, src_h = live.w;
    const int bh = bb.z >> 8, cpr = bb.z & 255;
    const unsigned lpitch = (unsigned)cpr * 16u;  // rows packed: chunk k = r * cpr + ci lands at LDS byte 16 * k
    // byte phase of the box origin inside its first 16-byte chunk; the same for every row because stride % 16 == 0
    const unsigned lo16 = (unsigned)(size_t)hot.h.src & 15u;
    const unsigned og = (unsigned)bb.y * stride + 3u * (unsigned)bb.x + lo16;  // from the 16-byte boundary below src
    const unsigned ph = og & 15u;
    // the workgroup's gain rows and row entries: requested BEFORE the box copy so that one wait covers both (behind it they
    // were a second round trip in front of the barrier: 9 us of a 75 us launch)
    f32x4 gstage = {0.f, 0.f, 0.f, 0.f};
    if (GAIN && gbase >= 0 && tid < kGainRows * 16 + 16) {
        // lane t < 64: row gbase + t / 16 of ghrow, columns 64 bx + 4 (t % 16) .. + 3 (the table's rows end on a multiple of
        // four columns: a chunk past the end repeats the last one - only inactive lanes would read it); lanes 64 .. 79: the row
        // entries of tile rows 16 by .. 16 by + 15.  One load instruction for both
        const unsigned gp = (unsigned)gq2.x;
        const unsigned row = (unsigned)min(gbase + (tid >> 4), gq2.y - 1), c4 = min((unsigned)(bx * 16 + (tid & 15)), (gp >> 2) - 1u);
        const unsigned ry = (unsigned)min(by * 16 + tid - kGainRows * 16, th - 1);
        const char PANO_GLOBAL* a = tid < kGainRows * 16 ? (const char PANO_GLOBAL*)gq.g.ghrow + ((row * gp + c4 * 4u) << 2)
                                                         : (const char PANO_GLOBAL*)gq.g.grow4 + (ry << 4);
        gstage = *reinterpret_cast<const f32x4 PANO_GLOBAL*>(a);
    }
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
                    16, 0, 0);  // aux 2 (nt: a frame is read once) gains 0.6 - 1.2 us per frame with four in flight and costs this kernel 1.5 us on its own (cold 20.25 -> 21.7 us): not set
            }
        }
    }
    // Everything that needs the table entry but not the box - the four codes and the LDS byte offsets of their taps - is done
    // HERE, in front of the barrier: a wave whose copies have landed waits for the slowest wave of its workgroup anyway (45 % of
    // this kernel's wave cycles are spent parked), and a third of its vector instructions fit into that wait
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
    unsigned loff[4] = {0u, 0u, 0u, 0u};
    if (bh) {
#pragma unroll
        for (int j = 0; j < 4; j++) loff[j] = __umul24(Y[j] >> 5, lpitch) + __umul24(X[j] >> 5, 3u) + ph;  // codes are box relative
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (GAIN && gbase >= 0 && tid < kGainRows * 16 + 16) sgain[tid] = gstage;
    if (bh || (GAIN && gbase >= 0)) __syncthreads();  // workgroup-uniform
    if (!active) return;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 gh0 = {0.f, 0.f, 0.f, 0.f}, gh1 = gh0;
    float2 gby = make_float2(0.f, 0.f);
    if (gains) {
        if (gbase >= 0) {  // the lane's row entry and the four columns of its two map rows, from the workgroup's copy
            const f32x4 re = sgain[kGainRows * 16 + threadIdx.y * 4 + (threadIdx.x >> 4)];
            gby = make_float2(re.z, re.w);
            gh0 = sgain[(__float_as_int(re.x) - gbase) * 16 + (int)(threadIdx.x & 15)];
            gh1 = sgain[(__float_as_int(re.y) - gbase) * 16 + (int)(threadIdx.x & 15)];
        } else {           // a map too fine for that (more than kGainRows rows per patch): everything per lane from global memory
            const int2 gy = cg->grow[y];
            gby = cg->groww[y];
            const float PANO_GLOBAL* gr = (const float PANO_GLOBAL*)gq.g.ghrow;
            const unsigned gp = (unsigned)gq2.x;
            gh0 = *reinterpret_cast<const f32x4 PANO_GLOBAL*>(gr + ((unsigned)gy.x * gp + (unsigned)x0));
            gh1 = *reinterpret_cast<const f32x4 PANO_GLOBAL*>(gr + ((unsigned)gy.y * gp + (unsigned)x0));
        }
    }
    uint8_t PANO_GLOBAL* d = dst + ((unsigned)y * dst_pitch + (unsigned)x0);  // 32-bit offsets: a tile is far below 4 GB
    uint2 t[4], u[4];
    if (bh) {
        // LDS byte offset of a pixel: (ys - ymin) * lpitch + 3 * (xs - xmin) + ph.  Three aligned dwords and
        // v_alignbyte, like the global taps: 8-byte ds reads at odd addresses work but run the kernel at half speed.
        const unsigned* sb = reinterpret_cast<const unsigned*>(sbox);
        // all sixteen LDS reads of the lane's four pixels go out BEFORE the first is used (the scheduler, left alone, waits for
        // each pixel's four reads before it issues the next pixel's: four LDS round trips per wave one after the other)
        unsigned k[4], ra[4][3], rb[4][3];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            k[j] = loff[j] & 3u;
            const unsigned* wt = sb + (loff[j] >> 2);
            const unsigned* wu = wt + (lpitch >> 2);  // row ys + 1; for ys == sh - 1 (weight 0) the spare row
#pragma unroll
            for (int q = 0; q < 3; q++) {
                ra[j][q] = wt[q];
                rb[j][q] = wu[q];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t[j] = make_uint2(__builtin_amdgcn_alignbyte(ra[j][1], ra[j][0], k[j]), __builtin_amdgcn_alignbyte(ra[j][2], ra[j][1], k[j]));
            u[j] = make_uint2(__builtin_amdgcn_alignbyte(rb[j][1], rb[j][0], k[j]), __builtin_amdgcn_alignbyte(rb[j][2], rb[j][1], k[j]));
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
    if (gains) {
        // VResizeLinear: g = h0 * (1 - fy) + h1 * fy, two pixels per instruction (v_pk_mul_f32 / v_pk_add_f32: IEEE per half, no
        // contraction - the same three roundings as the scalar expression)
        const f32x2 b0 = {gby.x, gby.x}, b1 = {gby.y, gby.y};
        const f32x2 g01 = f32x2{gh0.x, gh0.y} * b0 + f32x2{gh1.x, gh1.y} * b1;
        const f32x2 g23 = f32x2{gh0.z, gh0.w} * b0 + f32x2{gh1.z, gh1.w} * b1;
        // saturate_cast<uchar>(px * gain) = round-half-even + clamp is what v_cvt_pk_u8_f32 does, and it drops the byte
        // where the plane dword wants it: v_cvt_f32_ubyte2 per value, one v_pk_mul_f32 per two, v_cvt_pk_u8_f32 per value
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const f32x2 p01 = f32x2{(float)((r[0][ch] >> 16) & 0xffu), (float)((r[1][ch] >> 16) & 0xffu)} * g01;
            const f32x2 p23 = f32x2{(float)((r[2][ch] >> 16) & 0xffu), (float)((r[3][ch] >> 16) & 0xffu)} * g23;
            pk[ch] = __builtin_amdgcn_cvt_pk_u8_f32(p01.x, 0u, 0u);
            pk[ch] = __builtin_amdgcn_cvt_pk_u8_f32(p01.y, 1u, pk[ch]);
            pk[ch] = __builtin_amdgcn_cvt_pk_u8_f32(p23.x, 2u, pk[ch]);
            pk[ch] = __builtin_amdgcn_cvt_pk_u8_f32(p23.y, 3u, pk[ch]);
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
    // the LDS kernel's own deal (WarpDeal): every camera's live blocks in one list, an eighth of it per XCD
    WarpParams q = p;
    WarpDeal deal;
    {
        unsigned total = 0;
        for (int i = 0; i < kCams; i++) {
            if (i < ncam) {
                WarpCam& c = q.cam[i];
                const int nbx = (c.tw + 63) / 64, nby = (c.th + 15) / 16;
                const int cols = min(c.live_bx1, nbx - 1) - c.live_bx0 + 1 - (int)((unsigned)c.live_by0_gap >> 22);
                const int rows = min(c.live_by1, nby - 1) - (c.live_by0_gap & 0xfff) + 1;
                c.deal_cols = (unsigned)max(cols, 1);
                c.deal_mcols = c.deal_cols > 1 ? (unsigned)((1ull << 32) / c.deal_cols + 1ull) : 0u;
                if (cols > 0 && rows > 0) total += (unsigned)cols * (unsigned)rows;
            }
            deal.end[i] = total;
        }
        deal.per = max((total + 7u) / 8u, 1u);
    }
    const dim3 grid_deal(8, deal.per, 1);
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
        if (ev_start && ev_stop) hipExtLaunchKernelGGL(K, G, block, 0, s, ev_start, ev_stop, 0, q); \
        else hipLaunchKernelGGL(K, G, block, 0, s, q);                                        \
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
#define PANO_LAUNCH_K1_DEAL(K)                                                                                                    \
    do {                                                                                                                          \
        const unsigned* e = deal.end;                                                                                             \
        if (ev_start && ev_stop)                                                                                                  \
            hipExtLaunchKernelGGL(K, grid_deal, block, 0, s, ev_start, ev_stop, 0, e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], deal.per, q); \
        else hipLaunchKernelGGL(K, grid_deal, block, 0, s, e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], deal.per, q);         \
    } while (0)
        if (fast && gains) PANO_LAUNCH_K1_DEAL(warp_tiles_lut_kernel<true>);
        else if (fast) PANO_LAUNCH_K1_DEAL(warp_tiles_lut_kernel<false>);
#undef PANO_LAUNCH_K1_DEAL
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

}  // namespace pano
