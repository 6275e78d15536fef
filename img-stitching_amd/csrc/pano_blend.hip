// pano_blend.hip - K3: one blend level per launch (generic and vector forms), owner maps, launchers
// Device helpers: pano_dev.hpp; launch interface: pano_kernels.hpp.  Compile with -ffp-contract=off.

#include "pano_dev.hpp"

namespace pano {

// BlendLevel's pointers come out of scalar registers (asm loads) without an address space: every dereference says "global",
// or it is a flat_load / flat_store
#define PANO_G __attribute__((address_space(1)))
typedef unsigned blend_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned blend_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned blend_u32x3u __attribute__((ext_vector_type(3), aligned(1)));  // at any byte alignment

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
        for (int r = 0; r < 3; r++) {  // v_mul_lo_u32 is quarter rate
            const blend_u32x2 t = *(const blend_u32x2 PANO_G*)(S + (unsigned)(__mul24(yi[r], pitch) + ab));
            d[r] = make_uint2(t.x, t.y);
        }
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
        for (int r = 0; r < 3; r++) {
            const blend_u32x3 t = *(const blend_u32x3 PANO_G*)(S + (unsigned)(__mul24(yi[r], pitch) + ab));
            d[r] = make_uint3(t.x, t.y, t.z);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const unsigned w0 = __builtin_amdgcn_alignbyte(d[r].y, d[r].x, bs);
            const unsigned w1 = __builtin_amdgcn_alignbyte(d[r].z, d[r].y, bs);
            q[r][0] = __builtin_amdgcn_perm(w1, w0, selA);
            q[r][1] = __builtin_amdgcn_perm(w1, w0, selB);
        }
    }
}
// a . w + R for two packed int16 pairs, v_dot2_i32_i16 with an INLINE-CONSTANT accumulator R (w is a compile-time constant; R = 0, 8
// or 24: inline constants are free, the builtin would spend a v_mov on them)
template <int R>
__device__ __forceinline__ int sdot2_from(unsigned a, unsigned w) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(w), "n"(R));
    return d;
}
// pyrUp of a 4 x 2 block from the packed 4 x 3 neighbourhood: up[0][..] = fine row Y0 (even), up[1][..] = row Y0+1.
// Horizontal pass per coarse row: (p0 + 6 p1 + p2, 4 (p1 + p2), p1 + 6 p2 + p3, 4 (p2 + p3)) as dot products
// (v_dot4_u32_u8 on the byte window / v_dot2_i32_i16 on the short pairs); vertical pass in 32-bit ints:
//   even row (h0 + 6 h1 + h2 + 32) >> 6,   odd row (h1 + h2 + 8) >> 4.
// The rounding terms ride in the dot products' accumulators - +24 in h0, +8 in h2 (24 + 8 = 32 for the even row, 8 for the odd) -
// and the 6 h1 is one v_mad_*24: add, mad, shift | add, shift per sample.  On gfx950 an add or a right shift issues at twice the
// rate of a v_mad / v_add3 / v_mul_lo (profiles/r03_valu_instruction_rates.txt); the compiler's own v_mul_lo_u32 + v_add3_u32 chain
// for the plain expression cost half again as many issue cycles.
template <typename T>
__device__ __forceinline__ void up_block(const unsigned q[3][2], int up[2][4]) {
    int h[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        if (sizeof(T) == 1) {
            const unsigned w = q[r][0];
            const unsigned R = r == 0 ? 24u : (r == 2 ? 8u : 0u);
            h[r][0] = (int)__builtin_amdgcn_udot4(w, 0x00010601u, R, false);
            h[r][1] = (int)__builtin_amdgcn_udot4(w, 0x00040400u, R, false);
            h[r][2] = (int)__builtin_amdgcn_udot4(w, 0x01060100u, R, false);
            h[r][3] = (int)__builtin_amdgcn_udot4(w, 0x04040000u, R, false);
        } else {
            const s2_t A = __builtin_bit_cast(s2_t, q[r][0]), B = __builtin_bit_cast(s2_t, q[r][1]);
            const s2_t c16 = {1, 6}, c04 = {0, 4}, c01 = {0, 1};
            if (r == 0) {
                h[r][0] = __builtin_amdgcn_sdot2(A, c16, (int)B.x + 24, false);
                h[r][1] = __builtin_amdgcn_sdot2(A, c04, sdot2_from<24>(q[r][1], 0x00000004u), false);  // B . (4, 0)
                h[r][2] = __builtin_amdgcn_sdot2(A, c01, sdot2_from<24>(q[r][1], 0x00010006u), false);  // B . (6, 1)
                h[r][3] = sdot2_from<24>(q[r][1], 0x00040004u);                                         // B . (4, 4)
            } else if (r == 2) {
                h[r][0] = __builtin_amdgcn_sdot2(A, c16, (int)B.x + 8, false);
                h[r][1] = __builtin_amdgcn_sdot2(A, c04, sdot2_from<8>(q[r][1], 0x00000004u), false);
                h[r][2] = __builtin_amdgcn_sdot2(A, c01, sdot2_from<8>(q[r][1], 0x00010006u), false);
                h[r][3] = sdot2_from<8>(q[r][1], 0x00040004u);
            } else {
                h[r][0] = __builtin_amdgcn_sdot2(A, c16, (int)B.x, false);
                h[r][1] = __builtin_amdgcn_sdot2(A, c04, sdot2_from<0>(q[r][1], 0x00000004u), false);
                h[r][2] = __builtin_amdgcn_sdot2(A, c01, sdot2_from<0>(q[r][1], 0x00010006u), false);
                h[r][3] = sdot2_from<0>(q[r][1], 0x00040004u);
            }
        }
    }
    // No saturate_cast here: it cannot trigger.  Camera planes are 8-bit (h <= 8*255), and a collapsed canvas level
    // is bounded by 255 per remaining level (|norm_l| <= 255, pyrUp is a convex combination + rounding), i.e.
    // |out_l| <= 9*255 + 9 for the maximum of 8 bands - far inside int16 (and every h inside 24 bits).
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int h1 = h[1][k];
        if (sizeof(T) == 1) {
            up[0][k] = (int)((unsigned)(__mul24(h1, 6) + (h[0][k] + h[2][k])) >> 6);  // signed spelling: one v_mad_i32_i24; sums are non-negative
            up[1][k] = (int)((unsigned)(h1 + h[2][k]) >> 4);
        } else {
            up[0][k] = (__mul24(h1, 6) + (h[0][k] + h[2][k])) >> 6;
            up[1][k] = (h1 + h[2][k]) >> 4;
        }
    }
}

// four BGR pixels of an output row, stored with one instruction at any byte alignment
struct __attribute__((packed, aligned(1))) Bgr4 {
    unsigned x, y, z;
};
// store a finished 4 x 2 block: canvas level (planar int16) or, at level 0, dst_mask + convertTo(8U) + cut
// ALLON: every pixel of the block carries weight (dst_mask set) - the caller's guarantee, no per-pixel select
template <bool L0, bool ALLON = false, int NPL = 3>
__device__ __forceinline__ void store_block(const BlendLevel& C, int l, int X0, int Y0, const int v[3][2][4], bool o00,
                                            bool o01, bool o02, bool o03, bool o10, bool o11, bool o12, bool o13, int pb = 0) {
    if (!L0) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
#pragma unroll
            for (int r = 0; r < 2; r++) {
                uint2 pk;
                pk.x = ((unsigned)v[pl][r][0] & 0xffffu) | ((unsigned)v[pl][r][1] << 16);
                pk.y = ((unsigned)v[pl][r][2] & 0xffffu) | ((unsigned)v[pl][r][3] << 16);
                *(blend_u32x2 PANO_G*)(C.img + (size_t)(pb + pl) * C.cplane + (unsigned)(__mul24(Y0 + r, C.cpitch) + X0)) = blend_u32x2{pk.x, pk.y};
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
            uint8_t PANO_G* d = (uint8_t PANO_G*)C.out + (int)(__mul24(Y - C.cut_y, C.out_stride) + 3 * (X0 - C.cut_x));
            const bool whole = X0 >= C.cut_x && X0 + 4 <= C.cut_x + C.cut_w;
            if (whole) {
                // one 12-byte store at whatever alignment the row has: a cv::Mat panorama has rows of 3 * width bytes, so three
                // rows in four start off a dword boundary (gfx950 takes unaligned global stores; as twelve byte stores
                // those rows cost level 0 a quarter of its time)
                Bgr4 pk;
                pk.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                pk.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                pk.z = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
                // non-temporal (a cache hint only): nothing in the FRAME reads the panorama back (pano_stack_* and the D2H copy do, later and
                // once), and without the hint its 23 MB per frame pair push
                // the pyramids of the frames in flight out of the caches (-1.0 to -1.9 us per frame with four in flight; the same
                // hint on stores that ARE read back - G0, the pyramid levels, the canvas - or on any load of this kernel costs
                // 1 to 8 us: docs/EXPERIMENTS.md, round 4)
                __builtin_nontemporal_store(blend_u32x3u{pk.x, pk.y, pk.z}, (blend_u32x3u PANO_G*)d);
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
// UP: there is a coarser level (launch-uniform: BlendLevel::up).  A template parameter and not a test of it, because behind a run-time
// test every load_coarse - loads AND the permutes that consume them - sat in a conditional block of its own, and the wave waited for
// each plane's windows before it requested the next plane's: six dependent vector round trips where the code meant one.
template <bool L0, int NPL, bool UP>
__device__ __forceinline__ void blend_block(const PyrParams& P, const BlendLevel& C, const int l, const int X0, const int Y0,
                                            const int pb, const unsigned hint = 0xfu) {
    const int cam_lo = C.cam_lo;
    const int cw = C.cw, ch = C.ch;
    // Away from the seams a block belongs to exactly one camera with weight 1.0f everywhere (or to none):
    // the static owner map says so in one byte, and the block needs no weights, no float math and no division:
    //   acc = lap, W = 1  =>  norm = lap - sign(lap)  (see below)
    // one 16-bit entry per block: low byte = owner code, high byte = the cameras that carry weight anywhere on the block
    // hint (wave-uniform, from a static per-wave table): 0..7 / 0xE = every block of this wave has that single owner / none, so
    // the wave does not wait for its owner entries before it can issue its pixel loads; 0xF = look
    unsigned entry = 0, ucode;
    bool single;
    if (hint != 0xfu) {
        ucode = hint < 8u ? hint : 0xfeu;
        single = true;
    } else {
        entry = ((const uint16_t PANO_G*)C.owner)[(unsigned)(__mul24(Y0 >> 1, C.opitch) + (X0 >> 2))];
        const unsigned code = entry & 0xffu;
        ucode = __builtin_amdgcn_readfirstlane(code);
        single = ucode != 0xffu && __builtin_amdgcn_ballot_w64(code != ucode) == 0;
    }
    if (single) {
        // the whole wave (a 256 x 2 strip) has one owner: its parameters are scalar, the code is straight-line
        // and every load is in flight before the first use
        int v[3][2][4];
        unsigned cp[3][3][2];
        if (UP) {
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
                load_coarse<int16_t>(C.img_up + (size_t)(pb + pl) * C.cplane_up, cw >> 1, ch >> 1, C.cpitch_up,
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
                if (UP)
                    load_coarse<uint8_t>(c.lvl[l + 1] + (size_t)(pb + pl) * c.plane[l + 1], tw >> 1, th >> 1, c.pitch[l + 1],
                                         x >> 1, y >> 1, p[pl]);
            }
#pragma unroll
            for (int pl = 0; pl < NPL; pl++) {
                int up[2][4];
                if (UP) {
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
        if (UP) {
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
    if (!L0 && UP) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
            load_coarse<int16_t>(C.img_up + (size_t)(pb + pl) * C.cplane_up, cw >> 1, ch >> 1, C.cpitch_up, X0 >> 1,
                                 Y0 >> 1, cp[pl]);
    }
    // the accumulators are int16 by definition (dst += static_cast<short>(..) wraps): two per register (v_pk_add_i16) - the general
    // path sets the kernel's register allocation, and with it the waves in flight that the single-owner path lives on
    s2_t acc[3][2][2];
    float W[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            W[r][k] = 0.f;
            acc[0][r][k >> 1] = acc[1][r][k >> 1] = acc[2][r][k >> 1] = s2_t{0, 0};
        }
    // phase B: cameras with weight, in feed order
#pragma unroll
    for (int i = 0; i < kCams; i++) {
        if (!((live >> i) & 1u)) continue;
        const PyrCam& c = P.cam[cam_lo + i];
        const int x = X0 - (c.tx >> l), y = Y0 - (c.ty >> l);
        const int tw = c.w0 >> l, th = c.h0 >> l;
        // every load of this camera - its weights AND its pixels - goes out before any is used (with the weights converted and tested
        // in front of the pixel loads, as the source once read, the wave waited for the weights first: a round trip per camera)
        unsigned mk[2] = {0u, 0u};
        float4 wf[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        if (L0) {
            mk[0] = *reinterpret_cast<const unsigned*>(c.mask0 + (unsigned)(__mul24(y, c.pitch[0]) + x));
            mk[1] = *reinterpret_cast<const unsigned*>(c.mask0 + (unsigned)(__mul24(y + 1, c.pitch[0]) + x));
        } else {
#pragma unroll
            for (int r = 0; r < 2; r++) wf[r] = *reinterpret_cast<const float4*>(c.wgt[l] + (size_t)(y + r) * c.wpitch[l] + x);
        }
        unsigned g0[3], g1[3];
        unsigned p[3][3][2];
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            const uint8_t* g = c.lvl[l] + (size_t)(pb + pl) * c.plane[l] + (size_t)y * c.pitch[l] + x;
            g0[pl] = *reinterpret_cast<const unsigned*>(g);
            g1[pl] = *reinterpret_cast<const unsigned*>(g + c.pitch[l]);
            if (UP)
                load_coarse<uint8_t>(c.lvl[l + 1] + (size_t)(pb + pl) * c.plane[l + 1], tw >> 1, th >> 1, c.pitch[l + 1], x >> 1,
                                     y >> 1, p[pl]);
        }
        float w[2][4];
        if (L0) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) w[r][k] = (float)((mk[r] >> (8 * k)) & 0xffu) * (float)(1. / 255.);
        } else {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                w[r][0] = wf[r].x; w[r][1] = wf[r].y; w[r][2] = wf[r].z; w[r][3] = wf[r].w;
            }
        }
        // away from the seams the weight is exactly 1.0f on the whole block: no float path
        bool unit = true;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) unit &= w[r][k] == 1.0f;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) W[r][k] += w[r][k];  // + 0.f is exact
#pragma unroll
        for (int pl = 0; pl < NPL; pl++) {
            int up[2][4];
            if (UP) {
                up_block<uint8_t>(p[pl], up);
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) up[0][k] = up[1][k] = 0;
            }
            short t0[4], t1[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int l0 = sat16i((int)((g0[pl] >> (8 * k)) & 0xffu) - up[0][k]);
                const int l1 = sat16i((int)((g1[pl] >> (8 * k)) & 0xffu) - up[1][k]);
                // (short)(lap * 1.0f) == lap
                t0[k] = unit ? (short)l0 : (short)(int)((float)l0 * w[0][k]);
                t1[k] = unit ? (short)l1 : (short)(int)((float)l1 * w[1][k]);
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                acc[pl][0][h] += s2_t{t0[2 * h], t0[2 * h + 1]};  // wraps like the short += of the reference
                acc[pl][1][h] += s2_t{t1[2 * h], t1[2 * h + 1]};
            }
        }
    }
    if (L0 && UP) {
#pragma unroll
        for (int pl = 0; pl < NPL; pl++)
            load_coarse<int16_t>(C.img_up + (size_t)(pb + pl) * C.cplane_up, cw >> 1, ch >> 1, C.cpitch_up, X0 >> 1,
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
        if (UP) {
            if (L0) {
                // the coarse canvas windows stay as loaded (6 registers a plane) until HERE: left alone the scheduler forms their
                // horizontal sums (12 a plane) as soon as the loads land and the allocator carries them across the normalise - that
                // was the kernel's 97th register and its one spilled dword (96 VGPRs + 8 B of scratch -> 85, none)
#pragma unroll
                for (int r = 0; r < 3; r++) asm volatile("" : "+v"(cp[pl][r][0]), "+v"(cp[pl][r][1]));
            }
            up_block<int16_t>(cp[pl], up);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) up[0][k] = up[1][k] = 0;
        }
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int a = (k & 1) ? acc[pl][r][k >> 1].y : acc[pl][r][k >> 1].x;
                int nrm;
                if (unitW) nrm = toward_zero_by_one(a);
                else nrm = (int16_t)(int)((float)a / (W[r][k] + 1e-5f));
                v[pl][r][k] = UP ? sat16i(nrm + up[r][k]) : nrm;
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
// The scalars IN FRONT of the parameter blocks arrive in SGPRs at wave launch (kernarg preload, see the Makefile), and the
// canvas' fields of this level (BlendLevel = CanvasParams::hot, packed by the launcher) come with two scalar loads: one round
// trip where reading them field by field behind the early exits took five or six (tools/wave_timeline_l0.py).
//   shape 3 (XCD bands): a 1-D grid; a0 / a1 = the multipliers that divide by gx and gy (2^32 / d + 1; 0: d == 1), total = workgroups
//   shape 2: a plain 3-D grid
//   (the levels in seam-first order run a kernel of their own: blend_level_ordered_kernel)
struct BlendVecArgs {  // the kernel's argument list as the kernarg segment lays it out
    int lvl;
    unsigned a0, a1, total;
    PyrParams P;
    CanvasSet CS;
};
typedef int blend_i32x8 __attribute__((ext_vector_type(8)));
typedef int blend_i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void blend_level_from(const blend_i32x16& a, const blend_i32x8& b, BlendLevel& V) {
    int w[24];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = a[i];
#pragma unroll
    for (int i = 0; i < 8; i++) w[16 + i] = b[i];
    __builtin_memcpy(&V, w, sizeof(V));
}
__device__ __forceinline__ void load_blend_level(unsigned cs_at, unsigned canvas, BlendLevel& V) {
    blend_i32x16 a;
    blend_i32x8 b;
    const char __attribute__((address_space(4)))* hb = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
        (cs_at + (unsigned)offsetof(CanvasSet, c) + canvas * (unsigned)sizeof(CanvasParams) + (unsigned)offsetof(CanvasParams, hot));
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(hb) : "memory");
    blend_level_from(a, b, V);
}
template <bool L0, int NPL = 3>
__global__ __launch_bounds__(256) void blend_level_vec_kernel(int lvl, unsigned a0, unsigned a1, unsigned total, PyrParams P, CanvasSet CS) {
    static_assert(NPL == 3 || !L0, "level 0 writes interleaved BGR");
    unsigned bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
    if (((lvl >> 8) & 15) == 3) {
        // XCD bands (shape 3): a 1-D grid of 8 * per workgroups; the hardware deals consecutive ids round-robin
        // over the 8 XCDs, so XCD k is given the logical workgroups [k * per, (k + 1) * per) - a contiguous band of canvas
        // rows, whose neighbouring workgroups share their cache lines and pyrUp halos in ONE L2
        const unsigned gx = ((unsigned)lvl >> 12) & 0x3ffu, gy = ((unsigned)lvl >> 22) & 0x3ffu;
        const unsigned per = (total + 7u) / 8u;
        const unsigned logical = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per || logical >= total) return;
        const unsigned row = a0 ? __umulhi(logical, a0) : logical;  // logical / gx
        bxi = logical - row * gx;
        bzi = a1 ? __umulhi(row, a1) : row;                         // row / gy
        byi = row - bzi * gy;
    }
    const int pb = NPL == 3 ? 0 : (int)(bzi % 3);  // first plane of this lane
    BlendLevel C;
    load_blend_level((unsigned)offsetof(BlendVecArgs, CS), NPL == 3 ? bzi : bzi / 3, C);
    const int l = L0 ? 0 : (lvl & 0xff);
    const int cw = C.cw, ch = C.ch;
    // level 0 covers only the block-aligned hull of the cut rectangle
    const int bx0 = L0 ? (C.cut_x & ~3) : 0, by0 = L0 ? (C.cut_y & ~1) : 0;
    // a wave is 16 x 4 blocks = 64 x 8 pixels (not a 256-pixel strip): four times fewer waves straddle a seam,
    // and a wave that does not straddle one takes the single-owner fast path below.
    // The four waves of a workgroup form a 2 x 2 patch (128 x 16 pixels).  Consecutive workgroups go to different XCDs,
    // each with its own L2, so what a workgroup reads of a row should be whole 128-byte lines (four waves stacked read
    // half lines and fetched twice the bytes; four side by side straddle more seams: docs/EXPERIMENTS.md).  On top of
    // that the XCD bands above: neighbouring tiles share their lines and pyrUp halos in one L2.
    const int X0 = bx0 + ((bxi * 2 + (threadIdx.y & 1)) * 16 + (threadIdx.x & 15)) * 4;
    const int Y0 = by0 + ((byi * 2 + (threadIdx.y >> 1)) * 4 + (threadIdx.x >> 4)) * 2;
    if (L0) {
        if (X0 >= C.cut_x + C.cut_w || Y0 >= C.cut_y + C.cut_h) return;
    } else if (X0 >= cw || Y0 >= ch) {
        return;
    }
    if (C.up) blend_block<L0, NPL, true>(P, C, l, X0, Y0, pb, 0xfu);
    else blend_block<L0, NPL, false>(P, C, l, X0, Y0, pb, 0xfu);
}

// A vector level in seam-first order: XCD bands, the band of XCD k walked in the order of the static table CanvasParams::order[l] - tiles
// that hold a wave without a single owner (the general path: four times the instructions, two dependent rounds of loads) come
// first, so their long chains run under the bulk instead of behind it.  grid (8 * max entries per XCD, canvases).
// entry: bx | by << 8 | the four waves' owner nibbles << 16 (0xffff in the low half: no tile).
// A wave's scalar prologue is ONE round trip: the tables and their lengths are preloaded arguments, and the canvas' fields
// (BlendLevel = CanvasParams::hot) are requested by the wave's first instructions, in flight with the table entry.
struct BlendOrderedArgs {  // the kernel's argument list as the kernarg segment lays it out
    const uint32_t *ord0, *ord1;
    unsigned per0, per1;
    int l;
    PyrParams P;
    CanvasSet CS;
};
// L0 (NPL 3): 85 VGPRs, 5 waves / SIMD (at most 96), no scratch.  Until round 5 the allocator needed 97 and kept ONE dword of the seam
// path in scratch; five restructurings left it there, the sixth - the coarse canvas windows pinned as loaded until their use, see the
// seam path's last phase - took it out and eleven registers with it.  A sixth wave needs <= 80: the single-owner path's arithmetic
// phase holds 84 (at 80 the compiler spills 24 - 32 B there; the level-0 rows through LDS instead of registers do not help).
// Levels >= 1 (NPL 1): one plane per lane, l = the level.
// Read wave by wave in round 5 (tools/blend_timeline.py): level 0 is 15 128 waves of 4.5 us over 4 940 slots, a quarter of them empty at
// any time (1.2 us from a wave's end to its successor's first instruction on the same SIMD).  Built on that and not kept, each bit-exact:
// every wave a workgroup of its own, two or three tiles per wave in the same registers, long waves at the head of the list only
// (experiments/blend_one_wave_workgroups.patch, blend_tiles_per_wave.patch; docs/EXPERIMENTS.md).
template <bool L0, int NPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(L0 ? 5 : 8, 8))) void blend_level_ordered_kernel(
    const uint32_t* ord0, const uint32_t* ord1, unsigned per0, unsigned per1, int l, PyrParams P, CanvasSet CS) {
    static_assert((L0 && NPL == 3) || (!L0 && NPL == 1), "level 0 does three planes per lane, the levels above one");
    // grid (8 * canvases [* 3 planes], entries per XCD): blockIdx.x = XCD + 8 * (canvas [* 3 + plane]), blockIdx.y = position in the
    // XCD's list - so that the seam tiles of BOTH canvases are dispatched first (canvas-major, the second canvas' seam waves
    // started when the first canvas' whole list was out: half-way through the launch)
    const unsigned zi = blockIdx.x >> 3;
    const unsigned cvi = NPL == 3 ? zi : zi / 3u;
    const int pb = NPL == 3 ? 0 : (int)(zi - cvi * 3u);
    blend_i32x16 ha;
    blend_i32x8 hb8;
    {
        const char __attribute__((address_space(4)))* hb = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
            (offsetof(BlendOrderedArgs, CS) + offsetof(CanvasSet, c) + cvi * sizeof(CanvasParams) + offsetof(CanvasParams, hot));
        asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40" : "=&s"(ha), "=&s"(hb8) : "s"(hb) : "memory");
    }
    const unsigned k = blockIdx.x & 7u, jj = blockIdx.y;
    const unsigned per = cvi ? per1 : per0;
    const uint32_t* ord = cvi ? ord1 : ord0;
    if (jj >= per) return;
    unsigned ent;  // the table entry: a scalar load spelled out (behind the asm above the compiler would fetch it with a vector load)
    {
        const char __attribute__((address_space(4)))* ep = (const char __attribute__((address_space(4)))*)ord + (size_t)(k * per + jj) * 4u;
        asm volatile("s_load_dword %0, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ent), "+s"(ha), "+s"(hb8) : "s"(ep) : "memory");
    }
    if ((ent & 0xffffu) == 0xffffu) return;
    const unsigned bxi = ent & 0xffu, byi = (ent >> 8) & 0xffu;
    const unsigned hint = (ent >> (16 + 4 * __builtin_amdgcn_readfirstlane(threadIdx.y))) & 0xfu;
    BlendLevel C;
    blend_level_from(ha, hb8, C);
    // level 0 covers only the block-aligned hull of the cut rectangle; a wave is 16 x 4 blocks of 4 x 2 pixels, a workgroup 2 x 2 waves
    const int bx0 = L0 ? (C.cut_x & ~3) : 0, by0 = L0 ? (C.cut_y & ~1) : 0;
    const int X0 = bx0 + ((bxi * 2 + (threadIdx.y & 1)) * 16 + (threadIdx.x & 15)) * 4;
    const int Y0 = by0 + ((byi * 2 + (threadIdx.y >> 1)) * 4 + (threadIdx.x >> 4)) * 2;
    if (L0) {
        if (X0 >= C.cut_x + C.cut_w || Y0 >= C.cut_y + C.cut_h) return;
    } else if (X0 >= C.cw || Y0 >= C.ch) {
        return;
    }
    const int lv = L0 ? 0 : l;
    if (C.up) blend_block<L0, NPL, true>(P, C, lv, X0, Y0, pb, hint);
    else blend_block<L0, NPL, false>(P, C, lv, X0, Y0, pb, hint);
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

// What will the four waves of every 128 x 16-pixel workgroup tile of vector level l find in the owner map?  One nibble per wave
// (wave = threadIdx.y of the blend kernels' 2 x 2 shape; level 0: over the hull of the cut): 0..7 = every block of the wave
// inside the level / the cut has that single owner, 0xE = none of them has an owner, 0xF = anything else (the wave has to look).
// Static: it follows the owner map and the cut.
__global__ __launch_bounds__(256) void tile_mixed_kernel(CanvasParams C, int l, int gx, uint16_t* flags) {
    const int bx0 = l == 0 ? (C.cut_x & ~3) : 0, by0 = l == 0 ? (C.cut_y & ~1) : 0;
    const int xe = l == 0 ? C.cut_x + C.cut_w : (C.w0 >> l), ye = l == 0 ? C.cut_y + C.cut_h : (C.h0 >> l);
    const int X0 = bx0 + ((blockIdx.x * 2 + (threadIdx.y & 1)) * 16 + (threadIdx.x & 15)) * 4;
    const int Y0 = by0 + ((blockIdx.y * 2 + (threadIdx.y >> 1)) * 4 + (threadIdx.x >> 4)) * 2;
    const bool valid = X0 < xe && Y0 < ye;
    unsigned code = 0;
    if (valid) code = C.owner[l][(unsigned)(__mul24(Y0 >> 1, C.opitch[l]) + (X0 >> 2))] & 0xffu;
    const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
    unsigned nib = 0xEu;  // a wave with no block inside the cut does nothing at all
    if (vm) {
        const unsigned c0 = (unsigned)__shfl((int)code, __ffsll((long long)vm) - 1);
        const bool uniform = __builtin_amdgcn_ballot_w64(valid && code != c0) == 0;
        nib = !uniform ? 0xFu : (c0 < 8u ? c0 : (c0 == 0xFEu ? 0xEu : 0xFu));
    }
    __shared__ unsigned all;
    if (threadIdx.x == 0 && threadIdx.y == 0) all = 0;
    __syncthreads();
    if (threadIdx.x == 0) atomicOr(&all, nib << (4 * threadIdx.y));
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) flags[blockIdx.y * gx + blockIdx.x] = (uint16_t)all;
}
void launch_tile_mixed(const CanvasParams& c, int l, int gx, int gy, uint16_t* flags, hipStream_t s) {
    hipLaunchKernelGGL(tile_mixed_kernel, dim3(gx, gy, 1), dim3(64, 4, 1), 0, s, c, l, gx, flags);
}

void launch_blend_level(const PyrParams& p, const CanvasSet& cs, int l, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
#define PANO_LAUNCH_L0(K, G)                                                                         \
    do {                                                                                             \
        if (ev_start && ev_stop) hipExtLaunchKernelGGL(K, G, block, 0, s, ev_start, ev_stop, 0, karg, ka0, ka1, ktotal, p, q); \
        else hipLaunchKernelGGL(K, G, block, 0, s, karg, ka0, ka1, ktotal, p, q);                   \
    } while (0)
    auto magic = [](unsigned d) { return d > 1 ? (unsigned)((1ull << 32) / d + 1ull) : 0u; };
    const CanvasParams& c = cs.c[0];
    if (c.fast[l]) {
        CanvasSet q = cs;  // with each canvas' fields of this level packed for the kernels (BlendLevel)
        for (int g = 0; g < q.n; g++) pack_blend_level(q.c[g], l);
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
        // workgroups of 2 x 2 waves (128 x 16 pixels) dealt to the XCDs in bands (shape 3); extents beyond the 10 bits the band
        // form packs them in fall back to a plain 3-D grid (shape 2).  Canvas levels >= 1 run one colour plane per lane
        // (grid.z = canvas * 3 + plane): those launches are one round of waves whose seam waves set the duration
        int shape = 3;
        if ((w + 127) / 128 > 1023 || (h + 15) / 16 > 1023) shape = 2;
        dim3 block(64, 4, 1), grid((w + 127) / 128, (h + 15) / 16, cs.n);
        int larg = l | (shape << 8);
        const dim3 grid3 = grid;  // the logical extents
        if (shape == 3) {       // a 1-D grid of 8 * ceil(workgroups / 8)
            larg |= (int)((grid3.x & 0x3ffu) << 12) | (int)((grid3.y & 0x3ffu) << 22);
            const unsigned zext = l == 0 ? cs.n : cs.n * 3;
            grid = dim3(8u * ((grid3.x * grid3.y * zext + 7u) / 8u), 1, 1);
        }
        // XCD bands, seam tiles first, the waves told their owners (the ordered kernel), where every canvas of the set has the table
        bool ordered = shape == 3 && l < kOrderLevels;
        unsigned maxper = 0;
        for (int g = 0; ordered && g < cs.n; g++) {
            ordered = cs.c[g].order[l] != nullptr;
            maxper = max(maxper, (unsigned)cs.c[g].order_per[l]);
        }
        ordered = ordered && maxper > 0 && maxper < 65536u;  // grid.y
        const uint32_t *ko0 = ordered ? cs.c[0].order[l] : nullptr, *ko1 = ordered && cs.n > 1 ? cs.c[1].order[l] : nullptr;
        const unsigned kp0 = ordered ? (unsigned)cs.c[0].order_per[l] : 0u, kp1 = ordered && cs.n > 1 ? (unsigned)cs.c[1].order_per[l] : 0u;
        if (l == 0) {
            if (ordered) {
                const dim3 go(8u * cs.n, maxper, 1);
                if (ev_start && ev_stop) hipExtLaunchKernelGGL((blend_level_ordered_kernel<true, 3>), go, block, 0, s, ev_start, ev_stop, 0, ko0, ko1, kp0, kp1, 0, p, q);
                else hipLaunchKernelGGL((blend_level_ordered_kernel<true, 3>), go, block, 0, s, ko0, ko1, kp0, kp1, 0, p, q);
            } else {
                const int karg = larg;
                const unsigned ka0 = magic(grid3.x), ka1 = magic(grid3.y), ktotal = grid3.x * grid3.y * (unsigned)cs.n;
                PANO_LAUNCH_L0((blend_level_vec_kernel<true, 3>), grid);
            }
        } else if (ordered) {
            hipLaunchKernelGGL((blend_level_ordered_kernel<false, 1>), dim3(8u * cs.n * 3, maxper, 1), block, 0, s, ko0, ko1, kp0, kp1, l, p, q);
        } else {
            hipLaunchKernelGGL((blend_level_vec_kernel<false, 1>), shape == 3 ? grid : dim3(grid.x, grid.y, cs.n * 3), block, 0, s, larg,
                               magic(grid3.x), magic(grid3.y), grid3.x * grid3.y * (unsigned)(cs.n * 3), p, q);
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

}  // namespace pano
