// pano_dev.hpp - device-side helpers shared by the gfx950 (MI355X, CDNA4) kernel files of the panorama compose path
// (pano_warp.hip, pano_pyramid.hip, pano_blend.hip, pano_blend_small.hip, pano_init.hip).
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

#pragma once

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

// packed 16-bit pairs (v_pk_* / v_dot2)
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
typedef short s2_t __attribute__((ext_vector_type(2)));

// one row of cv::pyrDown's horizontal pass for four outputs (K2, and every kernel that rebuilds pyramid levels in LDS)
__device__ __forceinline__ void pyr_down_hrow(const uint4 q, int h[4]) {
    // q = 16 bytes starting at column 8t-4; the four outputs are centred on columns 8t, 8t+2, 8t+4, 8t+6, i.e. on
    // bytes 4, 6, 8, 10 of the window.  Each 5-tap window (1 4 6 4 1) straddles two dwords: two chained dot products
    // with the taps placed on the right bytes - no realignment, no byte extraction.
    h[0] = (int)__builtin_amdgcn_udot4(q.y, 0x00010406u, __builtin_amdgcn_udot4(q.x, 0x04010000u, 0u, false), false);  // bytes 2..6
    h[1] = (int)__builtin_amdgcn_udot4(q.z, 0x00000001u, __builtin_amdgcn_udot4(q.y, 0x04060401u, 0u, false), false);  // bytes 4..8
    h[2] = (int)__builtin_amdgcn_udot4(q.z, 0x00010406u, __builtin_amdgcn_udot4(q.y, 0x04010000u, 0u, false), false);  // bytes 6..10
    h[3] = (int)__builtin_amdgcn_udot4(q.w, 0x00000001u, __builtin_amdgcn_udot4(q.z, 0x04060401u, 0u, false), false);  // bytes 8..12
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

// weight of camera c at pixel (x, y) of its level-l tile
__device__ __forceinline__ float cam_weight(const PyrCam& c, int l, int x, int y) {
    if (l == 0) return (float)c.mask0[(size_t)y * c.pitch[0] + x] * (float)(1. / 255.);
    return c.wgt[l][(size_t)y * c.wpitch[l] + x];
}

}  // namespace pano
