// pano_blend_small.hip - K3, the small levels: one normalise launch + one LDS collapse launch
// Device helpers: pano_dev.hpp; launch interface: pano_kernels.hpp.  Compile with -ffp-contract=off.

#include "pano_dev.hpp"

namespace pano {

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
    // Weights: ONE unconditional load per camera slot (a pixel outside the camera's tile, or a slot past cam_n, reads the tile's
    // first weight and drops it) - as conditional loads each sat in a block of its own with its wait: eight round trips in a row
    float wv[kCams];
    {
        float wl[kCams];
        bool in[kCams];
#pragma unroll
        for (int i = 0; i < kCams; i++) {
            const PyrCam& c = P.cam[cam_lo + min(i, cam_n - 1)];  // a slot past cam_n re-reads the last camera's (cam_n >= 1)
            const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
            in[i] = i < cam_n && (unsigned)x < (unsigned)(c.w0 >> l) && (unsigned)y < (unsigned)(c.h0 >> l);
            wl[i] = c.wgt[l][in[i] ? (size_t)y * c.wpitch[l] + x : 0];  // small levels are never level 0
        }
#pragma unroll
        for (int i = 0; i < kCams; i++) wv[i] = in[i] ? wl[i] : 0.f;
    }
    int acc[3] = {0, 0, 0};
    float W = 0.f;
    const bool up = l < C.bands;  // block-uniform, tested OUTSIDE the plane loops: inside them every plane's ten loads were a
                                  // conditional block of their own with its wait
#pragma unroll
    for (int i = 0; i < kCams; i++) {
        const float w = wv[i];
        if (w == 0.f) continue;
        const PyrCam& c = P.cam[cam_lo + i];
        const int x = X - (c.tx >> l), y = Y - (c.ty >> l);
        const int tw = c.w0 >> l, th = c.h0 >> l;
        W += w;
        int lap[3];
#pragma unroll
        for (int k = 0; k < 3; k++) lap[k] = c.lvl[l][(size_t)k * c.plane[l] + (size_t)y * c.pitch[l] + x];
        if (up) {
            // the 3 x 3 taps of pyr_up_px for the three planes: all loads, then the arithmetic
            const int n = tw >> 1, m = th >> 1, cx = x >> 1, cy = y >> 1;
            int xi[3], wx[3], yi[3], wy[3];
            if (!(x & 1)) {
                xi[0] = cx > 0 ? cx - 1 : (n > 1 ? 1 : 0); xi[1] = cx; xi[2] = min(cx + 1, n - 1);
                wx[0] = 1; wx[1] = 6; wx[2] = 1;
            } else {
                xi[0] = cx; xi[1] = min(cx + 1, n - 1); xi[2] = cx;
                wx[0] = 4; wx[1] = 4; wx[2] = 0;
            }
            if (!(y & 1)) {
                yi[0] = cy > 0 ? cy - 1 : (m > 1 ? 1 : 0); yi[1] = cy; yi[2] = min(cy + 1, m - 1);
                wy[0] = 1; wy[1] = 6; wy[2] = 1;
            } else {
                yi[0] = cy; yi[1] = min(cy + 1, m - 1); yi[2] = cy;
                wy[0] = 4; wy[1] = 4; wy[2] = 0;
            }
            int tap[3][3][3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint8_t* S = c.lvl[l + 1] + (size_t)k * c.plane[l + 1];
#pragma unroll
                for (int j = 0; j < 3; j++)
#pragma unroll
                    for (int t = 0; t < 3; t++) tap[k][j][t] = S[(size_t)yi[j] * c.pitch[l + 1] + xi[t]];
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                int a = 0;
#pragma unroll
                for (int j = 0; j < 3; j++) a += (tap[k][j][0] * wx[0] + tap[k][j][1] * wx[1] + tap[k][j][2] * wx[2]) * wy[j];
                lap[k] = sat16i(lap[k] - sat16i((a + 32) >> 6));
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = (int16_t)(acc[k] + (int16_t)(int)((float)lap[k] * w));
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
__global__ __launch_bounds__(256) void collapse_small_kernel(CanvasSet CS) {
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
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
            nv[j][pl] = C.img[l][(size_t)pl * C.cplane[l] + (size_t)(ry0[j] + ty) * C.cpitch[l] + rx0[j] + tx];
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

void launch_blend_small(const PyrParams& p, const CanvasSet& cs, hipStream_t s) {
    const CanvasParams& c = cs.c[0];
    const int k0 = c.small_base;
    int cw = 0, ch = 0;
    for (int g = 0; g < cs.n; g++) {
        cw = max(cw, cs.c[g].w0 >> k0);
        ch = max(ch, cs.c[g].h0 >> k0);
    }
    dim3 block(64, 4, 1);
    dim3 g1((cw + 63) / 64, (ch + 3) / 4, (c.bands - k0 + 1) * cs.n);
    hipLaunchKernelGGL(norm_small_kernel, g1, block, 0, s, p, cs);
    dim3 g2((cw + kSmallTileW - 1) / kSmallTileW, (ch + kSmallTileH - 1) / kSmallTileH, cs.n);
    hipLaunchKernelGGL(collapse_small_kernel, g2, block, 0, s, cs);
}

}  // namespace pano
