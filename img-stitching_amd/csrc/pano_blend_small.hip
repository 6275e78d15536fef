// pano_blend_small.hip - K3, the small levels: normalise + LDS collapse, and the fused one-launch form
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

}  // namespace pano
