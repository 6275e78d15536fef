// pano_init.hip - init-time kernels: weights, caller-side stacking, mask pipeline, graph-cut weights, gain statistics, Voronoi
// Device helpers: pano_dev.hpp; launch interface: pano_kernels.hpp.  Compile with -ffp-contract=off.

#include "pano_dev.hpp"

namespace pano {

// ------------------------------------------------------------------------------------------------
// weights: mask * (1/255.f) with copyMakeBorder(CONSTANT 0); pyrDown CV_32F (scalar evaluation order unless told otherwise);
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

// cv::pyrDown CV_32F.  The five-tap sums are associated as the OpenCV build does that the weights are to match (F32Order: scalar
// code by default; the SSE2 / NEON vertical body, the universal-intrinsics horizontal body - oracle/pano_oracle.c has the forms and
// where each applies).  -ffp-contract=off: nothing is fused but the one multiply-add that such a build fuses itself.
__global__ __launch_bounds__(256) void pyr_down_f32_kernel(const float* __restrict__ src, int sw, int sh, int spitch,
                                                           float* __restrict__ dst, int dpitch, F32Order o) {
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int vend = o.vertical ? dw / o.vbody * o.vbody : 0;   // the vertical vector body ends here
    const int width0 = min((sw - 2 - 1) / 2 + 1, dw);           // pyrDown_: the columns whose five taps need no border table
    const int hend = o.horizontal && width0 > 1 ? 1 + (width0 - 1) / o.hbody * o.hbody : 0;
    const bool hvec = x >= 1 && x < hend, vvec = x < vend;
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = reflect101_idx(2 * x + k - 2, sw);
    float row[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float* r = src + (size_t)reflect101_idx(2 * y + k - 2, sh) * spitch;
        if (hvec) {
            const float inner = (r[xs[1]] + r[xs[3]]) * 4 + (r[xs[0]] + r[xs[4]]);
            row[k] = o.horizontal == 2 ? __builtin_fmaf(r[xs[2]], 6.f, inner) : r[xs[2]] * 6 + inner;
        } else {
            row[k] = r[xs[2]] * 6 + (r[xs[1]] + r[xs[3]]) * 4 + r[xs[0]] + r[xs[4]];
        }
    }
    float v;
    if (vvec) {
        const float a = (row[0] + row[4]) + (row[2] + row[2]);
        const float b = o.vertical == 2 ? (row[1] + row[2]) + row[3] : (row[1] + row[3]) + row[2];
        v = (a + b * 4) * (1.f / 256);
    } else {
        v = (row[2] * 6 + (row[1] + row[3]) * 4 + row[0] + row[4]) * (1.f / 256);
    }
    dst[(size_t)y * dpitch + x] = v;
}
void launch_pyr_down_f32(const float* src, int sw, int sh, int spitch, float* dst, int dpitch, F32Order o, hipStream_t s) {
    int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    dim3 block(64, 4, 1), grid((dw + 63) / 64, (dh + 3) / 4, 1);
    if (o.vbody < 1) o.vbody = 8;
    if (o.hbody < 1) o.hbody = 4;
    hipLaunchKernelGGL(pyr_down_f32_kernel, grid, block, 0, s, src, sw, sh, spitch, dst, dpitch, o);
}

// ------------------------------------------------------------------------------------------------
// the sharded exchange (pano_gather_slots): the LIVE rectangles of a camera's pyramid levels - what the blend on the root reads,
// 70 % of a slot on config 2 - packed into one contiguous message and unpacked in place on the root.  One segment = rows x
// width16 sixteen-byte chunks of one plane of one level; grid (segments, row blocks).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_segments_kernel(const XchSeg* __restrict__ segs, uint8_t* slots, uint8_t* stage, int unpack) {
    const XchSeg g = segs[blockIdx.x];
    const int rows_per_block = (g.rows + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, g.rows);
    for (int r = r0 + (int)threadIdx.y; r < r1; r += 4)
        for (int x = threadIdx.x; x < g.width16; x += 64) {
            uint4* a = reinterpret_cast<uint4*>(slots + g.slot_off + (size_t)r * g.pitch) + x;
            uint4* b = reinterpret_cast<uint4*>(stage + g.stage_off + (size_t)r * g.width16 * 16) + x;
            if (unpack) *a = *b;
            else *b = *a;
        }
}
void launch_copy_segments(const XchSeg* d_segs, int first, int count, int max_rows, uint8_t* slots, uint8_t* stage, bool unpack, hipStream_t s) {
    if (count <= 0) return;
    const int yb = max(1, min(16, (max_rows + 31) / 32));
    hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)count, (unsigned)yb, 1), dim3(64, 4, 1), 0, s, d_segs + first, slots, stage, unpack ? 1 : 0);
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
// One wave that occupies its stream for `ticks` of the 100 MHz wall clock and nothing else (the probe of pano_frame_streams: two
// of these on two streams take twice as long when the streams share a hardware queue).  Bounded: at most 2^20 naps of about 1 us.
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < (1 << 20); i++) {
        if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}
void launch_spin(unsigned long long ticks_100mhz, hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(1, 1, 1), dim3(64, 1, 1), 0, s, ticks_100mhz); }

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
