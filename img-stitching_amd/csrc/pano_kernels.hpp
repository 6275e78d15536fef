// pano_kernels.hpp - launch interface between the C-ABI host code (pano_api.cpp) and the gfx950
// kernels (pano_warp / pano_pyramid / pano_blend / pano_blend_small / pano_init .hip).  Plain structs passed by value as kernel arguments.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pano {

constexpr int kCams = 8;
constexpr int kLevels = 9;

// fused undistort front end of one camera (device copy; see include/pano.h pano_undistort)
struct FrontEndDev {
    int raw_w, raw_h, undist_w, undist_h, out_w, out_h;
    int rect[4];
    double K[9], newK[9], dist[4];
};

// one camera's slice of the fused warp launch (K1)
struct alignas(64) WarpCam {
    // --- the 64 bytes the table form of K1 needs, together so that a wave fetches them with one scalar load
    const uint8_t* src;   // BGR8 interleaved frame
    void* dst;            // pipeline: plane B of the level-0 tile (planar u8); stage warp: uint8x3 image
    const uint2* lutc;    // packed remap table, 8 bytes per 4-pixel group (see pack_warp_lut_kernel)
    const int4* box;      // per 64x16-pixel workgroup: its source box {xmin, ymin, rows<<8 | chunks, ceil(2^16/chunks)};
                          // the codes of lut / lutc are relative to (xmin, ymin)
    int tw, th;           // tile width/height in pixels
    // Blocks of 64 x 16 tile pixels (inclusive block coordinates) that anything downstream ever reads (see
    // pano_api.cpp live_rects): the K1 grid is laid over them.  {0, 0} .. {INT_MAX, INT_MAX} = everything.
    // A camera that straddles the +-pi seam of the projection has a full-width tile whose two ends are live and whose
    // middle is dead: gap_len block columns from gap_bx0 on are skipped (the grid is that much narrower).
    // live_by0_gap = live_by0 | gap_bx0 << 12 | gap_len << 22 (one dword, so that the hot part stays 64 bytes)
    int live_bx0, live_by0_gap;
    int src_stride;       // bytes
    int dst_pitch;        // bytes per row
    int dst_plane;        // pipeline: bytes between the B, G, R planes
    int lutc_pitch;       // groups per row = lut_pitch / 4
    // --- the next 16 bytes ride on a second scalar load
    int live_bx1, live_by1;
    int src_w, src_h;
    // --- 40 bytes the gain-applying instantiation of K1 fetches with two more scalar loads, together with the first
    const float* gain;    // [gh][gw] block gains (BlocksGainCompensator::apply) or nullptr
    // per 16-row block row of K1: the first of the (at most kGainRows) consecutive rows of ghrow that its 16 tile rows read, or -1
    // when they span more (a map finer than the blocks): K1 stages those rows' 64 columns in LDS once per workgroup
    const int* grow_base;
    // optional exposure gain: cv::resize(INTER_LINEAR) of the block map on the fly.  The horizontal pass is a table -
    // ghrow[gy][x] = S[gy][sx] * (1 - fx) + S[gy][sx1] * fx for every map row gy and tile column x (REFLECT folded like
    // colA) - so a pixel needs two coalesced reads and the vertical pass h0 * (1 - fy) + h1 * fy
    const float* ghrow;   // [gh][ghrow_pitch]
    const int4* grow4;    // [th] {sy0, sy1, bits of 1-fy, bits of fy}: grow and groww in one 16-byte entry per tile row
    int ghrow_pitch;      // floats per row, a multiple of 4; columns past tw repeat the last one
    int gh;               // rows of ghrow
    // --- set by launch_warp_tiles for the launch (WarpDeal): this camera's live block columns (its gap taken out) and the
    // multiplier that divides by them (2^32 / cols + 1; 0: one column)
    unsigned deal_cols, deal_mcols;
    // --- everything else
    // static remap table: one dword per tile pixel, see build_warp_table_kernel
    const uint32_t* lut;  // dense form, read where the packed form escapes; nullptr -> project on the fly
    int lut_pitch;        // dwords per row of lut (multiple of 8)
    float m[9];           // k_rinv = K * R^-1 (Projector::k_rinv)
    const float2* colA;   // [tw] {sin(u/s), cos(u/s)} with the REFLECT border of feed() folded in
    const float2* rowB;   // [th] {sin(pi - v/s) | 1, cos(pi - v/s) | v/s}
    const FrontEndDev* fe; // nullptr, or the undistort front end: src is then the RAW frame (src_w x src_h raw)
    int out_w, out_h;     // the stitcher's frame size (mask warp inside test); == src_w x src_h without a front end
    const int2* grow;     // [th] {sy0, sy1}
    const float2* groww;  // [th] {1-fy, fy}
    int gw;
};
constexpr int kGainRows = 4;
inline int warp_pack_live(int live_by0, int gap_bx0, int gap_len) { return live_by0 | gap_bx0 << 12 | (int)((unsigned)gap_len << 22); }
// How the table form of K1 deals its work: the live 64 x 16 blocks of every camera of the launch, camera after camera and row
// after row, form one list; XCD k (= workgroup id mod 8: grid.x is 8) takes the k-th eighth of it.  With as many cameras as XCDs
// that is "one camera per XCD" (an XCD's L2 holds one camera's frame) with the cameras' unequal live areas evened out at the
// joints: dealt camera = XCD, the XCDs of config 2's largest cameras ran 23 % more waves than those of the smallest (5208 against
// 4216) and finished 4 us after them (tools/wave_timeline.py).  Config 2: 18.2 -> 17.9 us warm, 21.8 -> 20.9 cold; config 4:
// 60.1 -> 56.3 us (66.0 -> 63.2 with gains).
struct WarpDeal {         // passed to the kernel as nine scalar arguments (preloaded into SGPRs at wave launch)
    unsigned end[kCams];  // end[c] = blocks of cameras 0..c (cameras past the last: the total)
    unsigned per;         // blocks per XCD = ceil(total / 8)
};
struct WarpParams {
    WarpCam cam[kCams];
};

// one camera's pyramid slot.  Every Gaussian level of an 8-bit image stays in [0,255], so the levels are
// stored as PLANAR uint8 (B plane, G plane, R plane): half the bytes of OpenCV's CV_16SC3 and
// dword-vectorisable stencils; the widening to int16 happens in registers where the Laplacian is formed.
struct PyrCam {
    uint8_t* lvl[kLevels];     // plane B of each Gaussian level; plane c at + c * plane[l]
    int pitch[kLevels];        // bytes per row (multiple of 16)
    int plane[kLevels];        // bytes per plane
    const uint8_t* mask0;      // level-0 weight source: blend mask with the CONSTANT 0 border of feed(), tile sized, pitch[0]
    const float* wgt[kLevels]; // f32 weight levels (level 0 = mask0 * (1/255.f), kept for the pyrDown chain)
    int wpitch[kLevels];       // floats per row
    int w0, h0;                // level-0 tile size (multiples of 2^bands)
    int tx, ty;                // tile origin in the padded canvas (level 0)
    int live[kLevels][4];      // per level {x0, y0, x1, y1} inclusive: the pixels of the level anything downstream reads
    int gap[kLevels][2];       // per level dead columns {x0, x1} inside live (x1 < x0: none): a +-pi straddler's middle
};
struct PyrParams {
    PyrCam cam[kCams];
    int ncam;
    // set by launch_pyr_down for the launch: per camera {live workgroup columns, 2^32 / columns + 1, workgroups per plane (columns x
    // rows), 2^32 / that + 1} (a multiplier of 0: divide by 1) - the kernel's own deal of its live workgroups over the XCDs
    unsigned deal[kCams][4];
};

constexpr int kOrderLevels = 3;  // blend levels that can run in seam-first tile order (the vector levels of five bands)
struct CanvasParams {
    int16_t* img[kLevels];     // collapsed canvas levels, planar int16 (level 0 is never materialised)
    int cpitch[kLevels];       // int16 elements per row
    int cplane[kLevels];       // int16 elements per plane
    int fast[kLevels];         // 1: every tile origin / size of this level is 4x2 aligned -> vector kernel
    const uint16_t* owner[kLevels]; // vector levels, per 4x2 block: low byte = the single unit-weight camera (0..7), 0xFE none,
                               // 0xFF mixed; high byte = bit i set when camera i carries weight anywhere on the block
    int opitch[kLevels];       // owner entries per block row
    int small_base;            // >0: levels small_base..bands run as one normalise launch + one LDS collapse launch
    // vector levels 0 .. kOrderLevels - 1, or nullptr: per XCD band (order_per entries each, low half 0xffff = none) the 128 x 16-pixel
    // workgroup tiles in the order they are dispatched - seam tiles first; low half = bx | by << 8; high half: what the tile's four
    // waves will find in the owner map, a nibble each (wave = threadIdx.y): 0..7 the single owner of every block of the wave,
    // 0xE no owner anywhere, 0xF look it up
    const uint32_t* order[kOrderLevels];
    int order_per[kOrderLevels];
    int cam_lo, cam_n;         // this canvas blends cameras [cam_lo, cam_lo + cam_n) of the PyrParams it is launched with
    int w0, h0;                // padded canvas size
    int bands;
    // final output
    uint8_t* out;
    int out_stride;            // bytes
    int cut_x, cut_y, cut_w, cut_h;  // in padded-canvas coordinates (pano rect origin == canvas origin)
    int final_w, final_h;      // dst_roi_final size (unpadded)
    // what the vector blend kernels read of this block for the level they are launched on (BlendLevel), packed by
    // launch_blend_level so that a wave fetches it with two scalar loads at its first instructions
    alignas(16) int hot[24];
};
// A canvas at one level, as the vector blend kernels see it (CanvasParams::hot holds exactly this)
struct BlendLevel {
    int cut_x, cut_y, cut_w, cut_h;  // level 0: the cut rectangle
    int cw, ch;                      // size of the level (w0 >> l, h0 >> l)
    int up;                          // != 0: there is a coarser level (l < bands)
    int cam_lo;
    uint8_t* out;                    // level 0: the panorama
    int out_stride;
    int opitch;                      // owner entries per block row of this level
    int16_t* img;                    // this level's collapsed canvas (levels >= 1: written here)
    const int16_t* img_up;           // level l + 1
    const uint16_t* owner;
    int cpitch, cplane, cpitch_up, cplane_up;
    int pad_[2];
};
static_assert(sizeof(BlendLevel) == 24 * sizeof(int), "BlendLevel is CanvasParams::hot");
inline void pack_blend_level(CanvasParams& C, int l) {
    BlendLevel v{};
    v.cut_x = C.cut_x; v.cut_y = C.cut_y; v.cut_w = C.cut_w; v.cut_h = C.cut_h;
    v.cw = C.w0 >> l; v.ch = C.h0 >> l;
    v.up = l < C.bands ? 1 : 0;
    v.cam_lo = C.cam_lo;
    v.out = C.out; v.out_stride = C.out_stride;
    v.opitch = C.opitch[l];
    v.img = C.img[l];
    v.img_up = l + 1 < kLevels ? C.img[l + 1] : nullptr;
    v.owner = C.owner[l];
    v.cpitch = C.cpitch[l]; v.cplane = C.cplane[l];
    v.cpitch_up = l + 1 < kLevels ? C.cpitch[l + 1] : 0; v.cplane_up = l + 1 < kLevels ? C.cplane[l + 1] : 0;
    static_assert(sizeof(v) == sizeof(C.hot), "hot block");
    __builtin_memcpy(C.hot, &v, sizeof(v));
}

// Up to two canvases (the reference's upper and lower stitcher) share every blend launch: grid.z picks the canvas.
// Their level structure (bands, vector levels, small_base) must be identical; their geometry need not be.
struct CanvasSet {
    CanvasParams c[2];
    int n;
};

// K1: fused REFLECT border + mapBackward + fixed-point bilinear remap + 8U->16S
// ev_start/ev_stop (optional) receive the dispatch's own begin/end timestamps (hipExtLaunchKernelGGL)
void launch_warp_tiles(const WarpParams& p, int ncam, int max_tw, int max_th, hipStream_t s,
                       hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// builds the static remap table and the source boxes of one camera tile (run once per pano_prepare);
// counters[0] = workgroups whose box does not fit LDS, counters[1] = workgroups whose source span does not fit the code
void launch_build_warp_table(const WarpCam& c, uint32_t* lut, int lut_pitch, int4* boxes, unsigned* counters, hipStream_t s);
void launch_pack_warp_lut(const uint32_t* lut, int lut_pitch, int tw, int th, uint2* lutc, int lutc_pitch, uint32_t* flags,
                          hipStream_t s);
// stage entry: plain RotationWarper::warp to an 8UC3 image
void launch_warp_image(const WarpCam& c, hipStream_t s);
// RotationWarper::warp(mask255, INTER_NEAREST, BORDER_CONSTANT)
void launch_warp_mask(const WarpCam& c, uint8_t* dst, int dst_stride, hipStream_t s);

// K2: one pyrDown level (16S x3) for the selected cameras: level `l` -> level `l+1`
void launch_pyr_down(const PyrParams& p, unsigned cam_bits, int l, hipStream_t s);
// K2 tail: levels b -> b+1 -> ... -> t (1 <= t - b <= 4) in ONE launch, tiled through LDS with the halo recomputed
constexpr int kPyrTailTile = 32;  // the tile of level b + 1 a workgroup owns
constexpr int kPyrTailBase = 2;   // the per-frame chain runs plain launches up to this level and the tail above it
void launch_pyr_tail(const PyrParams& p, unsigned cam_bits, int b, int t, hipStream_t s);
bool pyr_tail_ok(const PyrParams& p, unsigned cam_bits, int t);  // every source level of the tail at least 4 x 4
// the per-frame pyrDown chain 0 -> levels: plain launches up to level base = max(kPyrTailBase, levels - 4), levels base ..
// top as one pyr_tail launch (fewer than two levels left, or levels too small for it: plain launches)
inline void launch_pyr_chain(const PyrParams& p, unsigned cam_bits, int levels, hipStream_t s) {
    int plain = levels;
    const int b = kPyrTailBase > levels - 4 ? kPyrTailBase : levels - 4;
    if (levels - b >= 2 && (kPyrTailTile >> (levels - b - 1)) >= 4 && pyr_tail_ok(p, cam_bits, levels)) plain = b;
    int l = 0;
    for (; l < plain; l++) launch_pyr_down(p, cam_bits, l, s);
    if (l < levels) launch_pyr_tail(p, cam_bits, l, levels, s);
}
// K3: one blend level for the whole canvas (Laplacian, weight, accumulate, normalise, collapse);
// level 0 writes the cut 8U panorama
// ev_start / ev_stop (optional, level 0): the dispatch's own begin / end timestamps
void launch_blend_level(const PyrParams& p, const CanvasSet& cs, int l, hipStream_t s, hipEvent_t ev_start = nullptr,
                        hipEvent_t ev_stop = nullptr);
// the small levels small_base..bands in two launches: normalise, then the collapse chain through LDS
void launch_blend_small(const PyrParams& p, const CanvasSet& cs, hipStream_t s);
// per 128 x 16-pixel tile of vector level l (gx x gy of them; level 0: over the hull of the cut): the owner nibbles of its four
// waves (0xF: no single owner)
void launch_tile_mixed(const CanvasParams& c, int l, int gx, int gy, uint16_t* flags, hipStream_t s);
// owner map of a vector level (run when masks change)
void launch_build_owner(const PyrParams& p, const CanvasParams& c, int l, uint16_t* owner, hipStream_t s);
// Blender::NO path
void launch_no_blend(const PyrParams& p, const CanvasSet& cs, hipStream_t s);

// weights (run when masks change)
void launch_mask_to_weight(const uint8_t* mask, int mw, int mh, int mpitch, int left, int top,
                           float* w0, int wpitch, uint8_t* m0, int mpitch0, int tw, int th, hipStream_t s);
// which association of cv::pyrDown CV_32F's five-tap sums the weight pyramids follow (PANO_PYRDOWN32F_ORDER; oracle/pano_oracle.c):
// vertical 0 scalar / 1 SSE2, universal intrinsics / 2 NEON with a vector body of vbody floats; horizontal 0 scalar / 1 universal
// intrinsics / 2 the same with a fused multiply-add, body hbody floats
struct F32Order { int vertical = 0, vbody = 8, horizontal = 0, hbody = 4; };
void launch_pyr_down_f32(const float* src, int sw, int sh, int spitch, float* dst, int dpitch, F32Order o, hipStream_t s);
void launch_sum_weights(const PyrParams& p, int l, float* wsum, int cw, int ch, hipStream_t s);

// caller-side assembly: out rows [0, top_h) = up (resized to out_w x top_h with cv::resize INTER_LINEAR semantics, or
// copied from row up_y0 when up_w == out_w && up_rows == top_h), rows [top_h, 2*top_h) = down from row down_y0;
// rows [bar_y, bar_y + bar_h) black
void launch_stack(const uint8_t* up, int up_w, int up_h, int up_stride, int up_y0, bool resize_up,
                  const uint8_t* down, int down_stride, int down_y0, uint8_t* out, int out_w, int top_h, int out_stride,
                  int bar_y, int bar_h, hipStream_t s);

// mask preparation (Voronoi)
void launch_dilate3x3(const uint8_t* src, uint8_t* dst, int w, int h, hipStream_t s);
void launch_resize_linear_exact(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh,
                                const int* xofs, const int* xc1, const int* yofs, const int* yc1, int minx, int maxx,
                                int miny, int maxy, hipStream_t s);
// gain estimation (GainCompensator::feed): dense 8UC3 images / 8U masks of the seam-scale warps and the overlapping
// pairs of sub-images (blocks), each a rectangle of w x h pixels at (ax, ay) in image a and (bx, by) in image b
struct GainImages {
    const uint8_t* img[kCams];
    const uint8_t* mask[kCams];
    int w[kCams];
};
struct GainPair {
    int a, ax, ay, b, bx, by, w, h;
};
// graph-cut seam finder (GraphCutSeamFinder::Impl::findInPair): the padded overlap ROI of images a and b as a W x H
// grid; (ax, ay) / (bx, by) = image coordinates of grid vertex (0, 0) (may be negative: the gap reaches outside)
struct GcPair {
    int a, ax, ay, wa, ha, b, bx, by, wb, hb, W, H;
};
void launch_graphcut_weights(const GainImages& g, const GcPair& q, float* term, float* wh, float* wv, hipStream_t s);
void launch_graphcut_apply(const GcPair& q, uint8_t* mask_a, uint8_t* mask_b, const uint8_t* in_source, int gap, hipStream_t s);
void launch_gain_pairs(const GainImages& g, const GainPair* pairs, int npairs, int* count, double* sum_a, double* sum_b,
                       hipStream_t s);
void launch_and(const uint8_t* a, const uint8_t* b, uint8_t* dst, size_t n, hipStream_t s);
// one wave that keeps stream s busy for ticks / 100 MHz seconds (pano_frame_streams' probe)
void launch_spin(unsigned long long ticks_100mhz, hipStream_t s);
// one rectangle of the sharded exchange: `rows` rows of `width16` 16-byte chunks at slots + slot_off (row pitch `pitch` bytes) <-> packed
// at stage + stage_off (rows tight)
struct XchSeg {
    size_t slot_off, stage_off;
    int pitch, width16, rows, pad;
};
void launch_copy_segments(const XchSeg* d_segs, int first, int count, int max_rows, uint8_t* slots, uint8_t* stage, bool unpack, hipStream_t s);
// measurement probes (pano_probe.hip, pano_probe_copy): a float4 grid-stride copy and a copy in K1's traffic shape; each launch
// carries its own begin / end events like K1's (hipExtLaunchKernelGGL)
constexpr int kProbeBoxBytes = 4608;   // 18 rows x 256 B: config 2's K1 copies 4.4 KB of frame lines per workgroup
void launch_probe_copy_f4(const void* src, void* dst, size_t bytes, int blocks, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
void launch_probe_copy_k1_shape(const void* box_src, const void* table, void* dst, unsigned workgroups, hipStream_t s, hipEvent_t e0,
                                hipEvent_t e1);
// one VoronoiSeamFinder::findInPair on device masks
void launch_voronoi_pair(uint8_t* mask1, int w1, int h1, int tlx1, int tly1,
                         uint8_t* mask2, int w2, int h2, int tlx2, int tly2,
                         int rx, int ry, int rw, int rh, int* scratch, hipStream_t s);
size_t voronoi_scratch_ints(int rw, int rh);

}  // namespace pano
