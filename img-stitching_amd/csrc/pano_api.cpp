// pano_api.cpp - C-ABI of libpano_hip.so (include/pano.h): context, buffers, launch sequencing.
// Host code only; the arithmetic is in the kernel files (pano_warp / pano_pyramid / pano_blend / pano_blend_small / pano_init .hip), the init-time geometry in pano_plan.hpp.
// No CPU fallback exists: every compute entry point launches HIP kernels or fails.

#include "../../include/pano.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <mutex>
#include <new>
#include <stdexcept>
#include <sstream>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "pano_graphcut.hpp"
#include "pano_hostcopy.hpp"
#include "pano_rccl.hpp"
#include "pano_kernels.hpp"
#include "pano_plan.hpp"

using namespace pano;

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct MaskJob;  // a graph-cut mask refresh running beside the frame loop (pano_refresh_masks_*)

struct pano_ctx {
    pano_config cfg{};
    bool have_cam[kMaxCams] = {};
    float K[kMaxCams][9] = {}, R[kMaxCams][9] = {};
    float scale = 0.f;
    Plan plan;
    bool prepared = false;
    int device = -1;
    int levels = 0;  // bands + 1 (1 for Blender::NO)

    // per camera device data
    float2 *colA[kMaxCams] = {}, *rowB[kMaxCams] = {};          // bordered-tile tables (K1)
    float2 *colA_roi[kMaxCams] = {}, *rowB_roi[kMaxCams] = {};  // ROI tables (stage warp, mask warp)
    uint8_t* mask[kMaxCams] = {};                                // m_blenderMask, ROI sized, tight rows
    bool mask_set[kMaxCams] = {};
    bool weights_dirty = true;
    // gain
    float* gain[kMaxCams] = {};
    int gain_w[kMaxCams] = {}, gain_h[kMaxCams] = {};
    int2 *grow[kMaxCams] = {}, *grow_roi[kMaxCams] = {};
    float2 *groww[kMaxCams] = {}, *groww_roi[kMaxCams] = {};
    float *ghrow[kMaxCams] = {}, *ghrow_roi[kMaxCams] = {};  // horizontally resized gain map rows (tile / ROI columns)
    int* grow_base[kMaxCams] = {};                           // WarpCam::grow_base
    int4* grow4[kMaxCams] = {};                              // WarpCam::grow4
    int ghrow_pitch[kMaxCams] = {}, ghrow_roi_pitch[kMaxCams] = {};

    // pyramid slots (one allocation), weights, canvas
    char* pyr_base = nullptr;
    size_t slot_bytes = 0;
    size_t lvl_off[kMaxCams][kMaxLevels] = {};
    int lvl_pitch[kMaxCams][kMaxLevels] = {};
    int lvl_plane[kMaxCams][kMaxLevels] = {};
    int wpitch[kMaxCams][kMaxLevels] = {};
    float* wgt[kMaxCams][kMaxLevels] = {};
    uint8_t* mask0[kMaxCams] = {};  // level-0 tile-sized mask with the CONSTANT border of feed()
    // fused undistort front end
    bool have_fe[kMaxCams] = {};
    pano_undistort und[kMaxCams] = {};
    double newK[kMaxCams][9] = {};
    FrontEndDev* d_fe[kMaxCams] = {};
    int frame_w = 0, frame_h = 0;   // size of the frames pano_compose takes (raw size with a front end)
    uint32_t* lut[kMaxCams] = {};   // static remap tables of K1 (dense form, codes relative to the workgroup's source box)
    int lut_pitch[kMaxCams] = {};
    uint2* lutc[kMaxCams] = {};         // packed form of lut (8 bytes per 4 pixels), read by unflagged workgroups
    int4* box[kMaxCams] = {};           // source box of every 64x16-pixel workgroup of K1
    long long box_global[kMaxCams] = {}; // workgroups whose box does not fit LDS (global taps)
    std::vector<int4> h_box[kMaxCams];   // host copy of box[]: which frame bytes each K1 workgroup reads (static)
    // The frame bytes K1 reads with the present masks: byte columns [x0, x0 + w) of rows [y0, y0 + rows), x0 and w multiples of
    // 64 (a rectangular DMA runs at the link rate only when aligned: 50 GB/s against 5).  The host entries upload nothing else
    struct SrcRect { int x0, y0, w, rows; };
    SrcRect src_rect[kMaxCams] = {};
    uint32_t* k1_flags[kMaxCams] = {};  // per K1 workgroup: the table holds marked pixels there
    long long k1_blocks[kMaxCams] = {}, k1_flagged[kMaxCams] = {};
    bool use_lut = true;
    uint16_t* owner[kMaxLevels] = {};
    uint32_t* order[kOrderLevels] = {};   // CanvasParams::order
    size_t order_cap[kOrderLevels] = {};
    bool order_dirty = false;
    bool l0_order = true;            // PANO_L0_ORDER=0: plain band order
    float* wsum[kMaxLevels] = {};
    int16_t* canvas[kMaxLevels] = {};

    PyrParams pyr{};
    CanvasParams cv{};
    // frame slots (pano_set_frame_slots): extra sets of the per-frame buffers - pyramid slots and blend canvas - so
    // that several frames can be in flight on several streams.  Slot 0 is pyr_base / canvas[] above.
    // live rects: per camera and level the pixels {x0, y0, x1, y1} (inclusive, tile coordinates of the level) that the
    // blend ever reads, directly or through the pyramid chain; K1 / K2 do not produce the rest (see live_rects)
    int live[kMaxCams][kMaxLevels][4] = {};
    // dead columns {x0, x1} (inclusive, x1 < x0 = none) inside the live rect: the middle of a +-pi straddler's tile
    int gap[kMaxCams][kMaxLevels][2] = {};
    bool full_tiles = false;  // PANO_FULL_TILES=1: produce every pixel of every level (stage inspection)
    int nslots = 1, cur_slot = 0;
    char* slot_pyr[PANO_MAX_FRAME_SLOTS] = {};
    int16_t* slot_canvas[PANO_MAX_FRAME_SLOTS][kMaxLevels] = {};

    // host-buffer entry point staging
    uint8_t* stage_in[kMaxCams] = {};
    size_t stage_in_pitch = 0;
    uint8_t* stage_out = nullptr;
    size_t stage_out_pitch = 0, stage_out_bytes = 0;
    hipStream_t own_stream = nullptr;
    std::vector<hipStream_t> flight_streams;  // pano_frame_streams: owned here
    int flight_distinct = 0;
    // ... and its page-locked host side (pageable caller memory is copied through these by the pool's threads)
    uint8_t* pin_in[kMaxCams] = {};
    size_t pin_in_pitch = 0;
    uint8_t* pin_out = nullptr;
    hipStream_t host_h2d[2] = {};
    hipEvent_t host_in_ready[2] = {};
    uint8_t* stack_buf = nullptr;   // pano_stack_*_host: both halves + the stacked image on the device
    size_t stack_bytes = 0;
    double host_trace[5] = {};   // PANO_HOST_TRACE: stage in + queue H2D | queue kernels | H2D + kernels done | copy back | unstage
    long host_trace_n = 0;
    // streaming slots (pano_stream_*): pinned host buffers, per-slot device buffers, copy streams and events
    struct StreamSlot {
        uint8_t* h_in[kMaxCams] = {};
        uint8_t* d_in[kMaxCams] = {};
        uint8_t* h_out = nullptr;
        uint8_t* d_out = nullptr;
        hipStream_t h2d = nullptr, d2h = nullptr;
        hipEvent_t in_ready = nullptr, composed = nullptr, out_ready = nullptr;
        bool busy = false;
    };
    StreamSlot slots[PANO_STREAM_SLOTS];
    bool slots_ready = false;
    size_t slot_in_pitch = 0, slot_out_pitch = 0;

    // profiling: a ring of event quads so that the timed loop never has to wait for the GPU
    static constexpr int kEvRing = 64;
    struct EvSlot {
        hipEvent_t e[6];  // 0..3: K1 begin, K1 end, pyramid end, blend end; 4, 5: begin / end of the level-0 blend dispatch
        unsigned recorded;
    };
    bool profiling = false;
    EvSlot ring[kEvRing] = {};
    bool ev_valid = false;
    int ev_head = 0, ev_count = 0, ev_cur = -1;
    double acc_ms[PANO_NUM_STAGES] = {};
    uint64_t acc_n[PANO_NUM_STAGES] = {};
    float last_ms[PANO_NUM_STAGES] = {};

    // hipGraph cache of the per-frame launch sequence, keyed by the caller's buffers
    struct GraphEntry {
        const uint8_t* frames[kMaxCams];
        size_t strides[kMaxCams];
        uint8_t* out;
        size_t out_stride;
        int slot;  // the frame slot whose buffers the captured launches point at
        hipGraph_t graph;
        hipGraphExec_t exec;
    };
    std::vector<GraphEntry> graphs;
    bool use_graph = false;
    uint64_t graph_replays = 0;   // hipGraphLaunch calls so far (pano_debug_graph_stats)
    std::string gc_dump_path;     // pano_debug_graphcut_dump

    MaskJob* job = nullptr;
    MaskJob* job_trash = nullptr;  // (unused since the pool: kept for a refresh that failed half way)
    std::vector<std::pair<size_t, void*>> refresh_pool;  // device buffers of the last refresh, reused by the next (Scratch::pool)
    std::vector<std::pair<size_t, void*>> pairs_pool;    // ... and the graphs of its pairs: the refresh thread's while it runs

    std::string err;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return PANO_EHIP;                                                                   \
        }                                                                                       \
    } while (0)

pano_status fail(pano_ctx* c, pano_status s, const char* msg) {
    if (c) c->err = msg;
    return s;
}

template <typename T>
pano_status upload(pano_ctx* c, T** dptr, const void* h, size_t bytes) {
    if (*dptr) {
        HIP_TRY(c, hipFree(*dptr));
        *dptr = nullptr;
    }
    HIP_TRY(c, hipMalloc((void**)dptr, bytes ? bytes : 16));
    if (bytes) HIP_TRY(c, hipMemcpy(*dptr, h, bytes, hipMemcpyHostToDevice));
    return PANO_OK;
}

template <typename T>
void dfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void drop_graphs(pano_ctx* c) {
    for (auto& g : c->graphs) {
        (void)hipGraphExecDestroy(g.exec);
        (void)hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
}

void free_device(pano_ctx* c) {
    drop_graphs(c);
    for (auto& q : c->refresh_pool) (void)hipFree(q.second);
    c->refresh_pool.clear();
    for (auto& q : c->pairs_pool) (void)hipFree(q.second);
    c->pairs_pool.clear();
    // slot 0 owns what pyr_base / canvas[] were allocated as; the current slot may be another one
    if (c->nslots > 1) {
        c->pyr_base = c->slot_pyr[0];
        for (int l = 0; l < kMaxLevels; l++) c->canvas[l] = c->slot_canvas[0][l];
        for (int k = 1; k < c->nslots; k++) {
            dfree(c->slot_pyr[k]);
            for (int l = 0; l < kMaxLevels; l++) dfree(c->slot_canvas[k][l]);
        }
    }
    for (int k = 0; k < PANO_MAX_FRAME_SLOTS; k++) {
        c->slot_pyr[k] = nullptr;
        for (int l = 0; l < kMaxLevels; l++) c->slot_canvas[k][l] = nullptr;
    }
    c->nslots = 1;
    c->cur_slot = 0;
    for (int i = 0; i < kMaxCams; i++) {
        dfree(c->colA[i]); dfree(c->rowB[i]); dfree(c->colA_roi[i]); dfree(c->rowB_roi[i]);
        dfree(c->mask[i]); dfree(c->gain[i]); dfree(c->mask0[i]); dfree(c->lut[i]); dfree(c->lutc[i]); dfree(c->box[i]); dfree(c->k1_flags[i]); dfree(c->d_fe[i]);
        dfree(c->grow[i]); dfree(c->grow_roi[i]); dfree(c->groww[i]); dfree(c->groww_roi[i]);
        dfree(c->ghrow[i]); dfree(c->ghrow_roi[i]); dfree(c->grow_base[i]); dfree(c->grow4[i]);
        dfree(c->stage_in[i]);
        for (int l = 0; l < kMaxLevels; l++) dfree(c->wgt[i][l]);
    }
    for (int l = 0; l < kOrderLevels; l++) {
        dfree(c->order[l]);
        c->order_cap[l] = 0;
    }
    for (int l = 0; l < kMaxLevels; l++) {
        dfree(c->owner[l]);
        dfree(c->wsum[l]);
        dfree(c->canvas[l]);
    }
    for (auto& sl : c->slots) {
        for (int i = 0; i < kMaxCams; i++) {
            if (sl.h_in[i]) (void)hipHostFree(sl.h_in[i]);
            sl.h_in[i] = nullptr;
            dfree(sl.d_in[i]);
        }
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        sl.h_out = nullptr;
        dfree(sl.d_out);
        // (the copy streams are process-wide: not this context's to destroy)
        if (sl.in_ready) (void)hipEventDestroy(sl.in_ready);
        if (sl.composed) (void)hipEventDestroy(sl.composed);
        if (sl.out_ready) (void)hipEventDestroy(sl.out_ready);
        sl.h2d = sl.d2h = nullptr;
        sl.in_ready = sl.composed = sl.out_ready = nullptr;
    }
    c->slots_ready = false;
    dfree(c->pyr_base);
    dfree(c->stage_out);
    c->stage_out_bytes = 0;
    dfree(c->stack_buf);
    c->stack_bytes = 0;
    for (int i = 0; i < kMaxCams; i++) {
        if (c->pin_in[i]) (void)hipHostFree(c->pin_in[i]);
        c->pin_in[i] = nullptr;
    }
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    c->pin_out = nullptr;
    for (auto& hs : c->host_h2d) hs = nullptr;  // the device's shared copy queues: not this context's to destroy
    for (auto& he : c->host_in_ready) {
        if (he) (void)hipEventDestroy(he);
        he = nullptr;
    }
    if (c->ev_valid)
        for (auto& sl : c->ring)
            for (auto& e : sl.e) (void)hipEventDestroy(e);
    c->ev_valid = false;
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    c->own_stream = nullptr;
    for (hipStream_t fs : c->flight_streams) (void)hipStreamDestroy(fs);
    c->flight_streams.clear();
}

// WarpCam for the bordered feed() tile of camera i (pipeline) or its plain ROI (stage entries)
WarpCam make_warp_cam(const pano_ctx* c, int i, const uint8_t* src, size_t stride, bool roi_only) {
    WarpCam w{};
    w.src = src;
    w.src_w = c->frame_w;
    w.src_h = c->frame_h;
    w.out_w = c->plan.src_w;
    w.out_h = c->plan.src_h;
    w.fe = c->d_fe[i];
    w.src_stride = (int)stride;
    std::memcpy(w.m, c->plan.proj[i].k_rinv, sizeof(w.m));
    if (roi_only) {
        w.colA = c->colA_roi[i]; w.rowB = c->rowB_roi[i];
        w.tw = c->plan.roi[i].w; w.th = c->plan.roi[i].h;
        w.ghrow = c->ghrow_roi[i]; w.ghrow_pitch = c->ghrow_roi_pitch[i]; w.grow = c->grow_roi[i]; w.groww = c->groww_roi[i];
    } else {
        w.colA = c->colA[i]; w.rowB = c->rowB[i];
        w.tw = c->plan.tile[i].rect.w; w.th = c->plan.tile[i].rect.h;
        w.dst = c->pyr_base + (size_t)i * c->slot_bytes + c->lvl_off[i][0];
        w.dst_pitch = c->lvl_pitch[i][0];
        w.dst_plane = c->lvl_plane[i][0];
        w.lut = c->use_lut ? c->lut[i] : nullptr;
        w.lut_pitch = c->lut_pitch[i];
        w.lutc = c->use_lut ? c->lutc[i] : nullptr;
        w.box = c->use_lut ? c->box[i] : nullptr;
        w.lutc_pitch = c->lut_pitch[i] / 4;
        w.ghrow = c->ghrow[i]; w.ghrow_pitch = c->ghrow_pitch[i]; w.grow = c->grow[i]; w.groww = c->groww[i];
        w.grow_base = c->grow_base[i];
        w.grow4 = c->grow4[i];
    }
    w.gain = c->gain[i];
    w.gw = c->gain_w[i];
    w.gh = c->gain_h[i];
    w.live_bx0 = 0; w.live_bx1 = INT_MAX; w.live_by0_gap = 0; w.live_by1 = INT_MAX;
    if (!roi_only) {  // the bordered feed() tile: only the 64 x 16 blocks that overlap the live rect of level 0
        const int* L = c->live[i][0];
        if (L[2] < L[0] || L[3] < L[1]) { w.live_bx0 = 1; w.live_bx1 = 0; w.live_by0_gap = 1; w.live_by1 = 0; }  // nothing
        else {
            w.live_bx0 = L[0] >> 6; w.live_bx1 = L[2] >> 6; w.live_by1 = L[3] >> 4;
            // dead columns gap[0] .. gap[1] of level 0: the block columns that lie entirely inside them
            const int* G = c->gap[i][0];
            const int g0 = (G[0] + 63) >> 6, g1 = ((G[1] + 1) >> 6) - 1;
            const bool has = G[1] >= G[0] && g1 >= g0;
            w.live_by0_gap = warp_pack_live(L[1] >> 4, has ? g0 : 0, has ? g1 - g0 + 1 : 0);
        }
    }
    return w;
}

// Which pixels of which pyramid level does anything ever read?  MultiBandBlender::feed weighs the Laplacian of camera
// i with its weight pyramid, and those weights are zero away from the camera's blend mask: per level l
//   R_l   = support of w_l                           (R_0 = bounding box of the mask, R_{l+1} = pyrDown footprint of R_l)
//   N_l   = R_l  U  the pyrUp windows the blend opens at level l for the blocks of R_{l-1}
//   S_top = N_top,   S_l = N_l  U  the 5x5 pyrDown footprint of S_{l+1}
// S_l is what K1 (l = 0) and K2 (l >= 1) have to produce; the rest of the bordered tile (on config 2 a quarter of it)
// is never read by anything and is not produced.  Bounding boxes, all conservative; recomputed with the weights.
// A camera that straddles the +-pi seam of the projection has a full-width ROI (RotationWarper::warpRoi takes min / max
// of u) whose mask lives at the two ends.  The need sets are unions over the mask's support (w_l is linear in the mask),
// so the mask is cut at its widest run of empty columns (>= kMinDeadColumns) into two pieces, each piece gets its own
// chain of rectangles, and per level the columns between the two are recorded as the dead gap K1 / K2 step over.
static void live_source_rects(pano_ctx* c);
static void live_rects(pano_ctx* c, const std::vector<std::vector<uint8_t>>& masks) {
    constexpr int kMinDeadColumns = 256;
    const Plan& P = c->plan;
    const int top = std::max(P.bands, 0);
    for (int i = 0; i < P.n; i++) {
        const FeedTile& t = P.tile[i];
        auto full = [&](int l, int* r) { r[0] = 0; r[1] = 0; r[2] = (t.rect.w >> l) - 1; r[3] = (t.rect.h >> l) - 1; };
        for (int l = 0; l < kMaxLevels; l++) { c->gap[i][l][0] = 1; c->gap[i][l][1] = 0; }  // no gap
        if (c->full_tiles || masks.empty() || P.bands < 0) {
            for (int l = 0; l < c->levels; l++) full(l, c->live[i][l]);
            continue;
        }
        // per mask column: the rows it occupies
        const int mw = P.roi[i].w, mh = P.roi[i].h;
        std::vector<int> ctop(mw, INT_MAX), cbot(mw, -1);
        const uint8_t* m = masks[i].data();
        for (int y = 0; y < mh; y++)
            for (int x = 0; x < mw; x++)
                if (m[(size_t)y * mw + x]) { ctop[x] = std::min(ctop[x], y); cbot[x] = y; }
        int x0 = 0, x1 = mw - 1;
        while (x0 < mw && cbot[x0] < 0) x0++;
        while (x1 >= 0 && cbot[x1] < 0) x1--;
        if (x1 < x0) {  // empty mask: nothing is live
            for (int l = 0; l < c->levels; l++) { c->live[i][l][0] = c->live[i][l][1] = 1; c->live[i][l][2] = c->live[i][l][3] = 0; }
            continue;
        }
        int e0 = 0, e1 = -1;  // widest run of empty columns inside [x0, x1]
        for (int x = x0; x <= x1;) {
            if (cbot[x] >= 0) { x++; continue; }
            int r = x;
            while (cbot[r + 1] < 0) r++;  // column x1 is occupied: the run ends before it
            if (r - x > e1 - e0) { e0 = x; e1 = r; }
            x = r + 1;
        }
        const bool two = e1 - e0 + 1 >= kMinDeadColumns;
        auto clampl = [&](int l, int* r) {
            const int w = t.rect.w >> l, h = t.rect.h >> l;
            r[0] = std::max(r[0], 0); r[1] = std::max(r[1], 0); r[2] = std::min(r[2], w - 1); r[3] = std::min(r[3], h - 1);
        };
        auto uni = [](int* a, const int* b) { a[0] = std::min(a[0], b[0]); a[1] = std::min(a[1], b[1]); a[2] = std::max(a[2], b[2]); a[3] = std::max(a[3], b[3]); };
        // the chain of one piece: mask columns [cx0, cx1] -> S[l], what K1 (l = 0) and K2 (l >= 1) have to produce for it
        auto chain = [&](int cx0, int cx1, int S_out[kMaxLevels][4]) {
            int y0 = INT_MAX, y1 = -1;
            for (int x = cx0; x <= cx1; x++)
                if (cbot[x] >= 0) { y0 = std::min(y0, ctop[x]); y1 = std::max(y1, cbot[x]); }
            int R[kMaxLevels][4], N[kMaxLevels][4];
            R[0][0] = cx0 + t.left; R[0][1] = y0 + t.top; R[0][2] = cx1 + t.left; R[0][3] = y1 + t.top;
            for (int l = 0; l < top; l++) {  // w_{l+1}(p) != 0 only if w_l is != 0 somewhere in [2p-2, 2p+2]
                R[l + 1][0] = (R[l][0] - 1) >> 1; R[l + 1][1] = (R[l][1] - 1) >> 1;   // ceil((x0 - 2) / 2)
                R[l + 1][2] = (R[l][2] + 2) >> 1; R[l + 1][3] = (R[l][3] + 2) >> 1;
                clampl(l + 1, R[l + 1]);
            }
            for (int l = 0; l <= top; l++) {
                for (int k = 0; k < 4; k++) N[l][k] = R[l][k];
                if (l > 0) {
                    // the blend works on 4 x 2 blocks of level l-1 and opens the coarse window columns (X0>>1)-1 .. +2,
                    // rows (Y0>>1)-1 .. +1 around each (load_coarse); the scalar kernels' pyr_up_px windows lie inside
                    const int W[4] = {((R[l - 1][0] >> 2) << 1) - 1, (R[l - 1][1] >> 1) - 1, ((R[l - 1][2] >> 2) << 1) + 2, (R[l - 1][3] >> 1) + 1};
                    uni(N[l], W);
                    clampl(l, N[l]);
                }
            }
            int S[4] = {N[top][0], N[top][1], N[top][2], N[top][3]};
            for (int k = 0; k < 4; k++) S_out[top][k] = S[k];
            for (int l = top - 1; l >= 0; l--) {
                const int F[4] = {2 * S[0] - 2, 2 * S[1] - 2, 2 * S[2] + 2, 2 * S[3] + 2};  // REFLECT_101 stays inside this interval
                for (int k = 0; k < 4; k++) S[k] = N[l][k];
                uni(S, F);
                clampl(l, S);
                for (int k = 0; k < 4; k++) S_out[l][k] = S[k];
            }
        };
        int SA[kMaxLevels][4], SB[kMaxLevels][4];
        chain(x0, two ? e0 - 1 : x1, SA);
        if (two) chain(e1 + 1, x1, SB);
        for (int l = 0; l <= top; l++) {
            for (int k = 0; k < 4; k++) c->live[i][l][k] = SA[l][k];
            if (two) {
                uni(c->live[i][l], SB[l]);
                c->gap[i][l][0] = SA[l][2] + 1;   // dead columns of level l (inclusive); empty when the pieces meet
                c->gap[i][l][1] = SB[l][0] - 1;
            }
        }
        for (int l = top + 1; l < c->levels; l++) full(l, c->live[i][l]);
    }
    for (int i = 0; i < P.n; i++)
        for (int l = 0; l < c->levels; l++) {
            for (int k = 0; k < 4; k++) c->pyr.cam[i].live[l][k] = c->live[i][l][k];
            for (int k = 0; k < 2; k++) c->pyr.cam[i].gap[l][k] = c->gap[i][l][k];
        }
    live_source_rects(c);
}


// Which bytes of camera i's frame does K1 read?  The table is static, so every 64 x 16 workgroup's source box is (h_box), and the
// masks say which workgroups run (make_warp_cam: the live blocks of level 0 minus the dead middle of a +-pi straddler).  The
// union of the live boxes, widened to the bytes the copies and taps really touch: an LDS box copies `chunks` 16-byte pieces per
// row from the 16-byte boundary at or below its first byte; a patch on global taps fetches the aligned 12 bytes around each tap
// (3 bytes before it at most) in rows ys and ys + 1.  The staging buffers are 64-byte aligned with 64-byte pitches, so byte
// columns are frame columns.  No table (projecting kernel): the whole frame.
static void live_source_rects(pano_ctx* c) {
    const Plan& P = c->plan;
    const int row_bytes = c->frame_w * 3;
    for (int i = 0; i < P.n; i++) {
        pano_ctx::SrcRect& r = c->src_rect[i];
        r = {0, 0, (int)align_up((size_t)row_bytes, 64), c->frame_h};
        if (!c->use_lut || c->h_box[i].empty()) continue;
        const WarpCam w = make_warp_cam(c, i, nullptr, (size_t)row_bytes, false);
        const int nbx = (P.tile[i].rect.w + 63) / 64, nby = (P.tile[i].rect.h + 15) / 16;
        const unsigned lg = (unsigned)w.live_by0_gap;
        const int by0 = (int)(lg & 0xfffu), g0 = (int)((lg >> 12) & 0x3ffu), glen = (int)(lg >> 22);
        int x0 = INT_MAX, x1 = -1, y0 = INT_MAX, y1 = -1;
        for (int by = by0; by <= std::min(w.live_by1, nby - 1); by++)
            for (int bx = w.live_bx0; bx <= std::min(w.live_bx1, nbx - 1); bx++) {
                if (bx >= g0 && bx < g0 + glen) continue;
                const int4 b = c->h_box[i][(size_t)by * nbx + bx];
                int bx0, bx1, by1;
                if (b.z >> 8) {
                    bx0 = (3 * b.x) & ~15;
                    bx1 = bx0 + (b.z & 255) * 16 + 16;  // + a fetch of the general kernel's global taps on the box's last pixel
                    by1 = b.y + (b.z >> 8);
                } else {
                    bx0 = std::max(3 * b.x - 3, 0);
                    bx1 = 3 * (b.x + (b.w >> 16)) + 12;
                    by1 = std::min(b.y + (b.w & 0xffff) + 2, c->frame_h);
                }
                x0 = std::min(x0, bx0); x1 = std::max(x1, bx1);
                y0 = std::min(y0, b.y); y1 = std::max(y1, by1);
            }
        if (x1 < 0) { r = {0, 0, 0, 0}; continue; }  // nothing live: nothing to upload
        // a fetch near the end of a row may run into the first bytes of the next one: then whole rows
        y0 = std::max(y0, 0); y1 = std::min(y1, c->frame_h);
        x0 &= ~63;
        x1 = (int)align_up((size_t)x1, 64);
        if (x1 > row_bytes) { x0 = 0; x1 = (int)align_up((size_t)row_bytes, 64); }
        r = {x0, y0, x1 - x0, y1 - y0};
    }
}

// cv::resize INTER_LINEAR (f32) coefficients of dst index d for ssize -> dsize
inline void linear_coef(int d, int ssize, int dsize, int& s0, int& s1, float& a0, float& a1) {
    double scale = (double)ssize / dsize;
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0; s = 0; }
    if (s >= ssize - 1) { f = 0; s = ssize - 1; }
    s0 = s;
    s1 = std::min(s + 1, ssize - 1);
    a0 = 1.f - f;
    a1 = f;
}

pano_status upload_gain_tables(pano_ctx* c, int i, const float* h_gain) {
    const Rect& roi = c->plan.roi[i];
    const FeedTile& t = c->plan.tile[i];
    int gw = c->gain_w[i], gh = c->gain_h[i];
    auto build = [&](int n, int off, int len, int ssize, std::vector<int2>& idx, std::vector<float2>& wv, bool rows) {
        idx.resize(n);
        wv.resize(n);
        for (int k = 0; k < n; k++) {
            int d = reflect(k - off, len);
            int s0, s1;
            float a0, a1;
            if (rows) {
                // vertical: sy may be clamped on both taps independently (resizeGeneric_ row clipping)
                double scale = (double)ssize / len;
                float f = (float)((d + 0.5) * scale - 0.5);
                int s = (int)floorf(f);
                f -= s;
                s0 = std::min(std::max(s, 0), ssize - 1);
                s1 = std::min(std::max(s + 1, 0), ssize - 1);
                a0 = 1.f - f;
                a1 = f;
            } else {
                linear_coef(d, ssize, len, s0, s1, a0, a1);
            }
            idx[k] = make_int2(s0, s1);
            wv[k] = make_float2(a0, a1);
        }
    };
    // HResizeLinear<float> of every map row at every column: S[sx] * (1 - fx) + S[sx1] * fx, the f32 expression of
    // resizeGeneric_'s horizontal pass, once per gain map instead of once per pixel and frame
    auto hresize = [&](const std::vector<int2>& ix, const std::vector<float2>& wx, int n, std::vector<float>& out, int& pitch) {
        pitch = (int)align_up((size_t)n, 4);
        out.assign((size_t)gh * pitch, 0.f);
        for (int gy = 0; gy < gh; gy++) {
            const float* S = h_gain + (size_t)gy * gw;
            float* o = out.data() + (size_t)gy * pitch;
            for (int x = 0; x < pitch; x++) {
                const int k = std::min(x, n - 1);
                o[x] = S[ix[k].x] * wx[k].x + S[ix[k].y] * wx[k].y;
            }
        }
    };
    std::vector<int2> ix;
    std::vector<float2> wx;
    std::vector<float> hr;
    pano_status s;
    build(t.rect.w, t.left, roi.w, gw, ix, wx, false);
    hresize(ix, wx, t.rect.w, hr, c->ghrow_pitch[i]);
    if ((s = upload(c, &c->ghrow[i], hr.data(), hr.size() * sizeof(float)))) return s;
    build(t.rect.h, t.top, roi.h, gh, ix, wx, true);
    if ((s = upload(c, &c->grow[i], ix.data(), ix.size() * sizeof(int2)))) return s;
    if ((s = upload(c, &c->groww[i], wx.data(), wx.size() * sizeof(float2)))) return s;
    {   // per 16-row block row of K1: do its rows read at most kGainRows consecutive rows of ghrow?  (WarpCam::grow_base)
        std::vector<int> base((t.rect.h + 15) / 16);
        for (size_t b = 0; b < base.size(); b++) {
            int lo = INT_MAX, hi = -1;
            for (int y = (int)b * 16; y < std::min((int)b * 16 + 16, t.rect.h); y++) {
                lo = std::min(lo, std::min(ix[y].x, ix[y].y));
                hi = std::max(hi, std::max(ix[y].x, ix[y].y));
            }
            base[b] = hi - lo < kGainRows ? lo : -1;
        }
        if ((s = upload(c, &c->grow_base[i], base.data(), base.size() * sizeof(int)))) return s;
        std::vector<int4> both(ix.size());
        for (size_t y = 0; y < ix.size(); y++) {
            int wb[2];
            std::memcpy(wb, &wx[y], sizeof(wb));
            both[y] = make_int4(ix[y].x, ix[y].y, wb[0], wb[1]);
        }
        if ((s = upload(c, &c->grow4[i], both.data(), both.size() * sizeof(int4)))) return s;
    }
    build(roi.w, 0, roi.w, gw, ix, wx, false);
    hresize(ix, wx, roi.w, hr, c->ghrow_roi_pitch[i]);
    if ((s = upload(c, &c->ghrow_roi[i], hr.data(), hr.size() * sizeof(float)))) return s;
    build(roi.h, 0, roi.h, gh, ix, wx, true);
    if ((s = upload(c, &c->grow_roi[i], ix.data(), ix.size() * sizeof(int2)))) return s;
    if ((s = upload(c, &c->groww_roi[i], wx.data(), wx.size() * sizeof(float2)))) return s;
    return PANO_OK;
}

// weight pyramids + canvas weight sums; runs when masks changed
// dispatch order of the level-0 blend tiles: inside every XCD band the tiles with a general-path wave first (static: it follows
// the owner map and the cut).  PANO_L0_ORDER=0 keeps the plain band order
pano_status build_tile_order(pano_ctx* c, hipStream_t s) {
    c->order_dirty = false;
    for (int l = 0; l < kOrderLevels; l++) {
        c->cv.order[l] = nullptr;
        c->cv.order_per[l] = 0;
    }
    if (!c->l0_order || c->plan.bands < 0) return PANO_OK;
    for (int l = 0; l < kOrderLevels && l <= c->plan.bands; l++) {
        const CanvasParams& cv = c->cv;
        if (!cv.fast[l] || (cv.small_base > 0 && l >= cv.small_base)) break;  // the vector kernel's levels
        const int w = l == 0 ? cv.cut_x + cv.cut_w - (cv.cut_x & ~3) : (cv.w0 >> l);
        const int h = l == 0 ? cv.cut_y + cv.cut_h - (cv.cut_y & ~1) : (cv.h0 >> l);
        const int gx = (w + 127) / 128, gy = (h + 15) / 16;
        const size_t T = (size_t)gx * gy;
        if (T == 0 || gx >= 255 || gy >= 255) continue;  // an entry holds bx and by in a byte each
        const size_t per = (T + 7) / 8;
        uint16_t* d_flags = nullptr;
        HIP_TRY(c, hipMalloc((void**)&d_flags, T * sizeof(uint16_t)));
        launch_tile_mixed(cv, l, gx, gy, d_flags, s);
        std::vector<uint16_t> flags(T);  // four owner nibbles per tile, one per wave (0xF: that wave takes the general path)
        hipError_t fe = hipMemcpyAsync(flags.data(), d_flags, T * sizeof(uint16_t), hipMemcpyDeviceToHost, s);
        if (fe == hipSuccess) fe = hipStreamSynchronize(s);
        (void)hipFree(d_flags);
        HIP_TRY(c, fe);
        auto mixed = [](uint16_t f) { return (f & 0xf) == 0xf || ((f >> 4) & 0xf) == 0xf || ((f >> 8) & 0xf) == 0xf || (f >> 12) == 0xf; };
        std::vector<uint32_t> order(8 * per, 0xffffu);
        for (size_t k = 0; k < 8; k++) {
            const size_t lo = k * per, hi = std::min(T, lo + per);
            size_t o = lo;
            for (int pass = 1; pass >= 0; pass--)
                for (size_t t = lo; t < hi; t++)
                    if ((int)mixed(flags[t]) == pass) order[o++] = (uint32_t)(t % gx) | (uint32_t)(t / gx) << 8 | (uint32_t)flags[t] << 16;
        }
        if (c->order_cap[l] < order.size()) {
            dfree(c->order[l]);
            c->order_cap[l] = 0;
            HIP_TRY(c, hipMalloc((void**)&c->order[l], order.size() * sizeof(uint32_t)));
            c->order_cap[l] = order.size();
        }
        HIP_TRY(c, hipMemcpy(c->order[l], order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->cv.order[l] = c->order[l];
        c->cv.order_per[l] = (int)per;
    }
    return PANO_OK;
}

pano_status ensure_weights(pano_ctx* c, hipStream_t s) {
    if (!c->weights_dirty) {
        if (c->order_dirty) {  // the cut changed: frames in flight still walk the old table
            HIP_TRY(c, hipDeviceSynchronize());
            return build_tile_order(c, s);
        }
        return PANO_OK;
    }
    const Plan& P = c->plan;
    for (int i = 0; i < P.n; i++)
        if (!c->mask_set[i]) return fail(c, PANO_ESTATE, "blend masks not set (pano_set_mask / pano_build_masks_voronoi)");
    // the weights, bordered masks and owner maps are shared by every frame slot: frames still in flight on other streams
    // (composed under the old masks) have to finish before they are rewritten
    HIP_TRY(c, hipDeviceSynchronize());
    for (int i = 0; i < P.n; i++) {
        const FeedTile& t = P.tile[i];
        launch_mask_to_weight(c->mask[i], P.roi[i].w, P.roi[i].h, P.roi[i].w, t.left, t.top, c->wgt[i][0], c->wpitch[i][0],
                              c->mask0[i], c->lvl_pitch[i][0], t.rect.w, t.rect.h, s);
        for (int l = 0; l < P.bands; l++)
            launch_pyr_down_f32(c->wgt[i][l], t.rect.w >> l, t.rect.h >> l, c->wpitch[i][l], c->wgt[i][l + 1],
                                c->wpitch[i][l + 1], s);
    }
    // the summed canvas weights are not read by the blend (it re-adds the same f32 terms in the same
    // order); they are kept for stage inspection
    for (int l = 0; l <= P.bands; l++) launch_sum_weights(c->pyr, l, c->wsum[l], P.canvas.w >> l, P.canvas.h >> l, s);
    // owner maps of the vector levels
    for (int l = 0; l <= P.bands; l++)
        if (c->cv.fast[l]) launch_build_owner(c->pyr, c->cv, l, c->owner[l], s);
    {
        pano_status os = build_tile_order(c, s);
        if (os != PANO_OK) return os;
    }
    HIP_TRY(c, hipGetLastError());
    // one-time: later frames may run on other streams (frame slots) and must find the weights complete
    HIP_TRY(c, hipStreamSynchronize(s));
    {
        std::vector<std::vector<uint8_t>> hm(P.n);
        for (int i = 0; i < P.n; i++) {
            hm[i].resize((size_t)P.roi[i].w * P.roi[i].h);
            HIP_TRY(c, hipMemcpy(hm[i].data(), c->mask[i], hm[i].size(), hipMemcpyDeviceToHost));
        }
        live_rects(c, hm);
    }
    c->weights_dirty = false;
    return PANO_OK;
}

// fold the oldest pending event quad into the accumulators (waits for it if the GPU is still behind)
pano_status harvest_oldest(pano_ctx* c) {
    if (c->ev_count == 0) return PANO_OK;
    int idx = (c->ev_head - c->ev_count + pano_ctx::kEvRing) % pano_ctx::kEvRing;
    pano_ctx::EvSlot& sl = c->ring[idx];
    for (int k = 3; k >= 0; k--)
        if (sl.recorded & (1u << k)) {
            HIP_TRY(c, hipEventSynchronize(sl.e[k]));
            break;
        }
    if (sl.recorded & (1u << 5)) HIP_TRY(c, hipEventSynchronize(sl.e[5]));
    for (int k = 0; k < PANO_NUM_STAGES; k++) {
        const int a = k == PANO_STAGE_BLEND0 ? 4 : k, b = k == PANO_STAGE_BLEND0 ? 5 : k + 1;
        if ((sl.recorded & (1u << a)) && (sl.recorded & (1u << b))) {
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, sl.e[a], sl.e[b]));
            c->acc_ms[k] += ms;
            c->acc_n[k]++;
            c->last_ms[k] = ms;
        }
    }
    sl.recorded = 0;
    c->ev_count--;
    if (c->ev_cur == idx) c->ev_cur = -1;
    return PANO_OK;
}
pano_status begin_slot(pano_ctx* c) {
    if (c->ev_count == pano_ctx::kEvRing) {
        pano_status s = harvest_oldest(c);
        if (s != PANO_OK) return s;
    }
    c->ev_cur = c->ev_head;
    c->ev_head = (c->ev_head + 1) % pano_ctx::kEvRing;
    c->ev_count++;
    c->ring[c->ev_cur].recorded = 0;
    return PANO_OK;
}
pano_status record(pano_ctx* c, int k, hipStream_t s) {
    HIP_TRY(c, hipEventRecord(c->ring[c->ev_cur].e[k], s));
    c->ring[c->ev_cur].recorded |= 1u << k;
    return PANO_OK;
}

pano_status check_compute(pano_ctx* c) {
    if (!c) return PANO_EINVAL;
    if (c->device < 0) return fail(c, PANO_ENODEVICE, "plan-only context (config.device < 0): no CPU fallback exists");
    if (!c->prepared) return fail(c, PANO_ESTATE, "pano_prepare has not run");
    HIP_TRY(c, hipSetDevice(c->device));
    return PANO_OK;
}

bool parse_floats(const std::string& s, std::vector<float>& out) {
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        size_t b = tok.find_first_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        char* end = nullptr;
        float v = strtof(tok.c_str() + b, &end);
        if (end == tok.c_str() + b) return false;
        out.push_back(v);
    }
    return true;
}

// INTER_LINEAR_EXACT coefficient tables (resize.cpp interpolationLinear<uchar>::getCoeffs), IEEE double.
// inv_scale is what cv::resize hands down: dsize/ssize when the caller gave a dsize (pass 0), but the caller's fx when
// dsize was empty - resize(src, dst, Size(), fx, fy) samples on a 1/fx grid although dsize = cvRound(ssize*fx)
void linearExactCoeffs(int ssize, int dsize, double inv_scale, std::vector<int>& ofs, std::vector<int>& c1, int& mn, int& mx) {
    if (!(inv_scale > 0)) inv_scale = (double)dsize / ssize;
    double scale = 1.0 / inv_scale;
    ofs.assign(dsize, 0);
    c1.assign(dsize, 0);
    mn = 0;
    mx = dsize;
    for (int v = 0; v < dsize; v++) {
        double fval = scale * ((double)v + 0.5) - 0.5;
        int ival = (int)std::floor(fval);
        if (ival >= 0 && ssize > 1) {
            if (ival < ssize - 1) {
                ofs[v] = ival;
                c1[v] = (int)std::lrint((fval - (double)ival) * 256.0);
            } else {
                ofs[v] = ssize - 1;
                mx = std::min(mx, v);
            }
        } else {
            mn = std::max(mn, v + 1);
        }
    }
    if (mx < mn) mx = mn;
}

// device allocations that live for one init-time call.  With a pool (pano_ctx::refresh_pool: the mask refresh beside the frame
// loop, whose buffer sizes repeat from one refresh to the next) buffers come from it and go back to it instead of through
// hipMalloc / hipFree - every hipFree waits for the device, and a few dozen hipMallocs are milliseconds of a 16.7 ms tick
struct Scratch {
    typedef std::vector<std::pair<size_t, void*>> Pool;
    std::vector<std::pair<size_t, void*>> p;
    Pool* pool = nullptr;
    ~Scratch() { release(); }
    void release() {
        for (auto& q : p) {
            if (pool) pool->push_back(q);
            else (void)hipFree(q.second);
        }
        p.clear();
    }
    template <typename T>
    bool alloc(T** d, size_t bytes) {
        *d = nullptr;
        if (!bytes) bytes = 16;
        if (pool)
            for (size_t k = 0; k < pool->size(); k++)
                if ((*pool)[k].first == bytes) {
                    *d = (T*)(*pool)[k].second;
                    p.push_back((*pool)[k]);
                    pool->erase(pool->begin() + (long)k);
                    return true;
                }
        if (hipMalloc((void**)d, bytes) != hipSuccess) return false;
        p.push_back({bytes, (void*)*d});
        return true;
    }
    template <typename T>
    bool put(T** d, const void* h, size_t bytes) {
        return alloc(d, bytes) && (bytes == 0 || hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess);
    }
};

// cv::solve(A, b, x, DECOMP_LU) for CV_64F as OpenCV's own LU does it (core/src/matrix_decomp.cpp LUImpl, no LAPACK):
// partial pivoting on |a|, eps = 100 * DBL_EPSILON, elimination with alpha = a_ji * (-1 / a_ii), back substitution.
// The operation order is the result (f64 does not reassociate), so it is spelled out rather than delegated
bool solveLU(std::vector<double>& A, int m, std::vector<double>& x) {
    const double eps = 2.220446049250313e-16 * 100;
    auto at = [&](int r, int col) -> double& { return A[(size_t)r * m + col]; };
    for (int i = 0; i < m; i++) {
        int piv = i;
        for (int j = i + 1; j < m; j++)
            if (std::fabs(at(j, i)) > std::fabs(at(piv, i))) piv = j;
        if (std::fabs(at(piv, i)) < eps) return false;
        if (piv != i) {
            for (int j = i; j < m; j++) std::swap(at(i, j), at(piv, j));
            std::swap(x[i], x[piv]);
        }
        const double d = -1 / at(i, i);
        for (int j = i + 1; j < m; j++) {
            const double alpha = at(j, i) * d;
            for (int k = i + 1; k < m; k++) at(j, k) += alpha * at(i, k);
            x[j] += alpha * x[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double acc = x[i];
        for (int k = i + 1; k < m; k++) acc -= at(i, k) * x[k];
        x[i] = acc / at(i, i);
    }
    return true;
}

// cv::sepFilter2D(map, map, CV_32F, [.25 .5 .25], [.25 .5 .25]), BORDER_REFLECT_101: the symmetric small-kernel row and
// column filters both evaluate  centre * k0 + (left + right) * k1  in f32 (imgproc/src/filter.cpp)
void smooth121(std::vector<float>& m, int w, int h) {
    std::vector<float> t((size_t)w * h);
    auto r101 = [](int p, int len) { return len == 1 ? 0 : (p < 0 ? -p : (p >= len ? 2 * len - 2 - p : p)); };
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            t[(size_t)y * w + x] = m[(size_t)y * w + x] * 0.5f + (m[(size_t)y * w + r101(x - 1, w)] + m[(size_t)y * w + r101(x + 1, w)]) * 0.25f;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            m[(size_t)y * w + x] = (t[(size_t)r101(y - 1, h) * w + x] + t[(size_t)r101(y + 1, h) * w + x]) * 0.25f + t[(size_t)y * w + x] * 0.5f;
}

// What initSeam / updateMask put in front of the seam finder and the compensator (ocvstitcher.hpp:981-1017, :1228-1242):
// per camera the seam-scale ROI, the INTER_NEAREST / BORDER_CONSTANT warp of an all-255 mask and - when frames are given -
// resize(frame, seam_work_aspect, INTER_LINEAR_EXACT) warped INTER_LINEAR / BORDER_REFLECT.  Device buffers live in `tmp`
struct SeamWarps {
    std::vector<Rect> roi;
    std::vector<uint8_t*> img, mask;  // dense 8UC3 / 8U, roi[i].w x roi[i].h (img: nullptr without frames)
};
pano_status seam_scale_warps(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides, Scratch& tmp, hipStream_t s,
                             SeamWarps& out) {
    const Plan& P = c->plan;
    const int n = P.n, sw = P.src_w, sh = P.src_h;
    auto oom = [&]() { return fail(c, PANO_EHIP, "hipMalloc / hipMemcpy (seam-scale warps)"); };
    // seam scale (ocvstitcher.hpp:298, :988-1017)
    const double swa = std::min(1.0, std::sqrt(1e5 / ((double)sh * sw)));
    const int ssw = (int)std::lrint(sw * swa), ssh = (int)std::lrint(sh * swa);
    const float seam_scale = static_cast<float>(c->scale * swa), swa_f = (float)swa;
    // resize(imgs[i], seam_work_aspect, INTER_LINEAR_EXACT) (:988): one coefficient set for all cameras
    std::vector<int> xo, xc, yo, yc;
    int mnx = 0, mxx = 0, mny = 0, mxy = 0;
    int *dxo = nullptr, *dxc = nullptr, *dyo = nullptr, *dyc = nullptr;
    const bool shrink = h_frames && (ssw != sw || ssh != sh);
    if (shrink) {
        linearExactCoeffs(sw, ssw, swa, xo, xc, mnx, mxx);  // Size(), seam_work_aspect, seam_work_aspect
        linearExactCoeffs(sh, ssh, swa, yo, yc, mny, mxy);
        if (!tmp.put(&dxo, xo.data(), xo.size() * sizeof(int)) || !tmp.put(&dxc, xc.data(), xc.size() * sizeof(int)) ||
            !tmp.put(&dyo, yo.data(), yo.size() * sizeof(int)) || !tmp.put(&dyc, yc.data(), yc.size() * sizeof(int)))
            return oom();
    }
    out.roi.assign(n, Rect{});
    out.img.assign(n, nullptr);
    out.mask.assign(n, nullptr);
    std::vector<float> a, b;
    for (int i = 0; i < n; i++) {
        float K[9];
        std::memcpy(K, c->K[i], sizeof(K));
        K[0] *= swa_f; K[2] *= swa_f; K[4] *= swa_f; K[5] *= swa_f;
        Projector pj;
        pj.set(c->cfg.projector, seam_scale, K, c->R[i]);
        const Rect r = out.roi[i] = warpRoi(pj, ssw, ssh);
        trigTables(pj, r, 0, 0, r.w, r.h, a, b);
        float2 *dA = nullptr, *dB = nullptr;
        if (!tmp.put(&dA, a.data(), a.size() * sizeof(float)) || !tmp.put(&dB, b.data(), b.size() * sizeof(float)) ||
            !tmp.alloc(&out.mask[i], (size_t)r.w * r.h))
            return oom();
        // seamfinder_warper->warp(.., INTER_LINEAR, BORDER_REFLECT) and (.., INTER_NEAREST, BORDER_CONSTANT) (:1011-1014)
        WarpCam w{};
        w.src_w = ssw; w.src_h = ssh;
        w.out_w = ssw; w.out_h = ssh;
        std::memcpy(w.m, pj.k_rinv, sizeof(w.m));
        w.colA = dA; w.rowB = dB; w.tw = r.w; w.th = r.h;
        if (h_frames) {
            uint8_t *full = nullptr, *small = nullptr;
            if (!tmp.alloc(&full, (size_t)sw * sh * 3 + 16) || !tmp.alloc(&out.img[i], (size_t)r.w * r.h * 3)) return oom();
            HIP_TRY(c, hipMemcpy2DAsync(full, (size_t)sw * 3, h_frames[i], strides[i], (size_t)sw * 3, sh, hipMemcpyHostToDevice, s));
            small = full;
            if (shrink) {
                if (!tmp.alloc(&small, (size_t)ssw * ssh * 3 + 16)) return oom();
                launch_resize_linear_exact(full, sw, sh, 3, small, ssw, ssh, dxo, dxc, dyo, dyc, mnx, mxx, mny, mxy, s);
            }
            w.src = small; w.src_stride = ssw * 3;
            w.dst = out.img[i]; w.dst_pitch = r.w * 3;
            launch_warp_image(w, s);
        }
        launch_warp_mask(w, out.mask[i], r.w, s);
    }
    HIP_TRY(c, hipGetLastError());
    return PANO_OK;
}

// ... and behind the seam finder (ocvstitcher.hpp:1085, :1097-1101, :1246-1257): the full-scale NEAREST mask, the seam mask
// dilated 3 x 3 and resized INTER_LINEAR_EXACT to the ROI, their AND = m_blenderMask[i]
pano_status finish_seam_masks(pano_ctx* c, const SeamWarps& sm, Scratch& tmp, hipStream_t s) {
    const Plan& P = c->plan;
    auto oom = [&]() { return fail(c, PANO_EHIP, "hipMalloc / hipMemcpy (blend masks)"); };
    for (int i = 0; i < P.n; i++) {
        const Rect& r = P.roi[i];
        const Rect& q = sm.roi[i];
        uint8_t *full = nullptr, *dil = nullptr, *seam = nullptr;
        if (!tmp.alloc(&full, (size_t)r.w * r.h) || !tmp.alloc(&dil, (size_t)q.w * q.h) || !tmp.alloc(&seam, (size_t)r.w * r.h)) return oom();
        WarpCam w = make_warp_cam(c, i, nullptr, 0, true);
        launch_warp_mask(w, full, r.w, s);
        launch_dilate3x3(sm.mask[i], dil, q.w, q.h, s);
        std::vector<int> xo, xc, yo, yc;
        int mnx, mxx, mny, mxy;
        linearExactCoeffs(q.w, r.w, 0, xo, xc, mnx, mxx);  // explicit dsize (:1099, :1256)
        linearExactCoeffs(q.h, r.h, 0, yo, yc, mny, mxy);
        int *dxo = nullptr, *dxc = nullptr, *dyo = nullptr, *dyc = nullptr;
        if (!tmp.put(&dxo, xo.data(), xo.size() * sizeof(int)) || !tmp.put(&dxc, xc.data(), xc.size() * sizeof(int)) ||
            !tmp.put(&dyo, yo.data(), yo.size() * sizeof(int)) || !tmp.put(&dyc, yc.data(), yc.size() * sizeof(int)))
            return oom();
        launch_resize_linear_exact(dil, q.w, q.h, 1, seam, r.w, r.h, dxo, dxc, dyo, dyc, mnx, mxx, mny, mxy, s);
        launch_and(seam, full, c->mask[i], (size_t)r.w * r.h, s);
        c->mask_set[i] = true;
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    c->weights_dirty = true;
    live_rects(c, {});  // until the weights are rebuilt, produce every pixel
    drop_graphs(c);
    return PANO_OK;
}

// run an entry point's body; an exception becomes a status instead of crossing the C boundary
template <typename F>
pano_status guarded(pano_ctx* c, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return c ? fail(c, PANO_ENOMEM, "out of host memory") : PANO_ENOMEM;
    } catch (const std::exception& e) {
        return c ? fail(c, PANO_ERR, e.what()) : PANO_ERR;
    } catch (...) {
        return c ? fail(c, PANO_ERR, "unknown exception") : PANO_ERR;
    }
}

}  // namespace

extern "C" {

const char* pano_version(void) { return "pano-hip 0.1 (gfx950)"; }

const char* pano_last_error(const pano_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

static pano_status create_impl(const pano_config* cfg, pano_ctx** out) {
    if (!cfg || !out) return PANO_EINVAL;
    *out = nullptr;
    if (cfg->num_images < 1 || cfg->num_images > PANO_MAX_CAMS || cfg->width < 2 || cfg->height < 2) return PANO_EINVAL;
    if (cfg->projector != PANO_SPHERICAL && cfg->projector != PANO_CYLINDRICAL) return PANO_EINVAL;
    if (cfg->num_bands > PANO_MAX_BANDS || cfg->num_bands < PANO_BANDS_FROM_STRENGTH) return PANO_EINVAL;
    pano_ctx* c = new (std::nothrow) pano_ctx();
    if (!c) return PANO_ENOMEM;
    c->cfg = *cfg;
    c->scale = cfg->warped_image_scale;
    c->device = cfg->device;
    if (c->device >= 0) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || c->device >= ndev) {
            // fail loudly: there is no CPU path
            delete c;
            return PANO_EHIP;
        }
    }
    *out = c;
    return PANO_OK;
}

static void drop_job(pano_ctx* c);
void pano_destroy(pano_ctx* ctx) {
    if (!ctx) return;
    if (ctx->host_trace_n)
        fprintf(stderr, "pano_compose_host x %ld: stage in + queue H2D %.3f ms, queue kernels %.3f, wait H2D + kernels %.3f, copy back %.3f, unstage %.3f\n",
                ctx->host_trace_n, ctx->host_trace[0] / ctx->host_trace_n, ctx->host_trace[1] / ctx->host_trace_n,
                ctx->host_trace[2] / ctx->host_trace_n, ctx->host_trace[3] / ctx->host_trace_n, ctx->host_trace[4] / ctx->host_trace_n);
    if (ctx->device >= 0 && ctx->prepared) {
        (void)hipSetDevice(ctx->device);
        drop_job(ctx);
        (void)hipDeviceSynchronize();
        free_device(ctx);
    }
    delete ctx;
}

// What the reference only finds out as a CV_Assert or a garbage panorama: every value finite, K an intrinsic matrix, R a
// rotation (orthonormal to 1e-3 - the logs and YAMLs carry six significant digits - and not a reflection)
static pano_status validate_camera(pano_ctx* c, const float K[9], const float R[9]) {
    for (int k = 0; k < 9; k++)
        if (!std::isfinite(K[k]) || !std::isfinite(R[k])) return fail(c, PANO_EINVAL, "camera parameters must be finite");
    if (!(K[0] > 0.f) || !(K[4] > 0.f)) return fail(c, PANO_EINVAL, "focal length must be positive");
    double dev = 0.0;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double d = 0.0;
            for (int k = 0; k < 3; k++) d += (double)R[3 * k + a] * (double)R[3 * k + b];
            dev = std::max(dev, std::fabs(d - (a == b ? 1.0 : 0.0)));
        }
    if (dev > 1e-3) return fail(c, PANO_EINVAL, "R is not orthonormal (max |R^T R - I| > 1e-3)");
    const double det = (double)R[0] * ((double)R[4] * R[8] - (double)R[5] * R[7]) - (double)R[1] * ((double)R[3] * R[8] - (double)R[5] * R[6]) +
                       (double)R[2] * ((double)R[3] * R[7] - (double)R[4] * R[6]);
    if (!(det > 0.0)) return fail(c, PANO_EINVAL, "R is a reflection (det R < 0)");
    return PANO_OK;
}

pano_status pano_set_camera(pano_ctx* c, int i, const float K[9], const float R[9]) {
    if (!c || !K || !R || i < 0 || i >= c->cfg.num_images) return PANO_EINVAL;
    if (c->prepared) return fail(c, PANO_ESTATE, "cameras are fixed after pano_prepare");
    pano_status s = validate_camera(c, K, R);
    if (s != PANO_OK) return s;
    std::memcpy(c->K[i], K, 9 * sizeof(float));
    std::memcpy(c->R[i], R, 9 * sizeof(float));
    c->have_cam[i] = true;
    return PANO_OK;
}

// rotationMatrixToEulerAngles (ocvstitcher.hpp:229-253): degrees, f32 like the reference's Vec3f
static void euler_degrees(const float R[9], float out[3]) {
    const double r00 = R[0], r10 = R[3], r20 = R[6], r21 = R[7], r22 = R[8], r12 = R[5], r11 = R[4];
    const float sy = (float)std::sqrt(r00 * r00 + r10 * r10);
    float x, y, z;
    if (!(sy < 1e-6)) {
        x = (float)std::atan2(r21, r22);
        y = (float)std::atan2(-r20, (double)sy);
        z = (float)std::atan2(r10, r00);
    } else {
        x = (float)std::atan2(-r12, r11);
        y = (float)std::atan2(-r20, (double)sy);
        z = 0.f;
    }
    const float k = (float)(180.0 / M_PI);
    out[0] = x * k; out[1] = y * k; out[2] = z * k;
}

pano_status pano_verify_cameras(pano_ctx* c, const float* K_est, const float* R_est, float ex_thres, float in_thres, int* worst_camera) {
    if (!c || !K_est || !R_est) return PANO_EINVAL;
    if (worst_camera) *worst_camera = -1;
    const int n = c->cfg.num_images;
    for (int i = 0; i < n; i++)
        if (!c->have_cam[i]) return fail(c, PANO_ESTATE, "pano_verify_cameras: the context holds no camera to compare with");
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 9; k++)
            if (!std::isfinite(K_est[9 * i + k]) || !std::isfinite(R_est[9 * i + k])) {
                if (worst_camera) *worst_camera = i;
                return fail(c, PANO_ERR, "estimated camera parameters are not finite");
            }
        float a[3], b[3];
        euler_degrees(c->R[i], a);
        euler_degrees(R_est + 9 * i, b);
        double d2 = 0.0;
        for (int k = 0; k < 3; k++) d2 += ((double)a[k] - b[k]) * ((double)a[k] - b[k]);
        if (std::sqrt(d2) > (double)ex_thres) {
            if (worst_camera) *worst_camera = i;
            return fail(c, PANO_ERR, "extrinsic difference above the threshold: keep the default parameters");
        }
        const double dfx = (double)c->K[i][0] - K_est[9 * i], dfy = (double)c->K[i][4] - K_est[9 * i + 4];
        if (std::sqrt(dfx * dfx + dfy * dfy) > (double)in_thres) {
            if (worst_camera) *worst_camera = i;
            return fail(c, PANO_ERR, "intrinsic difference above the threshold: keep the default parameters");
        }
    }
    return PANO_OK;
}

static pano_status set_cameras_from_list_impl(pano_ctx* c, const char* list) {
    if (!c || !list) return PANO_EINVAL;
    std::vector<float> v;
    if (!parse_floats(list, v)) return fail(c, PANO_ERR, "camera list: not a number");
    const int n = c->cfg.num_images;
    if ((int)v.size() != 18 * n + 1) return fail(c, PANO_ERR, "camera list: expected 18*num_images+1 values");
    if (c->prepared) return fail(c, PANO_ESTATE, "cameras are fixed after pano_prepare");
    for (int i = 0; i < n; i++) {  // all or nothing
        pano_status s = validate_camera(c, &v[18 * i], &v[18 * i + 9]);
        if (s != PANO_OK) return s;
    }
    if (!std::isfinite(v.back()) || !(v.back() > 0.f)) return fail(c, PANO_EINVAL, "warped_image_scale must be positive");
    for (int i = 0; i < n; i++) {
        pano_status s = pano_set_camera(c, i, &v[18 * i], &v[18 * i + 9]);
        if (s != PANO_OK) return s;
    }
    c->scale = v.back();
    c->cfg.warped_image_scale = c->scale;
    return PANO_OK;
}

static pano_status load_camera_file_impl(pano_ctx* c, const char* path) {
    if (!c || !path) return PANO_EINVAL;
    std::ifstream fin(path);
    if (!fin.is_open()) return fail(c, PANO_ERR, "cannot open camera parameter file");
    std::vector<std::string> lines;
    std::string l;
    while (std::getline(fin, l)) {
        while (!l.empty() && (l.back() == '\r' || l.back() == ' ')) l.pop_back();
        if (!l.empty()) lines.push_back(l);
    }
    int last = -1;
    for (int i = 0; i < (int)lines.size(); i++)
        if (lines[i].find(':') != std::string::npos) last = i;
    if (last < 0) return fail(c, PANO_ERR, "no record in camera parameter file");
    if (c->prepared) return fail(c, PANO_ESTATE, "cameras are fixed after pano_prepare");
    const int n = c->cfg.num_images;
    std::vector<std::vector<float>> rec;
    for (int i = last + 1; i < (int)lines.size(); i++) {
        std::vector<float> v;
        if (!parse_floats(lines[i], v)) return fail(c, PANO_ERR, "camera parameter file: not a number");
        rec.push_back(v);
    }
    if (rec.empty()) return fail(c, PANO_ERR, "camera parameter file: truncated record");
    if (rec[0].size() == 18) {  // format written by saveCameraParams (ocvstitcher.hpp:522-562)
        if ((int)rec.size() < n + 1 || rec[n].size() != 1) return fail(c, PANO_ERR, "camera parameter file: record shape");
        for (int i = 0; i < n; i++) {  // all or nothing: shape and plausibility of every camera first
            if (rec[i].size() != 18) return fail(c, PANO_ERR, "camera parameter file: record shape");
            pano_status s = validate_camera(c, &rec[i][0], &rec[i][9]);
            if (s != PANO_OK) return s;
        }
        for (int i = 0; i < n; i++) {
            pano_status s = pano_set_camera(c, i, &rec[i][0], &rec[i][9]);
            if (s != PANO_OK) return s;
        }
        c->scale = rec[n][0];
    } else if (rec[0].size() == 9) {  // older shared-K format of 2222/cameraparaout_*.txt
        if ((int)rec.size() < n + 2 || rec[n + 1].size() != 1) return fail(c, PANO_ERR, "camera parameter file: record shape");
        for (int i = 0; i < n; i++) {
            if (rec[i + 1].size() != 9) return fail(c, PANO_ERR, "camera parameter file: record shape");
            pano_status s = validate_camera(c, &rec[0][0], &rec[i + 1][0]);
            if (s != PANO_OK) return s;
        }
        for (int i = 0; i < n; i++) {
            pano_status s = pano_set_camera(c, i, &rec[0][0], &rec[i + 1][0]);
            if (s != PANO_OK) return s;
        }
        c->scale = rec[n + 1][0];
    } else {
        return fail(c, PANO_ERR, "camera parameter file: record shape");
    }
    c->cfg.warped_image_scale = c->scale;
    return PANO_OK;
}

static pano_status set_undistort_impl(pano_ctx* c, int cam, const pano_undistort* u) {
    if (!c || !u || cam < 0 || cam >= c->cfg.num_images) return PANO_EINVAL;
    if (c->prepared) return fail(c, PANO_ESTATE, "the front end is fixed after pano_prepare");
    if (u->raw_w < 2 || u->raw_h < 2 || u->undist_w < 2 || u->undist_h < 2 || u->rect[2] < 1 || u->rect[3] < 1 ||
        u->rect[0] < 0 || u->rect[1] < 0 || u->rect[0] + u->rect[2] > u->undist_w || u->rect[1] + u->rect[3] > u->undist_h ||
        !(u->K[0] > 0) || !(u->K[4] > 0))
        return fail(c, PANO_EINVAL, "undistort parameters");
    if (u->raw_w > 8192 || u->raw_h > 8192) return fail(c, PANO_EINVAL, "front end needs raw frames <= 8192 x 8192");
    c->und[cam] = *u;
    optimalNewCameraMatrix(u->K, u->dist, u->undist_w, u->undist_h, c->newK[cam]);
    c->have_fe[cam] = true;
    return PANO_OK;
}

pano_status pano_get_new_camera_matrix(const pano_ctx* c, int cam, double newK[9]) {
    if (!c || !newK || cam < 0 || cam >= c->cfg.num_images || !c->have_fe[cam]) return PANO_EINVAL;
    std::memcpy(newK, c->newK[cam], 9 * sizeof(double));
    return PANO_OK;
}

static pano_status save_camera_file_impl(pano_ctx* c, const char* path) {
    if (!c || !path) return PANO_EINVAL;
    for (int i = 0; i < c->cfg.num_images; i++)
        if (!c->have_cam[i]) return fail(c, PANO_ESTATE, "camera parameters missing");
    FILE* f = fopen(path, "a");
    if (!f) return fail(c, PANO_ERR, "cannot open camera parameter file for append");
    time_t tt = time(nullptr);
    struct tm tmv;
    localtime_r(&tt, &tmv);
    char stamp[64];
    strftime(stamp, sizeof(stamp), "%F-%H-%M-%S:", &tmv);
    fprintf(f, "%s\n", stamp);
    for (int i = 0; i < c->cfg.num_images; i++) {
        for (int k = 0; k < 9; k++) fprintf(f, "%g,", c->K[i][k]);
        for (int k = 0; k < 9; k++) fprintf(f, "%g,", c->R[i][k]);
        fprintf(f, "\n");
    }
    fprintf(f, "%g\n", c->scale);
    fclose(f);
    return PANO_OK;
}

static pano_status prepare_impl(pano_ctx* c) {
    if (!c) return PANO_EINVAL;
    if (c->prepared) return fail(c, PANO_ESTATE, "already prepared");
    const int n = c->cfg.num_images;
    if (!(c->scale > 0.f)) return fail(c, PANO_EINVAL, "warped_image_scale must be positive");
    Plan& P = c->plan;
    P.n = n;
    P.src_w = c->cfg.width;
    P.src_h = c->cfg.height;
    c->frame_w = P.src_w;
    c->frame_h = P.src_h;
    {
        int nfe = 0;
        for (int i = 0; i < n; i++) nfe += c->have_fe[i] ? 1 : 0;
        if (nfe != 0 && nfe != n) return fail(c, PANO_ESTATE, "set the undistort front end for every camera or for none");
        if (nfe) {
            for (int i = 1; i < n; i++)
                if (c->und[i].raw_w != c->und[0].raw_w || c->und[i].raw_h != c->und[0].raw_h)
                    return fail(c, PANO_EINVAL, "all cameras must share one raw frame size");
            c->frame_w = c->und[0].raw_w;
            c->frame_h = c->und[0].raw_h;
        }
    }
    for (int i = 0; i < n; i++) {
        if (!c->have_cam[i]) return fail(c, PANO_ESTATE, "camera parameters missing");
        P.proj[i].set(c->cfg.projector, c->scale, c->K[i], c->R[i]);
        P.roi[i] = warpRoi(P.proj[i], P.src_w, P.src_h);
        // a camera that straddles the +-pi seam of the projection gets the ROI RotationWarper::warpRoi gives it: the whole u
        // range, its two ends live and the span between them dead (skipped, see live_rects).  PANO_WRAP_IS_ERROR=1
        // keeps the refusal for callers that want the reference's 2 x 4 grouping enforced (README.md:27-29)
        if (P.roi[i].w <= 0 || P.roi[i].h <= 0) return fail(c, PANO_EINVAL, "camera ROI is empty");
        if (P.roi[i].w >= (int)(2.0 * M_PI * c->scale) - 1 && getenv("PANO_WRAP_IS_ERROR") && atoi(getenv("PANO_WRAP_IS_ERROR")))
            return fail(c, PANO_EWRAP, "camera ROI wraps the projection seam; split the ring into groups");
    }
    Rect pano_rect = resultRoi(P.roi, n);
    int req = c->cfg.num_bands;
    if (req == PANO_BANDS_FROM_STRENGTH) req = bandsFromStrength(pano_rect.w, pano_rect.h, c->cfg.blend_strength);
    if (req < 0) req = -1;
    if (!makePlan(P, req)) return fail(c, PANO_EINVAL, "too many bands");
    P.cut = Rect{c->cfg.cut[0], c->cfg.cut[1], c->cfg.cut[2], c->cfg.cut[3]};
    if (P.cut.w == 0 || P.cut.h == 0) P.cut = Rect{0, 0, P.pano.w, P.pano.h};
    if (P.cut.x < 0 || P.cut.y < 0 || P.cut.w < 0 || P.cut.h < 0 || P.cut.x + P.cut.w > P.pano.w ||
        P.cut.y + P.cut.h > P.pano.h)
        return fail(c, PANO_EINVAL, "cut rectangle outside the panorama");
    c->levels = P.bands < 0 ? 1 : P.bands + 1;
    c->prepared = true;
    live_rects(c, {});  // no masks yet: every pixel of every level is live
    if (c->device < 0) return PANO_OK;  // plan-only

    HIP_TRY(c, hipSetDevice(c->device));
    // pyramid slots: every level of one camera contiguous, every camera the same slot size
    size_t slot = 0;
    for (int i = 0; i < n; i++) {
        size_t off = 0;
        for (int l = 0; l < c->levels; l++) {
            // planar u8: B, G, R planes, rows padded to 16 bytes, planes to 256 bytes
            int w = P.tile[i].rect.w >> l, h = P.tile[i].rect.h >> l;
            c->lvl_pitch[i][l] = (int)align_up((size_t)w, 16);
            c->lvl_plane[i][l] = (int)align_up((size_t)c->lvl_pitch[i][l] * h, 256);
            c->wpitch[i][l] = (int)align_up((size_t)w, 4);
            c->lvl_off[i][l] = off;
            off += (size_t)c->lvl_plane[i][l] * 3;
        }
        slot = std::max(slot, off);
    }
    c->cv = CanvasParams{};
    c->slot_bytes = align_up(slot, 4096);
    HIP_TRY(c, hipMalloc((void**)&c->pyr_base, c->slot_bytes * n + 256));  // + slack: edge lanes read up to 4 bytes past a row
    HIP_TRY(c, hipMemset(c->pyr_base, 0, c->slot_bytes * n));
    std::vector<float> a, b;
    for (int i = 0; i < n; i++) {
        const FeedTile& t = P.tile[i];
        pano_status s;
        trigTables(P.proj[i], P.roi[i], t.left, t.top, t.rect.w, t.rect.h, a, b);
        while ((a.size() / 2) % 4) {  // K1 reads the column table four entries at a time
            a.push_back(a[a.size() - 2]);
            a.push_back(a[a.size() - 2]);
        }
        if ((s = upload(c, &c->colA[i], a.data(), a.size() * sizeof(float)))) return s;
        if ((s = upload(c, &c->rowB[i], b.data(), b.size() * sizeof(float)))) return s;
        trigTables(P.proj[i], P.roi[i], 0, 0, P.roi[i].w, P.roi[i].h, a, b);
        if ((s = upload(c, &c->colA_roi[i], a.data(), a.size() * sizeof(float)))) return s;
        if ((s = upload(c, &c->rowB_roi[i], b.data(), b.size() * sizeof(float)))) return s;
        HIP_TRY(c, hipMalloc((void**)&c->mask[i], (size_t)P.roi[i].w * P.roi[i].h));
        HIP_TRY(c, hipMalloc((void**)&c->mask0[i], (size_t)c->lvl_pitch[i][0] * t.rect.h + 256));
        for (int l = 0; l < c->levels; l++)
            HIP_TRY(c, hipMalloc((void**)&c->wgt[i][l], ((size_t)c->wpitch[i][l] * (t.rect.h >> l) + 64) * sizeof(float)));
    }
    for (int l = 0; l < c->levels; l++) {
        int cw = P.canvas.w >> l, ch = P.canvas.h >> l;
        c->cv.cpitch[l] = (int)align_up((size_t)cw, 8);
        c->cv.cplane[l] = c->cv.cpitch[l] * ch;
        HIP_TRY(c, hipMalloc((void**)&c->wsum[l], (size_t)cw * ch * sizeof(float)));
        if (l > 0) HIP_TRY(c, hipMalloc((void**)&c->canvas[l], (size_t)c->cv.cplane[l] * 3 * sizeof(int16_t) + 256));
    }
    for (int i = 0; i < n; i++)
        if (c->have_fe[i]) {
            FrontEndDev fe{};
            const pano_undistort& u = c->und[i];
            fe.raw_w = u.raw_w; fe.raw_h = u.raw_h; fe.undist_w = u.undist_w; fe.undist_h = u.undist_h;
            fe.out_w = P.src_w; fe.out_h = P.src_h;
            std::memcpy(fe.rect, u.rect, sizeof(fe.rect));
            std::memcpy(fe.K, u.K, sizeof(fe.K));
            std::memcpy(fe.newK, c->newK[i], sizeof(fe.newK));
            std::memcpy(fe.dist, u.dist, sizeof(fe.dist));
            pano_status us = upload(c, &c->d_fe[i], &fe, sizeof(fe));
            if (us != PANO_OK) return us;
        }
    // static remap tables of the warp (K1): the projection of every tile pixel is fixed from here on.
    // PANO_WARP_ON_THE_FLY=1 keeps the projecting kernel (also what frames beyond 8192 x 8192 use).
    // hipGraph replay of the frame is opt-in (PANO_GRAPH=1): measured on MI355X the 12 stream-ordered launches
    // of a frame run 3 % faster than the replayed graph (0.268 vs 0.276 ms per 8-camera panorama) - the GPU,
    // not the host, is the limiter
    c->use_graph = getenv("PANO_GRAPH") && atoi(getenv("PANO_GRAPH"));
    // the codes are relative to each workgroup's source box, so the frame size does not limit the table; 8192 keeps the
    // byte offsets of a frame inside 24-bit multiplies
    c->use_lut = c->frame_w <= 8192 && c->frame_h <= 8192 && !(getenv("PANO_WARP_ON_THE_FLY") && atoi(getenv("PANO_WARP_ON_THE_FLY")));
    if (c->use_lut) {
        bool too_wide = false;
        for (int i = 0; i < n; i++) {
            const FeedTile& t = P.tile[i];
            const size_t nb = (size_t)((t.rect.w + 63) / 64) * ((t.rect.h + 15) / 16);
            c->lut_pitch[i] = (int)align_up((size_t)t.rect.w, 8);
            HIP_TRY(c, hipMalloc((void**)&c->lut[i], (size_t)c->lut_pitch[i] * t.rect.h * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc((void**)&c->box[i], nb * sizeof(int4)));
            unsigned* d_cnt = nullptr;
            HIP_TRY(c, hipMalloc((void**)&d_cnt, 2 * sizeof(unsigned)));
            HIP_TRY(c, hipMemset(d_cnt, 0, 2 * sizeof(unsigned)));
            // table and boxes are in pixels of the frame K1 samples (the RAW frame when a front end is set)
            WarpCam w = make_warp_cam(c, i, nullptr, (size_t)c->frame_w * 3, false);
            launch_build_warp_table(w, c->lut[i], c->lut_pitch[i], c->box[i], d_cnt, nullptr);
            unsigned h_cnt[2] = {0, 0};
            hipError_t ce = hipMemcpy(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost);
            (void)hipFree(d_cnt);
            HIP_TRY(c, ce);
            c->box_global[i] = h_cnt[0];
            too_wide |= h_cnt[1] != 0;
            c->h_box[i].resize(nb);
            HIP_TRY(c, hipMemcpy(c->h_box[i].data(), c->box[i], nb * sizeof(int4), hipMemcpyDeviceToHost));
            const int gp = c->lut_pitch[i] / 4;
            HIP_TRY(c, hipMalloc((void**)&c->k1_flags[i], nb * sizeof(uint32_t)));
            HIP_TRY(c, hipMemset(c->k1_flags[i], 0, nb * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc((void**)&c->lutc[i], (size_t)gp * t.rect.h * sizeof(uint2)));
            launch_pack_warp_lut(c->lut[i], c->lut_pitch[i], t.rect.w, t.rect.h, c->lutc[i], gp, c->k1_flags[i], nullptr);
            std::vector<uint32_t> hf(nb);
            HIP_TRY(c, hipMemcpy(hf.data(), c->k1_flags[i], nb * sizeof(uint32_t), hipMemcpyDeviceToHost));
            c->k1_blocks[i] = (long long)nb;
            c->k1_flagged[i] = 0;
            for (uint32_t f : hf) c->k1_flagged[i] += f != 0;
        }
        if (too_wide) {
            // a 64 x 16 patch that spans 2048 source pixels (a projection that magnifies 32 x): no table for this rig
            for (int i = 0; i < n; i++) { dfree(c->lut[i]); dfree(c->lutc[i]); dfree(c->box[i]); dfree(c->k1_flags[i]); c->h_box[i].clear(); }
            c->use_lut = false;
        }
        HIP_TRY(c, hipDeviceSynchronize());
    }
    // kernel parameter blocks
    c->pyr = PyrParams{};
    c->pyr.ncam = n;
    for (int i = 0; i < n; i++) {
        PyrCam& pc = c->pyr.cam[i];
        pc.w0 = P.tile[i].rect.w; pc.h0 = P.tile[i].rect.h;
        pc.tx = P.tile[i].rect.x; pc.ty = P.tile[i].rect.y;
        pc.mask0 = c->mask0[i];
        for (int l = 0; l < c->levels; l++) {
            pc.lvl[l] = (uint8_t*)(c->pyr_base + (size_t)i * c->slot_bytes + c->lvl_off[i][l]);
            pc.wgt[l] = c->wgt[i][l];
            pc.pitch[l] = c->lvl_pitch[i][l];
            pc.plane[l] = c->lvl_plane[i][l];
            pc.wpitch[l] = c->wpitch[i][l];
        }
    }
    for (int l = 0; l < c->levels; l++) {
        c->cv.img[l] = c->canvas[l];
        // the vector blend kernel needs every tile box of the level on a 4 x 2 grid
        // ... and only pays on big levels: small ones are latency bound and want one pixel per thread
        // 200 K pixels: on the 1080p rig levels 0..2 run the vector kernel, 3..5 the fused small-level pair.  Measured with
        // two frames in flight (the GPU is VALU-issue bound there and the small-level kernels spend 3x the instructions
        // per pixel): 600 K -> 8860, 200 K -> 9200, 50 K -> 9080 panoramas/s
        constexpr size_t vec_min_px = 200000;
        bool fast = P.bands >= 0 && ((P.canvas.w >> l) % 4 == 0) && ((P.canvas.h >> l) % 2 == 0) &&
                    (size_t)(P.canvas.w >> l) * (P.canvas.h >> l) >= vec_min_px;
        for (int i = 0; i < n && fast; i++) {
            const Rect& r = P.tile[i].rect;
            fast = ((r.x >> l) % 4 == 0) && ((r.y >> l) % 2 == 0) && ((r.w >> l) % 4 == 0) && ((r.h >> l) % 2 == 0) &&
                   ((r.w >> l) << l) == r.w && ((r.x >> l) << l) == r.x;
        }
        c->cv.fast[l] = fast ? 1 : 0;
        if (fast) {
            c->cv.opitch[l] = (int)align_up((size_t)(P.canvas.w >> l) / 4, 64);
            HIP_TRY(c, hipMalloc((void**)&c->owner[l], (size_t)c->cv.opitch[l] * ((P.canvas.h >> l) / 2) * sizeof(uint16_t)));
            c->cv.owner[l] = c->owner[l];
        }
    }
    // the levels above the last vector level run fused (one normalise launch + one LDS collapse launch)
    c->cv.small_base = 0;
    if (P.bands >= 1) {
        int k = 0;
        while (k <= P.bands && c->cv.fast[k]) k++;
        if (k >= 1 && P.bands - k + 1 >= 2) c->cv.small_base = k;
    }
    c->full_tiles = getenv("PANO_FULL_TILES") && atoi(getenv("PANO_FULL_TILES"));
    // level 0 walks its tiles seam tiles first, with the owner codes of their waves in the order table (A/B lever; PANO_L0_ORDER=0:
    // plain XCD-band order, every wave looks its owners up; no result changes)
    c->l0_order = !(getenv("PANO_L0_ORDER") && atoi(getenv("PANO_L0_ORDER")) == 0);
    live_rects(c, {});  // no masks yet: every pixel of every level is live
    c->cv.cam_lo = 0;
    c->cv.cam_n = n;
    c->cv.w0 = P.canvas.w; c->cv.h0 = P.canvas.h;
    c->cv.bands = P.bands < 0 ? 0 : P.bands;
    c->cv.cut_x = P.cut.x; c->cv.cut_y = P.cut.y; c->cv.cut_w = P.cut.w; c->cv.cut_h = P.cut.h;
    c->cv.final_w = P.pano.w; c->cv.final_h = P.pano.h;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    for (auto& sl : c->ring)
        for (auto& e : sl.e) HIP_TRY(c, hipEventCreate(&e));
    c->ev_valid = true;
    return PANO_OK;
}

// point the kernel parameter blocks at the buffers of frame slot k
static void bind_slot(pano_ctx* c, int k) {
    c->cur_slot = k;
    c->pyr_base = c->slot_pyr[k];
    for (int l = 0; l < c->levels; l++) {
        c->canvas[l] = c->slot_canvas[k][l];
        c->cv.img[l] = c->canvas[l];
    }
    for (int i = 0; i < c->plan.n; i++)
        for (int l = 0; l < c->levels; l++)
            c->pyr.cam[i].lvl[l] = (uint8_t*)(c->pyr_base + (size_t)i * c->slot_bytes + c->lvl_off[i][l]);
}

static pano_status set_frame_slots_impl(pano_ctx* c, int n) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (n < 1 || n > PANO_MAX_FRAME_SLOTS) return fail(c, PANO_EINVAL, "frame slots: 1 .. PANO_MAX_FRAME_SLOTS");
    HIP_TRY(c, hipDeviceSynchronize());
    if (c->nslots == 1) {  // slot 0 = the buffers pano_prepare allocated
        c->slot_pyr[0] = c->pyr_base;
        for (int l = 0; l < kMaxLevels; l++) c->slot_canvas[0][l] = c->canvas[l];
    }
    bind_slot(c, 0);
    for (int k = n; k < c->nslots; k++) {  // shrink
        dfree(c->slot_pyr[k]);
        for (int l = 0; l < kMaxLevels; l++) dfree(c->slot_canvas[k][l]);
    }
    const int have = std::min(c->nslots, n);
    c->nslots = have;
    drop_graphs(c);
    for (int k = have; k < n; k++) {  // grow: a slot counts only once every one of its buffers exists
        char* pyr = nullptr;
        int16_t* cvs[kMaxLevels] = {};
        bool ok = hipMalloc((void**)&pyr, c->slot_bytes * c->plan.n + 256) == hipSuccess &&
                  hipMemset(pyr, 0, c->slot_bytes * c->plan.n) == hipSuccess;
        for (int l = 1; ok && l < c->levels; l++)
            ok = hipMalloc((void**)&cvs[l], (size_t)c->cv.cplane[l] * 3 * sizeof(int16_t) + 256) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            dfree(pyr);
            for (int l = 1; l < c->levels; l++) dfree(cvs[l]);
            return fail(c, PANO_EHIP, "hipMalloc (frame slot): the slots allocated so far stay usable");
        }
        c->slot_pyr[k] = pyr;
        for (int l = 1; l < c->levels; l++) c->slot_canvas[k][l] = cvs[l];
        c->nslots = k + 1;
    }
    return PANO_OK;
}

pano_status pano_select_frame_slot(pano_ctx* c, int k) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (k < 0 || k >= c->nslots) return fail(c, PANO_EINVAL, "frame slot out of range");
    if (c->nslots > 1 && k != c->cur_slot) bind_slot(c, k);  // captured graphs are keyed by slot
    return PANO_OK;
}

pano_status pano_get_roi(const pano_ctx* c, int i, int r[4]) {
    if (!c || !r || !c->prepared || i < 0 || i >= c->plan.n) return PANO_EINVAL;
    r[0] = c->plan.roi[i].x; r[1] = c->plan.roi[i].y; r[2] = c->plan.roi[i].w; r[3] = c->plan.roi[i].h;
    return PANO_OK;
}

pano_status pano_get_pano_rect(const pano_ctx* c, int r[4]) {
    if (!c || !r || !c->prepared) return PANO_EINVAL;
    r[0] = c->plan.pano.x; r[1] = c->plan.pano.y; r[2] = c->plan.pano.w; r[3] = c->plan.pano.h;
    return PANO_OK;
}

pano_status pano_get_num_bands(const pano_ctx* c, int* nb) {
    if (!c || !nb || !c->prepared) return PANO_EINVAL;
    *nb = c->plan.bands;
    return PANO_OK;
}

pano_status pano_get_feed_tile(const pano_ctx* c, int i, int r[4], int tblr[4]) {
    if (!c || !r || !tblr || !c->prepared || i < 0 || i >= c->plan.n) return PANO_EINVAL;
    const FeedTile& t = c->plan.tile[i];
    r[0] = t.rect.x; r[1] = t.rect.y; r[2] = t.rect.w; r[3] = t.rect.h;
    tblr[0] = t.top; tblr[1] = t.bottom; tblr[2] = t.left; tblr[3] = t.right;
    return PANO_OK;
}

pano_status pano_set_cut(pano_ctx* c, const int r[4]) {
    if (!c || !r) return PANO_EINVAL;
    if (!c->prepared) {
        std::memcpy(c->cfg.cut, r, 4 * sizeof(int));
        return PANO_OK;
    }
    Rect cut{r[0], r[1], r[2], r[3]};
    if (cut.w == 0 || cut.h == 0) cut = Rect{0, 0, c->plan.pano.w, c->plan.pano.h};
    if (cut.x < 0 || cut.y < 0 || cut.w < 0 || cut.h < 0 || cut.x + cut.w > c->plan.pano.w || cut.y + cut.h > c->plan.pano.h)
        return fail(c, PANO_EINVAL, "cut rectangle outside the panorama");
    drop_graphs(c);
    c->plan.cut = cut;
    c->cv.cut_x = cut.x; c->cv.cut_y = cut.y; c->cv.cut_w = cut.w; c->cv.cut_h = cut.h;
    c->order_dirty = true;  // the level-0 tile grid lies over the hull of the cut
    return PANO_OK;
}

pano_status pano_get_output_size(const pano_ctx* c, int* w, int* h) {
    if (!c || !w || !h || !c->prepared) return PANO_EINVAL;
    *w = c->plan.cut.w;
    *h = c->plan.cut.h;
    return PANO_OK;
}

static pano_status set_mask_impl(pano_ctx* c, int i, const uint8_t* h_mask, int w, int h, size_t stride) {
    pano_status s = check_compute(c);
    if (s != PANO_OK) return s;
    if (!h_mask || i < 0 || i >= c->plan.n) return PANO_EINVAL;
    if (w != c->plan.roi[i].w || h != c->plan.roi[i].h || stride < (size_t)w)
        return fail(c, PANO_EINVAL, "mask must be ROI sized (pano_get_roi)");
    HIP_TRY(c, hipMemcpy2D(c->mask[i], (size_t)w, h_mask, stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice));
    c->mask_set[i] = true;
    c->weights_dirty = true;
    live_rects(c, {});  // until the weights are rebuilt, produce every pixel
    drop_graphs(c);
    return PANO_OK;
}

pano_status pano_get_mask(pano_ctx* c, int i, uint8_t* h_mask, size_t stride) {
    pano_status s = check_compute(c);
    if (s != PANO_OK) return s;
    if (!h_mask || i < 0 || i >= c->plan.n) return PANO_EINVAL;
    if (!c->mask_set[i]) return fail(c, PANO_ESTATE, "mask not set");
    int w = c->plan.roi[i].w, h = c->plan.roi[i].h;
    if (stride < (size_t)w) return PANO_EINVAL;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy2D(h_mask, stride, c->mask[i], (size_t)w, (size_t)w, (size_t)h, hipMemcpyDeviceToHost));
    return PANO_OK;
}

static pano_status build_masks_voronoi_impl(pano_ctx* c) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    const int n = c->plan.n;
    hipStream_t s = c->own_stream;
    Scratch tmp;
    SeamWarps sm;
    if ((st = seam_scale_warps(c, nullptr, nullptr, tmp, s, sm)) != PANO_OK) return st;
    const std::vector<Rect>& sroi = sm.roi;
    // PairwiseSeamFinder::run order
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            int x_tl = std::max(sroi[i].x, sroi[j].x), y_tl = std::max(sroi[i].y, sroi[j].y);
            int x_br = std::min(sroi[i].x + sroi[i].w, sroi[j].x + sroi[j].w);
            int y_br = std::min(sroi[i].y + sroi[i].h, sroi[j].y + sroi[j].h);
            if (!(x_tl < x_br && y_tl < y_br)) continue;
            int* scratch = nullptr;
            if (!tmp.alloc(&scratch, voronoi_scratch_ints(x_br - x_tl, y_br - y_tl) * sizeof(int))) return fail(c, PANO_EHIP, "hipMalloc");
            launch_voronoi_pair(sm.mask[i], sroi[i].w, sroi[i].h, sroi[i].x, sroi[i].y, sm.mask[j], sroi[j].w, sroi[j].h,
                                sroi[j].x, sroi[j].y, x_tl, y_tl, x_br - x_tl, y_br - y_tl, scratch, s);
        }
    return finish_seam_masks(c, sm, tmp, s);
}

// GraphCutSeamFinder over the seam-scale warps `sm` (PairwiseSeamFinder::run order; GraphCutSeamFinder::Impl::findInPair per
// overlapping pair: weights on the GPU, the max-flow on the host - pano_graphcut.hpp -, the mask update on the GPU; a later pair
// sees the masks the earlier left).  Touches nothing of a context: it also runs on the refresh thread (pano_refresh_masks_begin)
// dump (optional, pano_debug_graphcut_dump): every pair's graph AS THE GPU BUILT IT and the labels the host max-flow gave it are
// appended - int32 {i, j, W, H}, then W*H f32 term, wh, wv and W*H label bytes (1 = source side)
static pano_status graphcut_pairs(int n, SeamWarps& sm, Scratch& tmp, hipStream_t s, std::string& err, FILE* dump = nullptr) {
#define GC_TRY(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            return PANO_EHIP;                                                     \
        }                                                                         \
    } while (0)
    GainImages gi{};
    for (int i = 0; i < n; i++) { gi.img[i] = sm.img[i]; gi.mask[i] = sm.mask[i]; gi.w[i] = sm.roi[i].w; }
    const int gap = 10;
    std::vector<float> term, wh, wv;
    std::vector<uint8_t> in_source;
    for (int i = 0; i < n - 1; i++)
        for (int j = i + 1; j < n; j++) {
            const Rect &ra = sm.roi[i], &rb = sm.roi[j];
            const int x_tl = std::max(ra.x, rb.x), y_tl = std::max(ra.y, rb.y);
            const int x_br = std::min(ra.x + ra.w, rb.x + rb.w), y_br = std::min(ra.y + ra.h, rb.y + rb.h);
            if (!(x_tl < x_br && y_tl < y_br)) continue;
            GcPair q{};
            q.W = x_br - x_tl + 2 * gap; q.H = y_br - y_tl + 2 * gap;
            q.a = i; q.ax = x_tl - ra.x - gap; q.ay = y_tl - ra.y - gap; q.wa = ra.w; q.ha = ra.h;
            q.b = j; q.bx = x_tl - rb.x - gap; q.by = y_tl - rb.y - gap; q.wb = rb.w; q.hb = rb.h;
            const size_t nv = (size_t)q.W * q.H;
            float *d_term = nullptr, *d_wh = nullptr, *d_wv = nullptr;
            uint8_t* d_lab = nullptr;
            if (!tmp.alloc(&d_term, nv * sizeof(float)) || !tmp.alloc(&d_wh, nv * sizeof(float)) || !tmp.alloc(&d_wv, nv * sizeof(float)) ||
                !tmp.alloc(&d_lab, nv)) {
                err = "hipMalloc (graph cut)";
                return PANO_EHIP;
            }
            launch_graphcut_weights(gi, q, d_term, d_wh, d_wv, s);
            GC_TRY(hipGetLastError());
            term.resize(nv); wh.resize(nv); wv.resize(nv); in_source.resize(nv);
            GC_TRY(hipMemcpyAsync(term.data(), d_term, nv * sizeof(float), hipMemcpyDeviceToHost, s));
            GC_TRY(hipMemcpyAsync(wh.data(), d_wh, nv * sizeof(float), hipMemcpyDeviceToHost, s));
            GC_TRY(hipMemcpyAsync(wv.data(), d_wv, nv * sizeof(float), hipMemcpyDeviceToHost, s));
            GC_TRY(hipStreamSynchronize(s));
            GridMaxFlow flow(q.W, q.H, term.data(), wh.data(), wv.data());
            flow.run();
            for (size_t k = 0; k < nv; k++) in_source[k] = flow.inSource((int)k) ? 1 : 0;
            if (dump) {
                const int hdr[4] = {i, j, q.W, q.H};
                fwrite(hdr, sizeof(int), 4, dump);
                fwrite(term.data(), sizeof(float), nv, dump);
                fwrite(wh.data(), sizeof(float), nv, dump);
                fwrite(wv.data(), sizeof(float), nv, dump);
                fwrite(in_source.data(), 1, nv, dump);
            }
            GC_TRY(hipMemcpyAsync(d_lab, in_source.data(), nv, hipMemcpyHostToDevice, s));
            launch_graphcut_apply(q, sm.mask[i], sm.mask[j], d_lab, gap, s);
            GC_TRY(hipStreamSynchronize(s));  // in_source is reused by the next pair
        }
#undef GC_TRY
    return PANO_OK;
}

// pano_refresh_masks_*: updateMask beside the frame loop.  begin() uploads the frames and warps them at the seam scale (a few ms
// on a stream of the job's own), then a thread runs the graph cuts (the host max-flow: tens of ms); poll() installs the masks
// once the thread is through - on the caller's thread, like pano_build_masks_graphcut does at its end
struct MaskJob {
    std::thread th;
    std::atomic<int> state{0};  // 1 running, 2 masks ready, 3 failed
    Scratch tmp;
    SeamWarps sm;
    hipStream_t s = nullptr;
    pano_status st = PANO_OK;
    std::string err;
};
static void reap_trash(pano_ctx* c) {
    MaskJob* j = c->job_trash;
    if (!j) return;
    if (j->th.joinable()) j->th.join();
    c->job_trash = nullptr;
    delete j;
}
static void drop_job(pano_ctx* c) {
    reap_trash(c);
    MaskJob* j = c->job;
    if (!j) return;
    if (j->th.joinable()) j->th.join();
    if (j->s) (void)hipStreamDestroy(j->s);
    c->job = nullptr;
    delete j;  // frees the job's device scratch
}
// the masks are installed: the job's device buffers go back to the context's pool (the next refresh asks for the same sizes)
static void retire_job(pano_ctx* c) {
    reap_trash(c);
    MaskJob* j = c->job;
    c->job = nullptr;
    j->tmp.release();  // pooled: no hipFree
    if (j->s) (void)hipStreamDestroy(j->s);
    delete j;
}
static pano_status refresh_begin_impl(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!h_frames || !strides) return PANO_EINVAL;
    const int n = c->plan.n;
    for (int i = 0; i < n; i++)
        if (!h_frames[i] || strides[i] < (size_t)c->plan.src_w * 3) return fail(c, PANO_EINVAL, "frame pointer / stride");
    if (c->job) return fail(c, PANO_ESTATE, "a mask refresh is under way: pano_refresh_masks_poll / _wait first");
    reap_trash(c);
    MaskJob* j = new MaskJob;
    j->tmp.pool = &c->refresh_pool;
    c->job = j;
    if (hipStreamCreateWithFlags(&j->s, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        drop_job(c);
        return fail(c, PANO_EHIP, "hipStreamCreate (mask refresh)");
    }
    if ((st = seam_scale_warps(c, h_frames, strides, j->tmp, j->s, j->sm)) == PANO_OK && hipStreamSynchronize(j->s) != hipSuccess)
        st = fail(c, PANO_EHIP, "hipStreamSynchronize (mask refresh)");
    if (st != PANO_OK) {  // the caller's frames are no longer needed either way
        drop_job(c);
        return st;
    }
    j->state = 1;
    const int device = c->device;
    Scratch::Pool* pool = &c->pairs_pool;
    j->th = std::thread([j, n, device, pool]() {
        pano_status r = PANO_EHIP;
        try {
            if (hipSetDevice(device) == hipSuccess) {
                Scratch pairs;  // the graphs of the pairs, from a pool that is this thread's while it runs: a hipFree here
                pairs.pool = pool;  // would hold up the frame loop's launches too (it waits for the device under the runtime's lock)
                r = graphcut_pairs(n, j->sm, pairs, j->s, j->err);
            }
            else j->err = "hipSetDevice (mask refresh thread)";
        } catch (const std::exception& e) {
            r = PANO_ERR;
            j->err = e.what();
        } catch (...) {
            r = PANO_ERR;
            j->err = "unknown exception (mask refresh thread)";
        }
        j->st = r;
        j->state = r == PANO_OK ? 2 : 3;
    });
    return PANO_OK;
}
static pano_status refresh_poll_impl(pano_ctx* c, int* done, bool wait) {
    if (done) *done = 0;
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    MaskJob* j = c->job;
    if (!j) return PANO_OK;
    if (j->state == 1 && !wait) return PANO_OK;
    if (j->th.joinable()) j->th.join();
    if (j->state == 3) {
        st = fail(c, j->st, j->err.c_str());
        drop_job(c);
        return st;
    }
    st = finish_seam_masks(c, j->sm, j->tmp, c->own_stream);
    retire_job(c);
    if (st == PANO_OK && done) *done = 1;
    return st;
}

static pano_status build_masks_graphcut_impl(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!h_frames || !strides) return PANO_EINVAL;
    if (c->job && (st = refresh_poll_impl(c, nullptr, true)) != PANO_OK) return st;  // a refresh under way ends first
    const int n = c->plan.n;
    for (int i = 0; i < n; i++)
        if (!h_frames[i] || strides[i] < (size_t)c->plan.src_w * 3) return fail(c, PANO_EINVAL, "frame pointer / stride");
    hipStream_t s = c->own_stream;
    Scratch tmp;
    tmp.pool = &c->refresh_pool;  // calibration's cut leaves the buffers the refreshes beside the loop will ask for
    SeamWarps sm;
    if ((st = seam_scale_warps(c, h_frames, strides, tmp, s, sm)) != PANO_OK) return st;
    std::string err;
    {
        Scratch pairs;
        pairs.pool = &c->pairs_pool;  // no refresh thread is running (waited for above)
        FILE* dump = c->gc_dump_path.empty() ? nullptr : fopen(c->gc_dump_path.c_str(), "ab");
        st = graphcut_pairs(n, sm, pairs, s, err, dump);
        if (dump) fclose(dump);
        if (st != PANO_OK) return fail(c, st, err.c_str());
    }
    return finish_seam_masks(c, sm, tmp, s);
}

pano_status pano_set_gain_map(pano_ctx* c, int i, const float* h_gain, int gw, int gh) {
    pano_status s = check_compute(c);
    if (s != PANO_OK) return s;
    if (i < 0 || i >= c->plan.n) return PANO_EINVAL;
    HIP_TRY(c, hipDeviceSynchronize());
    drop_graphs(c);
    if (!h_gain) {
        dfree(c->gain[i]);
        return PANO_OK;
    }
    if (gw < 1 || gh < 1) return PANO_EINVAL;
    c->gain_w[i] = gw;
    c->gain_h[i] = gh;
    if ((s = upload(c, &c->gain[i], h_gain, (size_t)gw * gh * sizeof(float)))) return s;
    return upload_gain_tables(c, i, h_gain);
}

pano_status pano_get_gain_map(pano_ctx* c, int i, float* h_gain, int* gw, int* gh) {
    pano_status s = check_compute(c);
    if (s != PANO_OK) return s;
    if (i < 0 || i >= c->plan.n) return PANO_EINVAL;
    const bool have = c->gain[i] != nullptr;
    if (gw) *gw = have ? c->gain_w[i] : 0;
    if (gh) *gh = have ? c->gain_h[i] : 0;
    if (h_gain && have)
        HIP_TRY(c, hipMemcpy(h_gain, c->gain[i], (size_t)c->gain_w[i] * c->gain_h[i] * sizeof(float), hipMemcpyDeviceToHost));
    return PANO_OK;
}

static pano_status estimate_gains_impl(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides, int block_w,
                                int block_h) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!h_frames || !strides || block_w < 1 || block_h < 1) return PANO_EINVAL;
    const Plan& P = c->plan;
    const int n = P.n, sw = P.src_w;
    for (int i = 0; i < n; i++)
        if (!h_frames[i] || strides[i] < (size_t)sw * 3) return fail(c, PANO_EINVAL, "frame pointer / stride");
    hipStream_t s = c->own_stream;
    Scratch tmp;
    auto oom = [&]() { return fail(c, PANO_EHIP, "hipMalloc / hipMemcpy (gain estimation)"); };
    SeamWarps sm;
    if ((st = seam_scale_warps(c, h_frames, strides, tmp, s, sm)) != PANO_OK) return st;
    const std::vector<Rect>& sroi = sm.roi;
    GainImages gi{};
    for (int i = 0; i < n; i++) { gi.img[i] = sm.img[i]; gi.mask[i] = sm.mask[i]; gi.w[i] = sroi[i].w; }
    // BlocksGainCompensator::feed: equalised blocks of every image, in image order then row-major
    struct Block { int cam, x, y, w, h; };
    std::vector<Block> blk;
    std::vector<int> per_w(n), per_h(n);
    for (int i = 0; i < n; i++) {
        const int cols = sroi[i].w, rows = sroi[i].h;
        per_w[i] = (cols + block_w - 1) / block_w;
        per_h[i] = (rows + block_h - 1) / block_h;
        const int bw = (cols + per_w[i] - 1) / per_w[i], bh = (rows + per_h[i] - 1) / per_h[i];
        for (int by = 0; by < per_h[i]; by++)
            for (int bx = 0; bx < per_w[i]; bx++)
                blk.push_back({i, bx * bw, by * bh, std::min(bx * bw + bw, cols) - bx * bw, std::min(by * bh + bh, rows) - by * bh});
    }
    const int nb = (int)blk.size();
    if (nb > 4096) return fail(c, PANO_EINVAL, "gain estimation: more than 4096 blocks (raise the block size)");
    // GainCompensator::feed on the blocks: the overlapping pairs i <= j (a block overlaps itself)
    std::vector<GainPair> pairs;
    std::vector<int> pi, pj2;
    for (int i = 0; i < nb; i++)
        for (int j = i; j < nb; j++) {
            const Block &A = blk[i], &B = blk[j];
            const int ax = sroi[A.cam].x + A.x, ay = sroi[A.cam].y + A.y, bx = sroi[B.cam].x + B.x, by = sroi[B.cam].y + B.y;
            const int x0 = std::max(ax, bx), y0 = std::max(ay, by), x1 = std::min(ax + A.w, bx + B.w), y1 = std::min(ay + A.h, by + B.h);
            if (!(x0 < x1 && y0 < y1)) continue;
            pairs.push_back({A.cam, A.x + x0 - ax, A.y + y0 - ay, B.cam, B.x + x0 - bx, B.y + y0 - by, x1 - x0, y1 - y0});
            pi.push_back(i);
            pj2.push_back(j);
        }
    const int np = (int)pairs.size();
    GainPair* d_pairs = nullptr;
    int* d_cnt = nullptr;
    double *d_sa = nullptr, *d_sb = nullptr;
    if (!tmp.put(&d_pairs, pairs.data(), (size_t)np * sizeof(GainPair)) || !tmp.alloc(&d_cnt, (size_t)np * sizeof(int)) ||
        !tmp.alloc(&d_sa, (size_t)np * sizeof(double)) || !tmp.alloc(&d_sb, (size_t)np * sizeof(double)))
        return oom();
    launch_gain_pairs(gi, d_pairs, np, d_cnt, d_sa, d_sb, s);
    HIP_TRY(c, hipGetLastError());
    std::vector<int> cnt(np);
    std::vector<double> sa(np), sb(np);
    HIP_TRY(c, hipMemcpyAsync(cnt.data(), d_cnt, (size_t)np * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(sa.data(), d_sa, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(sb.data(), d_sb, (size_t)np * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    // N, I, then the normal equations of the gain model (alpha = 0.01, beta = 100) and cv::solve
    std::vector<int> N((size_t)nb * nb, 0);
    std::vector<double> I((size_t)nb * nb, 0.0), A((size_t)nb * nb, 0.0), g(nb, 0.0);
    for (int p = 0; p < np; p++) {
        const int i = pi[p], j = pj2[p], m = std::max(1, cnt[p]);
        N[(size_t)i * nb + j] = N[(size_t)j * nb + i] = m;
        I[(size_t)i * nb + j] = sa[p] / m;
        I[(size_t)j * nb + i] = sb[p] / m;
    }
    const double alpha = 0.01, beta = 100;
    for (int i = 0; i < nb; i++)
        for (int j = 0; j < nb; j++) {
            const double nij = N[(size_t)i * nb + j], iij = I[(size_t)i * nb + j], iji = I[(size_t)j * nb + i];
            g[i] += beta * nij;
            A[(size_t)i * nb + i] += beta * nij;
            if (j == i) continue;
            A[(size_t)i * nb + i] += 2 * alpha * iij * iij * nij;
            A[(size_t)i * nb + j] -= 2 * alpha * iij * iji * nij;
        }
    if (!solveLU(A, nb, g)) return fail(c, PANO_ESTATE, "gain estimation: singular system");
    // gain maps: the block gains as f32, smoothed twice (BlocksGainCompensator::feed tail), then installed like
    // pano_set_gain_map (apply = stitching_detailed.cpp:841)
    HIP_TRY(c, hipDeviceSynchronize());
    drop_graphs(c);
    int k = 0;
    for (int i = 0; i < n; i++) {
        std::vector<float> map((size_t)per_w[i] * per_h[i]);
        for (size_t q = 0; q < map.size(); q++) map[q] = static_cast<float>(g[k++]);
        smooth121(map, per_w[i], per_h[i]);
        smooth121(map, per_w[i], per_h[i]);
        c->gain_w[i] = per_w[i];
        c->gain_h[i] = per_h[i];
        if ((st = upload(c, &c->gain[i], map.data(), map.size() * sizeof(float)))) return st;
        if ((st = upload_gain_tables(c, i, map.data()))) return st;
    }
    return PANO_OK;
}

pano_status pano_warp(pano_ctx* c, int i, const uint8_t* d_src, size_t src_stride, uint8_t* d_dst, size_t dst_stride,
                      void* stream) {
    pano_status s = check_compute(c);
    if (s != PANO_OK) return s;
    if (!d_src || !d_dst || i < 0 || i >= c->plan.n) return PANO_EINVAL;
    if (src_stride < (size_t)c->frame_w * 3 || dst_stride < (size_t)c->plan.roi[i].w * 3) return PANO_EINVAL;
    WarpCam w = make_warp_cam(c, i, d_src, src_stride, true);
    w.dst = d_dst;
    w.dst_pitch = (int)dst_stride;
    launch_warp_image(w, (hipStream_t)stream);
    HIP_TRY(c, hipGetLastError());
    return PANO_OK;
}

pano_status pano_warp_mask(pano_ctx* c, int i, uint8_t* d_dst, size_t dst_stride, void* stream) {
    pano_status s = check_compute(c);
    if (s != PANO_OK) return s;
    if (!d_dst || i < 0 || i >= c->plan.n || dst_stride < (size_t)c->plan.roi[i].w) return PANO_EINVAL;
    WarpCam w = make_warp_cam(c, i, nullptr, 0, true);
    launch_warp_mask(w, d_dst, (int)dst_stride, (hipStream_t)stream);
    HIP_TRY(c, hipGetLastError());
    return PANO_OK;
}

// pyrDown launches per frame: every level, or - when the small levels run fused - only up to small_base (the fused kernel
// builds the levels above it in LDS)
static int pyr_levels(const pano_ctx* c) { return c->plan.bands; }

pano_status pano_feed_cameras(pano_ctx* c, unsigned cam_bits, const uint8_t* const* d_frames, const size_t* strides,
                              void* stream) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!d_frames || !strides) return PANO_EINVAL;
    const Plan& P = c->plan;
    hipStream_t s = (hipStream_t)stream;
    cam_bits &= (1u << P.n) - 1u;
    // the live rects follow the masks: once every mask is there, have them (and the weights) current, so that a rank
    // that only feeds (camera sharding) skips the same dead pixels as the rank that blends
    if (c->weights_dirty) {
        bool all = true;
        for (int i = 0; i < P.n; i++) all &= c->mask_set[i];
        if (all && (st = ensure_weights(c, s)) != PANO_OK) return st;
    }
    WarpParams wp{};
    int k = 0, mw = 0, mh = 0;
    for (int i = 0; i < P.n; i++) {
        if (!((cam_bits >> i) & 1u)) continue;
        if (!d_frames[i] || strides[i] < (size_t)c->frame_w * 3) return fail(c, PANO_EINVAL, "frame pointer / stride");
        wp.cam[k++] = make_warp_cam(c, i, d_frames[i], strides[i], false);
        mw = std::max(mw, P.tile[i].rect.w);
        mh = std::max(mh, P.tile[i].rect.h);
    }
    if (k == 0) return PANO_OK;
    if (c->profiling) {
        // K1's events carry the dispatch's own begin/end timestamps (what rocprofv3 reports per kernel)
        if ((st = begin_slot(c)) != PANO_OK) return st;
        pano_ctx::EvSlot& sl = c->ring[c->ev_cur];
        launch_warp_tiles(wp, k, mw, mh, s, sl.e[0], sl.e[1]);
        sl.recorded |= 3u;
    } else {
        launch_warp_tiles(wp, k, mw, mh, s);
    }
    launch_pyr_chain(c->pyr, cam_bits, pyr_levels(c), s);
    if (c->profiling && (st = record(c, 2, s)) != PANO_OK) return st;
    HIP_TRY(c, hipGetLastError());
    return PANO_OK;
}

pano_status pano_blend(pano_ctx* c, uint8_t* d_out, size_t out_stride, void* stream) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    const Plan& P = c->plan;
    if (!d_out || out_stride < (size_t)P.cut.w * 3) return PANO_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((st = ensure_weights(c, s)) != PANO_OK) return st;
    CanvasParams cv = c->cv;
    cv.out = d_out;
    cv.out_stride = (int)out_stride;
    CanvasSet cs{};
    cs.n = 1;
    cs.c[0] = cv;
    if (P.bands < 0) {
        launch_no_blend(c->pyr, cs, s);
    } else {
        int top = P.bands;
        if (cv.small_base > 0) {
            launch_blend_small(c->pyr, cs, s);
            top = cv.small_base - 1;
        }
        if (c->profiling && c->ev_cur < 0 && (st = begin_slot(c)) != PANO_OK) return st;
        for (int l = top; l >= 0; l--) {
            if (l == 0 && c->profiling) {
                pano_ctx::EvSlot& sl = c->ring[c->ev_cur];
                launch_blend_level(c->pyr, cs, l, s, sl.e[4], sl.e[5]);
                sl.recorded |= 3u << 4;
            } else {
                launch_blend_level(c->pyr, cs, l, s);
            }
        }
    }
    if (c->profiling) {
        if (c->ev_cur < 0 && (st = begin_slot(c)) != PANO_OK) return st;
        if ((st = record(c, 3, s)) != PANO_OK) return st;
        c->ev_cur = -1;  // frame closed
    }
    HIP_TRY(c, hipGetLastError());
    return PANO_OK;
}

pano_status pano_compose(pano_ctx* c, const uint8_t* const* d_frames, const size_t* strides, uint8_t* d_out,
                         size_t out_stride, void* stream) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    // weights first so that the profiled stages hold only per-frame work
    if ((st = ensure_weights(c, (hipStream_t)stream)) != PANO_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    const int n = c->plan.n;
    // Steady state: the ~12 launches of a frame are replayed as one hipGraph, captured once per set of
    // caller buffers (a capture needs a real stream; the legacy null stream and profiled runs launch directly).
    if (c->use_graph && !c->profiling && s != nullptr && d_frames && strides && d_out) {
        for (auto& g : c->graphs) {
            bool same = g.out == d_out && g.out_stride == out_stride && g.slot == c->cur_slot;
            for (int i = 0; i < n && same; i++) same = g.frames[i] == d_frames[i] && g.strides[i] == strides[i];
            if (same) {
                HIP_TRY(c, hipGraphLaunch(g.exec, s));
                c->graph_replays++;
                return PANO_OK;
            }
        }
        pano_ctx::GraphEntry g{};
        for (int i = 0; i < n; i++) {
            g.frames[i] = d_frames[i];
            g.strides[i] = strides[i];
        }
        g.out = d_out;
        g.out_stride = out_stride;
        g.slot = c->cur_slot;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            pano_status st1 = pano_feed_cameras(c, (1u << n) - 1u, d_frames, strides, stream);
            pano_status st2 = st1 == PANO_OK ? pano_blend(c, d_out, out_stride, stream) : st1;
            hipError_t e = hipStreamEndCapture(s, &g.graph);
            if (st2 == PANO_OK && e == hipSuccess && hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) == hipSuccess) {
                if (c->graphs.size() >= 8) drop_graphs(c);
                c->graphs.push_back(g);
                HIP_TRY(c, hipGraphLaunch(g.exec, s));
                c->graph_replays++;
                return PANO_OK;
            }
            if (e == hipSuccess && g.graph) (void)hipGraphDestroy(g.graph);
            (void)hipGetLastError();
            c->use_graph = false;  // capture is not available here: launch directly from now on
            if (st2 != PANO_OK) return st2;
        } else {
            (void)hipGetLastError();
            c->use_graph = false;
        }
    }
    if ((st = pano_feed_cameras(c, (1u << n) - 1u, d_frames, strides, stream)) != PANO_OK) return st;
    return pano_blend(c, d_out, out_stride, stream);
}

pano_status pano_debug_graphcut_dump(pano_ctx* c, const char* path) {
    if (!c) return PANO_EINVAL;
    c->gc_dump_path = path ? path : "";
    return PANO_OK;
}

pano_status pano_debug_graph_stats(const pano_ctx* c, int* graphs_held, uint64_t* replays) {
    if (!c) return PANO_EINVAL;
    if (graphs_held) *graphs_held = c->use_graph ? (int)c->graphs.size() : -1;
    if (replays) *replays = c->graph_replays;
    return PANO_OK;
}

pano_status pano_compose_pair(pano_ctx* a, pano_ctx* b, const uint8_t* const* fa, const size_t* sa, uint8_t* oa, size_t osa,
                              const uint8_t* const* fb, const size_t* sb, uint8_t* ob, size_t osb, void* stream) {
    pano_status st = check_compute(a);
    if (st != PANO_OK) return st;
    if ((st = check_compute(b)) != PANO_OK) return st;
    if (!fa || !sa || !oa || !fb || !sb || !ob) return PANO_EINVAL;
    const Plan &A = a->plan, &B = b->plan;
    bool same = a->device == b->device && A.bands == B.bands && A.n + B.n <= kCams && a->cv.small_base == b->cv.small_base &&
                (a->use_lut && b->use_lut) == (a->use_lut || b->use_lut);
    for (int l = 0; l < a->levels && same; l++) same = a->cv.fast[l] == b->cv.fast[l];
    if (!same) {  // different level structure: one after the other, same results
        if ((st = pano_compose(a, fa, sa, oa, osa, stream)) != PANO_OK) return st;
        return pano_compose(b, fb, sb, ob, osb, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    if ((st = ensure_weights(a, s)) != PANO_OK) return st;
    if ((st = ensure_weights(b, s)) != PANO_OK) return st;
    if (osa < (size_t)A.cut.w * 3 || osb < (size_t)B.cut.w * 3) return PANO_EINVAL;
    // K1: every camera of both contexts in one launch
    WarpParams wp{};
    int mw = 0, mh = 0;
    for (int i = 0; i < A.n; i++) {
        if (!fa[i] || sa[i] < (size_t)a->frame_w * 3) return fail(a, PANO_EINVAL, "frame pointer / stride");
        wp.cam[i] = make_warp_cam(a, i, fa[i], sa[i], false);
        mw = std::max(mw, A.tile[i].rect.w);
        mh = std::max(mh, A.tile[i].rect.h);
    }
    for (int i = 0; i < B.n; i++) {
        if (!fb[i] || sb[i] < (size_t)b->frame_w * 3) return fail(b, PANO_EINVAL, "frame pointer / stride");
        wp.cam[A.n + i] = make_warp_cam(b, i, fb[i], sb[i], false);
        mw = std::max(mw, B.tile[i].rect.w);
        mh = std::max(mh, B.tile[i].rect.h);
    }
    const bool prof = a->profiling;  // stage events of a pair go to the first context's ring
    if (prof) {
        if ((st = begin_slot(a)) != PANO_OK) return st;
        pano_ctx::EvSlot& sl = a->ring[a->ev_cur];
        launch_warp_tiles(wp, A.n + B.n, mw, mh, s, sl.e[0], sl.e[1]);
        sl.recorded |= 3u;
    } else {
        launch_warp_tiles(wp, A.n + B.n, mw, mh, s);
    }
    // K2: merged camera list
    PyrParams pp = a->pyr;
    for (int i = 0; i < B.n; i++) pp.cam[A.n + i] = b->pyr.cam[i];
    pp.ncam = A.n + B.n;
    const unsigned all = (1u << pp.ncam) - 1u;
    launch_pyr_chain(pp, all, pyr_levels(a), s);
    if (prof && (st = record(a, 2, s)) != PANO_OK) return st;
    // K3: both canvases per launch
    CanvasSet cs{};
    cs.n = 2;
    cs.c[0] = a->cv;
    cs.c[0].out = oa;
    cs.c[0].out_stride = (int)osa;
    cs.c[1] = b->cv;
    cs.c[1].out = ob;
    cs.c[1].out_stride = (int)osb;
    cs.c[1].cam_lo = A.n;
    if (A.bands < 0) {
        launch_no_blend(pp, cs, s);
    } else {
        int top = A.bands;
        if (cs.c[0].small_base > 0) {
            launch_blend_small(pp, cs, s);
            top = cs.c[0].small_base - 1;
        }
        for (int l = top; l >= 0; l--) {
            if (l == 0 && prof) {
                pano_ctx::EvSlot& sl = a->ring[a->ev_cur];
                launch_blend_level(pp, cs, l, s, sl.e[4], sl.e[5]);
                sl.recorded |= 3u << 4;
            } else {
                launch_blend_level(pp, cs, l, s);
            }
        }
    }
    if (prof) {
        if ((st = record(a, 3, s)) != PANO_OK) return st;
        a->ev_cur = -1;
    }
    HIP_TRY(a, hipGetLastError());
    return PANO_OK;
}

// is [p, p + bytes) page-locked memory known to HIP?  (plain malloc memory: an error or "unregistered", by ROCm version)
static bool is_pinned_host(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    auto one = [](const void* q) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, q) != hipSuccess) {
            (void)hipGetLastError();  // the failed query must not surface as a later launch error
            return false;
        }
        return a.type == hipMemoryTypeHost;
    };
    return one(p) && one(static_cast<const char*>(p) + bytes - 1);
}

// process(vector<Mat>&, Mat&) (ocvstitcher.hpp:1141): host frames in, host panorama out, synchronous.  Page-locked caller
// memory is DMA'd directly; pageable memory goes through the ctx's page-locked staging, copied by the pool's threads while the
// previous camera's DMA runs (pano_hostcopy.hpp).  Works in frame slot 0 (pano.h) whatever slot the caller has selected.
namespace {
hipError_t shared_copy_streams(int device, hipStream_t* h2d, hipStream_t* d2h);  // below, with the streaming slots
}
static pano_status compose_host_impl(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides, uint8_t* h_out,
                              size_t out_stride) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!h_frames || !strides || !h_out) return PANO_EINVAL;
    const Plan& P = c->plan;
    const size_t row_in = (size_t)c->frame_w * 3, row_out = (size_t)P.cut.w * 3;
    // staging pitches: multiples of 64 bytes (K1 wants strides % 16 == 0; a rectangular DMA runs at the link rate only on
    // 64-byte boundaries), equal to width * 3 for the usual frame widths: a caller stride of width*3 then needs no staging
    const size_t in_pitch = align_up(row_in, 64), out_pitch = align_up(row_out, 16);
    for (int i = 0; i < P.n; i++)
        if (!h_frames[i] || strides[i] < row_in) return PANO_EINVAL;
    if (out_stride < row_out) return PANO_EINVAL;
    if (!c->stage_in[0] || c->stage_in_pitch != in_pitch) {
        for (int i = 0; i < P.n; i++) {
            dfree(c->stage_in[i]);
            HIP_TRY(c, hipMalloc((void**)&c->stage_in[i], in_pitch * c->frame_h + 64));
        }
        c->stage_in_pitch = in_pitch;
    }
    // the output staging buffer follows the cut (pano_set_cut may grow it in either dimension)
    if (!c->stage_out || out_pitch * (size_t)P.cut.h > c->stage_out_bytes) {
        HIP_TRY(c, hipDeviceSynchronize());
        dfree(c->stage_out);
        if (c->pin_out) (void)hipHostFree(c->pin_out);
        c->pin_out = nullptr;
        c->stage_out_bytes = out_pitch * (size_t)P.cut.h;
        HIP_TRY(c, hipMalloc((void**)&c->stage_out, c->stage_out_bytes));
    }
    c->stage_out_pitch = out_pitch;
    if (!c->host_h2d[0]) {
        // the device's shared upload / download queues (see shared_copy_streams): both stitcher threads of a rig feed the same
        // two queues, so the link runs in both directions at once instead of the contexts' streams colliding on hardware queues
        hipError_t se = shared_copy_streams(c->device, &c->host_h2d[0], &c->host_h2d[1]);
        if (se != hipSuccess) HIP_TRY(c, se);
        for (auto& he : c->host_in_ready) HIP_TRY(c, hipEventCreateWithFlags(&he, hipEventDisableTiming));
    }
    hipStream_t up = c->host_h2d[0], down = c->host_h2d[1];
    const int prev_slot = c->cur_slot;
    if (c->nslots > 1 && prev_slot != 0) bind_slot(c, 0);
    struct RestoreSlot {  // every return below - the HIP_TRY ones included - leaves the caller's frame slot selected
        pano_ctx* c;
        int prev;
        ~RestoreSlot() {
            if (c->nslots > 1 && prev != 0 && c->cur_slot != prev) bind_slot(c, prev);
        }
    } restore_slot{c, prev_slot};
    hipStream_t s = c->own_stream;
    CopyPool& pool = CopyPool::instance();
    // PANO_HOST_TRACE=1: mean host-clock ms of the phases, printed by pano_destroy (diagnostic)
    static const bool trace = getenv("PANO_HOST_TRACE") && atoi(getenv("PANO_HOST_TRACE"));
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp[6] = {};
    if (trace) tp[0] = now();
    const uint8_t* frames[kMaxCams];
    size_t pitches[kMaxCams];
    // The reference calls process() of its two stitchers from two threads at the same moment (src/master.cpp:314-318).  Left
    // alone both would stage and upload in lockstep, halving each other's rate, and then download in lockstep.  One stitcher at
    // a time through the upload section staggers them: the second one's upload runs against the first one's kernels and
    // download (the link is full duplex)
    static std::mutex upload_turn;
    std::unique_lock<std::mutex> turn(upload_turn);
    // every pageable camera's rows go to the copy threads at once; each camera's DMA is queued the moment its rows are staged
    CopyPool::Latch staged[kMaxCams];
    struct WaitAll {  // an early return must not leave copy threads writing to latches of a dead stack frame
        CopyPool& pool;
        CopyPool::Latch* l;
        ~WaitAll() {
            for (int i = 0; i < kMaxCams; i++) pool.wait(l[i]);
        }
    } wait_all{pool, staged};
    const uint8_t* dma_src[kMaxCams];
    bool any_staged = false;
    // Only the bytes K1 reads cross the link (src_rect: with the masks of config 2, 70 % of a frame): byte columns [x0, x0 + w) of
    // rows [y0, y0 + rows), one rectangular DMA per camera - at the link rate, because x0, w and both pitches are multiples of 64
    // (misaligned rectangles run at a tenth of it, tools/pcie_2d.py).  The rest of the device frame keeps whatever it held.
    for (int i = 0; i < P.n; i++) {
        dma_src[i] = h_frames[i];
        const pano_ctx::SrcRect& r = c->src_rect[i];
        const size_t wcopy = std::min((size_t)r.w, row_in - (size_t)r.x0);  // the frame's own bytes of those columns
        // direct DMA from page-locked caller memory whose rows sit on the staging grid; anything else is staged
        if (strides[i] != in_pitch || ((size_t)h_frames[i] & 63) || !is_pinned_host(h_frames[i], strides[i] * (size_t)(c->frame_h - 1) + row_in)) {
            if (!c->pin_in[i] || c->pin_in_pitch != in_pitch) {
                HIP_TRY(c, hipStreamSynchronize(up));
                if (c->pin_in[i]) (void)hipHostFree(c->pin_in[i]);
                c->pin_in[i] = nullptr;
                HIP_TRY(c, hipHostMalloc((void**)&c->pin_in[i], in_pitch * c->frame_h, hipHostMallocDefault));
            }
            if (r.rows > 0)
                pool.submit(staged[i], c->pin_in[i] + (size_t)r.y0 * in_pitch + r.x0, in_pitch, h_frames[i] + (size_t)r.y0 * strides[i] + r.x0,
                            strides[i], wcopy, r.rows);
            dma_src[i] = c->pin_in[i];
            any_staged = true;
        }
    }
    for (int i = 0; i < P.n; i++) {
        pool.wait(staged[i]);
        const pano_ctx::SrcRect& r = c->src_rect[i];
        if (r.rows > 0) {
            const size_t off = (size_t)r.y0 * in_pitch + r.x0;
            if ((size_t)r.w >= in_pitch)  // whole rows: one linear transfer
                HIP_TRY(c, hipMemcpyAsync(c->stage_in[i] + off, dma_src[i] + off, in_pitch * (size_t)(r.rows - 1) + row_in, hipMemcpyHostToDevice, up));
            else
                HIP_TRY(c, hipMemcpy2DAsync(c->stage_in[i] + off, in_pitch, dma_src[i] + off, in_pitch, (size_t)r.w, (size_t)r.rows,
                                            hipMemcpyHostToDevice, up));
        }
        frames[i] = c->stage_in[i];
        pitches[i] = in_pitch;
    }
    c->pin_in_pitch = in_pitch;
    HIP_TRY(c, hipEventRecord(c->host_in_ready[0], up));
    HIP_TRY(c, hipStreamWaitEvent(s, c->host_in_ready[0], 0));
    if (!any_staged)  // nothing was staged, so queueing took no time: the turn lasts until the frames have crossed the link
        HIP_TRY(c, hipEventSynchronize(c->host_in_ready[0]));
    turn.unlock();
    if (trace) tp[1] = now();
    // a page-locked panorama buffer with TIGHT rows (a continuous cv::Mat: step == 3 * width): the blend writes rows at that
    // stride and the way back is one linear DMA.  Any other stride has bytes between the rows that are not the panorama's - a
    // ROI view's belong to its parent image - and a linear copy would overwrite them: those take the staged 2-D copy below
    const bool direct_out = out_stride == row_out && is_pinned_host(h_out, row_out * (size_t)P.cut.h);
    const size_t dev_pitch = direct_out ? out_stride : out_pitch;
    st = pano_compose(c, frames, pitches, c->stage_out, dev_pitch, s);
    if (st != PANO_OK) return st;
    if (trace) {
        tp[2] = now();
        HIP_TRY(c, hipStreamSynchronize(s));   // tracing only: separates the kernels from the copy back
        tp[3] = now();
    }
    auto account = [&]() {
        if (!trace) return;
        tp[5] = now();
        if (tp[4] == 0) tp[4] = tp[5];
        for (int k = 0; k < 5; k++) c->host_trace[k] += tp[k + 1] - tp[k];
        c->host_trace_n++;
    };
    const size_t out_bytes = dev_pitch * (size_t)(P.cut.h - 1) + row_out;
    // the way back runs on the device's download queue, behind the kernels of THIS context only
    HIP_TRY(c, hipEventRecord(c->host_in_ready[1], s));
    HIP_TRY(c, hipStreamWaitEvent(down, c->host_in_ready[1], 0));
    if (direct_out) {
        HIP_TRY(c, hipMemcpyAsync(h_out, c->stage_out, out_bytes, hipMemcpyDeviceToHost, down));
        HIP_TRY(c, hipEventRecord(c->host_in_ready[1], down));
        HIP_TRY(c, hipEventSynchronize(c->host_in_ready[1]));
        account();
        return PANO_OK;
    }
    if (!c->pin_out) HIP_TRY(c, hipHostMalloc((void**)&c->pin_out, c->stage_out_bytes, hipHostMallocDefault));
    // the panorama comes back in two halves so that the host copy of the first overlaps the DMA of the second
    const int h0 = P.cut.h / 2;
    const size_t b0 = out_pitch * (size_t)h0;
    if (h0 > 0) HIP_TRY(c, hipMemcpyAsync(c->pin_out, c->stage_out, b0, hipMemcpyDeviceToHost, down));
    HIP_TRY(c, hipEventRecord(c->host_in_ready[0], down));
    HIP_TRY(c, hipMemcpyAsync(c->pin_out + b0, c->stage_out + b0, out_bytes - b0, hipMemcpyDeviceToHost, down));
    HIP_TRY(c, hipEventRecord(c->host_in_ready[1], down));
    HIP_TRY(c, hipEventSynchronize(c->host_in_ready[0]));
    pool.copy2d(h_out, out_stride, c->pin_out, out_pitch, row_out, h0);
    HIP_TRY(c, hipEventSynchronize(c->host_in_ready[1]));
    if (trace) tp[4] = now();
    pool.copy2d(h_out + (size_t)h0 * out_stride, out_stride, c->pin_out + b0, out_pitch, row_out, P.cut.h - h0);
    account();
    return PANO_OK;
}

/* page-locked host memory for frames and panoramas (what cv::cuda::HostMem(PAGE_LOCKED) is to a CUDA OpenCV build):
 * pano_compose_host DMAs such buffers directly */
void* pano_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void pano_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

pano_status pano_stack_master(pano_ctx* c, const uint8_t* d_up, int up_w, int up_h, size_t up_stride, const uint8_t* d_down,
                              int down_w, int down_h, size_t down_stride, uint8_t* d_out, size_t out_stride, void* stream) {
    if (!c) return PANO_EINVAL;
    if (c->device < 0) return fail(c, PANO_ENODEVICE, "plan-only context");
    if (!d_up || !d_down || !d_out || up_w < 1 || up_h < 1 || down_w < 1 || down_h < 1 || up_stride < (size_t)up_w * 3 ||
        down_stride < (size_t)down_w * 3 || out_stride < (size_t)down_w * 3)
        return PANO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const bool resize_up = !(up_w == down_w && up_h == down_h);  // cv::resize to the same size is a copy
    const int rows = 2 * down_h;
    launch_stack(d_up, up_w, up_h, (int)up_stride, 0, resize_up, d_down, (int)down_stride, 0, d_out, down_w, down_h,
                 (int)out_stride, rows / 2 - 5, 10, (hipStream_t)stream);
    HIP_TRY(c, hipGetLastError());
    return PANO_OK;
}

pano_status pano_stack_finalcut(pano_ctx* c, const uint8_t* d_up, int up_w, int up_h, size_t up_stride, const uint8_t* d_down,
                                int down_w, int down_h, size_t down_stride, int finalcut, uint8_t* d_out, size_t out_stride,
                                void* stream) {
    if (!c) return PANO_EINVAL;
    if (c->device < 0) return fail(c, PANO_ENODEVICE, "plan-only context");
    const int width = std::min(up_w, down_w), height = std::min(up_h, down_h) - 2 * finalcut;
    if (!d_up || !d_down || !d_out || finalcut < 0 || width < 1 || height < 1 || up_stride < (size_t)up_w * 3 ||
        down_stride < (size_t)down_w * 3 || out_stride < (size_t)width * 3)
        return PANO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    launch_stack(d_up, width, height, (int)up_stride, finalcut, false, d_down, (int)down_stride, finalcut, d_out, width, height,
                 (int)out_stride, height - 2, 4, (hipStream_t)stream);
    HIP_TRY(c, hipGetLastError());
    return PANO_OK;
}

// master.cpp:321-326 on host cv::Mat-style buffers: the two half panoramas go up, pano_stack_master runs, the stacked image
// comes back (synchronous).  finalcut < 0: master.cpp's resize + vconcat + 10-row bar; >= 0: panocamimpl.cpp:354-360's crop
static pano_status stack_host_impl(pano_ctx* c, const uint8_t* h_up, int up_w, int up_h, size_t up_stride, const uint8_t* h_down,
                                   int down_w, int down_h, size_t down_stride, int finalcut, uint8_t* h_out, size_t out_stride) {
    if (!c) return PANO_EINVAL;
    if (c->device < 0) return fail(c, PANO_ENODEVICE, "plan-only context");
    if (!h_up || !h_down || !h_out || up_w < 1 || up_h < 1 || down_w < 1 || down_h < 1 || up_stride < (size_t)up_w * 3 ||
        down_stride < (size_t)down_w * 3)
        return PANO_EINVAL;
    const int ow = finalcut < 0 ? down_w : std::min(up_w, down_w);
    const int oh = finalcut < 0 ? 2 * down_h : 2 * (std::min(up_h, down_h) - 2 * finalcut);
    if (ow < 1 || oh < 2 || out_stride < (size_t)ow * 3) return PANO_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t pu = align_up((size_t)up_w * 3, 16), pd = align_up((size_t)down_w * 3, 16), po = align_up((size_t)ow * 3, 16);
    const size_t bu = align_up(pu * up_h, 256), bd = align_up(pd * down_h, 256), bo = po * oh;
    if (bu + bd + bo > c->stack_bytes) {
        dfree(c->stack_buf);
        c->stack_bytes = 0;
        HIP_TRY(c, hipMalloc((void**)&c->stack_buf, bu + bd + bo));
        c->stack_bytes = bu + bd + bo;
    }
    uint8_t *d_up = c->stack_buf, *d_down = d_up + bu, *d_out = d_down + bd;
    hipStream_t s = c->own_stream;
    HIP_TRY(c, hipMemcpy2DAsync(d_up, pu, h_up, up_stride, (size_t)up_w * 3, up_h, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpy2DAsync(d_down, pd, h_down, down_stride, (size_t)down_w * 3, down_h, hipMemcpyHostToDevice, s));
    pano_status st = finalcut < 0 ? pano_stack_master(c, d_up, up_w, up_h, pu, d_down, down_w, down_h, pd, d_out, po, s)
                                  : pano_stack_finalcut(c, d_up, up_w, up_h, pu, d_down, down_w, down_h, pd, finalcut, d_out, po, s);
    if (st != PANO_OK) return st;
    HIP_TRY(c, hipMemcpy2DAsync(h_out, out_stride, d_out, po, (size_t)ow * 3, oh, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return PANO_OK;
}
pano_status pano_stack_master_host(pano_ctx* c, const uint8_t* h_up, int up_w, int up_h, size_t up_stride, const uint8_t* h_down,
                                   int down_w, int down_h, size_t down_stride, uint8_t* h_out, size_t out_stride) {
    return stack_host_impl(c, h_up, up_w, up_h, up_stride, h_down, down_w, down_h, down_stride, -1, h_out, out_stride);
}
pano_status pano_stack_finalcut_host(pano_ctx* c, const uint8_t* h_up, int up_w, int up_h, size_t up_stride, const uint8_t* h_down,
                                     int down_w, int down_h, size_t down_stride, int finalcut, uint8_t* h_out, size_t out_stride) {
    if (finalcut < 0) return PANO_EINVAL;
    return stack_host_impl(c, h_up, up_w, up_h, up_stride, h_down, down_w, down_h, down_stride, finalcut, h_out, out_stride);
}

namespace {
// process-wide copy streams, one pair per device, created on first use and never destroyed
hipError_t shared_copy_streams(int device, hipStream_t* h2d, hipStream_t* d2h) {
    static std::mutex m;
    static hipStream_t up[64] = {}, down[64] = {};
    std::lock_guard<std::mutex> g(m);
    if (device < 0 || device >= 64) return hipErrorInvalidDevice;
    if (!up[device] || !down[device]) {
        // both or neither: a half-made pair would hand out a null download stream - the legacy default stream, on which
        // every copy back would serialise against everything, silently
        hipStream_t u = nullptr, d = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&u, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        e = hipStreamCreateWithFlags(&d, hipStreamNonBlocking);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(u);
            return e;
        }
        up[device] = u;
        down[device] = d;
    }
    *h2d = up[device];
    *d2h = down[device];
    return hipSuccess;
}
pano_status ensure_slots(pano_ctx* c) {
    if (c->slots_ready) return PANO_OK;
    hipError_t st_ = hipSuccess;
    const Plan& P = c->plan;
    c->slot_in_pitch = align_up((size_t)c->frame_w * 3, 256);
    c->slot_out_pitch = align_up((size_t)P.pano.w * 3, 256);  // room for any later cut
    for (auto& sl : c->slots) {
        for (int i = 0; i < P.n; i++) {
            HIP_TRY(c, hipHostMalloc((void**)&sl.h_in[i], c->slot_in_pitch * c->frame_h, hipHostMallocDefault));
            HIP_TRY(c, hipMalloc((void**)&sl.d_in[i], c->slot_in_pitch * c->frame_h + 64));
        }
        HIP_TRY(c, hipHostMalloc((void**)&sl.h_out, c->slot_out_pitch * P.pano.h, hipHostMallocDefault));
        HIP_TRY(c, hipMalloc((void**)&sl.d_out, c->slot_out_pitch * P.pano.h));
        // ONE upload and ONE download queue per device, shared by every slot of every context: uploads all cross the same link
        // anyway, and the runtime multiplexes streams onto a few hardware queues (4 by default) - with a pair of copy streams
        // per slot and context (8 + 2 compute streams for the two stitchers of a rig) uploads, downloads and kernels of
        // unrelated slots landed on the same hardware queue and the link ran in one direction at a time
        if ((st_ = shared_copy_streams(c->device, &sl.h2d, &sl.d2h)) != hipSuccess) HIP_TRY(c, st_);
        HIP_TRY(c, hipEventCreateWithFlags(&sl.in_ready, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&sl.composed, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&sl.out_ready, hipEventDisableTiming));
    }
    c->slots_ready = true;
    return PANO_OK;
}
}  // namespace

pano_status pano_stream_input(pano_ctx* c, int slot, int cam, uint8_t** h_ptr, size_t* stride) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (slot < 0 || slot >= PANO_STREAM_SLOTS || cam < 0 || cam >= c->plan.n || !h_ptr || !stride) return PANO_EINVAL;
    if ((st = ensure_slots(c)) != PANO_OK) return st;
    *h_ptr = c->slots[slot].h_in[cam];
    *stride = c->slot_in_pitch;
    return PANO_OK;
}

pano_status pano_stream_output(pano_ctx* c, int slot, uint8_t** h_ptr, size_t* stride) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (slot < 0 || slot >= PANO_STREAM_SLOTS || !h_ptr || !stride) return PANO_EINVAL;
    if ((st = ensure_slots(c)) != PANO_OK) return st;
    *h_ptr = c->slots[slot].h_out;
    *stride = c->slot_out_pitch;
    return PANO_OK;
}

pano_status pano_stream_submit(pano_ctx* c, int slot) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (slot < 0 || slot >= PANO_STREAM_SLOTS) return PANO_EINVAL;
    if ((st = ensure_slots(c)) != PANO_OK) return st;
    pano_ctx::StreamSlot& sl = c->slots[slot];
    if (sl.busy) return fail(c, PANO_ESTATE, "slot still in flight: pano_stream_wait it first");
    const Plan& P = c->plan;
    const uint8_t* frames[kMaxCams];
    size_t pitches[kMaxCams];
    for (int i = 0; i < P.n; i++) {
        // only the bytes K1 reads with the present masks cross the link: one aligned rectangular DMA (see pano_compose_host)
        const pano_ctx::SrcRect& r = c->src_rect[i];
        if (r.rows > 0) {
            const size_t off = (size_t)r.y0 * c->slot_in_pitch + r.x0;
            if ((size_t)r.w >= c->slot_in_pitch || (size_t)r.w >= align_up((size_t)c->frame_w * 3, 64))
                HIP_TRY(c, hipMemcpyAsync(sl.d_in[i] + (size_t)r.y0 * c->slot_in_pitch, sl.h_in[i] + (size_t)r.y0 * c->slot_in_pitch,
                                          c->slot_in_pitch * (size_t)r.rows, hipMemcpyHostToDevice, sl.h2d));
            else
                HIP_TRY(c, hipMemcpy2DAsync(sl.d_in[i] + off, c->slot_in_pitch, sl.h_in[i] + off, c->slot_in_pitch, (size_t)r.w, (size_t)r.rows,
                                            hipMemcpyHostToDevice, sl.h2d));
        }
        frames[i] = sl.d_in[i];
        pitches[i] = c->slot_in_pitch;
    }
    HIP_TRY(c, hipEventRecord(sl.in_ready, sl.h2d));
    HIP_TRY(c, hipStreamWaitEvent(c->own_stream, sl.in_ready, 0));
    const int prev_slot = c->cur_slot;   // the streaming form works in frame slot 0 (pano.h)
    if (c->nslots > 1 && prev_slot != 0) bind_slot(c, 0);
    st = pano_compose(c, frames, pitches, sl.d_out, c->slot_out_pitch, c->own_stream);
    if (c->nslots > 1 && prev_slot != 0) bind_slot(c, prev_slot);
    if (st != PANO_OK) return st;
    HIP_TRY(c, hipEventRecord(sl.composed, c->own_stream));
    HIP_TRY(c, hipStreamWaitEvent(sl.d2h, sl.composed, 0));
    HIP_TRY(c, hipMemcpyAsync(sl.h_out, sl.d_out, c->slot_out_pitch * P.cut.h, hipMemcpyDeviceToHost, sl.d2h));
    HIP_TRY(c, hipEventRecord(sl.out_ready, sl.d2h));
    sl.busy = true;
    return PANO_OK;
}

pano_status pano_stream_wait(pano_ctx* c, int slot) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (slot < 0 || slot >= PANO_STREAM_SLOTS || !c->slots_ready) return PANO_EINVAL;
    pano_ctx::StreamSlot& sl = c->slots[slot];
    if (!sl.busy) return fail(c, PANO_ESTATE, "slot was not submitted");
    HIP_TRY(c, hipEventSynchronize(sl.out_ready));
    sl.busy = false;
    return PANO_OK;
}

pano_status pano_get_pyramid_slots(pano_ctx* c, void** d_base, size_t* slot_bytes) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!d_base || !slot_bytes) return PANO_EINVAL;
    *d_base = c->pyr_base;
    *slot_bytes = c->slot_bytes;
    return PANO_OK;
}

/* ---- the camera-sharded exchange over RCCL (one process per GPU) ------------------------------------------- */
#define RCCL_TRY(ctx, expr)                                                                      \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            if (ctx) (ctx)->err = std::string(#expr) + ": " + Rccl::get().GetErrorString(r_);   \
            return PANO_EHIP;                                                                    \
        }                                                                                        \
    } while (0)

// the sharded path for callers whose frames are in host memory (a capture card per GPU host process): upload + feed, and
// blend + download, on the ctx's own stream
static pano_status feed_cameras_host_impl(pano_ctx* c, unsigned cam_bits, const uint8_t* const* h_frames, const size_t* strides) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!h_frames || !strides) return PANO_EINVAL;
    const Plan& P = c->plan;
    cam_bits &= (1u << P.n) - 1u;
    const size_t row_in = (size_t)c->frame_w * 3, in_pitch = align_up(row_in, 64);  // the staging grid of pano_compose_host
    if (!c->stage_in[0] || c->stage_in_pitch != in_pitch) {
        HIP_TRY(c, hipDeviceSynchronize());
        for (int i = 0; i < P.n; i++) {
            dfree(c->stage_in[i]);
            HIP_TRY(c, hipMalloc((void**)&c->stage_in[i], in_pitch * c->frame_h + 64));
        }
        c->stage_in_pitch = in_pitch;
    }
    const uint8_t* frames[kMaxCams] = {};
    size_t pitches[kMaxCams] = {};
    for (int i = 0; i < P.n; i++) {
        if (!((cam_bits >> i) & 1u)) continue;
        if (!h_frames[i] || strides[i] < row_in) return PANO_EINVAL;
        const pano_ctx::SrcRect& r = c->src_rect[i];  // only the bytes K1 reads (see pano_compose_host)
        if (r.rows > 0)
            HIP_TRY(c, hipMemcpy2DAsync(c->stage_in[i] + (size_t)r.y0 * in_pitch + r.x0, in_pitch, h_frames[i] + (size_t)r.y0 * strides[i] + r.x0,
                                        strides[i], std::min((size_t)r.w, row_in - (size_t)r.x0), (size_t)r.rows, hipMemcpyHostToDevice, c->own_stream));
        frames[i] = c->stage_in[i];
        pitches[i] = in_pitch;
    }
    return pano_feed_cameras(c, cam_bits, frames, pitches, c->own_stream);
}
pano_status pano_feed_cameras_host(pano_ctx* c, unsigned cam_bits, const uint8_t* const* h_frames, const size_t* strides) {
    return guarded(c, [&]() { return feed_cameras_host_impl(c, cam_bits, h_frames, strides); });
}
pano_status pano_blend_host(pano_ctx* c, uint8_t* h_out, size_t out_stride) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    const Plan& P = c->plan;
    const size_t row_out = (size_t)P.cut.w * 3, out_pitch = align_up(row_out, 16);
    if (!h_out || out_stride < row_out) return PANO_EINVAL;
    if (!c->stage_out || out_pitch * (size_t)P.cut.h > c->stage_out_bytes) {
        HIP_TRY(c, hipDeviceSynchronize());
        dfree(c->stage_out);
        if (c->pin_out) (void)hipHostFree(c->pin_out);
        c->pin_out = nullptr;
        c->stage_out_bytes = out_pitch * (size_t)P.cut.h;
        HIP_TRY(c, hipMalloc((void**)&c->stage_out, c->stage_out_bytes));
    }
    if ((st = pano_blend(c, c->stage_out, out_pitch, c->own_stream)) != PANO_OK) return st;
    HIP_TRY(c, hipMemcpy2DAsync(h_out, out_stride, c->stage_out, out_pitch, row_out, P.cut.h, hipMemcpyDeviceToHost, c->own_stream));
    HIP_TRY(c, hipStreamSynchronize(c->own_stream));
    return PANO_OK;
}

pano_status pano_rccl_unique_id(char id[PANO_RCCL_ID_BYTES]) {
    static_assert(PANO_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return PANO_ENODEVICE;
    ncclUniqueId u;
    if (R.GetUniqueId(&u) != ncclSuccess) return PANO_EHIP;
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return PANO_OK;
}

pano_status pano_rccl_comm_create(pano_ctx* c, const char id[PANO_RCCL_ID_BYTES], int world, int rank, void** comm) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!id || !comm || world < 1 || rank < 0 || rank >= world) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return fail(c, PANO_ENODEVICE, R.error.c_str());
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t nc = nullptr;
    RCCL_TRY(c, R.CommInitRank(&nc, world, u, rank));
    *comm = nc;
    return PANO_OK;
}

pano_status pano_rccl_comm_destroy(void* comm) {
    if (!comm) return PANO_OK;
    Rccl& R = Rccl::get();
    if (!R.ok) return PANO_ENODEVICE;
    return R.CommDestroy((ncclComm_t)comm) == ncclSuccess ? PANO_OK : PANO_EHIP;
}

pano_status pano_rccl_comm_count(void* comm, int* ranks) {
    if (!comm || !ranks) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return PANO_ENODEVICE;
    return R.CommCount((ncclComm_t)comm, ranks) == ncclSuccess ? PANO_OK : PANO_EHIP;
}

const char* pano_rccl_library(void) {
    Rccl& R = Rccl::get();
    return R.ok ? R.path.c_str() : "";
}

pano_status pano_gather_slots(pano_ctx* c, void* comm, int rank, int root, const int* owner_rank, void* stream) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!comm || !owner_rank || rank < 0 || root < 0) return PANO_EINVAL;
    Rccl& R = Rccl::get();
    if (!R.ok) return fail(c, PANO_ENODEVICE, R.error.c_str());
    const int n = c->plan.n;
    if (!stream) stream = c->own_stream;  // callers without HIP types: the stream pano_feed_cameras_host / pano_blend_host use
    // consecutive slots with the same peer travel as one message: a rank's cameras are a contiguous byte range
    RCCL_TRY(c, R.GroupStart());
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && owner_rank[j] == owner_rank[i]) j++;
        const int owner = owner_rank[i];
        char* base = c->pyr_base + (size_t)i * c->slot_bytes;
        const size_t bytes = (size_t)(j - i) * c->slot_bytes;
        ncclResult_t r = ncclSuccess;
        if (owner != root) {
            if (rank == root) r = R.Recv(base, bytes, ncclUint8, owner, (ncclComm_t)comm, (hipStream_t)stream);
            else if (rank == owner) r = R.Send(base, bytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream);
        }
        if (r != ncclSuccess) {
            (void)R.GroupEnd();
            c->err = std::string("ncclSend / ncclRecv: ") + R.GetErrorString(r);
            return PANO_EHIP;
        }
        i = j;
    }
    RCCL_TRY(c, R.GroupEnd());
    return PANO_OK;
}

pano_status pano_set_profiling(pano_ctx* c, int enabled) {
    if (!c) return PANO_EINVAL;
    c->profiling = enabled != 0;
    return PANO_OK;
}

pano_status pano_get_stage_ms(pano_ctx* c, float ms[PANO_NUM_STAGES]) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!ms || !c->profiling) return fail(c, PANO_ESTATE, "profiling is off");
    while (c->ev_count > 0)
        if ((st = harvest_oldest(c)) != PANO_OK) return st;
    for (int k = 0; k < PANO_NUM_STAGES; k++) ms[k] = c->last_ms[k];
    return PANO_OK;
}

pano_status pano_get_stage_stats(pano_ctx* c, double total_ms[PANO_NUM_STAGES], uint64_t launches[PANO_NUM_STAGES],
                                 int reset) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!total_ms || !launches) return PANO_EINVAL;
    while (c->ev_count > 0)
        if ((st = harvest_oldest(c)) != PANO_OK) return st;
    for (int k = 0; k < PANO_NUM_STAGES; k++) {
        total_ms[k] = c->acc_ms[k];
        launches[k] = c->acc_n[k];
        if (reset) {
            c->acc_ms[k] = 0;
            c->acc_n[k] = 0;
        }
    }
    return PANO_OK;
}

pano_status pano_get_warp_bytes(const pano_ctx* c, uint64_t* src_bytes, uint64_t* dst_bytes) {
    if (!c || !c->prepared || !src_bytes || !dst_bytes) return PANO_EINVAL;
    uint64_t s = 0, d = 0;
    for (int i = 0; i < c->plan.n; i++) {
        // K1 produces the 64 x 16 blocks that overlap the live rect of level 0 (pano_get_live_rect): those tile bytes
        // are written once (planar u8), and the same share of the frame is what they sample
        const int tw = c->plan.tile[i].rect.w, th = c->plan.tile[i].rect.h;
        const int* L = c->live[i][0];
        uint64_t live_px = 0, dead_px = 0;
        if (L[2] >= L[0] && L[3] >= L[1]) {
            const int x0 = (L[0] >> 6) << 6, x1 = std::min(((L[2] >> 6) + 1) << 6, tw);
            const int y0 = (L[1] >> 4) << 4, y1 = std::min(((L[3] >> 4) + 1) << 4, th);
            live_px = (uint64_t)(x1 - x0) * (uint64_t)(y1 - y0);
            // the dead middle of a +-pi straddler: its block columns are not produced, and its columns are no part of the
            // tile area the frame maps onto
            const int* G = c->gap[i][0];
            const int g0 = (G[0] + 63) >> 6, g1 = ((G[1] + 1) >> 6) - 1;
            if (G[1] >= G[0] && g1 >= g0) {
                live_px -= (uint64_t)(g1 - g0 + 1) * 64 * (uint64_t)(y1 - y0);
                dead_px = (uint64_t)(G[1] - G[0] + 1) * (uint64_t)th;
            }
        }
        d += live_px * 3;
        s += (uint64_t)((double)c->frame_w * c->frame_h * 3 * std::min(1.0, (double)live_px / ((double)tw * th - (double)dead_px)));
    }
    *src_bytes = s;
    *dst_bytes = d;
    return PANO_OK;
}

pano_status pano_get_warp_table_stats(const pano_ctx* c, uint64_t* table_bytes, uint64_t* blocks, uint64_t* blocks_checked) {
    if (!c || !c->prepared || !table_bytes || !blocks || !blocks_checked) return PANO_EINVAL;
    uint64_t g = 0, nb = 0, nf = 0;
    if (c->use_lut)
        for (int i = 0; i < c->plan.n; i++) {
            // unflagged workgroups read 2 bytes per pixel (packed groups), flagged ones 4 (dense table)
            const uint64_t px = (uint64_t)c->plan.tile[i].rect.w * c->plan.tile[i].rect.h;
            const uint64_t fpx = std::min<uint64_t>(px, (uint64_t)c->k1_flagged[i] * 64 * 16);
            g += (px - fpx) * 2 + fpx * 4;
            nb += (uint64_t)c->k1_blocks[i];
            nf += (uint64_t)c->k1_flagged[i];
        }
    *table_bytes = g;
    *blocks = nb;
    *blocks_checked = nf;
    return PANO_OK;
}

pano_status pano_get_source_rect(const pano_ctx* c, int i, int rect[4]) {
    if (!c || !c->prepared || !rect || i < 0 || i >= c->plan.n) return PANO_EINVAL;
    const pano_ctx::SrcRect& r = c->src_rect[i];
    rect[0] = r.x0; rect[1] = r.y0; rect[2] = r.w; rect[3] = r.rows;
    return PANO_OK;
}

pano_status pano_get_live_rect(const pano_ctx* c, int i, int level, int rect[4]) {
    if (!c || !c->prepared || !rect || i < 0 || i >= c->plan.n || level < 0 || level >= c->levels) return PANO_EINVAL;
    const int* L = c->live[i][level];
    rect[0] = L[0]; rect[1] = L[1];
    rect[2] = L[2] >= L[0] ? L[2] - L[0] + 1 : 0;
    rect[3] = L[3] >= L[1] ? L[3] - L[1] + 1 : 0;
    return PANO_OK;
}

pano_status pano_get_live_gap(const pano_ctx* c, int i, int level, int gap[2]) {
    if (!c || !gap) return PANO_EINVAL;
    if (!c->prepared) return PANO_ESTATE;
    if (i < 0 || i >= c->plan.n || level < 0 || level >= c->levels) return PANO_EINVAL;
    const int* G = c->gap[i][level];
    gap[0] = G[1] >= G[0] ? G[0] : 0;
    gap[1] = G[1] >= G[0] ? G[1] - G[0] + 1 : 0;
    return PANO_OK;
}

pano_status pano_debug_get_level(pano_ctx* c, int i, int level, int16_t* h_dst, int* w, int* h) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (i < 0 || i >= c->plan.n || level < 0 || level >= c->levels || !w || !h) return PANO_EINVAL;
    *w = c->plan.tile[i].rect.w >> level;
    *h = c->plan.tile[i].rect.h >> level;
    if (!h_dst) return PANO_OK;
    HIP_TRY(c, hipDeviceSynchronize());
    // with the small levels fused, camera levels above small_base only ever exist in LDS: build them for the inspection
    for (int l = pyr_levels(c); l < level; l++) launch_pyr_down(c->pyr, 1u << i, l, nullptr);
    HIP_TRY(c, hipDeviceSynchronize());
    std::vector<uint8_t> tmp((size_t)*w * *h);
    for (int pl = 0; pl < 3; pl++) {  // planar u8 on the device -> CV_16SC3 for the caller
        HIP_TRY(c, hipMemcpy2D(tmp.data(), (size_t)*w, c->pyr.cam[i].lvl[level] + (size_t)pl * c->lvl_plane[i][level],
                               (size_t)c->lvl_pitch[i][level], (size_t)*w, (size_t)*h, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < tmp.size(); k++) h_dst[k * 3 + pl] = tmp[k];
    }
    return PANO_OK;
}

pano_status pano_debug_get_weights(pano_ctx* c, int i, int level, float* h_dst, int* w, int* h) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (i < 0 || i >= c->plan.n || level < 0 || level >= c->levels || !w || !h || c->plan.bands < 0) return PANO_EINVAL;
    *w = c->plan.tile[i].rect.w >> level;
    *h = c->plan.tile[i].rect.h >> level;
    if (!h_dst) return PANO_OK;
    if ((st = ensure_weights(c, c->own_stream)) != PANO_OK) return st;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy2D(h_dst, (size_t)*w * 4, c->wgt[i][level], (size_t)c->wpitch[i][level] * 4, (size_t)*w * 4,
                           (size_t)*h, hipMemcpyDeviceToHost));
    return PANO_OK;
}

pano_status pano_debug_get_canvas_weights(pano_ctx* c, int level, float* h_dst, int* w, int* h) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (level < 0 || level >= c->levels || !w || !h || c->plan.bands < 0) return PANO_EINVAL;
    *w = c->plan.canvas.w >> level;
    *h = c->plan.canvas.h >> level;
    if (!h_dst) return PANO_OK;
    if ((st = ensure_weights(c, c->own_stream)) != PANO_OK) return st;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(h_dst, c->wsum[level], (size_t)*w * *h * 4, hipMemcpyDeviceToHost));
    return PANO_OK;
}

pano_status pano_debug_get_canvas(pano_ctx* c, int level, int16_t* h_dst, int* w, int* h) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    // level 0 is written straight to the 8U panorama and never materialised; levels above the fused base
    // (CanvasParams::small_base) hold the normalised Laplacian, not the collapsed image
    if (level < 1 || level >= c->levels || !w || !h || c->plan.bands < 0) return PANO_EINVAL;
    *w = c->plan.canvas.w >> level;
    *h = c->plan.canvas.h >> level;
    if (!h_dst) return PANO_OK;
    HIP_TRY(c, hipDeviceSynchronize());
    std::vector<int16_t> tmp((size_t)*w * *h);
    for (int pl = 0; pl < 3; pl++) {
        HIP_TRY(c, hipMemcpy2D(tmp.data(), (size_t)*w * 2, c->canvas[level] + (size_t)pl * c->cv.cplane[level],
                               (size_t)c->cv.cpitch[level] * 2, (size_t)*w * 2, (size_t)*h, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < tmp.size(); k++) h_dst[k * 3 + pl] = tmp[k];
    }
    return PANO_OK;
}

// ---- entry points that allocate on the host: nothing may unwind through the C boundary ("no exceptions across
// the ABI") - std::bad_alloc and friends become a status
pano_status pano_create(const pano_config* cfg, pano_ctx** out) {
    return guarded(nullptr, [&]() { return create_impl(cfg, out); });
}

pano_status pano_set_cameras_from_list(pano_ctx* c, const char* list) {
    return guarded(c, [&]() { return set_cameras_from_list_impl(c, list); });
}

pano_status pano_load_camera_file(pano_ctx* c, const char* path) {
    return guarded(c, [&]() { return load_camera_file_impl(c, path); });
}
pano_status pano_get_camera(const pano_ctx* c, int i, float K[9], float R[9], float* scale) {
    if (!c || i < 0 || i >= c->cfg.num_images) return PANO_EINVAL;
    if (!c->have_cam[i]) return PANO_ESTATE;
    if (K) std::memcpy(K, c->K[i], 9 * sizeof(float));
    if (R) std::memcpy(R, c->R[i], 9 * sizeof(float));
    if (scale) *scale = c->scale;
    return PANO_OK;
}

pano_status pano_save_camera_file(pano_ctx* c, const char* path) {
    return guarded(c, [&]() { return save_camera_file_impl(c, path); });
}

pano_status pano_prepare(pano_ctx* c) {
    // a failure half way (an allocation, an upload) must not leave a ctx that claims to be prepared with null buffers
    // a second pano_prepare on a prepared context is a harmless error (PANO_ESTATE): it must not tear down a context that may
    // have frames in flight and a mask refresh running - only a prepare that failed half way is cleaned up
    const bool was_prepared = c && c->prepared;
    pano_status st = guarded(c, [&]() { return prepare_impl(c); });
    if (st != PANO_OK && c && !was_prepared && c->prepared) {
        std::string why = c->err;
        if (c->device >= 0) {
            drop_job(c);
            (void)hipDeviceSynchronize();
            free_device(c);
        }
        c->prepared = false;
        c->err = why;
    }
    return st;
}

pano_status pano_set_frame_slots(pano_ctx* c, int n) {
    return guarded(c, [&]() { return set_frame_slots_impl(c, n); });
}

// Streams for frames in flight that really run side by side.  The HIP runtime multiplexes a process's streams onto a few hardware
// queues (GPU_MAX_HW_QUEUES, default 4), by an order the caller does not control; two flight streams on one queue run their frames
// one after the other (15 % fewer panoramas/s on config 2, docs/EXPERIMENTS.md).  So: candidates are created one by one and PROBED
// against the streams already taken - a 150 us one-wave spin on each of the pair; on a shared queue the two take 300 us - and kept
// when they overlap with all of them.  Rejected candidates stay alive until the search is over (the runtime gives a new stream the
// least-used queue: destroying a reject would hand its queue to the next candidate again).
static pano_status frame_streams_impl(pano_ctx* c, int n, void** streams, int* distinct) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (n < 1 || n > PANO_MAX_FRAME_SLOTS || !streams) return PANO_EINVAL;
    if ((int)c->flight_streams.size() < n) {
        HIP_TRY(c, hipDeviceSynchronize());
        for (hipStream_t fs : c->flight_streams) (void)hipStreamDestroy(fs);
        c->flight_streams.clear();
        constexpr double kSpinUs = 150.0;
        auto pair_us = [&](hipStream_t a, hipStream_t b) {
            double best = 1e30;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipStreamSynchronize(a);
                (void)hipStreamSynchronize(b);
                const auto t0 = std::chrono::steady_clock::now();
                launch_spin((unsigned long long)(kSpinUs * 100.0), a);
                launch_spin((unsigned long long)(kSpinUs * 100.0), b);
                (void)hipStreamSynchronize(a);
                (void)hipStreamSynchronize(b);
                best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            return best;
        };
        std::vector<hipStream_t> rejects;
        for (int tries = 0; tries < 6 * n + 8 && (int)c->flight_streams.size() < n; tries++) {
            hipStream_t cand = nullptr;
            if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            launch_spin(100ull, cand);  // first use: the queue is bound (and the kernel's code loaded) before anything is timed
            (void)hipStreamSynchronize(cand);
            bool alone = true;
            for (hipStream_t taken : c->flight_streams)
                if (pair_us(taken, cand) > 1.6 * kSpinUs) { alone = false; break; }
            (alone ? c->flight_streams : rejects).push_back(cand);
        }
        c->flight_distinct = (int)c->flight_streams.size();
        // fewer hardware queues than streams asked for: the rest share (they work, they do not overlap)
        while ((int)c->flight_streams.size() < n && !rejects.empty()) { c->flight_streams.push_back(rejects.back()); rejects.pop_back(); }
        for (hipStream_t r : rejects) (void)hipStreamDestroy(r);
        (void)hipGetLastError();
        if ((int)c->flight_streams.size() < n) return fail(c, PANO_EHIP, "hipStreamCreate (flight streams)");
    }
    for (int i = 0; i < n; i++) streams[i] = (void*)c->flight_streams[i];
    if (distinct) *distinct = std::min(c->flight_distinct, n);
    return PANO_OK;
}
pano_status pano_frame_streams(pano_ctx* c, int n, void** streams, int* distinct) {
    return guarded(c, [&]() { return frame_streams_impl(c, n, streams, distinct); });
}

pano_status pano_set_mask(pano_ctx* c, int i, const uint8_t* h_mask, int w, int h, size_t stride) {
    return guarded(c, [&]() { return set_mask_impl(c, i, h_mask, w, h, stride); });
}

pano_status pano_build_masks_voronoi(pano_ctx* c) {
    return guarded(c, [&]() { return build_masks_voronoi_impl(c); });
}

pano_status pano_refresh_masks_begin(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides) {
    return guarded(c, [&] { return refresh_begin_impl(c, h_frames, strides); });
}
pano_status pano_refresh_masks_poll(pano_ctx* c, int* done) {
    return guarded(c, [&] { return refresh_poll_impl(c, done, false); });
}
pano_status pano_refresh_masks_wait(pano_ctx* c) {
    return guarded(c, [&] { return refresh_poll_impl(c, nullptr, true); });
}
pano_status pano_build_masks_graphcut(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides) {
    return guarded(c, [&]() { return build_masks_graphcut_impl(c, h_frames, strides); });
}

pano_status pano_estimate_gains(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides, int block_w, int block_h) {
    return guarded(c, [&]() { return estimate_gains_impl(c, h_frames, strides, block_w, block_h); });
}

pano_status pano_compose_host(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides, uint8_t* h_out, size_t out_stride) {
    return guarded(c, [&]() { return compose_host_impl(c, h_frames, strides, h_out, out_stride); });
}

pano_status pano_set_undistort(pano_ctx* c, int cam, const pano_undistort* u) {
    return guarded(c, [&]() { return set_undistort_impl(c, cam, u); });
}

}  // extern "C"
