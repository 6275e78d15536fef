// pano_ctx.hpp - INTERNAL to libpano_hip.so: the context behind include/pano.h's opaque pano_ctx, the helpers every translation unit
// of the C-ABI uses (status / HIP error plumbing, device buffers) and the functions they share.  Not installed, not part of the ABI:
// everything declared here has hidden visibility.
//   pano_api.cpp      context life cycle, pano_prepare, frame slots, the per-frame launch sequence (feed / blend / compose), stage
//                     inspection, profiling
//   pano_cameras.cpp  camera parameters: validation, verifyCamParams, the 18 N + 1 list, cameraparaout_<id>.txt reader / writer
//   pano_masks.cpp    seam-scale warps, Voronoi / graph-cut seam finders, the mask refresh beside the frame loop, exposure gains
//   pano_host.cpp     entries on host memory: pano_compose_host, page-locked staging, the streaming slots, caller-side stacking
//   pano_sharded.cpp  the camera-sharded exchange over RCCL
#pragma once

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

#pragma GCC visibility push(hidden)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

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
    F32Order f32_order;       // PANO_PYRDOWN32F_ORDER=v,vbody,h,hbody: the association of cv::pyrDown CV_32F the weights follow
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

    // the sharded exchange's packed form (pano_gather_slots): per camera the live rectangles of its levels as copy segments, where the
    // camera's packed bytes start in the staging buffer and how many they are; rebuilt when the live rects change
    std::vector<XchSeg> xch_segs;
    int xch_first[kMaxCams + 1] = {};      // segments of camera i: [xch_first[i], xch_first[i + 1])
    int xch_rows[kMaxCams] = {};           // the tallest segment of camera i
    size_t xch_off[kMaxCams + 1] = {};     // packed bytes of camera i: [xch_off[i], xch_off[i + 1]) of the staging buffer
    XchSeg* d_xch_segs = nullptr;
    uint8_t* xch_stage = nullptr;
    size_t xch_stage_bytes = 0;
    bool xch_dirty = true;
    bool xch_whole_slots = false;          // PANO_GATHER_WHOLE_SLOTS=1: whole slots travel, in place (rounds 1 - 4)
    uint64_t xch_bytes_moved = 0;          // bytes this rank handed to ncclSend / ncclRecv so far (pano_get_exchange_stats)

    MaskJob* job = nullptr;
    MaskJob* job_trash = nullptr;  // (unused since the pool: kept for a refresh that failed half way)
    std::vector<std::pair<size_t, void*>> refresh_pool;  // device buffers of the last refresh, reused by the next (Scratch::pool)
    std::vector<std::pair<size_t, void*>> pairs_pool;    // ... and the graphs of its pairs: the refresh thread's while it runs

    std::string err;
};

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return PANO_EHIP;                                                                   \
        }                                                                                       \
    } while (0)

inline pano_status fail(pano_ctx* c, pano_status s, const char* msg) {
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


// ---- shared between the translation units (defined in the file named) ----
// pano_api.cpp
void drop_graphs(pano_ctx* c);
void free_device(pano_ctx* c);
WarpCam make_warp_cam(const pano_ctx* c, int i, const uint8_t* src, size_t stride, bool roi_only);
void live_rects(pano_ctx* c, const std::vector<std::vector<uint8_t>>& masks);
pano_status upload_gain_tables(pano_ctx* c, int i, const float* h_gain);
pano_status ensure_weights(pano_ctx* c, hipStream_t s);
pano_status check_compute(pano_ctx* c);
void bind_slot(pano_ctx* c, int k);
// pano_cameras.cpp
bool parse_floats(const std::string& s, std::vector<float>& out);
// pano_masks.cpp
void drop_job(pano_ctx* c);  // ends a mask refresh under way (joins its thread, frees its buffers)
// pano_host.cpp
hipError_t shared_copy_streams(int device, hipStream_t* h2d, hipStream_t* d2h);  // process-wide copy streams, one pair per device

#pragma GCC visibility pop
