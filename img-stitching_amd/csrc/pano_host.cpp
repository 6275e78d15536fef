// pano_host.cpp - the entries of the C-ABI (include/pano.h) that take HOST memory: pano_compose_host (process(vector<Mat>&, Mat&),
// reference include/ocvstitcher.hpp:1141), page-locked staging, caller-side stacking on host buffers, the streaming slots of a
// capture loop (pano_stream_*), and the host forms of the sharded feed / blend.

#include "pano_ctx.hpp"

extern "C" {

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

}  // extern "C"

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

namespace {
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

extern "C" {

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

pano_status pano_compose_host(pano_ctx* c, const uint8_t* const* h_frames, const size_t* strides, uint8_t* h_out, size_t out_stride) {
    return guarded(c, [&]() { return compose_host_impl(c, h_frames, strides, h_out, out_stride); });
}

}  // extern "C"
