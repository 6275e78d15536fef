// pano_api.cpp - C-ABI of libpano_hip.so (include/pano.h): context life cycle, pano_prepare, frame slots, the per-frame launch
// sequence (feed / blend / compose), stage inspection, profiling.  The other entry points: pano_cameras.cpp, pano_masks.cpp,
// pano_host.cpp, pano_sharded.cpp (pano_ctx.hpp names them).
// Host code only; the arithmetic is in the kernel files (pano_warp / pano_pyramid / pano_blend / pano_blend_small / pano_init .hip), the init-time geometry in pano_plan.hpp.
// No CPU fallback exists: every compute entry point launches HIP kernels or fails.

#include "pano_ctx.hpp"

#pragma GCC visibility push(hidden)  // helpers of this file and of its siblings (declared in pano_ctx.hpp): not part of the ABI

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
    dfree(c->d_xch_segs); dfree(c->xch_stage);
    c->xch_stage_bytes = 0; c->xch_dirty = true; c->xch_segs.clear();
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
void live_rects(pano_ctx* c, const std::vector<std::vector<uint8_t>>& masks) {
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
    c->xch_dirty = true;   // the sharded exchange packs these rectangles (pano_gather_slots)
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
                                c->wpitch[i][l + 1], c->f32_order, s);
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

#pragma GCC visibility pop

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
    c->xch_whole_slots = getenv("PANO_GATHER_WHOLE_SLOTS") && atoi(getenv("PANO_GATHER_WHOLE_SLOTS"));
    // the association of cv::pyrDown CV_32F the blend weights follow: "vertical,vbody,horizontal,hbody" (default: scalar order, what
    // an OpenCV build without SIMD computes; the day the pin kit says which build the reference's results come from, this is where
    // the product follows it - tests/pin_stages.py PYRDOWN32F_VARIANTS names the forms)
    c->f32_order = F32Order{};
    if (const char* fo = getenv("PANO_PYRDOWN32F_ORDER")) {
        int v[4] = {0, 8, 0, 4};
        if (sscanf(fo, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) >= 1 && v[0] >= 0 && v[0] <= 2 && v[2] >= 0 && v[2] <= 2 && v[1] >= 1 && v[1] <= 64 &&
            v[3] >= 1 && v[3] <= 64)
            c->f32_order = F32Order{v[0], v[1], v[2], v[3]};
        else
            return fail(c, PANO_EINVAL, "PANO_PYRDOWN32F_ORDER wants vertical,vbody,horizontal,hbody with vertical, horizontal in 0..2 and bodies in 1..64");
    }
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

}  // extern "C"
// point the kernel parameter blocks at the buffers of frame slot k
void bind_slot(pano_ctx* c, int k) {
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
extern "C" {

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
    return guarded(c, [&]() { c->gc_dump_path = path ? path : ""; return PANO_OK; });  // the assignment allocates
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
pano_status pano_get_pyramid_slots(pano_ctx* c, void** d_base, size_t* slot_bytes) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if (!d_base || !slot_bytes) return PANO_EINVAL;
    *d_base = c->pyr_base;
    *slot_bytes = c->slot_bytes;
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

// Device-copy ceilings, measured in the library with the kernels' own launch and timing machinery (per-launch begin / end events of
// hipExtLaunchKernelGGL - the interval rocprofv3 reports per dispatch).  `sets` rotating buffer sets: with sets x the bytes of one
// launch above the 256 MiB Infinity Cache no launch finds its input where an earlier one left it (the COLD figure K1's roofline leads
// with); sets = 1 is the warm figure.
static pano_status probe_copy_impl(pano_ctx* c, int kind, uint64_t units, int sets, int reps, double* gbps, double* us, uint64_t* bytes_moved) {
    pano_status st = check_compute(c);
    if (st != PANO_OK) return st;
    if ((kind != PANO_PROBE_COPY_F4 && kind != PANO_PROBE_COPY_K1_SHAPE && kind != PANO_PROBE_COPY_F4_FLAT) || units == 0 || sets < 1 || sets > 64 || reps < 1 || reps > 4096 ||
        !gbps || !us)
        return PANO_EINVAL;
    size_t in_bytes, out_bytes, tab_bytes = 0;
    const bool f4 = kind == PANO_PROBE_COPY_F4 || kind == PANO_PROBE_COPY_F4_FLAT;
    if (f4) {
        in_bytes = out_bytes = (size_t)((units + 15) / 16 * 16);
    } else {
        if (units > (1u << 22)) return PANO_EINVAL;
        in_bytes = (size_t)units * kProbeBoxBytes + 1024;  // the last workgroup's surplus lanes re-read its last chunk only
        tab_bytes = (size_t)units * 256 * 8;
        out_bytes = (size_t)units * 3072;
    }
    if ((in_bytes + tab_bytes + out_bytes) * (size_t)sets > ((size_t)6 << 30)) return PANO_EINVAL;
    std::vector<void*> in(sets, nullptr), tab(sets, nullptr), out(sets, nullptr);
    std::vector<hipEvent_t> ev;
    hipStream_t s = c->own_stream;
    auto cleanup = [&]() {
        for (void* p : in) if (p) (void)hipFree(p);
        for (void* p : tab) if (p) (void)hipFree(p);
        for (void* p : out) if (p) (void)hipFree(p);
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    };
    hipError_t err = hipSuccess;
#define PROBE_TRY(expr)                         \
    do {                                        \
        if ((err = (expr)) != hipSuccess) {     \
            cleanup();                          \
            return fail(c, PANO_EHIP, #expr);   \
        }                                       \
    } while (0)
    for (int k = 0; k < sets; k++) {
        PROBE_TRY(hipMalloc(&in[k], in_bytes));
        PROBE_TRY(hipMalloc(&out[k], out_bytes));
        PROBE_TRY(hipMemsetAsync(in[k], 0x5a, in_bytes, s));
        PROBE_TRY(hipMemsetAsync(out[k], 0, out_bytes, s));
        if (tab_bytes) {
            PROBE_TRY(hipMalloc(&tab[k], tab_bytes));
            PROBE_TRY(hipMemsetAsync(tab[k], 0x33, tab_bytes, s));
        }
    }
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * 8;  // 8 workgroups of 4 waves per CU: every wave slot taken
    const int warm = std::max(2 * sets, 4);
    ev.resize(2 * (size_t)reps, nullptr);
    for (auto& e : ev) PROBE_TRY(hipEventCreate(&e));
    for (int k = 0; k < warm + reps; k++) {
        const int b = k % sets;
        hipEvent_t e0 = k >= warm ? ev[2 * (size_t)(k - warm)] : nullptr, e1 = k >= warm ? ev[2 * (size_t)(k - warm) + 1] : nullptr;
        if (f4) launch_probe_copy_f4(in[b], out[b], in_bytes, kind == PANO_PROBE_COPY_F4 ? blocks : 0, s, e0, e1);
        else launch_probe_copy_k1_shape(in[b], tab[b], out[b], (unsigned)units, s, e0, e1);
    }
    PROBE_TRY(hipGetLastError());
    PROBE_TRY(hipStreamSynchronize(s));
    double total_ms = 0.0;
    for (int k = 0; k < reps; k++) {
        float ms = 0.f;
        PROBE_TRY(hipEventElapsedTime(&ms, ev[2 * (size_t)k], ev[2 * (size_t)k + 1]));
        total_ms += ms;
    }
#undef PROBE_TRY
    cleanup();
    const uint64_t moved = f4 ? (uint64_t)in_bytes * 2 : (uint64_t)units * (kProbeBoxBytes + 2048 + 3072);
    *us = total_ms / reps * 1e3;
    *gbps = (double)moved / (*us * 1e-6) / 1e9;
    if (bytes_moved) *bytes_moved = moved;
    return PANO_OK;
}
pano_status pano_probe_copy(pano_ctx* c, int kind, uint64_t units, int sets, int reps, double* gbps, double* us_per_launch, uint64_t* bytes_moved) {
    return guarded(c, [&]() { return probe_copy_impl(c, kind, units, sets, reps, gbps, us_per_launch, bytes_moved); });
}

// the identity of the device code this library was built from: the first 16 hex digits of the SHA-256 over the kernel sources
// (csrc/*.hip, pano_dev.hpp, pano_kernels.hpp), computed by the Makefile.  profiles/warp_traffic.json records the id its counters
// were taken with; bench.py prints a counter figure only beside the id it belongs to.
const char* pano_kernel_source_id(void) {
#ifdef PANO_KERNEL_SOURCE_ID
    return PANO_KERNEL_SOURCE_ID;
#else
    return "unknown";
#endif
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
        // < 0: the probe itself failed (a launch or a wait returned an error, or the pair came back faster than ONE spin can -
        // the kernels did not run): such a pair is never counted as overlapping
        auto pair_us = [&](hipStream_t a, hipStream_t b) {
            double best = 1e30;
            for (int rep = 0; rep < 2; rep++) {
                if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1.0;
                const auto t0 = std::chrono::steady_clock::now();
                launch_spin((unsigned long long)(kSpinUs * 100.0), a);
                launch_spin((unsigned long long)(kSpinUs * 100.0), b);
                if (hipGetLastError() != hipSuccess) return -1.0;
                if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1.0;
                best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
            }
            return best < 0.8 * kSpinUs ? -1.0 : best;
        };
        std::vector<hipStream_t> rejects;
        bool probe_failed = false;
        for (int tries = 0; tries < 6 * n + 8 && (int)c->flight_streams.size() < n; tries++) {
            hipStream_t cand = nullptr;
            if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            launch_spin(100ull, cand);  // first use: the queue is bound (and the kernel's code loaded) before anything is timed
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(cand) != hipSuccess) probe_failed = true;
            bool alone = !probe_failed || c->flight_streams.empty();
            for (hipStream_t taken : c->flight_streams) {
                if (probe_failed) break;
                const double us = pair_us(taken, cand);
                if (us < 0.0) { probe_failed = true; alone = false; break; }
                if (us > 1.6 * kSpinUs) { alone = false; break; }
            }
            (alone ? c->flight_streams : rejects).push_back(cand);
        }
        if (probe_failed) {
            // nothing was measured: the streams still work, but nothing is known about their queues - say one, not n
            (void)hipGetLastError();
            while ((int)c->flight_streams.size() > 1) { rejects.push_back(c->flight_streams.back()); c->flight_streams.pop_back(); }
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

pano_status pano_set_undistort(pano_ctx* c, int cam, const pano_undistort* u) {
    return guarded(c, [&]() { return set_undistort_impl(c, cam, u); });
}

}  // extern "C"
