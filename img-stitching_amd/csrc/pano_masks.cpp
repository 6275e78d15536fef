// pano_masks.cpp - the mask pipeline and the exposure compensator of the C-ABI (include/pano.h): seam-scale warps, Voronoi and
// graph-cut seam finders (the max-flow on the host: pano_graphcut.hpp), dilate / resize / AND, the mask refresh beside the frame
// loop (pano_refresh_masks_*), gain maps and their estimation (reference include/ocvstitcher.hpp:975-1101, :1218-1261).

#include "pano_ctx.hpp"

namespace {

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


}  // namespace

extern "C" {

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
}  // extern "C"
void drop_job(pano_ctx* c) {
    reap_trash(c);
    MaskJob* j = c->job;
    if (!j) return;
    if (j->th.joinable()) j->th.join();
    if (j->s) (void)hipStreamDestroy(j->s);
    c->job = nullptr;
    delete j;  // frees the job's device scratch
}
extern "C" {
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
        if (!dump && !c->gc_dump_path.empty())  // a dump that was asked for and cannot be written is an error, not a silent OK
            return fail(c, PANO_ERR, ("pano_debug_graphcut_dump: cannot open " + c->gc_dump_path).c_str());
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

}  // extern "C"
