// stitcher.hpp - C++ host-side mirror of the reference's `ocvStitcher` (reference
// include/ocvstitcher.hpp:254-1306) over the C-ABI of include/pano.h.  Header-only, needs only the C++
// standard library and libpano_hip.so - OpenCV, yaml-cpp and spdlog are not required.
//
// A master.cpp / replay.cpp style loop (reference src/master.cpp:258-326, src/replay.cpp:206-288) ports by
// replacing `ocvStitcher` with `pano::Stitcher` and `cv::Mat` with `pano::Mat`:
//
//     pano::Stitcher up, down;
//     up.init(cfgPath); down.init(cfgPath);                 // RET_OK / RET_ERR            (:262)
//     up.calibration(upImgs); down.calibration(downImgs);   // seam masks                  (:592)
//     std::thread t1(&pano::Stitcher::process, &up, std::ref(upImgs), std::ref(out[0]));   // (:1141)
//
// Differences that are deliberate: K/R are never re-estimated (init modes 2 and 3 only - the north star fixes
// the cameras); the seam finder is the reference's GraphCutSeamFinder(COST_COLOR) by default (ocvstitcher.hpp:1033),
// with the geometry-only Voronoi finder of the one-shot twin (stitching_detailed.cpp:728-729) selectable; there
// is no function-static frame counter shared between instances (reference :1150 is racy): each instance
// refreshes its masks every `maskRefreshPeriod` frames on its own; and process() reports what went wrong
// (lastStatus() / lastError()) instead of leaving `ret` undefined.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pano.h"

namespace pano {

constexpr int RET_OK = 0, RET_ERR = -1;             // include/stitcherglobal.h:13-14
enum enStitcherInitMode { enInitALL = 1, enInitByDefault = 2, enInitByCfg = 3 };  // stitcherglobal.h:110-115

// the slice of cv::Mat the compose path uses: 8UC3, row-major, refcounted storage, ROI views
struct Mat {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    std::shared_ptr<uint8_t> owner;
    Mat() {}
    Mat(int r, int c) { create(r, c); }
    Mat(int r, int c, uint8_t* ext, size_t st) : rows(r), cols(c), step(st), data(ext) {}  // borrowed
    // storage is page-locked (pano_host_alloc) when the library can provide it - pano_compose_host then DMAs it directly
    // instead of staging it - and plain heap memory otherwise (plan-only use without a GPU)
    void create(int r, int c) {
        if (r == rows && c == cols && owner) return;
        rows = r; cols = c; step = (size_t)c * 3;
        const size_t bytes = step * (size_t)r;
        if (void* p = pano_host_alloc(bytes)) owner.reset((uint8_t*)p, [](uint8_t* q) { pano_host_free(q); });
        else owner.reset(new uint8_t[bytes], std::default_delete<uint8_t[]>());
        data = owner.get();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; owner.reset(); }
    bool empty() const { return data == nullptr; }
    Mat roi(int x, int y, int w, int h) const {  // cv::Mat::operator()(Rect): a view, no copy
        Mat m = *this;
        m.data = data + (size_t)y * step + (size_t)x * 3;
        m.rows = h; m.cols = w;
        return m;
    }
};

// mirrors stStitcherCfg (stitcherglobal.h:68-81)
struct StitcherCfg {
    int width = 0, height = 0;
    short id = 0, num_images = 0;
    float matchConf = 0.3f, adjusterConf = 0.7f, blendStrength = 1.f;
    float stitchercameraExThres = 0.f, stitchercameraInThres = 0.f;
    std::string cfgPath;
    int initMode = enInitByDefault;
};

namespace detail {
inline std::string trim(const std::string& s) {
    size_t b = s.find_first_not_of(" \t\r\n\"'"), e = s.find_last_not_of(" \t\r\n\"'");
    return b == std::string::npos ? std::string() : s.substr(b, e - b + 1);
}
inline std::string strip_comment(const std::string& l) {
    size_t p = l.find('#');
    return p == std::string::npos ? l : l.substr(0, p);
}
// flat `key: value` lines (the stitcher cfg, cfg/stitcher-imx390cfg.yaml)
inline bool read_flat_yaml(const std::string& path, std::map<std::string, std::string>& kv) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string l;
    while (std::getline(f, l)) {
        l = strip_comment(l);
        size_t c = l.find(':');
        if (c == std::string::npos) continue;
        std::string k = trim(l.substr(0, c)), v = trim(l.substr(c + 1));
        if (!k.empty() && k[0] != '-') kv[k] = v;
    }
    return true;
}
struct Structure {
    std::map<std::string, std::string> key;          // vendor sensor sttype undistor fov inputsz
    std::vector<std::string> cams;                   // one comma list per stitcher id
    std::vector<std::vector<int>> cut;
};
// the `structures:` list of cfg/cameras.yaml (:147-300): entries start with " -" at the list indent,
// each has scalar keys and `params:` with per-stitcher `cams: [...]` (may span lines) and `cut: [...]`
inline bool read_structures(const std::string& path, std::vector<Structure>& out) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::string l, bracket;
    bool in_structs = false, in_cams = false;
    int list_indent = -1;
    while (std::getline(f, l)) {
        l = strip_comment(l);
        if (trim(l).empty()) continue;
        if (l.compare(0, 11, "structures:") == 0) { in_structs = true; continue; }
        if (!in_structs) continue;
        if (in_cams) {
            bracket += " " + l;
            if (l.find(']') != std::string::npos) {
                size_t a = bracket.find('['), b = bracket.find(']');
                out.back().cams.push_back(bracket.substr(a + 1, b - a - 1));
                in_cams = false;
            }
            continue;
        }
        size_t ind = l.find_first_not_of(' ');
        std::string t = trim(l);
        if (t == "-") {
            if (list_indent < 0) list_indent = (int)ind;
            if ((int)ind == list_indent) out.emplace_back();
            continue;
        }
        if (out.empty()) continue;
        size_t c = t.find(':');
        if (c == std::string::npos) continue;
        std::string k = trim(t.substr(0, c)), v = trim(t.substr(c + 1));
        if (!k.empty() && k[0] == '-') k = trim(k.substr(1));
        if (k == "cams") {
            bracket = v;
            if (v.find(']') != std::string::npos) {
                size_t a = v.find('['), b = v.find(']');
                out.back().cams.push_back(v.substr(a + 1, b - a - 1));
            } else {
                in_cams = true;
            }
        } else if (k == "cut") {
            std::vector<int> r;
            size_t a = v.find('['), b = v.find(']');
            std::stringstream ss(a == std::string::npos ? std::string() : v.substr(a + 1, b - a - 1));
            std::string tok;
            while (std::getline(ss, tok, ',')) r.push_back(atoi(tok.c_str()));
            out.back().cut.push_back(r);
        } else if (k != "params") {
            out.back().key[k] = v;
        }
    }
    return true;
}
}  // namespace detail

class Stitcher {
  public:
    Stitcher() {}
    ~Stitcher() { pano_destroy(ctx_); }
    Stitcher(const Stitcher&) = delete;
    Stitcher& operator=(const Stitcher&) = delete;

    int projector = PANO_SPHERICAL;   // reference: SphericalWarperGpu (ocvstitcher.hpp:1000)
    int device = 0;
    int maskRefreshPeriod = 200;      // process() refreshes the masks every 200 calls (ocvstitcher.hpp:1152)
    bool asyncMaskRefresh = false;    // ... beside the frame loop (pano_refresh_masks_*) instead of inside process(): the frame
                                      // that triggers it and its successors are composed with the old masks until the cuts are done
    // seam finder of initSeam / updateMask: the reference creates GraphCutSeamFinder(COST_COLOR) (ocvstitcher.hpp:1033,
    // :860) and has NoSeamFinder commented beside it; the one-shot twin also offers Voronoi (stitching_detailed.cpp:728)
    enum SeamFinder { SeamGraphCut = 0, SeamVoronoi = 1 };
    int seamFinder = SeamGraphCut;
    // initSeam feeds a GAIN_BLOCKS compensator (ocvstitcher.hpp:1031-1032) but process() keeps apply() commented out
    // (:1178); the one-shot twin applies it (src/stitching_detailed.cpp:841).  true = estimate in calibration(), apply
    // in every process()
    bool exposureCompensation = false;

    // init(yaml) (ocvstitcher.hpp:262-358): stitcher cfg -> size, num_images, blend strength, init mode; the
    // matching `structures:` entry of the camera cfg -> default cams (18N+1 floats) and cut
    int init(const std::string& stitcherCfgPath, int stitcher_id = 0) {
        std::map<std::string, std::string> kv;
        if (!detail::read_flat_yaml(stitcherCfgPath, kv)) return RET_ERR;
        try {
            cfg_.width = std::stoi(kv.at("outPutWidth"));
            cfg_.height = std::stoi(kv.at("outPutHeight"));
            cfg_.id = (short)stitcher_id;
            cfg_.num_images = (short)std::stoi(kv.at("num_images"));
            cfg_.blendStrength = std::stof(kv.at("stitcherBlenderStrength"));
            cfg_.cfgPath = kv.count("camcfgpath") ? kv["camcfgpath"] : std::string();
            cfg_.initMode = kv.count("initMode") ? std::stoi(kv["initMode"]) : (int)enInitByDefault;
            if (kv.count("stitcherMatchConf")) cfg_.matchConf = std::stof(kv["stitcherMatchConf"]);
            if (kv.count("stitcherAdjusterConf")) cfg_.adjusterConf = std::stof(kv["stitcherAdjusterConf"]);
            if (kv.count("stitcherCameraExThres")) cfg_.stitchercameraExThres = std::stof(kv["stitcherCameraExThres"]);
            if (kv.count("stitcherCameraInThres")) cfg_.stitchercameraInThres = std::stof(kv["stitcherCameraInThres"]);
            std::vector<detail::Structure> st;
            if (!detail::read_structures(kv.at("cameraparams"), st)) return RET_ERR;
            defaultCamParams_.clear();
            for (auto& s : st) {
                auto eq = [&](const char* a, const char* b) { return s.key.count(a) && kv.count(b) && s.key[a] == kv[b]; };
                if (eq("vendor", "vendor") && eq("sensor", "sensor") && eq("sttype", "sttype") && eq("undistor", "undistor") &&
                    eq("fov", "fov") && eq("inputsz", "outPutWidth") && (int)s.cams.size() > stitcher_id) {
                    defaultCamParams_ = s.cams[stitcher_id];
                    if ((int)s.cut.size() > stitcher_id && s.cut[stitcher_id].size() == 4) cut_ = s.cut[stitcher_id];
                }
            }
        } catch (...) {
            return RET_ERR;  // "stitcher yml pars failed" (ocvstitcher.hpp:344-348)
        }
        if (defaultCamParams_.empty()) return RET_ERR;  // useDefaultCamParams (ocvstitcher.hpp:428-432)
        return RET_OK;
    }

    // programmatic init: what init(yaml) extracts, passed directly
    int init(const StitcherCfg& cfg, const std::string& defaultCamParams, const int cut[4] = nullptr) {
        cfg_ = cfg;
        defaultCamParams_ = defaultCamParams;
        if (cut) cut_.assign(cut, cut + 4);
        return defaultCamParams_.empty() ? RET_ERR : RET_OK;
    }

    // calibration(imgs) (ocvstitcher.hpp:592-650).  K/R come from the defaults (mode 2) or from the last
    // record of <cfgPath>cameraparaout_<id>.txt (mode 3, falling back to the defaults: the state the reference ends in
    // when its own fallback, initAll, has failed five times and it flips to mode 2, :639-643); then the mask half of
    // initSeam (:975-1101) runs on the GPU.  Mode 1 (initAll: features, matching, bundle adjustment) is not rebuilt
    // here - the north star fixes K and R - and is served like the reference's "calibration failed due to environment,
    // use default parameters"; a caller that runs its own bundle adjustment hands the result to the overload below.
    //
    // The cut (m_cutParams).  init(yaml) loads the structure's `cut` in EVERY mode (:333-337) and initSeam never touches
    // it, so modes 2 and 3 and every fallback onto the defaults crop with the yaml cut; only a successful initAll
    // rewrites it to [0, (rows - cut_h) / 2, cols, cut_h] (:959-964) - here: the overload with estimated cameras.
    int calibration(const std::vector<Mat>& imgs) { return calibrate(imgs, nullptr, nullptr, 0.f); }

    // The tail of initAll (ocvstitcher.hpp:783-964) for a caller that estimated the cameras itself: K_est / R_est are
    // N x 9 row-major f32, scale_est the median focal (:736-751).  verifyCamParams against the defaults (:783-795) with
    // the yaml thresholds; a plausible estimate is installed and the cut becomes initAll's (:959-964); an implausible one
    // is RET_ERR, where the reference's loop retries and finally takes the defaults - calibration(imgs) does that.
    int calibration(const std::vector<Mat>& imgs, const float* K_est, const float* R_est, float scale_est) {
        if (!K_est || !R_est) return RET_ERR;
        return calibrate(imgs, K_est, R_est, scale_est);
    }

    // process(imgs, ret) (ocvstitcher.hpp:1141-1216).  The reference's returns nothing and cannot fail visibly; here a frame
    // that could not be composed leaves `ret` EMPTY and the reason in lastStatus() / lastError()
    void process(std::vector<Mat>& imgs, Mat& ret) {
        const uint8_t* frames[PANO_MAX_CAMS];
        size_t strides[PANO_MAX_CAMS];
        status_ = PANO_OK;
        if (!ctx_) { status_ = PANO_ESTATE; ret.release(); return; }
        if (!borrow(imgs, frames, strides)) { status_ = PANO_EINVAL; ret.release(); return; }  // count, size, null data
        if (refreshing_) {  // a refresh started on an earlier frame: its masks go in as soon as its thread is through
            int done = 0;
            if (pano_refresh_masks_poll(ctx_, &done) != PANO_OK || done) refreshing_ = false;
        }
        bool begin = false;
        if (maskRefreshPeriod > 0 && ++frame_ > maskRefreshPeriod) {  // updateMask cadence (:1152-1159)
            frame_ = 0;
            if (asyncMaskRefresh && seamFinder == SeamGraphCut) begin = !refreshing_;
            else if (buildMasks(imgs) != RET_OK) { status_ = PANO_ERR; ret.release(); return; }
        }
        int w = 0, h = 0;
        if ((status_ = pano_get_output_size(ctx_, &w, &h)) != PANO_OK) { ret.release(); return; }
        ret.create(h, w);
        if ((status_ = pano_compose_host(ctx_, frames, strides, ret.data, ret.step)) != PANO_OK) ret.release();
        // beside the frame loop, and behind this frame's panorama: the reference's inline updateMask costs several frame periods
        if (begin && pano_refresh_masks_begin(ctx_, frames, strides) == PANO_OK) refreshing_ = true;
    }

    pano_ctx* handle() { return ctx_; }
    const StitcherCfg& config() const { return cfg_; }
    pano_status lastStatus() const { return status_; }   // of the last process()
    const char* lastError() const { return pano_last_error(ctx_); }

  private:
    int calibrate(const std::vector<Mat>& imgs, const float* K_est, const float* R_est, float scale_est) {
        // the frames feed the graph-cut seam finder and, when it is on, the exposure compensator
        pano_destroy(ctx_);
        ctx_ = nullptr;
        pano_config c{};
        c.num_images = cfg_.num_images; c.width = cfg_.width; c.height = cfg_.height;
        c.projector = projector; c.blend_strength = cfg_.blendStrength; c.num_bands = PANO_BANDS_FROM_STRENGTH;
        c.device = device;
        if (cut_.size() == 4)  // the yaml cut, whatever the mode (:333-337); initAll's rewrite follows pano_prepare below
            for (int i = 0; i < 4; i++) c.cut[i] = cut_[i];
        if (pano_create(&c, &ctx_) != PANO_OK) return RET_ERR;
        if (pano_set_cameras_from_list(ctx_, defaultCamParams_.c_str()) != PANO_OK) return RET_ERR;  // useDefaultCamParams
        if (K_est) {
            if (pano_verify_cameras(ctx_, K_est, R_est, cfg_.stitchercameraExThres, cfg_.stitchercameraInThres, nullptr) != PANO_OK)
                return RET_ERR;  // "environment is not suitable for calibration" (:415-418)
            std::string list;
            char buf[32];
            for (int i = 0; i < cfg_.num_images; i++)
                for (int k = 0; k < 18; k++) {
                    snprintf(buf, sizeof buf, "%.9g,", k < 9 ? K_est[9 * i + k] : R_est[9 * i + k - 9]);
                    list += buf;
                }
            snprintf(buf, sizeof buf, "%.9g", scale_est);
            list += buf;
            if (pano_set_cameras_from_list(ctx_, list.c_str()) != PANO_OK) return RET_ERR;
        } else if (cfg_.initMode == enInitByCfg) {
            std::string file = cfg_.cfgPath + "cameraparaout_" + std::to_string(cfg_.id) + ".txt";
            (void)pano_load_camera_file(ctx_, file.c_str());  // all or nothing: a record that does not load leaves the defaults
        }
        if (pano_prepare(ctx_) != PANO_OK) return RET_ERR;
        if (K_est && cut_.size() == 4) {  // initAll's cut (:959-964): full width, the yaml height centred
            int rect[4];
            if (pano_get_pano_rect(ctx_, rect) != PANO_OK) return RET_ERR;
            const int cut[4] = {0, (rect[3] - cut_[3]) / 2, rect[2], cut_[3]};
            if (pano_set_cut(ctx_, cut) != PANO_OK) return RET_ERR;
        }
        if (device >= 0) {
            if (buildMasks(imgs) != RET_OK) return RET_ERR;
            if (exposureCompensation) {
                const uint8_t* frames[PANO_MAX_CAMS];
                size_t strides[PANO_MAX_CAMS];
                if (!borrow(imgs, frames, strides)) return RET_ERR;
                if (pano_estimate_gains(ctx_, frames, strides, 32, 32) != PANO_OK) return RET_ERR;
            }
        }
        frame_ = 0;
        refreshing_ = false;  // a refresh that was under way died with the old context
        return RET_OK;
    }
    // the frames as the C-ABI takes them; false when there are too few or they are not stitcher sized
    bool borrow(const std::vector<Mat>& imgs, const uint8_t** frames, size_t* strides) const {
        if ((int)imgs.size() < cfg_.num_images) return false;
        for (int i = 0; i < cfg_.num_images; i++) {
            if (!imgs[i].data || imgs[i].cols != cfg_.width || imgs[i].rows != cfg_.height) return false;
            frames[i] = imgs[i].data;
            strides[i] = imgs[i].step;
        }
        return true;
    }
    // the mask half of initSeam / updateMask (ocvstitcher.hpp:975-1101, :1218-1261)
    int buildMasks(const std::vector<Mat>& imgs) {
        const uint8_t* frames[PANO_MAX_CAMS];
        size_t strides[PANO_MAX_CAMS];
        if (seamFinder == SeamGraphCut && borrow(imgs, frames, strides))
            return pano_build_masks_graphcut(ctx_, frames, strides) == PANO_OK ? RET_OK : RET_ERR;
        return pano_build_masks_voronoi(ctx_) == PANO_OK ? RET_OK : RET_ERR;  // geometry only: needs no frames
    }
    pano_ctx* ctx_ = nullptr;
    StitcherCfg cfg_;
    std::string defaultCamParams_;
    std::vector<int> cut_;
    int frame_ = 0;
    bool refreshing_ = false;
    pano_status status_ = PANO_OK;
};

}  // namespace pano
