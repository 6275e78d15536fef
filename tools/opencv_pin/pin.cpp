// pin.cpp - the OpenCV pin kit: produces, WITH OPENCV ITSELF, the golden vectors this repository cannot produce.
//
// NOT BUILT OR RUN IN THIS REPOSITORY'S CONTAINER OR ON ITS GPU BOX: neither has OpenCV (no headers, no libraries, no cv2, no
// network).  It is for a holder of an OpenCV 3.4.x build (>= 3.4.0 for INTER_LINEAR_EXACT; the reference's CMakeLists.txt:58 asks
// for "OpenCV 3", README.md:4-5 for >= 3.4.0):
//
//     g++ -O2 -std=c++11 tools/opencv_pin/pin.cpp -o pin $(pkg-config --cflags --libs opencv)
//     ./pin tests/golden tests/golden/opencv          # writes tests/golden/opencv/{c1,c1b,r,s}_golden.json (+ PNGs of every stage)
//     python -m pytest tests/test_oracle.py -k opencv_pin    # compares the oracle with them, key by key
//
// Until those files exist the oracle's parity is UNPINNED and the loader test is skipped with that reason.
//
// What it runs is exactly the reference's sequence, on the CPU classes the north star names as the parity target:
//   * ocvStitcher::initSeam / updateMask (include/ocvstitcher.hpp:975-1136, :1218-1261): cv::resize(INTER_LINEAR_EXACT) to the seam
//     scale, detail::SphericalWarper / CylindricalWarper::warp (LINEAR / REFLECT images, NEAREST / CONSTANT masks),
//     detail::BlocksGainCompensator::feed, detail::GraphCutSeamFinder(COST_COLOR)::find (and VoronoiSeamFinder, the CLI's other
//     option, src/stitching_detailed.cpp:728-729), dilate, resize(INTER_LINEAR_EXACT), AND;
//   * ocvStitcher::process (:1141-1216): RotationWarper::warpRoi / warp, convertTo(CV_16S), the band rule, MultiBandBlender
//     prepare / feed / blend (Blender::NO below one band), convertTo(CV_8U), the cut;
//   * src/stitching_detailed.cpp:841: ExposureCompensator::apply between warp and feed;
//   * src/master.cpp:321-326: cv::resize of the upper panorama to the lower one's size, vconcat, the 10-row bar.
// Inputs are the committed fixtures tests/golden/<prefix>_cam<i>.png + <prefix>_cams.json.  Outputs, two kinds:
//   (1) <prefix>_golden.json in the schema of tests/golden/<prefix>_golden.json (tests/golden/make_golden.py), with SHA-256 over the
//       same bytes (BGR, row-major, tight) - small, comparable key by key;
//   (2) EVERY STAGE AS RAW DATA: <out>/<group>/manifest.json + one .npy per array, in the schema of
//       tests/pin_stages.py (compute_group names the arrays; groups c1, c1b, r0, r1, s0, s1, r_stack, s_stack): ROIs, the float maps
//       of buildMaps, the warps, the NEAREST masks, the seam-scale frames / maps / warps / masks, both seam finders' masks, the blend
//       masks, gain maps and gain-applied warps, cv::pyrDown / pyrUp CV_16S and cv::pyrDown CV_32F by themselves ("unit" stages: they
//       tell WHICH association of the f32 sum this OpenCV build runs - scalar, SSE2, NEON, universal intrinsics), MultiBandBlender's
//       accumulated Laplacian and weight pyramids (private members, read through #define private public), its result, the panoramas.
//       tests/test_oracle.py::test_opencv_pin_files_when_present runs the oracle stage by stage ON THESE INPUTS and names the first
//       stage that diverges.  A few hundred MB; commit the manifests and what you can, or keep them beside the run.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/imgcodecs.hpp>
#include <opencv2/imgproc.hpp>
// BlocksGainCompensator keeps its gain maps private in 3.4 (getMatGains arrived in 4.x), MultiBandBlender its pyramids
// (dst_pyr_laplace_, dst_band_weights_): the kit reads them as they are
#define private public
#define protected public
#include <opencv2/stitching/detail/blenders.hpp>
#include <opencv2/stitching/detail/exposure_compensate.hpp>
#undef protected
#undef private
#include <opencv2/stitching/detail/seam_finders.hpp>
#include <opencv2/stitching/detail/util.hpp>
#include <opencv2/stitching/detail/warpers.hpp>

using namespace cv;
using namespace cv::detail;

// ---- SHA-256 (FIPS 180-4) ---------------------------------------------------------------------------------------------------
namespace {
struct Sha256 {
    uint32_t h[8];
    uint64_t len = 0;
    uint8_t buf[64];
    size_t fill = 0;
    Sha256() {
        static const uint32_t init[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        std::memcpy(h, init, sizeof(h));
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t k[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
            0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
            0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
            0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
            0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
            0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + k[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            const size_t take = std::min(n, 64 - fill);
            std::memcpy(buf + fill, p, take);
            fill += take; p += take; n -= take;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex() {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(l, 8);
        char out[65];
        for (int i = 0; i < 8; i++) std::snprintf(out + 8 * i, 9, "%08x", h[i]);
        return std::string(out, 64);
    }
};
// the bytes numpy's ascontiguousarray(a).tobytes() gives: rows tight, whatever the Mat's step
std::string sha(const Mat& m) {
    Sha256 s;
    const size_t row = (size_t)m.cols * m.elemSize();
    for (int y = 0; y < m.rows; y++) s.update(m.ptr<uint8_t>(y), row);
    return s.hex();
}

// ---- the little of JSON the *_cams.json fixtures use: "key": number | [numbers | [numbers]] ---------------------------------
std::string slurp(const std::string& path) {
    std::ifstream f(path.c_str());
    if (!f) { std::cerr << "cannot open " << path << "\n"; std::exit(2); }
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}
// all numbers of the value that follows "key" (from `from` on), flattened; *end = offset behind the value
std::vector<double> numbers_after(const std::string& js, const std::string& key, size_t from = 0, size_t* end = nullptr) {
    size_t p = js.find("\"" + key + "\"", from);
    if (p == std::string::npos) { std::cerr << "key " << key << " missing\n"; std::exit(2); }
    p = js.find(':', p) + 1;
    while (std::isspace((unsigned char)js[p])) p++;
    std::vector<double> out;
    int depth = 0;
    do {
        const char c = js[p];
        if (c == '[') { depth++; p++; }
        else if (c == ']') { depth--; p++; }
        else if (c == '-' || c == '+' || std::isdigit((unsigned char)c)) {
            char* e = nullptr;
            out.push_back(std::strtod(js.c_str() + p, &e));
            p = (size_t)(e - js.c_str());
        } else p++;
    } while (depth > 0);
    if (end) *end = p;
    return out;
}

Mat_<float> mat3(const double* v) {
    Mat_<float> m(3, 3);
    for (int i = 0; i < 9; i++) m(i / 3, i % 3) = (float)v[i];
    return m;
}

// ---- JSON out --------------------------------------------------------------------------------------------------------------
struct Json {
    std::ostringstream o;
    bool first = true;
    void key(const std::string& k) { o << (first ? "" : ",\n") << " \"" << k << "\": "; first = false; }
    template <class T> void list(const std::string& k, const std::vector<T>& v, bool quote) {
        key(k); o << "[";
        for (size_t i = 0; i < v.size(); i++) o << (i ? ", " : "") << (quote ? "\"" : "") << v[i] << (quote ? "\"" : "");
        o << "]";
    }
    void str(const std::string& k, const std::string& v) { key(k); o << "\"" << v << "\""; }
    void raw(const std::string& k, const std::string& v) { key(k); o << v; }
};
std::string rect_list(const std::vector<Rect>& r) {
    std::ostringstream o; o << "[";
    for (size_t i = 0; i < r.size(); i++) o << (i ? ", " : "") << "[" << r[i].x << ", " << r[i].y << ", " << r[i].width << ", " << r[i].height << "]";
    o << "]"; return o.str();
}

// ---- the reference's steps ------------------------------------------------------------------------------------------------
Ptr<RotationWarper> make_warper(int kind, float scale) {
    if (kind == 1) return makePtr<detail::CylindricalWarper>(scale);
    return makePtr<detail::SphericalWarper>(scale);
}

struct Rig {
    int n, w, h, kind;
    std::vector<Mat_<float>> K, R;
    float scale;
    std::vector<Mat> frames;
};

// initSeam / updateMask up to m_blenderMask (ocvstitcher.hpp:975-1101, :1218-1257).  seam: 0 graph cut (the reference), 1 Voronoi.
// gains (optional): the compensator fed like initSeam feeds it (:1031-1032), BEFORE the seam finder touches the masks
std::vector<Mat> blend_masks(const Rig& g, int seam, Ptr<ExposureCompensator>* gains = nullptr, std::vector<Point>* seam_corners = nullptr) {
    const double swa_d = std::min(1.0, std::sqrt(1e5 / ((double)g.h * g.w)));   // ocvstitcher.hpp:298
    std::vector<Mat> seamSized(g.n);
    std::vector<UMat> masks(g.n), images_warped(g.n), masks_warped(g.n), images_warped_f(g.n);
    std::vector<Point> corners(g.n);
    Ptr<RotationWarper> sw = make_warper(g.kind, static_cast<float>(g.scale * swa_d));
    for (int i = 0; i < g.n; i++) {
        resize(g.frames[i], seamSized[i], Size(), swa_d, swa_d, INTER_LINEAR_EXACT);
        masks[i].create(seamSized[i].size(), CV_8U);
        masks[i].setTo(Scalar::all(255));
        Mat_<float> K = g.K[i].clone();
        const float swa = (float)swa_d;
        K(0, 0) *= swa; K(0, 2) *= swa; K(1, 1) *= swa; K(1, 2) *= swa;
        corners[i] = sw->warp(seamSized[i], K, g.R[i], INTER_LINEAR, BORDER_REFLECT, images_warped[i]);
        sw->warp(masks[i], K, g.R[i], INTER_NEAREST, BORDER_CONSTANT, masks_warped[i]);
        images_warped[i].convertTo(images_warped_f[i], CV_32F);
    }
    if (gains) {
        *gains = ExposureCompensator::createDefault(ExposureCompensator::GAIN_BLOCKS);
        (*gains)->feed(corners, images_warped, masks_warped);
    }
    if (seam_corners) *seam_corners = corners;
    Ptr<SeamFinder> finder;
    if (seam == 0) finder = makePtr<detail::GraphCutSeamFinder>(GraphCutSeamFinderBase::COST_COLOR);
    else finder = makePtr<detail::VoronoiSeamFinder>();
    finder->find(images_warped_f, corners, masks_warped);
    Ptr<RotationWarper> bw = make_warper(g.kind, g.scale);
    std::vector<Mat> out(g.n);
    for (int i = 0; i < g.n; i++) {
        Mat mask(g.frames[i].size(), CV_8U, Scalar::all(255)), full, dilated, seam_mask;
        bw->warp(mask, g.K[i], g.R[i], INTER_NEAREST, BORDER_CONSTANT, full);
        dilate(masks_warped[i], dilated, Mat());
        resize(dilated, seam_mask, full.size(), 0, 0, INTER_LINEAR_EXACT);
        out[i] = seam_mask & full;
    }
    return out;
}

// process() (ocvstitcher.hpp:1141-1216).  bands: >= 0 explicit, -1 Blender::NO, -2 the reference's rule from `strength`;
// comp (optional): apply the compensator between warp and feed (src/stitching_detailed.cpp:841); cut: {x, y, w, h} or empty
Mat process(const Rig& g, const std::vector<Mat>& blend_mask, int bands, float strength, ExposureCompensator* comp, const std::vector<int>& cut,
            int* bands_used = nullptr, std::vector<Rect>* rois = nullptr, Rect* pano_roi = nullptr, std::vector<Mat>* warps = nullptr) {
    Ptr<RotationWarper> bw = make_warper(g.kind, g.scale);
    std::vector<Point> corners(g.n);
    std::vector<Size> sizes(g.n);
    std::vector<Rect> rr(g.n);
    for (int i = 0; i < g.n; i++) {
        rr[i] = bw->warpRoi(Size(g.w, g.h), g.K[i], g.R[i]);
        corners[i] = rr[i].tl();
        sizes[i] = rr[i].size();
    }
    if (rois) *rois = rr;
    const Rect full = resultRoi(corners, sizes);
    if (pano_roi) *pano_roi = full;
    Ptr<Blender> blender;
    int nb = bands;
    if (bands == -2) {
        const float blend_width = std::sqrt(static_cast<float>(full.size().area())) * strength / 100.f;
        nb = blend_width < 1.f ? -1 : static_cast<int>(std::ceil(std::log(blend_width) / std::log(2.)) - 1.);
    }
    if (nb < 0) blender = Blender::createDefault(Blender::NO, false);
    else blender = makePtr<MultiBandBlender>(false, nb);   // the CPU class: the parity target (try_gpu = false)
    blender->prepare(corners, sizes);
    if (bands_used) *bands_used = nb < 0 ? -1 : dynamic_cast<MultiBandBlender*>(blender.get())->numBands();
    for (int i = 0; i < g.n; i++) {
        Mat img_warped, img_warped_s;
        bw->warp(g.frames[i], g.K[i], g.R[i], INTER_LINEAR, BORDER_REFLECT, img_warped);
        if (comp) {
            Mat mask(g.frames[i].size(), CV_8U, Scalar::all(255)), mask_warped;
            bw->warp(mask, g.K[i], g.R[i], INTER_NEAREST, BORDER_CONSTANT, mask_warped);
            comp->apply(i, corners[i], img_warped, mask_warped);
        }
        if (warps) warps->push_back(img_warped.clone());
        img_warped.convertTo(img_warped_s, CV_16S);
        blender->feed(img_warped_s, blend_mask[i], corners[i]);
    }
    Mat result, result_mask, ret;
    blender->blend(result, result_mask);
    result.convertTo(ret, CV_8U);
    if (cut.size() == 4) ret = ret(Rect(cut[0], cut[1], cut[2], cut[3])).clone();
    return ret;
}

void save(const std::string& dir, const std::string& name, const Mat& m) { imwrite(dir + "/" + name + ".png", m); }

// ---- raw stages: .npy files + manifest.json, the schema of tests/pin_stages.py -------------------------------------------------
struct StageDir {
    std::string dir, group;
    std::ostringstream man;
    bool first = true;
    StageDir(const std::string& out, const std::string& g) : dir(out + "/" + g), group(g) {
#ifdef _WIN32
        const std::string cmd = "mkdir \"" + dir + "\"";
#else
        const std::string cmd = "mkdir -p \"" + dir + "\"";
#endif
        if (std::system(cmd.c_str()) != 0) { std::cerr << "cannot create " << dir << "\n"; std::exit(2); }
    }
    // NumPy format 1.0: magic, version, little-endian u16 header length, a Python dict literal padded with spaces to a multiple of
    // 64 bytes and ended by a newline, then the array C-contiguous
    void put(const std::string& name, const Mat& m_in) {
        Mat m = m_in.isContinuous() ? m_in : m_in.clone();
        const char* descr = nullptr;
        const char* dtype = nullptr;
        switch (m.depth()) {
            case CV_8U: descr = "|u1"; dtype = "uint8"; break;
            case CV_16S: descr = "<i2"; dtype = "int16"; break;
            case CV_32S: descr = "<i4"; dtype = "int32"; break;
            case CV_32F: descr = "<f4"; dtype = "float32"; break;
            default: std::cerr << "stage " << name << ": depth " << m.depth() << " has no .npy form here\n"; std::exit(2);
        }
        std::ostringstream shape, jshape;
        shape << "(" << m.rows << ", " << m.cols;
        jshape << "[" << m.rows << ", " << m.cols;
        if (m.channels() > 1) { shape << ", " << m.channels(); jshape << ", " << m.channels(); }
        shape << ")"; jshape << "]";
        std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + shape.str() + ", }";
        while ((10 + hdr.size() + 1) % 64) hdr += ' ';
        hdr += '\n';
        std::string fn = name;
        for (size_t p = 0; (p = fn.find('/', p)) != std::string::npos;) fn.replace(p, 1, "__");
        fn += ".npy";
        std::ofstream f((dir + "/" + fn).c_str(), std::ios::binary);
        if (!f) { std::cerr << "cannot write " << dir << "/" << fn << "\n"; std::exit(2); }
        const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
        f.write((const char*)magic, 8);
        const unsigned short hl = (unsigned short)hdr.size();
        const unsigned char le[2] = {(unsigned char)(hl & 0xff), (unsigned char)(hl >> 8)};
        f.write((const char*)le, 2);
        f.write(hdr.data(), (std::streamsize)hdr.size());
        f.write((const char*)m.data, (std::streamsize)(m.total() * m.elemSize()));
        man << (first ? "" : ",\n") << "  \"" << name << "\": {\"file\": \"" << fn << "\", \"dtype\": \"" << dtype << "\", \"shape\": " << jshape.str() << "}";
        first = false;
    }
    void put(const std::string& name, const UMat& u) { put(name, u.getMat(ACCESS_READ)); }
    void finish() {
        std::ofstream f((dir + "/manifest.json").c_str());
        f << "{\n \"group\": \"" << group << "\",\n \"meta\": {\"generator\": \"tools/opencv_pin/pin.cpp\", \"opencv\": \"" << CV_VERSION << "\"},\n \"arrays\": {\n"
          << man.str() << "\n }\n}\n";
    }
};

struct Run {
    std::string tag;
    int seam;          // 0 graph cut, 1 Voronoi
    int bands;         // >= 0, -1 Blender::NO, -2 from strength
    float strength;
    bool gains;
    std::vector<int> cut;
    bool levels;       // dump the blender's pyramids
};

// every stage of one fixture group, in the order and under the names of tests/pin_stages.py compute_group; returns the panoramas of
// the runs (the rig's cut panorama is stacked by the caller)
std::vector<Mat> dump_group(const std::string& out, const std::string& name, const Rig& g, const std::vector<Run>& runs) {
    StageDir sd(out, name);
    const int n = g.n;
    auto cam = [](int i, const char* what) { return "cam" + std::to_string(i) + "/" + what; };
    Ptr<RotationWarper> bw = make_warper(g.kind, g.scale);
    // ocvstitcher.hpp:1054-1063 (warpRoi), :1171 (warp = buildMaps + remap), :1085 (mask warp)
    Mat_<int> roi(n, 4);
    std::vector<Point> corners(n);
    std::vector<Size> sizes(n);
    for (int i = 0; i < n; i++) {
        const Rect r = bw->warpRoi(Size(g.w, g.h), g.K[i], g.R[i]);
        roi(i, 0) = r.x; roi(i, 1) = r.y; roi(i, 2) = r.width; roi(i, 3) = r.height;
        corners[i] = r.tl(); sizes[i] = r.size();
    }
    sd.put("roi", roi);
    std::vector<Mat> warps(n), full_masks(n);
    for (int i = 0; i < n; i++) {
        Mat xmap, ymap;
        bw->buildMaps(Size(g.w, g.h), g.K[i], g.R[i], xmap, ymap);
        sd.put(cam(i, "xmap"), xmap); sd.put(cam(i, "ymap"), ymap);
        bw->warp(g.frames[i], g.K[i], g.R[i], INTER_LINEAR, BORDER_REFLECT, warps[i]);
        sd.put(cam(i, "warp"), warps[i]);
        Mat mask(g.frames[i].size(), CV_8U, Scalar::all(255));
        bw->warp(mask, g.K[i], g.R[i], INTER_NEAREST, BORDER_CONSTANT, full_masks[i]);
        sd.put(cam(i, "full_mask"), full_masks[i]);
    }
    // initSeam / updateMask at the seam scale (ocvstitcher.hpp:988-1017, :1218-1243)
    const double swa_d = std::min(1.0, std::sqrt(1e5 / ((double)g.h * g.w)));
    Ptr<RotationWarper> sw = make_warper(g.kind, static_cast<float>(g.scale * swa_d));
    std::vector<UMat> images_warped(n), masks_warped(n), images_warped_f(n);
    std::vector<Point> scorners(n);
    for (int i = 0; i < n; i++) {
        Mat seamSized;
        resize(g.frames[i], seamSized, Size(), swa_d, swa_d, INTER_LINEAR_EXACT);
        sd.put(cam(i, "seam_frame"), seamSized);
        Mat_<float> K = g.K[i].clone();
        const float swa = (float)swa_d;
        K(0, 0) *= swa; K(0, 2) *= swa; K(1, 1) *= swa; K(1, 2) *= swa;
        Mat sx, sy;
        sw->buildMaps(seamSized.size(), K, g.R[i], sx, sy);
        sd.put(cam(i, "seam_xmap"), sx); sd.put(cam(i, "seam_ymap"), sy);
        UMat m; m.create(seamSized.size(), CV_8U); m.setTo(Scalar::all(255));
        scorners[i] = sw->warp(seamSized, K, g.R[i], INTER_LINEAR, BORDER_REFLECT, images_warped[i]);
        sw->warp(m, K, g.R[i], INTER_NEAREST, BORDER_CONSTANT, masks_warped[i]);
        images_warped[i].convertTo(images_warped_f[i], CV_32F);
        sd.put(cam(i, "seam_warp"), images_warped[i]); sd.put(cam(i, "seam_mask_warp"), masks_warped[i]);
    }
    Mat_<int> sc(n, 2);
    for (int i = 0; i < n; i++) { sc(i, 0) = scorners[i].x; sc(i, 1) = scorners[i].y; }
    sd.put("seam_corners", sc);
    bool need_gains = false;
    for (const Run& r : runs) need_gains |= r.gains;
    Ptr<ExposureCompensator> comp;
    if (need_gains) {   // fed BEFORE the seam finder touches the masks (ocvstitcher.hpp:1031-1032)
        comp = ExposureCompensator::createDefault(ExposureCompensator::GAIN_BLOCKS);
        comp->feed(scorners, images_warped, masks_warped);
    }
    // both seam finders on copies of the NEAREST masks; dilate, resize, AND (ocvstitcher.hpp:1033-1035, :1097-1101)
    std::vector<Mat> blend[2];   // [seam kind: 0 graph cut, 1 Voronoi]
    const char* kname[2] = {"graphcut", "voronoi"};
    for (int kind = 1; kind >= 0; kind--) {
        std::vector<UMat> mv(n);
        for (int i = 0; i < n; i++) masks_warped[i].copyTo(mv[i]);
        Ptr<SeamFinder> finder;
        if (kind == 0) finder = makePtr<detail::GraphCutSeamFinder>(GraphCutSeamFinderBase::COST_COLOR);
        else finder = makePtr<detail::VoronoiSeamFinder>();
        finder->find(images_warped_f, scorners, mv);
        blend[kind].resize(n);
        for (int i = 0; i < n; i++) {
            sd.put(cam(i, (std::string(kname[kind]) + "_seam_mask").c_str()), mv[i]);
            Mat dilated, seam_mask;
            dilate(mv[i], dilated, Mat());
            resize(dilated, seam_mask, full_masks[i].size(), 0, 0, INTER_LINEAR_EXACT);
            blend[kind][i] = seam_mask & full_masks[i];
            sd.put(cam(i, (std::string(kname[kind]) + "_blend_mask").c_str()), blend[kind][i]);
        }
    }
    std::vector<Mat> gain_warps;
    if (need_gains) {
        BlocksGainCompensator* bc = dynamic_cast<BlocksGainCompensator*>(comp.get());
        gain_warps.resize(n);
        for (int i = 0; i < n; i++) {
            sd.put(cam(i, "gain_map"), bc->gain_maps_[i]);
            gain_warps[i] = warps[i].clone();
            comp->apply(i, corners[i], gain_warps[i], full_masks[i]);   // src/stitching_detailed.cpp:841
            sd.put(cam(i, "warp_gain"), gain_warps[i]);
        }
    }
    // the pyramid primitives by themselves, on real data
    {
        Mat u_in, down, up;
        warps[0].convertTo(u_in, CV_16S);
        pyrDown(u_in, down);
        pyrUp(down, up);
        sd.put("unit/pyrdown16s_in", u_in); sd.put("unit/pyrdown16s_out", down); sd.put("unit/pyrup16s_out", up);
        Mat wm, nx;
        blend[runs[0].seam][0].convertTo(wm, CV_32F, 1. / 255.);   // MultiBandBlender::feed's weight map (blenders.cpp)
        sd.put("unit/pyrdown32f_in", wm);
        for (int l = 1; l <= 3; l++) {
            pyrDown(wm, nx);
            sd.put("unit/pyrdown32f_l" + std::to_string(l), nx);
            wm = nx.clone();
        }
    }
    // process() per run (ocvstitcher.hpp:1141-1216)
    std::vector<Mat> panos;
    for (const Run& r : runs) {
        const std::string pre = "blend_" + r.tag + "/";
        const Rect full = resultRoi(corners, sizes);
        int nb = r.bands;
        if (r.bands == -2) {
            const float blend_width = std::sqrt(static_cast<float>(full.size().area())) * r.strength / 100.f;
            nb = blend_width < 1.f ? -1 : static_cast<int>(std::ceil(std::log(blend_width) / std::log(2.)) - 1.);
        }
        Ptr<Blender> blender;
        if (nb < 0) blender = Blender::createDefault(Blender::NO, false);
        else blender = makePtr<MultiBandBlender>(false, nb);
        blender->prepare(corners, sizes);
        for (int i = 0; i < n; i++) {
            Mat s16;
            (r.gains ? gain_warps[i] : warps[i]).convertTo(s16, CV_16S);
            blender->feed(s16, blend[r.seam][i], corners[i]);
        }
        if (r.levels && nb >= 0) {
            MultiBandBlender* mb = dynamic_cast<MultiBandBlender*>(blender.get());
            for (int l = 0; l <= mb->numBands(); l++) {
                sd.put(pre + "laplace_l" + std::to_string(l), mb->dst_pyr_laplace_[l]);
                sd.put(pre + "weights_l" + std::to_string(l), mb->dst_band_weights_[l]);
            }
        }
        Mat result, result_mask, pano;
        blender->blend(result, result_mask);
        sd.put(pre + "result", result); sd.put(pre + "result_mask", result_mask);
        result.convertTo(pano, CV_8U);
        if (r.cut.size() == 4) pano = pano(Rect(r.cut[0], r.cut[1], r.cut[2], r.cut[3])).clone();
        sd.put(pre + "pano", pano);
        panos.push_back(pano);
    }
    sd.finish();
    std::cout << name << ": " << "raw stages written\n";
    return panos;
}

// config 1 / 1b: 4 x 480 x 270 under one shared K (tests/golden/make_golden.py group_480)
void group_480(const std::string& in, const std::string& out, const std::string& prefix, bool all_bands) {
    const std::string js = slurp(in + "/" + prefix + "_cams.json");
    const std::vector<double> K = numbers_after(js, "K"), R = numbers_after(js, "R"), sc = numbers_after(js, "scale");
    Rig g; g.n = 4; g.w = 480; g.h = 270; g.kind = 0; g.scale = (float)sc[0];
    for (int i = 0; i < 4; i++) {
        g.K.push_back(mat3(K.data())); g.R.push_back(mat3(R.data() + 9 * i));
        g.frames.push_back(imread(in + "/" + prefix + "_cam" + std::to_string(i) + ".png", IMREAD_COLOR));
        if (g.frames.back().empty()) { std::cerr << "missing frame\n"; std::exit(2); }
    }
    {   // every stage as raw data (tests/pin_stages.py load_groups: the same runs under the same tags)
        std::vector<Run> runs;
        if (all_bands) {
            runs.push_back(Run{"bNO", 1, -1, 0.f, false, {}, false});
            runs.push_back(Run{"b0", 1, 0, 0.f, false, {}, false});
            runs.push_back(Run{"b2", 1, 2, 0.f, false, {}, false});
        }
        runs.push_back(Run{"b4", 1, 4, 0.f, false, {}, true});
        if (all_bands) runs.push_back(Run{"b2cut", 1, 2, 0.f, false, {100, 20, 1000, 200}, false});
        runs.push_back(Run{"gc4", 0, 4, 0.f, false, {}, false});
        runs.push_back(Run{"gain4", 1, 4, 0.f, true, {}, false});
        dump_group(out, prefix, g, runs);
    }
    Json j;
    std::vector<Mat> vor = blend_masks(g, 1);
    std::vector<Rect> rois; std::vector<Mat> warps;
    std::vector<std::string> wsha, msha;
    std::ostringstream panos; panos << "{";
    const int bands_all[4] = {-1, 0, 2, 4};
    bool firstp = true;
    for (int b = 0; b < 4; b++) {
        const int nb = bands_all[b];
        if (!all_bands && nb != 4) continue;
        warps.clear();
        Mat p = process(g, vor, nb, 0.f, nullptr, {}, nullptr, &rois, nullptr, &warps);
        panos << (firstp ? "" : ", ") << "\"" << nb << "\": \"" << sha(p) << "\""; firstp = false;
        save(out, prefix + "_pano_b" + std::to_string(nb), p);
        if (nb == 4) j.raw("pano_size", "[" + std::to_string(p.cols) + ", " + std::to_string(p.rows) + "]");
    }
    panos << "}";
    for (int i = 0; i < 4; i++) {
        wsha.push_back(sha(warps[i])); msha.push_back(sha(vor[i]));
        save(out, prefix + "_warp" + std::to_string(i), warps[i]); save(out, prefix + "_voronoi_mask" + std::to_string(i), vor[i]);
    }
    j.raw("rois", rect_list(rois));
    j.list("warp_sha256", wsha, true);
    j.list("mask_sha256", msha, true);
    j.raw("pano_sha256", panos.str());
    if (all_bands) {
        const std::vector<int> cut = {100, 20, 1000, 200};
        j.list("cut", cut, false);
        j.str("pano_cut_sha256", sha(process(g, vor, 2, 0.f, nullptr, cut)));
    }
    Ptr<ExposureCompensator> comp;
    std::vector<Mat> gc = blend_masks(g, 0, &comp);
    std::vector<std::string> gsha;
    for (int i = 0; i < 4; i++) { gsha.push_back(sha(gc[i])); save(out, prefix + "_graphcut_mask" + std::to_string(i), gc[i]); }
    j.list("graphcut_mask_sha256", gsha, true);
    Mat pg = process(g, gc, 4, 0.f, nullptr, {});
    j.str("graphcut_pano_b4_sha256", sha(pg)); save(out, prefix + "_graphcut_pano_b4", pg);
    // the block gain maps as BlocksGainCompensator::feed leaves them (one f32 per 32 x 32 block, smoothed): raw f32 bytes
    BlocksGainCompensator* bc = dynamic_cast<BlocksGainCompensator*>(comp.get());
    std::vector<std::string> gmsha; std::ostringstream shapes; shapes << "[";
    for (int i = 0; i < 4; i++) {
        Mat gm = bc->gain_maps_[i].getMat(ACCESS_READ).clone();
        gmsha.push_back(sha(gm));
        shapes << (i ? ", " : "") << "[" << gm.rows << ", " << gm.cols << "]";
        std::ofstream f((out + "/" + prefix + "_gain_map" + std::to_string(i) + ".f32").c_str(), std::ios::binary);
        for (int y = 0; y < gm.rows; y++) f.write((const char*)gm.ptr<float>(y), (std::streamsize)gm.cols * 4);
    }
    shapes << "]";
    j.raw("gain_map_shape", shapes.str());
    j.list("gain_map_sha256", gmsha, true);
    Mat pe = process(g, vor, 4, 0.f, comp.get(), {});
    j.str("gain_pano_b4_sha256", sha(pe)); save(out, prefix + "_gain_pano_b4", pe);
    std::ofstream f((out + "/" + prefix + "_golden.json").c_str());
    f << "{\n" << j.o.str() << "\n}\n";
    std::cout << prefix << ": done\n";
}

// rig R / S: what replay.cpp does with 2222/4cam (tests/golden/make_golden.py rig): frames 0,1 -> stitcher 0, 2,3 -> stitcher 1
void rig(const std::string& in, const std::string& out, const std::string& prefix) {
    const std::string js = slurp(in + "/" + prefix + "_cams.json");
    const int W = (int)numbers_after(js, "width")[0], H = (int)numbers_after(js, "height")[0];
    std::ostringstream st; st << "[";
    std::vector<Mat> halves, raw_halves;
    size_t pos = js.find("\"stitchers\"");
    for (int s = 0; s < 2; s++) {
        size_t e1 = 0, e2 = 0;
        const std::vector<double> v = numbers_after(js, "cams", pos, &e1);
        const std::vector<double> cutd = numbers_after(js, "cut", pos, &e2);
        pos = std::max(e1, e2);
        Rig g; g.n = 2; g.w = W; g.h = H; g.kind = 0; g.scale = (float)v.back();
        for (int i = 0; i < 2; i++) {
            g.K.push_back(mat3(v.data() + 18 * i)); g.R.push_back(mat3(v.data() + 18 * i + 9));
            g.frames.push_back(imread(in + "/" + prefix + "_cam" + std::to_string(2 * s + i) + ".png", IMREAD_COLOR));
            if (g.frames.back().empty()) { std::cerr << "missing frame\n"; std::exit(2); }
        }
        const std::vector<int> cut = {(int)cutd[0], (int)cutd[1], (int)cutd[2], (int)cutd[3]};
        raw_halves.push_back(dump_group(out, prefix + std::to_string(s), g, {Run{"rig", 0, -2, 1.0f, false, cut, true}})[0]);
        std::vector<Mat> gc = blend_masks(g, 0);
        int bands = 0; std::vector<Rect> rois; Rect full;
        Mat pano = process(g, gc, -2, 1.0f, nullptr, cut, &bands, &rois, &full);   // stitcherBlenderStrength: 1 (cfg/stitcher-imx390cfg.yaml:49)
        halves.push_back(pano);
        save(out, prefix + "_pano_cut" + std::to_string(s), pano);
        st << (s ? ", " : "") << "{\"rois\": " << rect_list(rois) << ", \"pano_roi\": [" << full.x << ", " << full.y << ", " << full.width << ", "
           << full.height << "], \"bands\": " << bands << ", \"cut\": [" << cut[0] << ", " << cut[1] << ", " << cut[2] << ", " << cut[3]
           << "], \"graphcut_mask_sha256\": [\"" << sha(gc[0]) << "\", \"" << sha(gc[1]) << "\"], \"pano_cut_sha256\": \"" << sha(pano)
           << "\", \"pano_cut_size\": [" << pano.cols << ", " << pano.rows << "]}";
    }
    st << "]";
    // master.cpp:321-326
    Mat up, ret;
    cv::resize(halves[0], up, halves[1].size());
    cv::vconcat(up, halves[1], ret);
    cv::rectangle(ret, cv::Rect(0, ret.rows / 2 - 5, ret.cols, 10), cv::Scalar(0, 0, 0), -1);
    save(out, prefix + "_stacked", ret);
    {
        Mat up2, ret2;
        cv::resize(raw_halves[0], up2, raw_halves[1].size());
        cv::vconcat(up2, raw_halves[1], ret2);
        cv::rectangle(ret2, cv::Rect(0, ret2.rows / 2 - 5, ret2.cols, 10), cv::Scalar(0, 0, 0), -1);
        StageDir sd(out, prefix + "_stack");
        sd.put("stacked", ret2);
        sd.finish();
    }
    Json j;
    j.raw("stitchers", st.str());
    j.str("stack_master_sha256", sha(ret));
    j.raw("stack_master_size", "[" + std::to_string(ret.cols) + ", " + std::to_string(ret.rows) + "]");
    std::ofstream f((out + "/" + prefix + "_golden.json").c_str());
    f << "{\n" << j.o.str() << "\n}\n";
    std::cout << prefix << ": done\n";
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: pin <tests/golden> <out dir, e.g. tests/golden/opencv>\n"; return 2; }
    const std::string in = argv[1], out = argv[2];
    std::cout << "OpenCV " << CV_VERSION << "\n";
    {
        std::ofstream f((out + "/VERSION.txt").c_str());
        if (!f) { std::cerr << "cannot write into " << out << " (create the directory first)\n"; return 2; }
        f << "OpenCV " << CV_VERSION << "\n" << cv::getBuildInformation();
    }
    group_480(in, out, "c1", true);
    group_480(in, out, "c1b", false);
    rig(in, out, "r");
    rig(in, out, "s");
    return 0;
}
