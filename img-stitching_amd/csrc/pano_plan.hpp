// pano_plan.hpp - host-side geometry of the compose path (init-time, exact f32 like the reference's
// OpenCV calls): projector matrices, warp ROI, feed() tile boxes, band rule, trig tables.
//
// Replaces, on the host, what ocvStitcher computes through OpenCV at init (reference
// include/ocvstitcher.hpp:1054-1063 warpRoi, :1110 resultRoi, :1186-1198 band rule + prepare) and
// what RotationWarper::warp re-derives every frame (detectResultRoi + buildMaps, :1171).  The
// per-pixel trigonometry of mapBackward is separable (sin/cos of u/scale per column, of v/scale per
// row), so it is tabulated here once with the host libm - the same sinf/cosf an OpenCV CPU build
// calls - and the kernels only multiply.
//
// Must be compiled with -ffp-contract=off (f32 expressions are in OpenCV's evaluation order).
#pragma once

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pano {

constexpr int kMaxCams = 8;
constexpr int kMaxLevels = 9;  // bands 0..8 -> levels 0..8

struct Rect {
    int x = 0, y = 0, w = 0, h = 0;
};

// cv::detail::ProjectorBase + Spherical/CylindricalProjector (stitching/detail/warpers_inl.hpp)
struct Projector {
    int kind = 0;  // 0 spherical, 1 cylindrical
    float scale = 1.f;
    float k[9], rinv[9], r_kinv[9], k_rinv[9];

    static void invert3x3(const float* S, float* D) {
        // cv::invert 3x3 CV_32F: cofactors in double, rounded to float
        auto s = [&](int r, int c) { return (double)S[r * 3 + c]; };
        double det = S[0] * (s(1, 1) * S[8] - s(1, 2) * S[7]) - S[1] * (s(1, 0) * S[8] - s(1, 2) * S[6]) +
                     S[2] * (s(1, 0) * S[7] - s(1, 1) * S[6]);
        if (det == 0.) {
            for (int i = 0; i < 9; i++) D[i] = 0.f;
            return;
        }
        double d = 1. / det;
        D[0] = (float)((s(1, 1) * S[8] - s(1, 2) * S[7]) * d);
        D[1] = (float)((s(0, 2) * S[7] - s(0, 1) * S[8]) * d);
        D[2] = (float)((s(0, 1) * S[5] - s(0, 2) * S[4]) * d);
        D[3] = (float)((s(1, 2) * S[6] - s(1, 0) * S[8]) * d);
        D[4] = (float)((s(0, 0) * S[8] - s(0, 2) * S[6]) * d);
        D[5] = (float)((s(0, 2) * S[3] - s(0, 0) * S[5]) * d);
        D[6] = (float)((s(1, 0) * S[7] - s(1, 1) * S[6]) * d);
        D[7] = (float)((s(0, 1) * S[6] - s(0, 0) * S[7]) * d);
        D[8] = (float)((s(0, 0) * S[4] - s(0, 1) * S[3]) * d);
    }
    static void mul3x3(const float* a, const float* b, float* d) {
        // cv::gemm small-matrix f32 path: products summed left to right in f32
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) d[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    }
    void set(int kind_, float scale_, const float* K, const float* R) {
        kind = kind_;
        scale = scale_;
        float kinv[9];
        std::memcpy(k, K, sizeof(k));
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) rinv[r * 3 + c] = R[c * 3 + r];
        invert3x3(K, kinv);
        mul3x3(R, kinv, r_kinv);
        mul3x3(K, rinv, k_rinv);
    }
    void mapForward(float x, float y, float& u, float& v) const {
        float x_ = r_kinv[0] * x + r_kinv[1] * y + r_kinv[2];
        float y_ = r_kinv[3] * x + r_kinv[4] * y + r_kinv[5];
        float z_ = r_kinv[6] * x + r_kinv[7] * y + r_kinv[8];
        u = scale * atan2f(x_, z_);
        if (kind == 0) {
            float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
            v = scale * ((float)M_PI - acosf(w == w ? w : 0));
        } else {
            v = scale * y_ / sqrtf(x_ * x_ + z_ * z_);
        }
    }
};

// RotationWarperBase::detectResultRoiByBorder + SphericalWarper::detectResultRoi pole fix-up
inline void detectResultRoi(const Projector& p, int w, int h, int tl[2], int br[2]) {
    float tlu = std::numeric_limits<float>::max(), tlv = tlu, bru = -tlu, brv = -tlu;
    auto acc = [&](float x, float y) {
        float u, v;
        p.mapForward(x, y, u, v);
        tlu = (std::min)(tlu, u);
        tlv = (std::min)(tlv, v);
        bru = (std::max)(bru, u);
        brv = (std::max)(brv, v);
    };
    for (int x = 0; x < w; ++x) {
        acc((float)x, 0.f);
        acc((float)x, (float)(h - 1));
    }
    for (int y = 0; y < h; ++y) {
        acc(0.f, (float)y);
        acc((float)(w - 1), (float)y);
    }
    tl[0] = (int)tlu; tl[1] = (int)tlv; br[0] = (int)bru; br[1] = (int)brv;
    if (p.kind != 0) return;
    tlu = (float)tl[0]; tlv = (float)tl[1]; bru = (float)br[0]; brv = (float)br[1];
    for (int pole = 0; pole < 2; pole++) {
        float x = p.rinv[1], y = pole == 0 ? p.rinv[4] : -p.rinv[4], z = p.rinv[7];
        if (y > 0.f) {
            float x_ = (p.k[0] * x + p.k[1] * y) / z + p.k[2];
            float y_ = p.k[4] * y / z + p.k[5];
            if (x_ > 0.f && x_ < w && y_ > 0.f && y_ < h) {
                float pv = pole == 0 ? (float)(M_PI * p.scale) : 0.f;
                tlu = (std::min)(tlu, 0.f); tlv = (std::min)(tlv, pv);
                bru = (std::max)(bru, 0.f); brv = (std::max)(brv, pv);
            }
        }
    }
    tl[0] = (int)tlu; tl[1] = (int)tlv; br[0] = (int)bru; br[1] = (int)brv;
}

// RotationWarperBase::warpRoi
inline Rect warpRoi(const Projector& p, int w, int h) {
    int tl[2], br[2];
    detectResultRoi(p, w, h, tl, br);
    return Rect{tl[0], tl[1], br[0] + 1 - tl[0], br[1] + 1 - tl[1]};
}

// cv::detail::resultRoi(corners, sizes)
inline Rect resultRoi(const Rect* r, int n) {
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; i++) {
        tlx = std::min(tlx, r[i].x); tly = std::min(tly, r[i].y);
        brx = std::max(brx, r[i].x + r[i].w); bry = std::max(bry, r[i].y + r[i].h);
    }
    return Rect{tlx, tly, brx - tlx, bry - tly};
}

// band rule of the callers, ocvstitcher.hpp:1188-1195: -1 = Blender::NO
inline int bandsFromStrength(int w, int h, float strength) {
    float blend_width = std::sqrt(static_cast<float>(w * h)) * strength / 100.f;
    if (blend_width < 1.f) return -1;
    return static_cast<int>(std::ceil(std::log(blend_width) / std::log(2.)) - 1.);
}

// cv::borderInterpolate, BORDER_REFLECT
inline int reflect(int p, int len) {
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p - 1 : 2 * len - 1 - p;
    return p;
}

struct FeedTile {
    Rect rect;  // bordered tile in padded-canvas coordinates (level 0)
    int top = 0, bottom = 0, left = 0, right = 0;
};

// Everything fixed once K, R, scale and the frame size are known.
struct Plan {
    int n = 0, src_w = 0, src_h = 0;
    Projector proj[kMaxCams];
    Rect roi[kMaxCams];   // m_corners / m_sizes
    Rect pano;            // dst_roi_final (warp coordinates)
    Rect canvas;          // dst_roi padded to a multiple of 2^bands
    int bands = 0;        // num_bands_ after the crop in MultiBandBlender::prepare; -1 = Blender::NO
    FeedTile tile[kMaxCams];
    Rect cut;             // in pano coordinates
};

// MultiBandBlender::prepare(Rect) padding + MultiBandBlender::feed tile box (blenders.cpp)
inline bool makePlan(Plan& P, int requested_bands) {
    P.pano = resultRoi(P.roi, P.n);
    P.canvas = P.pano;
    if (requested_bands < 0) {
        P.bands = -1;
        for (int i = 0; i < P.n; i++) {
            P.tile[i].rect = Rect{P.roi[i].x - P.pano.x, P.roi[i].y - P.pano.y, P.roi[i].w, P.roi[i].h};
            P.tile[i].top = P.tile[i].bottom = P.tile[i].left = P.tile[i].right = 0;
        }
        return true;
    }
    double max_len = (double)std::max(P.pano.w, P.pano.h);
    int nb = std::min(requested_bands, (int)std::ceil(std::log(max_len) / std::log(2.0)));
    if (nb >= kMaxLevels) return false;
    P.bands = nb;
    const int m = 1 << nb;
    P.canvas.w += (m - P.canvas.w % m) % m;
    P.canvas.h += (m - P.canvas.h % m) % m;
    const Rect& R = P.canvas;
    for (int i = 0; i < P.n; i++) {
        const Rect& r = P.roi[i];
        int gap = 3 * m;
        int tlx = std::max(R.x, r.x - gap), tly = std::max(R.y, r.y - gap);
        int brx = std::min(R.x + R.w, r.x + r.w + gap), bry = std::min(R.y + R.h, r.y + r.h + gap);
        tlx = R.x + (((tlx - R.x) >> nb) << nb);
        tly = R.y + (((tly - R.y) >> nb) << nb);
        int width = brx - tlx, height = bry - tly;
        width += (m - width % m) % m;
        height += (m - height % m) % m;
        brx = tlx + width;
        bry = tly + height;
        int dy = std::max(bry - (R.y + R.h), 0), dx = std::max(brx - (R.x + R.w), 0);
        tlx -= dx; brx -= dx; tly -= dy; bry -= dy;
        FeedTile& t = P.tile[i];
        t.top = r.y - tly; t.left = r.x - tlx;
        t.bottom = bry - r.y - r.h; t.right = brx - r.x - r.w;
        t.rect = Rect{tlx - R.x, tly - R.y, width, height};
    }
    return true;
}

// cvUndistortPoints without R / P: 5 fixed-point iterations (imgproc/src/undistort.cpp)
inline void undistortPoint(const double K[9], const double d[4], double u, double v, double& ox, double& oy) {
    const double fx = K[0], fy = K[4], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[5];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        double r2 = x * x + y * y;
        double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((0 * r2 + d[1]) * r2 + d[0]) * r2);
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        double deltaX = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x);
        double deltaY = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    ox = x;
    oy = y;
}

// cv::getOptimalNewCameraMatrix(K, dist, size, alpha = 1, size): 9x9 grid of f32 points -> outer rectangle -> the
// projection that maps it onto the viewport (calib3d/src/calibration.cpp; reference nvcam.hpp:830)
inline void optimalNewCameraMatrix(const double K[9], const double dist[4], int w, int h, double newK[9]) {
    const int N = 9;
    float oX0 = std::numeric_limits<float>::max(), oX1 = -oX0, oY0 = oX0, oY1 = -oX0;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            float px = (float)x * w / (N - 1), py = (float)y * h / (N - 1);
            double ux, uy;
            undistortPoint(K, dist, (double)px, (double)py, ux, uy);
            float fx_ = (float)ux, fy_ = (float)uy;
            oX0 = (std::min)(oX0, fx_); oX1 = (std::max)(oX1, fx_);
            oY0 = (std::min)(oY0, fy_); oY1 = (std::max)(oY1, fy_);
        }
    const float ow = oX1 - oX0, oh = oY1 - oY0;
    const double fx1 = (w - 1) / ow, fy1 = (h - 1) / oh;  // int / float: an f32 quotient, as in OpenCV
    for (int i = 0; i < 9; i++) newK[i] = 0;
    newK[0] = fx1; newK[4] = fy1; newK[2] = -fx1 * oX0; newK[5] = -fy1 * oY0; newK[8] = 1;
}

// Separable factors of Spherical/CylindricalProjector::mapBackward on the integer (u, v) grid:
//   spherical:   x_ = sinf(pi - v/s) * sinf(u/s),  y_ = cosf(pi - v/s),  z_ = sinf(pi - v/s) * cosf(u/s)
//   cylindrical: x_ = sinf(u/s),                   y_ = v/s,             z_ = cosf(u/s)
// colA[i] = {sinf(u/s), cosf(u/s)}, rowB[j] = {sinf(pi - v/s) | 1.0f, cosf(pi - v/s) | v/s}.
// idx maps a table slot to a warp coordinate, which is how copyMakeBorder(BORDER_REFLECT) of
// feed() is folded into the tables: slot t of a bordered tile uses u = roi.x + reflect(t - left, roi.w).
inline void trigTables(const Projector& p, const Rect& roi, int left, int top, int tw, int th,
                       std::vector<float>& colA, std::vector<float>& rowB) {
    colA.resize((size_t)tw * 2);
    rowB.resize((size_t)th * 2);
    for (int t = 0; t < tw; t++) {
        float u = (float)(roi.x + reflect(t - left, roi.w));
        u /= p.scale;
        colA[2 * t] = sinf(u);
        colA[2 * t + 1] = cosf(u);
    }
    for (int t = 0; t < th; t++) {
        float v = (float)(roi.y + reflect(t - top, roi.h));
        v /= p.scale;
        if (p.kind == 0) {
            rowB[2 * t] = sinf((float)M_PI - v);
            rowB[2 * t + 1] = cosf((float)M_PI - v);
        } else {
            rowB[2 * t] = 1.0f;
            rowB[2 * t + 1] = v;
        }
    }
}

}  // namespace pano
