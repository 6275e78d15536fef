// pano_cameras.cpp - camera parameters of the C-ABI (include/pano.h): validation at the boundary, verifyCamParams, the 18 N + 1 list
// of cameras.yaml, the cameraparaout_<id>.txt reader and writer (reference include/ocvstitcher.hpp:365-562).  Host code only.

#include "pano_ctx.hpp"

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

extern "C" {

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

}  // extern "C"
