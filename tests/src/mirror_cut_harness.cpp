// mirror_cut_harness.cpp - test infrastructure: pano::Stitcher's cut rule (m_cutParams) per init mode, in plan mode (no GPU).
// Reference: include/ocvstitcher.hpp:333-337 (init(yaml) loads `cut` in every mode), :627-631 (mode 3 -> initCamParams ->
// initSeam, cut untouched), :959-964 (only a successful initAll rewrites it to [0, (rows - cut_h) / 2, cols, cut_h]).
//   mirror_cut_harness <stitcher-cfg.yaml> <id> plain            calibration(imgs)
//   mirror_cut_harness <stitcher-cfg.yaml> <id> estimated <deg>  calibration(imgs, K_est, R_est, scale): the defaults with every
//                                                                camera yawed by <deg> degrees - a stand-in for a caller's BA
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../img-stitching_amd/csrc/stitcher.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    pano::Stitcher st;
    st.device = -1;
    const int id = atoi(argv[2]);
    if (st.init(argv[1], id) != pano::RET_OK) { printf("init failed\n"); return 1; }
    std::vector<pano::Mat> imgs;  // plan mode: no frames are read
    int rc;
    if (std::string(argv[3]) == "estimated") {
        // the defaults as a plan-only context parses them, then perturbed
        pano::Stitcher probe;
        probe.device = -1;
        if (probe.init(argv[1], id) != pano::RET_OK || probe.calibration(imgs) != pano::RET_OK) return 1;
        const int n = st.config().num_images;
        std::vector<float> K(9 * n), R(9 * n);
        float scale = 0.f;
        const double a = atof(argc > 4 ? argv[4] : "0") * 3.14159265358979323846 / 180.0;
        const float c = (float)cos(a), s = (float)sin(a);
        for (int i = 0; i < n; i++) {
            float r[9];
            pano_get_camera(probe.handle(), i, &K[9 * i], r, &scale);
            const float ry[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
            for (int y = 0; y < 3; y++)
                for (int x = 0; x < 3; x++) R[9 * i + 3 * y + x] = ry[3 * y] * r[x] + ry[3 * y + 1] * r[3 + x] + ry[3 * y + 2] * r[6 + x];
        }
        rc = st.calibration(imgs, K.data(), R.data(), scale);
    } else {
        rc = st.calibration(imgs);
    }
    if (rc != pano::RET_OK) { printf("calibration RET_ERR: %s\n", st.lastError()); return 1; }
    int w = 0, h = 0, r[4];
    float K[9], scale = 0.f;
    pano_get_output_size(st.handle(), &w, &h);
    pano_get_pano_rect(st.handle(), r);
    pano_get_camera(st.handle(), 0, K, nullptr, &scale);
    printf("pano %dx%d output %dx%d fx0 %.6g scale %.6g\n", r[2], r[3], w, h, K[0], scale);
    return 0;
}
