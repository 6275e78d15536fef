// host_entry_bench.cpp - pano_compose_host the way the reference drives process(): two stitchers (config 2: 2 x 4 x 1080p,
// 5 bands), one std::thread each per frame, joined (src/master.cpp:314-318); pageable or page-locked caller memory.
//   g++ -O2 -std=c++17 tools/src/host_entry_bench.cpp -o /tmp/heb -Iinclude -Limg-stitching_amd -lpano_hip -Wl,-rpath,$PWD/img-stitching_amd -lpthread
//   /tmp/heb [frames] [pinned] [serial]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pano.h"

int main(int argc, char** argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 200;
    const bool pinned = argc > 2 && atoi(argv[2]);
    const bool serial = argc > 3 && atoi(argv[3]);
    const int W = 1920, H = 1080, N = 4;
    const float f = 1002.416f;
    const float yaw[4] = {67.5f, 22.5f, -22.5f, -67.5f};
    pano_ctx* ctx[2] = {nullptr, nullptr};
    std::vector<uint8_t*> in[2];
    uint8_t* out[2];
    int ow = 0, oh = 0;
    for (int g = 0; g < 2; g++) {
        pano_config c{};
        c.num_images = N; c.width = W; c.height = H; c.projector = PANO_SPHERICAL; c.warped_image_scale = f;
        c.num_bands = 5; c.device = 0;
        if (pano_create(&c, &ctx[g]) != PANO_OK) return 1;
        for (int i = 0; i < N; i++) {
            const float K[9] = {f, 0, W / 2.f, 0, f, H / 2.f, 0, 0, 1};
            const double t = yaw[i] * M_PI / 180.0;
            const float R[9] = {(float)cos(t), 0, (float)sin(t), 0, 1, 0, (float)-sin(t), 0, (float)cos(t)};
            pano_set_camera(ctx[g], i, K, R);
        }
        if (pano_prepare(ctx[g]) != PANO_OK || pano_build_masks_voronoi(ctx[g]) != PANO_OK) { fprintf(stderr, "%s\n", pano_last_error(ctx[g])); return 1; }
        pano_get_output_size(ctx[g], &ow, &oh);
        for (int i = 0; i < N; i++) {
            uint8_t* p = pinned ? (uint8_t*)pano_host_alloc((size_t)W * H * 3) : (uint8_t*)malloc((size_t)W * H * 3);
            for (size_t k = 0; k < (size_t)W * H * 3; k++) p[k] = (uint8_t)(k * 7 + i * 31 + g);
            in[g].push_back(p);
        }
        out[g] = pinned ? (uint8_t*)pano_host_alloc((size_t)ow * oh * 3) : (uint8_t*)malloc((size_t)ow * oh * 3);
        memset(out[g], 0, (size_t)ow * oh * 3);
    }
    size_t strides[4] = {(size_t)W * 3, (size_t)W * 3, (size_t)W * 3, (size_t)W * 3};
    auto one = [&](int g) { pano_compose_host(ctx[g], in[g].data(), strides, out[g], (size_t)ow * 3); };
    auto frame = [&]() {
        if (serial) { one(0); one(1); return; }
        std::thread a(one, 0), b(one, 1);
        a.join(); b.join();
    };
    for (int k = 0; k < 5; k++) frame();
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < frames; k++) frame();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
    unsigned sum = 0;
    for (size_t k = 0; k < (size_t)ow * oh * 3; k += 997) sum += out[0][k] + out[1][k];
    printf("%s caller memory, %s: %.3f ms per panorama = %.1f panoramas/s (%dx%d x2, checksum %u)\n", pinned ? "page-locked" : "pageable",
           serial ? "one thread" : "two threads", ms, 1e3 / ms, ow, oh, sum);
    pano_destroy(ctx[0]);
    pano_destroy(ctx[1]);
    return 0;
}
