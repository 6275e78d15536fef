// replay.cpp - the reference's replay/master main loop (src/replay.cpp:192-404, src/master.cpp:258-326)
// on top of pano::Stitcher: two stitchers (upper / lower camera group), one thread each per frame, the two
// half-panoramas stacked with a black divider.  No OpenCV: frames are binary PPM (P6) files or synthetic.
//
//   g++ -O2 -std=c++17 examples/replay.cpp -o replay -Limg-stitching_amd -lpano_hip -Wl,-rpath,$PWD/img-stitching_amd -lpthread
//   ./replay <stitcher-cfg.yaml> [--plan] [--exposure] [--voronoi] [--frames N] [--fps F] [--refresh-every N] [--async-refresh]
//            [up0.ppm up1.ppm ... down0.ppm ...]
//
// --plan: geometry only (no GPU): prints the panorama size of both stitchers and exits.
// --fps F: the capture loop of src/master.cpp:302-411 paced at F frames per second (a tick the loop reaches more than a period late is
//            a dropped frame); --refresh-every N: updateMask every N frames (the reference: 200, ocvstitcher.hpp:1152-1159) - inside
//            process() like the reference, or with --async-refresh beside the loop (pano::Stitcher::asyncMaskRefresh)
// --exposure: estimate block gains in calibration() and apply them in process() (the reference estimates, ocvstitcher.hpp:1031-1032,
//             but leaves apply commented out, :1178).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

#include "../img-stitching_amd/csrc/stitcher.hpp"

static bool read_ppm(const std::string& path, pano::Mat& m) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    int w = 0, h = 0, mx = 0;
    if (fscanf(f, "P6 %d %d %d", &w, &h, &mx) != 3 || mx != 255) { fclose(f); return false; }
    fgetc(f);
    m.create(h, w);
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    size_t n = fread(rgb.data(), 1, rgb.size(), f);
    fclose(f);
    if (n != rgb.size()) return false;
    for (size_t i = 0; i < (size_t)w * h; i++) {  // RGB file -> BGR like cv::imread
        m.data[i * 3] = rgb[i * 3 + 2]; m.data[i * 3 + 1] = rgb[i * 3 + 1]; m.data[i * 3 + 2] = rgb[i * 3];
    }
    return true;
}
static void write_ppm(const std::string& path, const pano::Mat& m) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return;
    fprintf(f, "P6\n%d %d\n255\n", m.cols, m.rows);
    std::vector<uint8_t> row((size_t)m.cols * 3);
    for (int y = 0; y < m.rows; y++) {
        const uint8_t* s = m.data + (size_t)y * m.step;
        for (int x = 0; x < m.cols; x++) { row[x * 3] = s[x * 3 + 2]; row[x * 3 + 1] = s[x * 3 + 1]; row[x * 3 + 2] = s[x * 3]; }
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
}
static void synthetic(pano::Mat& m, int w, int h, int seed) {
    m.create(h, w);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t* p = m.data + (size_t)y * m.step + x * 3;
            p[0] = (uint8_t)(x * 255 / w); p[1] = (uint8_t)(y * 255 / h); p[2] = (uint8_t)((((x >> 5) + (y >> 5) + seed) & 1) * 200);
        }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: replay <stitcher-cfg.yaml> [--plan] [--exposure] [--voronoi] [--frames N] [ppm files]\n"); return 2; }
    std::string cfg = argv[1];
    bool plan = false, exposure = false, voronoi = false, async_refresh = false;
    int nframes = 3, refresh_every = 0;
    double fps = 0.0;
    std::vector<std::string> files;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--plan") plan = true;
        else if (a == "--exposure") exposure = true;
        else if (a == "--voronoi") voronoi = true;
        else if (a == "--frames" && i + 1 < argc) nframes = atoi(argv[++i]);
        else if (a == "--fps" && i + 1 < argc) fps = atof(argv[++i]);
        else if (a == "--refresh-every" && i + 1 < argc) refresh_every = atoi(argv[++i]);
        else if (a == "--async-refresh") async_refresh = true;
        else files.push_back(a);
    }
    pano::Stitcher st[2];
    for (int s = 0; s < 2; s++) {
        if (plan) st[s].device = -1;
        st[s].exposureCompensation = exposure;
        if (voronoi) st[s].seamFinder = pano::Stitcher::SeamVoronoi;  // default: graph cut, like the reference
        if (refresh_every > 0) st[s].maskRefreshPeriod = refresh_every;
        st[s].asyncMaskRefresh = async_refresh;
        if (st[s].init(cfg, s) != pano::RET_OK) { fprintf(stderr, "stitcher %d init failed\n", s); return 1; }
    }
    const int n = st[0].config().num_images, W = st[0].config().width, H = st[0].config().height;
    std::vector<pano::Mat> imgs[2];
    for (int s = 0; s < 2; s++)
        for (int i = 0; i < n; i++) {
            pano::Mat m;
            size_t k = (size_t)s * n + i;
            if (k >= files.size() || !read_ppm(files[k], m) || m.cols != W || m.rows != H) synthetic(m, W, H, (int)k);
            imgs[s].push_back(m);
        }
    for (int s = 0; s < 2; s++)
        if (st[s].calibration(imgs[s]) != pano::RET_OK) { fprintf(stderr, "stitcher %d calibration failed: %s\n", s, st[s].lastError()); return 1; }
    for (int s = 0; s < 2; s++) {
        int w = 0, h = 0, nb = 0, r[4];
        pano_get_output_size(st[s].handle(), &w, &h);
        pano_get_num_bands(st[s].handle(), &nb);
        pano_get_pano_rect(st[s].handle(), r);
        printf("stitcher %d: pano %dx%d at (%d,%d), output %dx%d, bands %d\n", s, r[2], r[3], r[0], r[1], w, h, nb);
        int gw = 0, gh = 0;
        if (!plan && pano_get_gain_map(st[s].handle(), 0, nullptr, &gw, &gh) == PANO_OK && gw > 0)
            printf("stitcher %d: exposure gain maps %dx%d blocks\n", s, gw, gh);
    }
    if (plan) return 0;
    pano::Mat out[2];
    if (fps > 0.0) {
        // the paced capture loop: a frame set "arrives" every 1 / fps seconds and both stitchers process it on a thread each
        using clk = std::chrono::steady_clock;
        const double period = 1.0 / fps;
        std::vector<double> lat;
        int dropped = 0;
        const auto t0 = clk::now() + std::chrono::milliseconds(10);
        for (int f = 0; f < nframes; f++) {
            const auto target = t0 + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(f * period));
            if (clk::now() > target + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(period))) {
                dropped++;  // this tick is over before the loop got here
                continue;
            }
            std::this_thread::sleep_until(target);
            std::thread t1(&pano::Stitcher::process, &st[0], std::ref(imgs[0]), std::ref(out[0]));
            std::thread t2(&pano::Stitcher::process, &st[1], std::ref(imgs[1]), std::ref(out[1]));
            t1.join();
            t2.join();
            lat.push_back(std::chrono::duration<double, std::milli>(clk::now() - target).count());
        }
        std::sort(lat.begin(), lat.end());
        const double total = std::chrono::duration<double>(clk::now() - t0).count();
        if (!lat.empty())
            printf("paced at %.1f fps: %d frames offered, %d composed, dropped %d, achieved %.2f fps, latency ms p50 %.3f p99 %.3f max %.3f, "
                   "mask refresh every %d frames %s\n", fps, nframes, (int)lat.size(), dropped, lat.size() / total, lat[lat.size() / 2],
                   lat[std::min(lat.size() - 1, lat.size() * 99 / 100)], lat.back(), st[0].maskRefreshPeriod,
                   async_refresh ? "beside the loop" : "inside process()");
    } else
    for (int f = 0; f < nframes; f++) {
        auto t0 = std::chrono::steady_clock::now();
        std::thread t1(&pano::Stitcher::process, &st[0], std::ref(imgs[0]), std::ref(out[0]));  // master.cpp:314-318
        std::thread t2(&pano::Stitcher::process, &st[1], std::ref(imgs[1]), std::ref(out[1]));
        t1.join();
        t2.join();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("frame %d: stitching takes %.3f ms\n", f, ms);
    }
    if (out[0].empty() || out[1].empty()) { fprintf(stderr, "process failed: %s / %s\n", st[0].lastError(), st[1].lastError()); return 1; }
    // master.cpp:321-326: cv::resize(up -> down.size()), vconcat, black 10-row bar - the library's bit-exact twin of it
    pano::Mat fin(2 * out[1].rows, out[1].cols);
    if (pano_stack_master_host(st[1].handle(), out[0].data, out[0].cols, out[0].rows, out[0].step, out[1].data, out[1].cols,
                               out[1].rows, out[1].step, fin.data, fin.step) != PANO_OK) {
        fprintf(stderr, "pano_stack_master_host: %s\n", st[1].lastError());
        return 1;
    }
    write_ppm("final.ppm", fin);
    printf("wrote final.ppm %dx%d\n", fin.cols, fin.rows);
    return 0;
}
