// sharded_replay.cpp - one stitcher's cameras spread over W ranks (one process per GPU): every rank warps and
// builds the pyramids of ITS cameras (pano_feed_cameras_host), ONE grouped RCCL exchange lands their pyramid slots on rank 0
// (pano_gather_slots), rank 0 blends (pano_blend_host).  The panorama equals the single-GPU one bit for bit.
// Plain C++ against the C-ABI - no HIP, RCCL or OpenCV headers; the ranks find each other through a file that carries the
// 128-byte RCCL unique id (any channel the application already has will do).
//
//   g++ -O2 -std=c++17 examples/sharded_replay.cpp -o sharded_replay -Iinclude -Limg-stitching_amd -lpano_hip -Wl,-rpath,$PWD/img-stitching_amd
//   ./sharded_replay 0 2 /tmp/pano.id &  ./sharded_replay 1 2 /tmp/pano.id        (rank r uses GPU r)
//   ./sharded_replay --single          the same panorama on one GPU, for comparison of the checksum
//   ... <frames> --device D            every rank on GPU D: a rehearsal on one GPU (RCCL itself refuses that; PANO_RCCL_LIB
//                                      can name a library that does not, e.g. the test double of tests/src/fake_rccl.cpp)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pano.h"

static const int W = 1920, H = 1080, N = 4;

static pano_ctx* make_ctx(int device) {
    const float f = 1002.416f, yaw[4] = {67.5f, 22.5f, -22.5f, -67.5f};   // config 2, one group
    pano_config c{};
    c.num_images = N; c.width = W; c.height = H; c.projector = PANO_SPHERICAL; c.warped_image_scale = f; c.num_bands = 5;
    c.device = device;
    pano_ctx* ctx = nullptr;
    if (pano_create(&c, &ctx) != PANO_OK) return nullptr;
    for (int i = 0; i < N; i++) {
        const float K[9] = {f, 0, W / 2.f, 0, f, H / 2.f, 0, 0, 1};
        const double t = yaw[i] * M_PI / 180.0;
        const float R[9] = {(float)cos(t), 0, (float)sin(t), 0, 1, 0, (float)-sin(t), 0, (float)cos(t)};
        pano_set_camera(ctx, i, K, R);
    }
    if (pano_prepare(ctx) != PANO_OK || pano_build_masks_voronoi(ctx) != PANO_OK) {
        fprintf(stderr, "%s\n", pano_last_error(ctx));
        pano_destroy(ctx);
        return nullptr;
    }
    return ctx;
}
static void fill(std::vector<uint8_t>& fr, int cam, int frame) {
    fr.resize((size_t)W * H * 3);
    for (size_t k = 0; k < fr.size(); k++) fr[k] = (uint8_t)((k * 7 + cam * 31 + frame * 13) >> 3);
}
static unsigned checksum(const std::vector<uint8_t>& v) {  // position-weighted byte sum mod 2^32
    unsigned s = 0;
    for (size_t k = 0; k < v.size(); k++) s += v[k] * ((unsigned)k * 2654435761u + 1u);
    return s;
}

int main(int argc, char** argv) {
    const bool single = argc > 1 && std::string(argv[1]) == "--single";
    if (!single && argc < 4) {
        fprintf(stderr, "usage: sharded_replay <rank> <world> <id-file> [frames] [--device D]   |   sharded_replay --single [frames]\n");
        return 2;
    }
    const int rank = single ? 0 : atoi(argv[1]), world = single ? 1 : atoi(argv[2]);
    const int frames = argc > (single ? 2 : 4) ? atoi(argv[single ? 2 : 4]) : 3;
    int device = rank;
    for (int a = 1; a + 1 < argc; a++)
        if (std::string(argv[a]) == "--device") device = atoi(argv[a + 1]);
    if (world < 1 || N % world || rank < 0 || rank >= world) { fprintf(stderr, "world must divide %d cameras\n", N); return 2; }
    pano_ctx* ctx = make_ctx(device);
    if (!ctx) return 1;
    // camera c belongs to rank c / (N / world)
    int owner[N];
    unsigned mine = 0;
    for (int c = 0; c < N; c++) {
        owner[c] = c / (N / world);
        if (owner[c] == rank) mine |= 1u << c;
    }
    void* comm = nullptr;
    if (world > 1) {
        char id[PANO_RCCL_ID_BYTES];
        const std::string path = argv[3];
        if (rank == 0) {
            if (pano_rccl_unique_id(id) != PANO_OK) { fprintf(stderr, "no RCCL\n"); return 1; }
            FILE* f = fopen((path + ".tmp").c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 1;
            fclose(f);
            rename((path + ".tmp").c_str(), path.c_str());
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 600 && !(f = fopen(path.c_str(), "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "rank %d: no unique id in %s\n", rank, path.c_str()); return 1; }
            fclose(f);
        }
        if (pano_rccl_comm_create(ctx, id, world, rank, &comm) != PANO_OK) { fprintf(stderr, "rank %d: %s\n", rank, pano_last_error(ctx)); return 1; }
    }
    int ow = 0, oh = 0;
    pano_get_output_size(ctx, &ow, &oh);
    std::vector<uint8_t> fr[N], out((size_t)ow * oh * 3);
    const uint8_t* ptr[N];
    size_t stride[N];
    for (int f = 0; f < frames; f++) {
        for (int c = 0; c < N; c++) {
            if ((mine >> c) & 1u) fill(fr[c], c, f);      // a rank only ever sees its own cameras' frames
            ptr[c] = ((mine >> c) & 1u) ? fr[c].data() : nullptr;
            stride[c] = (size_t)W * 3;
        }
        auto t0 = std::chrono::steady_clock::now();
        if (pano_feed_cameras_host(ctx, mine, ptr, stride) != PANO_OK) { fprintf(stderr, "rank %d feed: %s\n", rank, pano_last_error(ctx)); return 1; }
        if (world > 1 && pano_gather_slots(ctx, comm, rank, 0, owner, nullptr) != PANO_OK) { fprintf(stderr, "rank %d gather: %s\n", rank, pano_last_error(ctx)); return 1; }
        if (rank == 0) {
            if (pano_blend_host(ctx, out.data(), (size_t)ow * 3) != PANO_OK) { fprintf(stderr, "blend: %s\n", pano_last_error(ctx)); return 1; }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("frame %d: %dx%d panorama, checksum %08x, %.3f ms (%d rank%s)\n", f, ow, oh, checksum(out), ms, world, world > 1 ? "s" : "");
        }
    }
    pano_rccl_comm_destroy(comm);
    pano_destroy(ctx);
    return 0;
}
