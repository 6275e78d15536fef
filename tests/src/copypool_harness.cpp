// the copy-thread pool of pano_compose_host (csrc/pano_hostcopy.hpp) as a plain C++ unit under ThreadSanitizer: several caller
// threads (the reference calls process() of its two stitchers from two threads, src/master.cpp:314-318) push batches of strided
// row copies through the one process-wide pool at once; every destination must hold its source's bytes.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../img-stitching_amd/csrc/pano_hostcopy.hpp"

int main() {
    pano::CopyPool& pool = pano::CopyPool::instance();
    int bad = 0;
    auto caller = [&](unsigned seed) {
        for (int round = 0; round < 12; round++) {
            const int jobs = 1 + (int)(seed % 4);
            std::vector<std::vector<uint8_t>> src(jobs), dst(jobs);
            std::vector<size_t> width(jobs), spitch(jobs), dpitch(jobs);
            std::vector<int> rows(jobs);
            std::vector<pano::CopyPool::Latch> latch(jobs);
            for (int j = 0; j < jobs; j++) {
                seed = seed * 1664525u + 1013904223u;
                width[j] = 1 + seed % 5000; rows[j] = 1 + (seed >> 16) % 300;
                spitch[j] = width[j] + (seed >> 8) % 17; dpitch[j] = width[j] + (seed >> 12) % 13;
                src[j].resize(spitch[j] * rows[j]); dst[j].assign(dpitch[j] * rows[j], 0xee);
                for (size_t k = 0; k < src[j].size(); k++) src[j][k] = (uint8_t)(k * 31 + j + seed);
                pool.submit(latch[j], dst[j].data(), dpitch[j], src[j].data(), spitch[j], width[j], rows[j]);
            }
            for (int j = 0; j < jobs; j++) pool.wait(latch[j]);
            for (int j = 0; j < jobs; j++)
                for (int y = 0; y < rows[j]; y++)
                    for (size_t x = 0; x < width[j]; x++)
                        if (dst[j][y * dpitch[j] + x] != src[j][y * spitch[j] + x]) { bad++; break; }
            // and the blocking form
            std::vector<uint8_t> a(700000), b(700000, 0);
            for (size_t k = 0; k < a.size(); k++) a[k] = (uint8_t)(k ^ seed);
            pool.copy2d(b.data(), 7000, a.data(), 7000, 7000, 100);
            if (a != b) bad++;
        }
    };
    std::thread t1(caller, 1u), t2(caller, 2u), t3(caller, 3u);
    t1.join(); t2.join(); t3.join();
    printf("threads %d bad %d\n", pool.threads(), bad);
    return bad ? 1 : 0;
}
