// test harness: pano::GridMaxFlow (the product's host max-flow, csrc/pano_graphcut.hpp) on grids read from a binary file:
// int32 W, H, then W*H f32 term, wh, wv; repeated until EOF.  Writes W*H label bytes per grid to stdout.
#include <cstdio>
#include <vector>

#include "../../img-stitching_amd/csrc/pano_graphcut.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int wh2[2];
    while (fread(wh2, sizeof(int), 2, f) == 2) {
        const size_t n = (size_t)wh2[0] * wh2[1];
        std::vector<float> term(n), wh(n), wv(n);
        if (fread(term.data(), 4, n, f) != n || fread(wh.data(), 4, n, f) != n || fread(wv.data(), 4, n, f) != n) return 3;
        pano::GridMaxFlow g(wh2[0], wh2[1], term.data(), wh.data(), wv.data());
        g.run();
        std::vector<unsigned char> lab(n);
        for (size_t k = 0; k < n; k++) lab[k] = g.inSource((int)k) ? 1 : 0;
        fwrite(lab.data(), 1, n, stdout);
    }
    fclose(f);
    return 0;
}
