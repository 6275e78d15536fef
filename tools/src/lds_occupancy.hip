// How many workgroups does a CU of this device hold at once, as a function of the workgroup's LDS bytes and waves?  Every workgroup
// stamps s_memrealtime at entry and exit around a 20-us spin; the host counts the workgroups resident at the same time and divides by
// the CU count.  hipcc --offload-arch=gfx950 -O2 tools/src/lds_occupancy.hip -o lds_occupancy && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
extern __shared__ unsigned char dyn[];
__global__ void spin(unsigned long long* stamps, unsigned ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) dyn[0] = 1;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    const int cus = p.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"lds_per_cu_reported\": %zu, \"rows\": [\n", p.gcnArchName, cus, (size_t)p.maxSharedMemoryPerMultiProcessor);
    const int sizes[] = {1024, 8192, 12288, 16384, 16385, 17664, 18432, 20480, 20704, 21760, 23040, 24576, 26624, 27648, 32768, 32769, 40960, 53248, 54272, 65536};
    const int threads[] = {256, 320};
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    bool first = true;
    for (int th : threads)
        for (int sz : sizes) {
            const int nwg = cus * 12;
            unsigned long long* d;
            if (hipMalloc(&d, nwg * 16) != hipSuccess) return 1;
            hipMemset(d, 0, nwg * 16);
            hipLaunchKernelGGL(spin, dim3(nwg), dim3(th), sz, 0, d, 2000u);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed at %d\n", sz); return 1; }
            std::vector<unsigned long long> h(2 * nwg);
            hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
            hipFree(d);
            std::vector<std::pair<unsigned long long, int>> ev;
            for (int i = 0; i < nwg; i++) { ev.push_back({h[2 * i], 1}); ev.push_back({h[2 * i + 1], -1}); }
            std::sort(ev.begin(), ev.end());
            int cur = 0, mx = 0;
            for (auto& e : ev) { cur += e.second; mx = std::max(mx, cur); }
            printf("%s {\"threads\": %d, \"lds_bytes\": %d, \"workgroups_resident_max\": %d, \"per_cu\": %.2f}", first ? "" : ",\n", th, sz, mx, (double)mx / cus);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
