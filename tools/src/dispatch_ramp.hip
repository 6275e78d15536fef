// dispatch_ramp.hip - what launching N workgroups costs on its own: an empty body (one scalar load, exit) over 37.5 K waves as
// workgroups of 1 / 4 / 8 / 16 waves, and a body that waits ~2 us (s_sleep) so that residency matters.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(const int* p, int* out) {
    if (p[blockIdx.x & 7] == 12345) out[0] = 1;
}
__global__ void k_sleep(const int* p, int* out) {
    for (int i = 0; i < 40; i++) __builtin_amdgcn_s_sleep(127);   // ~ 40 x 127 x 64 clocks... bounded
    if (p[blockIdx.x & 7] == 12345) out[0] = 1;
}
int main() {
    int *p, *o;
    (void)hipMalloc(&p, 64); (void)hipMemset(p, 0, 64); (void)hipMalloc(&o, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int waves = 37632;
    for (int wpb : {1, 4, 8, 16}) {
        const int blocks = waves / wpb, threads = 64 * wpb;
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, 0, p, o);
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, 0, p, o);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("empty body, %5d workgroups of %2d waves: %.2f us per launch (back to back, incl. the ~1.5 us launch boundary)\n", blocks, wpb, ms * 1e3 / 20);
    }
    return 0;
}
