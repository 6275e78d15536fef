// pano_probe.hip - measurement probes: what the part streams, in the two shapes K1's bytes move in ("also report
// against a measured device-copy ceiling").  Not on the compose path; launched by pano_probe_copy only.
//
//   copy_f4_kernel       the plain ceiling: grid-stride copy, 16-byte loads and stores, four loads of a lane in flight - the form
//                        MI355X_MICROARCH.md quotes 6.29 TB/s for; copy_f4_flat_kernel: one element per lane, no loop.
//   copy_k1_shape_kernel a copy with K1's traffic shape and none of its arithmetic: per 256-thread workgroup (a 64 x 16 pixel
//                        patch) a source box of kProbeBoxRows rows x 256 B lands in LDS by global_load_lds_dwordx4 (coalesced
//                        16-byte direct-to-LDS loads, as K1 copies its box), every lane loads one 8-byte table entry, and
//                        after the barrier every lane reads three dwords of the box back from LDS and stores one dword into
//                        each of three planes (as K1 stores its tile).  In : out = 6656 : 3072 bytes per workgroup - config 2's
//                        K1 moves 4.4 KB of frame lines + 1.9 KB of table in and 3.1 KB out per workgroup.  Workgroups are dealt to
//                        the XCDs in contiguous eighths like K1's (blockIdx.x = XCD).
#include "pano_dev.hpp"

namespace pano {

__global__ __launch_bounds__(256) void copy_f4_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    // four 16-byte loads of a lane in flight before its first store
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n16; i += 4 * step) {
        const uint4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
        dst[i] = a; dst[i + step] = b; dst[i + 2 * step] = c; dst[i + 3 * step] = d;
    }
    for (; i < n16; i += step) dst[i] = src[i];
}
// the same bytes with one 16-byte element per lane and as many workgroups as it takes (no loop)
__global__ __launch_bounds__(256) void copy_f4_flat_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void copy_k1_shape_kernel(
    const uint8_t* __restrict__ box_src, const uint2* __restrict__ table, uint8_t* __restrict__ dst, unsigned per, unsigned total,
    unsigned dst_plane) {
    __shared__ uint4 sbox[kProbeBoxBytes / 16 + 64];
    const unsigned wg = blockIdx.x * per + blockIdx.y;  // XCD-major, like K1's deal
    if (wg >= total) return;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const uint2 e = table[(size_t)wg * 256 + tid];
    const uint8_t* const b = box_src + (size_t)wg * kProbeBoxBytes;
    constexpr int chunks = kProbeBoxBytes / 16;
#pragma unroll
    for (int it = 0; it < (chunks + 255) / 256; it++) {
        if (wv * 64 + 256 * it < chunks) {
            const unsigned k = (unsigned)min(tid + 256 * it, chunks - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b + k * 16u),
                                             (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) uint8_t*)&sbox[0] + (wv * 64 + 256 * it) * 16),
                                             16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned* sb = reinterpret_cast<const unsigned*>(sbox);
    // three dwords of the box per lane (an offset that depends on the table entry keeps the table load alive, as K1's does)
    const unsigned o = ((unsigned)tid * 3u + (e.x & 1u) + (e.y & 1u)) % (kProbeBoxBytes / 4 - 3);
    const unsigned v0 = sb[o], v1 = sb[o + 1], v2 = sb[o + 2];
    // a workgroup's slice of each plane: 16 rows x 64 lanes x 4 bytes, row after row (K1: 16 tile rows of 256 bytes)
    uint8_t* d = dst + (size_t)wg * 1024 + (size_t)tid * 4;
    *reinterpret_cast<unsigned*>(d) = v0;
    *reinterpret_cast<unsigned*>(d + dst_plane) = v1;
    *reinterpret_cast<unsigned*>(d + 2 * (size_t)dst_plane) = v2;
}

void launch_probe_copy_f4(const void* src, void* dst, size_t bytes, int blocks, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (blocks > 0) hipExtLaunchKernelGGL(copy_f4_kernel, dim3(blocks), dim3(256), 0, s, e0, e1, 0, (const uint4*)src, (uint4*)dst, bytes / 16);
    else hipExtLaunchKernelGGL(copy_f4_flat_kernel, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, s, e0, e1, 0, (const uint4*)src, (uint4*)dst, bytes / 16);
}
void launch_probe_copy_k1_shape(const void* box_src, const void* table, void* dst, unsigned workgroups, hipStream_t s, hipEvent_t e0,
                                hipEvent_t e1) {
    const unsigned per = (workgroups + 7u) / 8u;
    hipExtLaunchKernelGGL(copy_k1_shape_kernel, dim3(8, per, 1), dim3(64, 4, 1), 0, s, e0, e1, 0, (const uint8_t*)box_src,
                          (const uint2*)table, (uint8_t*)dst, per, workgroups, workgroups * 1024u);
}

}  // namespace pano
