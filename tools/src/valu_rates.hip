// valu_rates.hip - what the vector instructions of the compose kernels cost on gfx950: wave-instructions per cycle and CU for a
// stream of independent instructions of one kind, at 1 / 2 / 4 / 8 waves per SIMD (every CU busy).  Build + run on the GPU box:
//   hipcc -O2 --offload-arch=gfx950 tools/src/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(INS)                                                                              \
    for (int it = 0; it < iters; it++) {                                                       \
        REP8(asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t" \
                          INS " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8"     \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) \
    }
#define BODY3(INS)                                                                             \
    for (int it = 0; it < iters; it++) {                                                       \
        REP8(asm volatile(INS " %0, %0, %8, %0\n\t" INS " %1, %1, %8, %1\n\t" INS " %2, %2, %8, %2\n\t" INS " %3, %3, %8, %3\n\t" \
                          INS " %4, %4, %8, %4\n\t" INS " %5, %5, %8, %5\n\t" INS " %6, %6, %8, %6\n\t" INS " %7, %7, %8, %7"     \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) \
    }
#define KERNEL(NAME, BODYM, INS)                                                               \
    __global__ void NAME(unsigned* out, int iters) {                                           \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 3; \
        BODYM(INS)                                                                             \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;                  \
    }
#define BODYM(INS, MOD)                                                                        \
    for (int it = 0; it < iters; it++) {                                                       \
        REP8(asm volatile(INS " %0, %0, %8 " MOD "\n\t" INS " %1, %1, %8 " MOD "\n\t" INS " %2, %2, %8 " MOD "\n\t" INS " %3, %3, %8 " MOD "\n\t" \
                          INS " %4, %4, %8 " MOD "\n\t" INS " %5, %5, %8 " MOD "\n\t" INS " %6, %6, %8 " MOD "\n\t" INS " %7, %7, %8 " MOD     \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) \
    }
#define BODY1(INS)                                                                             \
    for (int it = 0; it < iters; it++) {                                                       \
        REP8(asm volatile(INS " %0, %0\n\t" INS " %1, %1\n\t" INS " %2, %2\n\t" INS " %3, %3\n\t" \
                          INS " %4, %4\n\t" INS " %5, %5\n\t" INS " %6, %6\n\t" INS " %7, %7"     \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) \
    }
#define KERNELM(NAME, INS, MOD)                                                                \
    __global__ void NAME(unsigned* out, int iters) {                                           \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 3; \
        BODYM(INS, MOD)                                                                        \
        if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;                  \
    }
KERNEL(k_add, BODY, "v_add_u32")
KERNEL(k_sub, BODY, "v_sub_u32")
KERNEL(k_or, BODY, "v_or_b32")
KERNEL(k_xor, BODY, "v_xor_b32")
KERNEL(k_minu, BODY, "v_min_u32")
KERNEL(k_maxi, BODY, "v_max_i32")
KERNEL(k_lshr, BODY, "v_lshrrev_b32")
KERNEL(k_ashr, BODY, "v_ashrrev_i32")
KERNEL(k_addf, BODY, "v_add_f32")
KERNEL(k_add16, BODY, "v_add_u16")
KERNEL(k_mullo16, BODY, "v_mul_lo_u16")
KERNEL(k_lshl16, BODY, "v_lshlrev_b16")
KERNEL(k_mov, BODY1, "v_mov_b32")
KERNEL(k_cvtub0, BODY1, "v_cvt_f32_ubyte0")
KERNEL(k_cvtu32, BODY1, "v_cvt_u32_f32")
KERNEL(k_rndne, BODY1, "v_rndne_f32")
KERNELM(k_add_sdwa, "v_add_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNELM(k_sub_sdwa, "v_sub_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD")
KERNELM(k_mul24_sdwa, "v_mul_u32_u24_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD")
KERNELM(k_add_dpp, "v_add_u32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNELM(k_cndmask, "v_cndmask_b32", ", vcc")
KERNEL(k_fma, BODY3, "v_fma_f32")
KERNEL(k_or3, BODY3, "v_or3_b32")
KERNEL(k_and_or, BODY3, "v_and_or_b32")
KERNEL(k_lshl_or, BODY3, "v_lshl_or_b32")
KERNEL(k_mad16, BODY3, "v_mad_u16")
KERNEL(k_min3, BODY3, "v_min3_i32")
KERNEL(k_sad, BODY3, "v_sad_u8")
KERNEL(k_and, BODY, "v_and_b32")
KERNEL(k_lshl, BODY, "v_lshlrev_b32")
KERNEL(k_mul24, BODY, "v_mul_u32_u24")
KERNEL(k_mullo, BODY, "v_mul_lo_u32")
KERNEL(k_pk_add16, BODY, "v_pk_add_u16")
KERNEL(k_pk_mul16, BODY, "v_pk_mul_lo_u16")
KERNEL(k_pk_min16, BODY, "v_pk_min_i16")
KERNEL(k_mulf32, BODY, "v_mul_f32")
KERNEL(k_pk_mulf32, BODY, "v_mul_f32")  // placeholder: packed f32 needs register pairs (measured separately below)
KERNEL(k_perm, BODY3, "v_perm_b32")
KERNEL(k_alignbyte, BODY3, "v_alignbyte_b32")
KERNEL(k_mad24, BODY3, "v_mad_u32_u24")
KERNEL(k_add3, BODY3, "v_add3_u32")
KERNEL(k_lshl_add, BODY3, "v_lshl_add_u32")
KERNEL(k_dot4, BODY3, "v_dot4_u32_u8")
KERNEL(k_dot2, BODY3, "v_dot2_u32_u16")
KERNEL(k_sdot2, BODY3, "v_dot2_i32_i16")
KERNEL(k_pk_mad16, BODY3, "v_pk_mad_u16")
KERNEL(k_med3, BODY3, "v_med3_i32")
KERNEL(k_bfe, BODY3, "v_bfe_u32")
KERNEL(k_cvt_pk_u8, BODY3, "v_cvt_pk_u8_f32")

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d;
    (void)hipMalloc(&d, 64);
    struct K { const char* name; void (*fn)(unsigned*, int); } ks[] = {
        {"v_add_u32", k_add}, {"v_sub_u32", k_sub}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_min_u32", k_minu}, {"v_max_i32", k_maxi},
        {"v_lshrrev_b32", k_lshr}, {"v_ashrrev_i32", k_ashr}, {"v_add_f32", k_addf}, {"v_fma_f32", k_fma}, {"v_add_u16", k_add16}, {"v_mul_lo_u16", k_mullo16},
        {"v_lshlrev_b16", k_lshl16}, {"v_mad_u16", k_mad16}, {"v_mov_b32", k_mov}, {"v_cvt_f32_ubyte0", k_cvtub0}, {"v_cvt_u32_f32", k_cvtu32}, {"v_rndne_f32", k_rndne},
        {"v_add_u32_sdwa", k_add_sdwa}, {"v_sub_u32_sdwa", k_sub_sdwa}, {"v_mul_u32_u24_sdwa", k_mul24_sdwa}, {"v_add_u32_dpp", k_add_dpp}, {"v_cndmask_b32", k_cndmask},
        {"v_or3_b32", k_or3}, {"v_and_or_b32", k_and_or}, {"v_lshl_or_b32", k_lshl_or}, {"v_min3_i32", k_min3}, {"v_sad_u8", k_sad}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_mul_u32_u24", k_mul24}, {"v_mul_lo_u32", k_mullo},
        {"v_mad_u32_u24", k_mad24}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshl_add}, {"v_bfe_u32", k_bfe}, {"v_med3_i32", k_med3},
        {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignbyte}, {"v_dot4_u32_u8", k_dot4}, {"v_dot2_u32_u16", k_dot2}, {"v_dot2_i32_i16", k_sdot2},
        {"v_pk_add_u16", k_pk_add16}, {"v_pk_mul_lo_u16", k_pk_mul16}, {"v_pk_mad_u16", k_pk_mad16}, {"v_pk_min_i16", k_pk_min16},
        {"v_mul_f32", k_mulf32}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8}};
    const int iters = 2000;  // x 64 instructions
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("cycles per wave-instruction and SIMD at N waves per SIMD (at the nominal %d MHz of the device properties)\n", p.clockRate / 1000);
    printf("%-18s %8s %8s %8s %8s\n", "instruction", "1", "2", "4", "8");
    for (auto& k : ks) {
        printf("%-18s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            const int threads = 256, blocks = cus * wps;  // 4 waves per block = one per SIMD; wps blocks per CU
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, d, 10);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, d, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 64 * wps;
            const double cycles = ms * 1e-3 * (double)p.clockRate * 1e3;
            printf(" %8.2f", cycles / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
