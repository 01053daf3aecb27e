// Microbenchmark: issue rate of the VALU instructions the DP kernel is made of (gfx950).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 64
#define ITER 2000
#define BODY(INS)                                                                                   \
    for (int it = 0; it < ITER; it++) {                                                             \
        _Pragma("unroll") for (int r = 0; r < REP / 8; r++) {                                       \
            asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t" \
                         INS " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8"   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); \
        }                                                                                           \
    }
#define BODY3(INS)                                                                                  \
    for (int it = 0; it < ITER; it++) {                                                             \
        _Pragma("unroll") for (int r = 0; r < REP / 8; r++) {                                       \
            asm volatile(INS " %0, %0, %8, %8\n\t" INS " %1, %1, %8, %8\n\t" INS " %2, %2, %8, %8\n\t" INS " %3, %3, %8, %8\n\t" \
                         INS " %4, %4, %8, %8\n\t" INS " %5, %5, %8, %8\n\t" INS " %6, %6, %8, %8\n\t" INS " %7, %7, %8, %8"   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k)); \
        }                                                                                           \
    }
#define KERN(name, B, INS) __global__ void name(int *out, int k) { int a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; \
    B(INS) out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; }
KERN(k_add_u32, BODY, "v_add_u32")
KERN(k_max_i32, BODY, "v_max_i32")
KERN(k_pk_add_i16, BODY, "v_pk_add_i16")
KERN(k_pk_max_i16, BODY, "v_pk_max_i16")
KERN(k_pk_min_u16, BODY, "v_pk_min_u16")
KERN(k_pk_sub_i16, BODY, "v_pk_sub_i16")
KERN(k_pk_mad_i16, BODY3, "v_pk_mad_i16")
KERN(k_max3_i32, BODY3, "v_max3_i32")
KERN(k_add3_u32, BODY3, "v_add3_u32")
KERN(k_lshl_or, BODY3, "v_lshl_or_b32")
KERN(k_sub_u32, BODY, "v_sub_u32")
KERN(k_and_b32, BODY, "v_and_b32")
KERN(k_or_b32, BODY, "v_or_b32")
KERN(k_xor_b32, BODY, "v_xor_b32")
KERN(k_lshrrev, BODY, "v_lshrrev_b32")
KERN(k_lshlrev, BODY, "v_lshlrev_b32")
KERN(k_min_u32, BODY, "v_min_u32")
KERN(k_max_u32, BODY, "v_max_u32")
KERN(k_pk_max_u16, BODY, "v_pk_max_u16")
KERN(k_pk_lshr16, BODY, "v_pk_lshrrev_b16")
KERN(k_pk_ashr16, BODY, "v_pk_ashrrev_i16")
KERN(k_pk_mul16, BODY, "v_pk_mul_lo_u16")
KERN(k_mul_u24, BODY, "v_mul_u32_u24")
KERN(k_bfi, BODY3, "v_bfi_b32")
KERN(k_and_or, BODY3, "v_and_or_b32")
KERN(k_or3, BODY3, "v_or3_b32")
KERN(k_perm, BODY3, "v_perm_b32")
KERN(k_alignbit, BODY3, "v_alignbit_b32")
KERN(k_lshl_add, BODY3, "v_lshl_add_u32")
KERN(k_mad_u24, BODY3, "v_mad_u32_u24")
KERN(k_xad, BODY3, "v_xad_u32")
KERN(k_min3_u32, BODY3, "v_min3_u32")
KERN(k_med3_i32, BODY3, "v_med3_i32")
// floating-point forms: small integers (|x| <= 2048) are exact in f16, so max/add recurrences could run on them
KERN(k_pk_add_f16, BODY, "v_pk_add_f16")
KERN(k_pk_max_f16, BODY, "v_pk_max_f16")
KERN(k_pk_min_f16, BODY, "v_pk_min_f16")
KERN(k_pk_mul_f16, BODY, "v_pk_mul_f16")
KERN(k_pk_fma_f16, BODY3, "v_pk_fma_f16")
KERN(k_max_f32, BODY, "v_max_f32")
KERN(k_add_f32, BODY, "v_add_f32")
KERN(k_max_f16, BODY, "v_max_f16")
KERN(k_max3_f32, BODY3, "v_max3_f32")
KERN(k_sad_u8, BODY3, "v_sad_u8")
KERN(k_max_i16, BODY, "v_max_i16")
KERN(k_max_u16, BODY, "v_max_u16")
// dependent chain (1 accumulator) to see the latency one wave pays
__global__ void k_dep_pk(int *out, int k) { int a = threadIdx.x; for (int it = 0; it < ITER * REP; it++) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a) : "v"(k)); out[blockIdx.x * blockDim.x + threadIdx.x] = a; }
__global__ void k_dep_u32(int *out, int k) { int a = threadIdx.x; for (int it = 0; it < ITER * REP; it++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(k)); out[blockIdx.x * blockDim.x + threadIdx.x] = a; }
template <typename F> void run(const char *name, F kern, int *d, int waves_per_simd) {
    int blocks = 256 * 4 * waves_per_simd / 4;      // 256-thread blocks = 4 waves
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1);
    hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double wave_instr = (double)blocks * 4 * ITER * REP;
    printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f cycles@2.4GHz per wave-instruction per SIMD\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / (wave_instr / 1024.0));
}
int main() {
    int *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4 * 4);
    for (int w : {1, 2, 4, 8}) {
        run("v_add_u32", k_add_u32, d, w); run("v_max_i32", k_max_i32, d, w); run("v_pk_add_i16", k_pk_add_i16, d, w); run("v_pk_max_i16", k_pk_max_i16, d, w);
        run("v_pk_min_u16", k_pk_min_u16, d, w); run("v_pk_sub_i16", k_pk_sub_i16, d, w); run("v_pk_mad_i16", k_pk_mad_i16, d, w);
        run("v_max3_i32", k_max3_i32, d, w); run("v_add3_u32", k_add3_u32, d, w); run("v_lshl_or_b32", k_lshl_or, d, w);
        run("dep pk_add", k_dep_pk, d, w); run("dep add_u32", k_dep_u32, d, w);
        if (w == 8) {
            run("v_sub_u32", k_sub_u32, d, w); run("v_and_b32", k_and_b32, d, w); run("v_or_b32", k_or_b32, d, w); run("v_xor_b32", k_xor_b32, d, w);
            run("v_lshrrev_b32", k_lshrrev, d, w); run("v_lshlrev_b32", k_lshlrev, d, w); run("v_min_u32", k_min_u32, d, w); run("v_max_u32", k_max_u32, d, w);
            run("v_pk_max_u16", k_pk_max_u16, d, w); run("v_pk_lshrrev_b16", k_pk_lshr16, d, w); run("v_pk_ashrrev_i16", k_pk_ashr16, d, w);
            run("v_pk_mul_lo_u16", k_pk_mul16, d, w); run("v_mul_u32_u24", k_mul_u24, d, w); run("v_bfi_b32", k_bfi, d, w); run("v_and_or_b32", k_and_or, d, w);
            run("v_or3_b32", k_or3, d, w); run("v_perm_b32", k_perm, d, w); run("v_alignbit_b32", k_alignbit, d, w); run("v_lshl_add_u32", k_lshl_add, d, w);
            run("v_mad_u32_u24", k_mad_u24, d, w); run("v_xad_u32", k_xad, d, w); run("v_min3_u32", k_min3_u32, d, w); run("v_med3_i32", k_med3_i32, d, w);
            run("v_pk_add_f16", k_pk_add_f16, d, w); run("v_pk_max_f16", k_pk_max_f16, d, w); run("v_pk_min_f16", k_pk_min_f16, d, w); run("v_pk_mul_f16", k_pk_mul_f16, d, w);
            run("v_max_f16", k_max_f16, d, w); run("v_max3_f32", k_max3_f32, d, w); run("v_sad_u8", k_sad_u8, d, w); run("v_max_i16", k_max_i16, d, w); run("v_max_u16", k_max_u16, d, w);
        }
    }
    return 0;
}
