/* Issue-rate microbenchmark for the VALU forms the FAST corner-score kernel could be written in.
 * Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu ; prints cycles per
 * wave-instruction (one wave per SIMD, 8 independent accumulators). */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 4096

#define KERNEL(name, insn)                                                                            \
    __global__ void name(uint32_t* out, uint64_t* cyc)                                                \
    {                                                                                                 \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7, b = threadIdx.x * 3 + 1, c = threadIdx.x * 5 + 2;                       \
        const uint64_t t0 = __builtin_readcyclecounter();                                             \
        for (int r = 0; r < REP; r++) {                                                               \
            asm volatile(insn(%0) insn(%1) insn(%2) insn(%3) insn(%4) insn(%5) insn(%6) insn(%7)      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c));                                                           \
        }                                                                                             \
        const uint64_t t1 = __builtin_readcyclecounter();                                             \
        out[threadIdx.x + blockIdx.x * blockDim.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
        if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                      \
    }

#define I_MAX_I32(x) "v_max_i32 " #x ", " #x ", %8\n"
#define I_PK_MAX_I16(x) "v_pk_max_i16 " #x ", " #x ", %8\n"
#define I_PK_MIN_U16(x) "v_pk_min_u16 " #x ", " #x ", %8\n"
#define I_PK_SUB_I16(x) "v_pk_sub_i16 " #x ", " #x ", %8\n"
#define I_PK_MAX_F16(x) "v_pk_max_f16 " #x ", " #x ", %8\n"
#define I_PK_ADD_F16(x) "v_pk_add_f16 " #x ", " #x ", %8\n"
#define I_PK_MAX3_F16(x) "v_pk_maximum3_f16 " #x ", " #x ", %8, %9\n"
#define I_MAX3_I32(x) "v_max3_i32 " #x ", " #x ", %8, %9\n"
#define I_MED3_I32(x) "v_med3_i32 " #x ", " #x ", %8, %9\n"
#define I_PERM(x) "v_perm_b32 " #x ", " #x ", %8, %9\n"
#define I_LSHL_OR(x) "v_lshl_or_b32 " #x ", " #x ", 16, %8\n"
#define I_MAX_I16(x) "v_max_i16 " #x ", " #x ", %8\n"
#define I_SAD_U8(x) "v_sad_u8 " #x ", " #x ", %8, %9\n"

KERNEL(k_max_i32, I_MAX_I32)
KERNEL(k_pk_max_i16, I_PK_MAX_I16)
KERNEL(k_pk_min_u16, I_PK_MIN_U16)
KERNEL(k_pk_sub_i16, I_PK_SUB_I16)
KERNEL(k_pk_max_f16, I_PK_MAX_F16)
KERNEL(k_pk_add_f16, I_PK_ADD_F16)
KERNEL(k_pk_max3_f16, I_PK_MAX3_F16)
KERNEL(k_max3_i32, I_MAX3_I32)
KERNEL(k_med3_i32, I_MED3_I32)
KERNEL(k_perm, I_PERM)
KERNEL(k_lshl_or, I_LSHL_OR)
KERNEL(k_max_i16, I_MAX_I16)
KERNEL(k_sad_u8, I_SAD_U8)

template <typename K> static void run(const char* name, K k, uint32_t* d_out, uint64_t* d_cyc, int wavesPerSimd)
{
    /* 256 CUs x 4 SIMDs; blocks of 256 threads (one wave per SIMD), wavesPerSimd blocks per CU */
    const int blocks = 256 * wavesPerSimd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instrPerSimd = (double)wavesPerSimd * REP * 8.0;
    printf("%-18s waves/SIMD %d  %.1f us  ns per wave-instr per SIMD %.3f (= %.2f cycles at 2.4 GHz)\n", name,
           wavesPerSimd, ms * 1000, ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4);
}

int main()
{
    uint32_t* d_out; uint64_t* d_cyc;
    hipMalloc(&d_out, 256 * 8 * 256 * 4); hipMalloc(&d_cyc, 8);
    for (int waves : {1, 4, 8}) {
        run("v_max_i32", k_max_i32, d_out, d_cyc, waves);
        run("v_max3_i32", k_max3_i32, d_out, d_cyc, waves);
        run("v_max_i16", k_max_i16, d_out, d_cyc, waves);
        run("v_pk_max_i16", k_pk_max_i16, d_out, d_cyc, waves);
        run("v_pk_min_u16", k_pk_min_u16, d_out, d_cyc, waves);
        run("v_pk_sub_i16", k_pk_sub_i16, d_out, d_cyc, waves);
        run("v_pk_max_f16", k_pk_max_f16, d_out, d_cyc, waves);
        run("v_pk_add_f16", k_pk_add_f16, d_out, d_cyc, waves);
        run("v_pk_maximum3_f16", k_pk_max3_f16, d_out, d_cyc, waves);
        run("v_perm_b32", k_perm, d_out, d_cyc, waves);
        run("v_lshl_or_b32", k_lshl_or, d_out, d_cyc, waves);
        run("v_sad_u8", k_sad_u8, d_out, d_cyc, waves);
    }
    return 0;
}
