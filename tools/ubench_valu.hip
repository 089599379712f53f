/* VALU issue-rate calibration for gfx950 (MI355X): cycles per wave64 instruction per SIMD for the instruction classes the
 * front-end kernels are made of, at 1 / 2 / 4 / 8 resident waves per SIMD.
 *
 * Build + run: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
 * (tools/valu_calibration.sh writes profiles/rNN_valu_issue.txt)
 *
 * Method: every wave runs REP iterations of 8 back-to-back instructions on 8 independent accumulators (no dependent issue
 * inside the group of eight; an accumulator is reused 8 instructions later).  A launch puts `waves` workgroups of 256 threads
 * (= one wave per SIMD each) on every CU, so a SIMD hosts exactly `waves` waves.  Two clocks are read:
 *   - s_memtime (clock64, the shader-engine counter) inside the kernel around the loop of wave 0 of block 0: the cycles one
 *     wave saw for its own REP x 8 instructions.  cycles_per_instr_per_SIMD = that / (REP x 8 x waves) if the SIMD were
 *     shared fairly - printed as `cyc/instr/SIMD (in-kernel)`;
 *   - HIP events around the launch: wall time of the whole grid / (REP x 8 x waves) in ns per instruction per SIMD, and the
 *     same in cycles using the clock frequency MEASURED by the kernel (s_memtime ticks per wall_clock64 tick at 100 MHz).
 * A wave64 op on a 16-lane pipe needs 4 cycles per instruction per SIMD, on a 32-lane pipe 2 (with >= 2 waves to alternate),
 * a quarter-rate op 16, and so on - the table says which holds for which instruction on this part. */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP 8192

#define KERNEL32(name, insn)                                                                          \
    __global__ void name(uint32_t* out, uint64_t* cyc)                                                \
    {                                                                                                 \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7, b = threadIdx.x * 3 + 1, c = threadIdx.x * 5 + 2;                       \
        const uint64_t w0 = wall_clock64();                                                           \
        const uint64_t t0 = clock64();                                                                \
        for (int r = 0; r < REP; r++) {                                                               \
            asm volatile(insn(%0) insn(%1) insn(%2) insn(%3) insn(%4) insn(%5) insn(%6) insn(%7)      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c) : "vcc");                                                   \
        }                                                                                             \
        const uint64_t t1 = clock64();                                                                \
        const uint64_t w1 = wall_clock64();                                                           \
        out[threadIdx.x + blockIdx.x * blockDim.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
        if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }              \
    }

#define KERNEL64(name, insn)                                                                          \
    __global__ void name(uint32_t* out, uint64_t* cyc)                                                \
    {                                                                                                 \
        double a0 = threadIdx.x + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
               a7 = a0 + 7, b = 1.0000001, c = 1e-9;                                                  \
        const uint64_t w0 = wall_clock64();                                                           \
        const uint64_t t0 = clock64();                                                                \
        for (int r = 0; r < REP; r++) {                                                               \
            asm volatile(insn(%0) insn(%1) insn(%2) insn(%3) insn(%4) insn(%5) insn(%6) insn(%7)      \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(b), "v"(c) : "vcc");                                                   \
        }                                                                                             \
        const uint64_t t1 = clock64();                                                                \
        const uint64_t w1 = wall_clock64();                                                           \
        out[threadIdx.x + blockIdx.x * blockDim.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
        if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }              \
    }

#define I_FMA_F32(x) "v_fma_f32 " #x ", " #x ", %8, %9\n"
#define I_ADD_U32(x) "v_add_u32 " #x ", " #x ", %8\n"
#define I_MAX_I32(x) "v_max_i32 " #x ", " #x ", %8\n"
#define I_MAX3_I32(x) "v_max3_i32 " #x ", " #x ", %8, %9\n"
#define I_MED3_I32(x) "v_med3_i32 " #x ", " #x ", %8, %9\n"
#define I_MAX_I16(x) "v_max_i16 " #x ", " #x ", %8\n"
#define I_PK_MAX_I16(x) "v_pk_max_i16 " #x ", " #x ", %8\n"
#define I_PK_MAX_U16(x) "v_pk_max_u16 " #x ", " #x ", %8\n"
#define I_PK_MIN_U16(x) "v_pk_min_u16 " #x ", " #x ", %8\n"
#define I_PK_SUB_I16(x) "v_pk_sub_i16 " #x ", " #x ", %8\n"
#define I_PK_MAX_F16(x) "v_pk_max_f16 " #x ", " #x ", %8\n"
#define I_PK_ADD_F16(x) "v_pk_add_f16 " #x ", " #x ", %8\n"
#define I_PK_MAX3_F16(x) "v_pk_maximum3_f16 " #x ", " #x ", %8, %9\n"
#define I_PK_MIN3_F16(x) "v_pk_minimum3_f16 " #x ", " #x ", %8, %9\n"
#define I_PERM(x) "v_perm_b32 " #x ", " #x ", %8, %9\n"
#define I_LSHL_OR(x) "v_lshl_or_b32 " #x ", " #x ", 16, %8\n"
#define I_SAD_U8(x) "v_sad_u8 " #x ", " #x ", %8, %9\n"
#define I_DOT4_U8(x) "v_dot4_u32_u8 " #x ", " #x ", %8, %9\n"
#define I_DOT2_U16(x) "v_dot2_u32_u16 " #x ", " #x ", %8, %9\n"
#define I_ALIGNBIT(x) "v_alignbit_b32 " #x ", " #x ", %8, 16\n"
#define I_ALIGNBYTE(x) "v_alignbyte_b32 " #x ", " #x ", %8, 1\n"
#define I_MUL_LO(x) "v_mul_lo_u32 " #x ", " #x ", %8\n"
#define I_MUL_HI(x) "v_mul_hi_u32 " #x ", " #x ", %8\n"
#define I_MAD_U24(x) "v_mad_u32_u24 " #x ", " #x ", %8, %9\n"
#define I_BFE(x) "v_bfe_u32 " #x ", " #x ", 3, 8\n"
#define I_MOV_DPP(x) "v_mov_b32_dpp " #x ", " #x " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_BCNT(x) "v_bcnt_u32_b32 " #x ", " #x ", %8\n"
#define I_CVT_F32_U32(x) "v_cvt_f32_u32 " #x ", " #x "\n"
#define I_RCP_F32(x) "v_rcp_f32 " #x ", " #x "\n"
#define I_AND(x) "v_and_b32 " #x ", " #x ", %8\n"
#define I_OR(x) "v_or_b32 " #x ", " #x ", %8\n"
#define I_XOR(x) "v_xor_b32 " #x ", " #x ", %8\n"
#define I_LSHLREV(x) "v_lshlrev_b32 " #x ", 3, " #x "\n"
#define I_LSHRREV(x) "v_lshrrev_b32 " #x ", 3, " #x "\n"
#define I_SUB_U32(x) "v_sub_u32 " #x ", " #x ", %8\n"
#define I_ADD_F32(x) "v_add_f32 " #x ", " #x ", %8\n"
#define I_MUL_F32(x) "v_mul_f32 " #x ", " #x ", %8\n"
#define I_MAX_F32(x) "v_max_f32 " #x ", " #x ", %8\n"
#define I_MIN_U32(x) "v_min_u32 " #x ", " #x ", %8\n"
#define I_MAX_U16(x) "v_max_u16 " #x ", " #x ", %8\n"
#define I_ADD_U16(x) "v_add_u16 " #x ", " #x ", %8\n"
#define I_LSHL_ADD(x) "v_lshl_add_u32 " #x ", " #x ", 2, %8\n"
#define I_ADD3(x) "v_add3_u32 " #x ", " #x ", %8, %9\n"
#define I_AND_OR(x) "v_and_or_b32 " #x ", " #x ", %8, %9\n"
#define I_MOV(x) "v_mov_b32 " #x ", %8\n"
#define I_CNDMASK(x) "v_cndmask_b32 " #x ", " #x ", %8, vcc\n"
#define I_CMP(x) "v_cmp_gt_u32 vcc, " #x ", %8\n"
#define I_MBCNT(x) "v_mbcnt_lo_u32_b32 " #x ", -1, " #x "\n"
#define I_CVT_PK_U8(x) "v_cvt_pk_u8_f32 " #x ", " #x ", %8, %9\n"
#define I_MAD_I32_I24(x) "v_mad_i32_i24 " #x ", " #x ", %8, %9\n"
#define I_MUL_U24(x) "v_mul_u32_u24 " #x ", " #x ", %8\n"
#define I_FMAC_F32(x) "v_fmac_f32 " #x ", %8, %9\n"
#define I_MED3_F32(x) "v_med3_f32 " #x ", " #x ", %8, %9\n"
#define I_CVT_F16_F32(x) "v_cvt_f16_f32 " #x ", " #x "\n"
#define I_RNDNE_F32(x) "v_rndne_f32 " #x ", " #x "\n"
#define I_SQRT_F32(x) "v_sqrt_f32 " #x ", " #x "\n"
#define I_ADD_F64(x) "v_add_f64 " #x ", " #x ", %8\n"
#define I_MUL_F64(x) "v_mul_f64 " #x ", " #x ", %8\n"
#define I_FMA_F64(x) "v_fma_f64 " #x ", " #x ", %8, %9\n"
#define I_PK_FMA_F32(x) "v_pk_fma_f32 " #x ", " #x ", %8, %9\n"
#define I_PK_ADD_F32(x) "v_pk_add_f32 " #x ", " #x ", %8\n"

KERNEL32(k_fma_f32, I_FMA_F32)
KERNEL32(k_add_u32, I_ADD_U32)
KERNEL32(k_max_i32, I_MAX_I32)
KERNEL32(k_max3_i32, I_MAX3_I32)
KERNEL32(k_med3_i32, I_MED3_I32)
KERNEL32(k_max_i16, I_MAX_I16)
KERNEL32(k_pk_max_i16, I_PK_MAX_I16)
KERNEL32(k_pk_max_u16, I_PK_MAX_U16)
KERNEL32(k_pk_min_u16, I_PK_MIN_U16)
KERNEL32(k_pk_sub_i16, I_PK_SUB_I16)
KERNEL32(k_pk_max_f16, I_PK_MAX_F16)
KERNEL32(k_pk_add_f16, I_PK_ADD_F16)
KERNEL32(k_pk_max3_f16, I_PK_MAX3_F16)
KERNEL32(k_pk_min3_f16, I_PK_MIN3_F16)
KERNEL32(k_perm, I_PERM)
KERNEL32(k_lshl_or, I_LSHL_OR)
KERNEL32(k_sad_u8, I_SAD_U8)
KERNEL32(k_dot4_u8, I_DOT4_U8)
KERNEL32(k_dot2_u16, I_DOT2_U16)
KERNEL32(k_alignbit, I_ALIGNBIT)
KERNEL32(k_alignbyte, I_ALIGNBYTE)
KERNEL32(k_mul_lo, I_MUL_LO)
KERNEL32(k_mul_hi, I_MUL_HI)
KERNEL32(k_mad_u24, I_MAD_U24)
KERNEL32(k_bfe, I_BFE)
KERNEL32(k_mov_dpp, I_MOV_DPP)
KERNEL32(k_bcnt, I_BCNT)
KERNEL32(k_cvt_f32_u32, I_CVT_F32_U32)
KERNEL32(k_rcp_f32, I_RCP_F32)
KERNEL32(k_and, I_AND)
KERNEL32(k_or, I_OR)
KERNEL32(k_xor, I_XOR)
KERNEL32(k_lshlrev, I_LSHLREV)
KERNEL32(k_lshrrev, I_LSHRREV)
KERNEL32(k_sub_u32, I_SUB_U32)
KERNEL32(k_add_f32, I_ADD_F32)
KERNEL32(k_mul_f32, I_MUL_F32)
KERNEL32(k_max_f32, I_MAX_F32)
KERNEL32(k_min_u32, I_MIN_U32)
KERNEL32(k_max_u16, I_MAX_U16)
KERNEL32(k_add_u16, I_ADD_U16)
KERNEL32(k_lshl_add, I_LSHL_ADD)
KERNEL32(k_add3, I_ADD3)
KERNEL32(k_and_or, I_AND_OR)
KERNEL32(k_mov, I_MOV)
KERNEL32(k_cndmask, I_CNDMASK)
KERNEL32(k_cmp, I_CMP)
KERNEL32(k_mbcnt, I_MBCNT)
KERNEL32(k_cvt_pk_u8, I_CVT_PK_U8)
KERNEL32(k_mad_i24, I_MAD_I32_I24)
KERNEL32(k_mul_u24, I_MUL_U24)
KERNEL32(k_fmac_f32, I_FMAC_F32)
KERNEL32(k_med3_f32, I_MED3_F32)
KERNEL32(k_cvt_f16, I_CVT_F16_F32)
KERNEL32(k_rndne, I_RNDNE_F32)
KERNEL32(k_sqrt_f32, I_SQRT_F32)
KERNEL64(k_add_f64, I_ADD_F64)
KERNEL64(k_mul_f64, I_MUL_F64)
KERNEL64(k_fma_f64, I_FMA_F64)
KERNEL64(k_pk_fma_f32, I_PK_FMA_F32)
KERNEL64(k_pk_add_f32, I_PK_ADD_F32)

typedef void (*Kern)(uint32_t*, uint64_t*);
struct Row { const char* name; Kern k; };

int main()
{
    uint32_t* d_out; uint64_t* d_cyc;
    hipMalloc(&d_out, 256 * 8 * 256 * 4); hipMalloc(&d_cyc, 16);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz; REP %d x 8 instructions per wave; one 256-thread workgroup = one wave per SIMD\n",
           prop.gcnArchName, cus, prop.clockRate, REP);
    printf("# %-20s %5s %10s %22s %26s %10s\n", "instruction", "waves", "grid us", "cyc/instr/SIMD(in-kernel)", "cyc/instr/SIMD(events,MHz)", "MHz");
    const Row rows[] = {
        {"v_fma_f32", k_fma_f32}, {"v_add_u32", k_add_u32}, {"v_max_i32", k_max_i32}, {"v_max3_i32", k_max3_i32},
        {"v_med3_i32", k_med3_i32}, {"v_max_i16", k_max_i16}, {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_max_u16", k_pk_max_u16},
        {"v_pk_min_u16", k_pk_min_u16}, {"v_pk_sub_i16", k_pk_sub_i16}, {"v_pk_max_f16", k_pk_max_f16}, {"v_pk_add_f16", k_pk_add_f16},
        {"v_pk_maximum3_f16", k_pk_max3_f16}, {"v_pk_minimum3_f16", k_pk_min3_f16}, {"v_perm_b32", k_perm}, {"v_lshl_or_b32", k_lshl_or},
        {"v_sad_u8", k_sad_u8}, {"v_dot4_u32_u8", k_dot4_u8}, {"v_dot2_u32_u16", k_dot2_u16}, {"v_alignbit_b32", k_alignbit},
        {"v_alignbyte_b32", k_alignbyte}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi}, {"v_mad_u32_u24", k_mad_u24},
        {"v_bfe_u32", k_bfe}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_bcnt_u32_b32", k_bcnt}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_rcp_f32", k_rcp_f32},
        {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_lshlrev}, {"v_lshrrev_b32", k_lshrrev}, {"v_sub_u32", k_sub_u32}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_max_f32", k_max_f32}, {"v_min_u32", k_min_u32}, {"v_max_u16", k_max_u16}, {"v_add_u16", k_add_u16}, {"v_lshl_add_u32", k_lshl_add}, {"v_add3_u32", k_add3}, {"v_and_or_b32", k_and_or}, {"v_mov_b32", k_mov}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_u32", k_cmp}, {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8}, {"v_mad_i32_i24", k_mad_i24}, {"v_mul_u32_u24", k_mul_u24}, {"v_fmac_f32", k_fmac_f32}, {"v_med3_f32", k_med3_f32}, {"v_cvt_f16_f32", k_cvt_f16}, {"v_rndne_f32", k_rndne}, {"v_sqrt_f32", k_sqrt_f32},
        {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64},
        {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_add_f32", k_pk_add_f32}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Row& r : rows)
        for (int waves : {1, 2, 4, 8}) {
            const int blocks = cus * waves;
            hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);     /* warm-up */
            hipDeviceSynchronize();
            float best = 1e30f;
            uint64_t cyc[2] = {0, 0};
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(r.k, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) { best = ms; hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost); }
            }
            const double n = (double)waves * REP * 8.0;
            const double mhz = cyc[1] ? (double)cyc[0] / ((double)cyc[1] / 100.0) : 0;    /* wall_clock64 ticks at 100 MHz */
            printf("  %-20s %5d %10.1f %22.2f %26.2f %10.0f\n", r.name, waves, best * 1000, (double)cyc[0] / n,
                   best * 1e3 * mhz / n, mhz);
        }
    return 0;
}
