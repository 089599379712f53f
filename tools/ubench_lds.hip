/* LDS instruction issue rate per CU for the read widths the FAST kernel could use.
 * Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_lds.hip -o tools/bin/ubench_lds ; prints cycles per
 * wave-instruction per CU (conflict-free addresses: lane i reads byte / half / dword i of a row). */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 2048

#define KERNEL(name, insn, scale)                                                                     \
    __global__ __launch_bounds__(256) void name(uint32_t* out)                                        \
    {                                                                                                 \
        __shared__ uint32_t lds[4096];                                                                \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;                       \
        __syncthreads();                                                                              \
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;                      \
        uint32_t addr = (threadIdx.x & 63) * scale + (threadIdx.x >> 6) * 1024;                       \
        for (int r = 0; r < REP; r++) {                                                               \
            asm volatile(insn " %0, %8\n" insn " %1, %8 offset:256\n" insn " %2, %8 offset:512\n"     \
                         insn " %3, %8 offset:768\n" insn " %4, %8 offset:1024\n"                     \
                         insn " %5, %8 offset:1280\n" insn " %6, %8 offset:1536\n"                    \
                         insn " %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)\n"                          \
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) \
                         : "v"(addr));                                                                \
            addr ^= (a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) & 0;                                      \
        }                                                                                             \
        out[threadIdx.x + blockIdx.x * 256] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                  \
    }

KERNEL(k_u8, "ds_read_u8", 1)
KERNEL(k_u16, "ds_read_u16", 2)
KERNEL(k_b32, "ds_read_b32", 4)
KERNEL(k_u8_stride, "ds_read_u8", 4)

template <typename K> static void run(const char* name, K k, uint32_t* d_out, int blocksPerCu)
{
    const int blocks = 256 * blocksPerCu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instrPerCu = (double)blocksPerCu * 4 * REP * 8.0;       /* wave-instructions per CU */
    printf("%-14s blocks/CU %d  %.1f us  ns per wave-instr per CU %.3f (= %.2f cycles at 2.4 GHz)\n", name, blocksPerCu,
           ms * 1000, ms * 1e6 / instrPerCu, ms * 1e6 / instrPerCu * 2.4);
}

int main()
{
    uint32_t* d_out;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    for (int b : {1, 2, 4}) {
        run("ds_read_u8", k_u8, d_out, b);
        run("ds_read_u16", k_u16, d_out, b);
        run("ds_read_b32", k_b32, d_out, b);
        run("ds_read_u8 x4", k_u8_stride, d_out, b);
    }
    return 0;
}
