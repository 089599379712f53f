// Checks on the device that the DPP / permlane-swap forms of "the key of lane ^ j" equal __shfl_xor for j = 1 .. 32 (gfx950), and times
// a 64-key bitonic sort built on each.   hipcc --offload-arch=gfx950 -O3 tools/ubench_permlane.hip -o tools/bin/ubench_permlane
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t partner_fast(uint32_t key, int j, int lane)
{
    if (j == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xF, 0xF, false);
    if (j == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xF, 0xF, false);
    if (j == 4) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x124, 0xF, 0xF, false);   /* row_ror:4 */
        const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x12C, 0xF, 0xF, false);   /* row_ror:12 */
        return (lane & 4) ? a : b;
    }
    if (j == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xF, 0xF, false);
    if (j == 16) { auto r = __builtin_amdgcn_permlane16_swap(key, key, false, false); return (lane & 16) ? r[0] : r[1]; }
    auto r = __builtin_amdgcn_permlane32_swap(key, key, false, false);
    return (lane & 32) ? r[0] : r[1];
}
__global__ void check(const uint32_t* in, uint32_t* bad)
{
    const int lane = threadIdx.x;
    const uint32_t key = in[lane];
#pragma unroll
    for (int j = 1; j <= 32; j <<= 1) {
        const uint32_t want = (uint32_t)__shfl_xor((int)key, j), got = partner_fast(key, j, lane);
        if (want != got) atomicOr(&bad[0], (uint32_t)j);
    }
}
template <bool FAST> __global__ void sortk(const uint32_t* in, uint32_t* out, int reps)
{
    const int lane = threadIdx.x;
    uint32_t key = in[lane], acc = 0;
    for (int r = 0; r < reps; r++) {
        key = key * 1664525u + 1013904223u + (uint32_t)lane;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t other = FAST ? partner_fast(key, j, lane) : (uint32_t)__shfl_xor((int)key, j);
                const bool keepMin = ((lane & j) == 0) == ((lane & k) == 0);
                const uint32_t lo = key < other ? key : other, hi = key < other ? other : key;
                key = keepMin ? lo : hi;
            }
        acc += key;
    }
    out[lane] = acc;
}
int main()
{
    uint32_t h[64], *d, *b, *o, hb = 0;
    for (int i = 0; i < 64; i++) h[i] = 0x9E3779B9u * (uint32_t)(i + 1);
    hipMalloc(&d, 256); hipMalloc(&b, 4); hipMalloc(&o, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice); hipMemset(b, 0, 4);
    check<<<1, 64>>>(d, b);
    hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost);
    std::printf("partner forms that differ from __shfl_xor (bit j set): 0x%x\n", hb);
    uint32_t r0[64], r1[64];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; v++) {
        const int reps = 20000;
        if (v == 0) sortk<false><<<1, 64>>>(d, o, 10); else sortk<true><<<1, 64>>>(d, o, 10);
        hipEventRecord(e0);
        if (v == 0) sortk<false><<<1, 64>>>(d, o, reps); else sortk<true><<<1, 64>>>(d, o, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(v ? r1 : r0, o, 256, hipMemcpyDeviceToHost);
        std::printf("%s: %.3f us per 64-key sort (one wavefront alone)\n", v ? "DPP / permlane swap" : "__shfl_xor (ds_bpermute beyond the row)", ms * 1e3 / reps);
    }
    int same = 1; for (int i = 0; i < 64; i++) same &= r0[i] == r1[i];
    std::printf("results equal: %d\n", same);
    return hb != 0 || !same;
}
