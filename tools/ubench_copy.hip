/* Streaming-copy microbenchmark: bytes per lane per access vs achieved HBM/Infinity-Cache bandwidth, for sizing the
 * per-thread work of the image kernels.  hipcc -O3 --offload-arch=gfx950 tools/ubench_copy.hip -o tools/bin/ubench_copy */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> __global__ void k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
/* each thread copies R elements spaced one "row" (pitch elements) apart: the 4-rows-per-thread shape of the pyramid kernels */
template <typename T, int R> __global__ void k_copy_rows(const T* __restrict__ a, T* __restrict__ b, size_t pitch, size_t rows)
{
    const size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x, y0 = (size_t)blockIdx.y * R;
    if (x >= pitch) return;
    T v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = (y0 + r < rows) ? a[(y0 + r) * pitch + x] : T{};
#pragma unroll
    for (int r = 0; r < R; r++) if (y0 + r < rows) b[(y0 + r) * pitch + x] = v[r];
}

template <typename F> static float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; i++) f();
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main()
{
    for (size_t mb : {84, 512}) {
        const size_t bytes = mb << 20;
        uint8_t *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes);
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_copy<uint32_t>, dim3(bytes / 4 / 256), dim3(256), 0, 0, (const uint32_t*)a, (uint32_t*)b, bytes / 4); });
        printf("%4zu MB  4 B/lane            %7.1f us  %6.0f GB/s (R+W)\n", mb, ms * 1e3, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_copy<uint2>, dim3(bytes / 8 / 256), dim3(256), 0, 0, (const uint2*)a, (uint2*)b, bytes / 8); });
        printf("%4zu MB  8 B/lane            %7.1f us  %6.0f GB/s\n", mb, ms * 1e3, 2.0 * bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_copy<uint4>, dim3(bytes / 16 / 256), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16); });
        printf("%4zu MB 16 B/lane            %7.1f us  %6.0f GB/s\n", mb, ms * 1e3, 2.0 * bytes / ms / 1e6);
        const size_t pitch4 = 176, rows4 = bytes / 4 / pitch4;     /* 704-byte rows like pyramid level 0 */
        ms = timeit([&] { hipLaunchKernelGGL((k_copy_rows<uint32_t, 4>), dim3(3, (rows4 + 3) / 4), dim3(64), 0, 0, (const uint32_t*)a, (uint32_t*)b, pitch4, rows4); });
        printf("%4zu MB  4 B/lane x 4 rows, 704-B rows, 64-thread blocks %7.1f us  %6.0f GB/s\n", mb, ms * 1e3, 2.0 * bytes / ms / 1e6);
        const size_t pitch16 = 44, rows16 = bytes / 16 / pitch16;
        ms = timeit([&] { hipLaunchKernelGGL((k_copy_rows<uint4, 4>), dim3(1, (rows16 + 3) / 4), dim3(64), 0, 0, (const uint4*)a, (uint4*)b, pitch16, rows16); });
        printf("%4zu MB 16 B/lane x 4 rows, 704-B rows, 64-thread blocks %7.1f us  %6.0f GB/s\n", mb, ms * 1e3, 2.0 * bytes / ms / 1e6);
        hipFree(a); hipFree(b);
    }
    return 0;
}
