/* pipeline_caller.cpp - a native caller of the THROUGHPUT path: device-resident batches through drfe_pipeline_* (three contexts
 * round robin, each on its own stream), results home through drfe_batch_download_async, all from C++ over include/drfe.h and the
 * HIP runtime (no Python, no torch).  Five batches of four frames (two alternating frames, so consecutive frames match against each
 * other) go in back to back; every batch must come back with the same bytes, and batch 0 is written out for
 * tests/test_gpu_native.py to compare with the ctypes path.
 * usage: pipeline_caller <gray0.raw> <depth0.raw> <gray1.raw> <depth1.raw> <w> <h> <out.bin> */
#include "drfe.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 10; } } while (0)

static std::vector<uint8_t> slurp(const char* path, size_t n)
{
    std::vector<uint8_t> b(n);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(b.data(), 1, n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::fclose(f);
    return b;
}

int main(int argc, char** argv)
{
    if (argc != 8) return 2;
    const int w = std::atoi(argv[5]), h = std::atoi(argv[6]), B = 4, NB = 5, DEPTH = 3;
    const size_t px = (size_t)w * h;
    std::vector<uint8_t> g[2] = {slurp(argv[1], px), slurp(argv[3], px)};
    std::vector<uint8_t> d[2] = {slurp(argv[2], px * 2), slurp(argv[4], px * 2)};
    uint8_t* dg = nullptr; uint16_t* dd = nullptr;
    CHK(hipMalloc((void**)&dg, px * B)); CHK(hipMalloc((void**)&dd, px * 2 * B));
    for (int f = 0; f < B; f++) {
        CHK(hipMemcpy(dg + px * f, g[f & 1].data(), px, hipMemcpyHostToDevice));
        CHK(hipMemcpy(dd + px * f, d[f & 1].data(), px * 2, hipMemcpyHostToDevice));
    }
    drfe_config cfg = {0, w, h, B, 1000, 1.2f, 8, 20, 7};
    drfe_pipeline* pipe = nullptr;
    if (drfe_pipeline_create(&cfg, DEPTH, &pipe) != DRFE_OK) { std::fprintf(stderr, "create: %s\n", drfe_last_error(nullptr)); return 3; }
    if (drfe_pipeline_depth(pipe) != DEPTH) return 4;
    const drfe_camera cam = {535.4f, 539.2f, 320.1f, 247.6f, 40.0f, 1.0f / 5000.0f, 0.0f, (float)w, 0.0f, (float)h};
    std::vector<float> T((size_t)B * 16, 0.0f);                       /* identity poses: the camera does not move between the frames */
    for (int f = 0; f < B; f++) for (int i = 0; i < 4; i++) T[(size_t)f * 16 + 5 * i] = 1.0f;
    const int K = drfe_orb_max_keypoints(drfe_pipeline_context(pipe, 0));
    const size_t kb = (size_t)B * K * sizeof(drfe_keypoint), db = (size_t)B * K * 32, mb = (size_t)B * K * 4;
    struct Res { uint8_t* p; };
    std::vector<Res> res(NB);
    for (int i = 0; i < NB; i++) CHK(hipHostMalloc((void**)&res[i].p, kb + db + mb + 8 * B, hipHostMallocDefault));
    for (int i = 0; i < NB; i++) {
        const int k = drfe_pipeline_submit(pipe, dg, dd, px, (size_t)w, w, h, T.data(), T.data(), &cam, 15.0f, 0, 1, B);
        if (k != i % DEPTH) { std::fprintf(stderr, "submit %d -> %d: %s\n", i, k, drfe_pipeline_last_error(pipe)); return 5; }
        uint8_t* r = res[i].p;
        /* the download is queued on the same stream as the batch: context k is free for batch i + DEPTH once it has run */
        if (drfe_batch_download_async(drfe_pipeline_context(pipe, k), B, (drfe_keypoint*)r, r + kb, (int32_t*)(r + kb + db + mb),
                                      (int32_t*)(r + kb + db), (int32_t*)(r + kb + db + mb + 4 * B), nullptr) != DRFE_OK) {
            std::fprintf(stderr, "download: %s\n", drfe_last_error(drfe_pipeline_context(pipe, k)));
            return 6;
        }
    }
    if (drfe_pipeline_sync(pipe, -1) != DRFE_OK) return 7;
    const int32_t* cnt0 = (const int32_t*)(res[0].p + kb + db + mb);
    const int32_t* mc0 = cnt0 + B;
    for (int i = 1; i < NB; i++) {                                       /* five identical batches: identical counts, keypoints, matches */
        const int32_t* cnt = (const int32_t*)(res[i].p + kb + db + mb);
        if (std::memcmp(cnt, cnt0, 8 * B) != 0) return 8;
        for (int f = 0; f < B; f++) {
            const size_t n = (size_t)cnt0[f];
            if (std::memcmp(res[i].p + (size_t)f * K * sizeof(drfe_keypoint), res[0].p + (size_t)f * K * sizeof(drfe_keypoint), n * sizeof(drfe_keypoint)) ||
                std::memcmp(res[i].p + kb + (size_t)f * K * 32, res[0].p + kb + (size_t)f * K * 32, n * 32) ||
                (f > 0 && std::memcmp(res[i].p + kb + db + (size_t)f * K * 4, res[0].p + kb + db + (size_t)f * K * 4, n * 4))) return 9;
        }
    }
    FILE* f = std::fopen(argv[7], "wb");
    std::fwrite(cnt0, 4, B, f);
    std::fwrite(mc0, 4, B, f);
    for (int s = 0; s < 2; s++) {                                         /* frames 0 and 1: keypoints, descriptors; frame 1: its matches */
        std::fwrite(res[0].p + (size_t)s * K * sizeof(drfe_keypoint), sizeof(drfe_keypoint), (size_t)cnt0[s], f);
        std::fwrite(res[0].p + kb + (size_t)s * K * 32, 32, (size_t)cnt0[s], f);
    }
    std::fwrite(res[0].p + kb + db + (size_t)1 * K * 4, 4, (size_t)cnt0[1], f);
    std::fclose(f);
    std::printf("pipeline_caller ok: %d batches of %d frames through %d contexts, %d / %d keypoints, %d matches\n", NB, B, DEPTH, cnt0[0], cnt0[1], mc0[1]);
    drfe_pipeline_destroy(pipe);
    return 0;
}
