/* capi.cpp — the C-ABI of libdrfe.so (include/drfe.h): context lifetime, HBM arenas, host<->device
 * staging.  No compute happens on the host: every entry point either launches the HIP kernels or
 * copies their results.  There is no CPU fallback — a missing GPU / HIP failure is DRFE_ERR_HIP. */
#include "drfe_internal.h"
#include <atomic>
#include "post_internal.h"
#include "match_internal.h"
#include "planes_internal.h"
#include "bow_internal.h"
#include "lines_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <cstdio>
#include <thread>
#include <new>

static std::string g_create_err;

static const int8_t kPatternHost[1024] = {
#include "../../include/drfe_orb_pattern.inc"
};

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return DRFE_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

template <class T>
static hipError_t dalloc(T** p, size_t n)
{
    *p = nullptr;
    return hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
}

static int upload_geometry(drfe_ctx* c, int w, int h)
{
    if (c->geom.imgW == w && c->geom.imgH == h) return DRFE_OK;
    if (w > c->cfg.max_width || h > c->cfg.max_height) {
        c->err = "frame larger than drfe_config.max_width/max_height";
        return DRFE_ERR_INVALID;
    }
    DevGeom g;
    std::memset(&g, 0, sizeof(g));
    std::vector<FastCell> cells;
    std::vector<BlurTile> tiles;
    std::vector<ResizeTap> taps;
    int rc = drfe_build_geometry(c, w, h, &g, &cells, &tiles, &taps);
    if (rc != DRFE_OK) return rc;
    if ((size_t)g.pyrSlotBytes > c->pyrSlotBytesMax || (size_t)g.blurSlotBytes > c->blurSlotBytesMax ||
        (size_t)g.candSlotElems > c->candSlotElemsMax || g.kpSlotElems > c->maxKp ||
        (int)cells.size() > c->cellsCap || (int)tiles.size() > c->tilesCap || (int)taps.size() > c->tapsCap) {
        c->err = "geometry exceeds the arenas sized at drfe_create";
        return DRFE_ERR_CAPACITY;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(c->d_geom, &g, sizeof(g), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_cells, cells.data(), cells.size() * sizeof(FastCell), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_tiles, tiles.data(), tiles.size() * sizeof(BlurTile), hipMemcpyHostToDevice));
    if (!taps.empty())
        HIPCHK(c, hipMemcpy(c->d_taps, taps.data(), taps.size() * sizeof(ResizeTap), hipMemcpyHostToDevice));
    c->geom = g;
    c->lastBatch = 0;
    c->glueValid = false;
    /* no slot holds a frame of this geometry yet: a slot-addressed call on a slot that drfe_frame_submit has not filled
     * (lastBatch only remembers the highest one) finds zero keypoints instead of another geometry's leftovers */
    HIPCHK(c, hipMemset(c->d_kpCount, 0, sizeof(int) * (size_t)c->cfg.max_batch));
    return DRFE_OK;
}

int drfe_default_host_threads()
{
    int n = 0;
#if defined(__linux__)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long period = 0;
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long long quota = std::atoll(q);
            const int lim = (int)std::max<long long>(1, quota / period);
            if (n <= 0 || lim < n) n = lim;
        }
        std::fclose(f);
    }
#endif
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return std::max(1, n);
}

hipError_t drfe_long_kernel_stream(hipStream_t* s, int part)
{
    static const int share = [] { const char* e = std::getenv("DRFE_CU_SPLIT"); const int v = e ? std::atoi(e) : 0; return v < 0 ? 0 : v > 95 ? 95 : v; }();
    if (share <= 0) {
        int prLow = 0, prHigh = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&prLow, &prHigh);
        if (e != hipSuccess) return e;
        /* The runtime maps the streams of one priority onto FOUR hardware queues: with eight long streams alive (four steps in
         * flight x lines + planes) two of them share a queue and their 0.1 s kernels run one behind the other (tools/queue_probe.py:
         * 4 contexts x 64 frames side by side 165 ms, 8 contexts 331 ms; 168 ms with DRFE_LONG_PRIO=split, which puts the plane
         * path's long kernels on the middle priority level and so on four queues of their own).  With 512-frame steps it does
         * not matter - the full front-end is bound by the LDS its long kernels hold for as long as they run,
         * the same rate either way (profiles/r04_queue_probe.txt) - so the default stays the lowest level for both: the ORB batch,
         * CAPE and the pools' lanes keep the middle level to themselves.  DRFE_CU_SPLIT=<percent> gives the line path that share of
         * every 32 CUs and the plane path the rest (an experiment: 3 350-3 520 frames/s at 50 / 62 against 4 580-4 850 with every
         * kernel free to run anywhere - the mix packs the LDS better than a partition).  Either path's long kernels one priority
         * level above the other's (DRFE_LONG_PRIO=split / lines) is slower as well: 4 310-4 390 / 3 860-4 190 against 4 940-5 020. */
        static const int split = [] { const char* m = std::getenv("DRFE_LONG_PRIO"); return !m ? 0 : (m[0] == 's' && m[1] == 'p') ? 1 : (m[0] == 'l' ? 2 : 0); }();      /* "split": planes up, "lines": lines up */
        const int mid = (prLow + prHigh) / 2;
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, (mid != prLow && ((part == 1 && split == 1) || (part == 0 && split == 2))) ? mid : prLow);
    }
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    /* the same share of every group of 32 CUs, whichever way the mask's bits map onto XCDs */
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
    const int cut = 32 * share / 100;
    for (int i = 0; i < cus; i++) {
        const bool first = (i % 32) < cut;
        if (first == (part == 0)) mask[(size_t)i / 32] |= 1u << (i % 32);
    }
    return hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data());
}

static std::atomic<long long> g_poolCpuNs[3];
void drfe_pool_cpu_add(int pool, long long ns) { if (pool >= 0 && pool < 3) g_poolCpuNs[pool] += ns; }
extern "C" void drfe_debug_pool_cpu_ns(long long* out3) { for (int k = 0; k < 3; k++) out3[k] = g_poolCpuNs[k].exchange(0); }

extern "C" {

const char* drfe_version(void) { return "drfe 0.1 (gfx950)"; }

const char* drfe_last_error(const drfe_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

void drfe_destroy(drfe_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drfe_match_buffers_free(c);
    drfe_planes_free(c);
    drfe_bow_free(c);
    drfe_lines_free(c);
    drfe_cape_lanes_free(c);
    drfe_post_free(c);
    drfe_one_shot_free(c);
    drfe_frame_lanes_free(c);
    void* ptrs[] = {c->d_geom, c->d_cells, c->d_tiles, c->d_taps, c->d_pattern, c->d_disc, c->d_pyr, c->d_blur,
                    c->d_cand0, c->d_cand1, c->d_node, c->d_candCount, c->d_sel, c->d_selCount, c->d_kps, c->d_kpsUn, c->d_desc,
                    c->d_kpCount, c->d_status, c->d_uRight, c->d_depth, c->d_gridOff, c->d_gridIdx, c->d_cellKp, c->d_cellDesc, c->d_match,
                    c->d_matchCount, c->d_poses, c->d_stage, c->d_callScratch, c->d_kpUV, c->d_kpDepth};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int i = 0; i < DRFE_STAGE_COUNT; i++)
        for (int j = 0; j < 2; j++)
            if (c->ev[i][j]) (void)hipEventDestroy(c->ev[i][j]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int drfe_create(const drfe_config* cfg, drfe_ctx** out)
{
    if (!cfg || !out) { g_create_err = "null argument"; return DRFE_ERR_INVALID; }
    *out = nullptr;
    if (cfg->nlevels < 1 || cfg->nlevels > DRFE_MAX_LEVELS || cfg->max_batch < 1 || cfg->nfeatures < 1 ||
        !(cfg->scale_factor > 1.0f) || cfg->max_width < 64 || cfg->max_height < 64 || cfg->max_width > 4095 ||
        cfg->max_height > 4095) {
        g_create_err = "invalid drfe_config";
        return DRFE_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_err = "no HIP device: libdrfe has no CPU path";
        return DRFE_ERR_HIP;
    }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "device ordinal out of range"; return DRFE_ERR_INVALID; }
    drfe_ctx* c = new (std::nothrow) drfe_ctx();
    if (!c) { g_create_err = "out of host memory"; return DRFE_ERR_INVALID; }
    c->cfg = *cfg;
    c->device = cfg->device;
    c->stream = nullptr;
    c->profile = false;
    c->lastBatch = 0;
    c->glueValid = false;
    c->mb = nullptr;
    c->ps = nullptr;
    c->planeLanes = nullptr;
    c->bow = nullptr;
    c->ls = nullptr;
    c->lsBatch = nullptr;
    c->capeLanes = nullptr;
    c->lsdDeviceGrow = 1;
    c->lsdRectMode = 0;
    c->capeBatch = nullptr;
    c->planesDeviceCape = 1;
    c->lsdDeviceNfa = 1;
    c->lsdStats[0] = c->lsdStats[1] = c->lsdStats[2] = c->lsdStats[3] = 0;
    c->capeStats[0] = c->capeStats[1] = 0;
    c->ahcStats[0] = c->ahcStats[1] = c->ahcStats[2] = c->ahcStats[3] = 0;
    c->planesDeviceVoxel = 1;
    c->planesDeviceRefit = 1;
    c->ahcRefitStats[0] = c->ahcRefitStats[1] = 0;
    c->planesDeviceAhc = 1;
    c->ahcArena = nullptr;
    c->lineWorkers = nullptr;
    c->frameLanes = nullptr;
    c->lineHost = nullptr;
    std::memset(&c->cam, 0, sizeof(c->cam));
    std::memset(&c->geom, 0, sizeof(c->geom));
    std::memset(c->ev, 0, sizeof(c->ev));
    std::memset(c->evUsed, 0, sizeof(c->evUsed));
    c->d_geom = nullptr; c->d_cells = nullptr; c->d_tiles = nullptr; c->d_taps = nullptr; c->d_pattern = nullptr;
    c->d_callScratch = nullptr; c->callScratchBytes = 0;
    c->d_disc = nullptr; c->d_pyr = nullptr; c->d_blur = nullptr; c->d_cand0 = nullptr; c->d_cand1 = nullptr;
    c->d_node = nullptr; c->d_candCount = nullptr; c->d_sel = nullptr; c->d_selCount = nullptr; c->d_kps = nullptr; c->d_kpsUn = nullptr; std::memset(&c->dist, 0, sizeof(c->dist));
    c->d_desc = nullptr; c->d_kpCount = nullptr; c->d_status = nullptr; c->d_uRight = nullptr; c->d_depth = nullptr;
    c->d_gridOff = nullptr; c->d_gridIdx = nullptr; c->d_cellKp = nullptr; c->d_cellDesc = nullptr; c->d_match = nullptr; c->d_matchCount = nullptr;
    c->d_poses = nullptr; c->d_stage = nullptr;

#define CREATE_FAIL(code)                 \
    do {                                  \
        g_create_err = c->err;            \
        drfe_destroy(c);                  \
        return (code);                    \
    } while (0)
#define CHIP(call)                                                                  \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess) {                                                    \
            c->err = std::string(#call) + ": " + hipGetErrorString(e__);            \
            CREATE_FAIL(DRFE_ERR_HIP);                                              \
        }                                                                           \
    } while (0)

    CHIP(hipSetDevice(c->device));
    CHIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i < DRFE_STAGE_COUNT; i++)
        for (int j = 0; j < 2; j++) CHIP(hipEventCreate(&c->ev[i][j]));
    drfe_build_tables(c);
    {
        const char* e = std::getenv("DRFE_FAST_GENERIC");
        c->fastGeneric = (e && e[0] == '1') ? 1 : 0;
        const char* es = std::getenv("DRFE_FAST_SCREEN");
        c->fastScreen = (es && es[0] >= '0' && es[0] <= '2') ? es[0] - '0' : 2;     /* 0 never, 1 at minThFAST only (rounds 3-4), 2 at iniThFAST first */
    }

    /* size the arenas with the geometry of the largest frame */
    DevGeom gmax;
    std::memset(&gmax, 0, sizeof(gmax));
    std::vector<FastCell> cells;
    std::vector<BlurTile> tiles;
    std::vector<ResizeTap> taps;
    int rc = drfe_build_geometry(c, cfg->max_width, cfg->max_height, &gmax, &cells, &tiles, &taps);
    if (rc != DRFE_OK) CREATE_FAIL(rc);
    const size_t B = (size_t)cfg->max_batch;
    c->pyrSlotBytesMax = (size_t)gmax.pyrSlotBytes;
    c->blurSlotBytesMax = (size_t)gmax.blurSlotBytes;
    c->candSlotElemsMax = (size_t)gmax.candSlotElems;
    c->maxKp = gmax.kpSlotElems;
    c->cellsCap = (int)cells.size() + 64;
    c->tilesCap = (int)tiles.size() + 64;
    c->tapsCap = (int)taps.size() + 64;
    const int nl = cfg->nlevels;

    CHIP(dalloc(&c->d_geom, 1));
    CHIP(dalloc(&c->d_cells, (size_t)c->cellsCap));
    CHIP(dalloc(&c->d_tiles, (size_t)c->tilesCap));
    CHIP(dalloc(&c->d_taps, (size_t)c->tapsCap));
    CHIP(dalloc(&c->d_pattern, 1024));
    CHIP(hipMemcpy(c->d_pattern, kPatternHost, 1024, hipMemcpyHostToDevice));
    {   /* disc offsets of IC_Angle: rows v = -15..15, u = -umax[|v|]..umax[|v|] (749 pixels) */
        std::vector<int16_t> disc;
        for (int v = -DRFE_HALF_PATCH; v <= DRFE_HALF_PATCH; v++) {
            const int d = c->umax[v < 0 ? -v : v];
            for (int u = -d; u <= d; u++) { disc.push_back((int16_t)u); disc.push_back((int16_t)v); }
        }
        c->discCount = (int)disc.size() / 2;
        if (c->discCount > 12 * 64) { c->err = "IC_Angle disc exceeds 12 offsets per lane"; CREATE_FAIL(DRFE_ERR_INVALID); }
        CHIP(dalloc(&c->d_disc, disc.size()));
        CHIP(hipMemcpy(c->d_disc, disc.data(), disc.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    /* + 256: k_blur reads whole dwords up to 24 bytes past a tile's last column without clamping; only the last row of the last
     * level of the last slot can take that past the arena (the values are never stored) */
    CHIP(dalloc(&c->d_pyr, B * c->pyrSlotBytesMax + 256));
    CHIP(dalloc(&c->d_blur, B * c->blurSlotBytesMax));
    CHIP(dalloc(&c->d_cand0, B * c->candSlotElemsMax));
    CHIP(dalloc(&c->d_cand1, B * c->candSlotElemsMax));
    CHIP(dalloc(&c->d_node, B * c->candSlotElemsMax));
    CHIP(dalloc(&c->d_candCount, B * DRFE_CC_SLOT));
    CHIP(dalloc(&c->d_sel, B * (size_t)c->maxKp));
    CHIP(dalloc(&c->d_selCount, B * nl));
    CHIP(dalloc(&c->d_kps, B * (size_t)c->maxKp));
    CHIP(dalloc(&c->d_desc, B * (size_t)c->maxKp * 32));
    CHIP(dalloc(&c->d_kpCount, B));
    CHIP(dalloc(&c->d_status, 4));
    CHIP(dalloc(&c->d_uRight, B * (size_t)c->maxKp));
    CHIP(dalloc(&c->d_depth, B * (size_t)c->maxKp));
    CHIP(dalloc(&c->d_gridOff, B * (DRFE_GRID_CELLS + 1)));
    CHIP(dalloc(&c->d_gridIdx, B * (size_t)c->maxKp));
    CHIP(dalloc(&c->d_cellKp, B * (size_t)c->maxKp));
    CHIP(dalloc(&c->d_cellDesc, B * (size_t)c->maxKp * 2));
    CHIP(dalloc(&c->d_match, B * (size_t)c->maxKp));
    CHIP(dalloc(&c->d_matchCount, B));
    CHIP(dalloc(&c->d_poses, 2 * B * 16));
    c->stageBytes = (size_t)cfg->max_width * cfg->max_height * 2;
    CHIP(dalloc(&c->d_stage, c->stageBytes));
    CHIP(hipMemset(c->d_kpCount, 0, sizeof(int) * B));
    CHIP(hipMemset(c->d_selCount, 0, sizeof(int) * B * nl));
    CHIP(hipMemset(c->d_status, 0, sizeof(int) * 4));
#undef CHIP
#undef CREATE_FAIL
    *out = c;
    return DRFE_OK;
}

int drfe_orb_scale_tables(const drfe_ctx* c, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2)
{
    if (!c) return DRFE_ERR_INVALID;
    for (int i = 0; i < c->cfg.nlevels; i++) {
        if (scale) scale[i] = c->scale[i];
        if (inv_scale) inv_scale[i] = c->invScale[i];
        if (sigma2) sigma2[i] = c->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = c->invSigma2[i];
    }
    return DRFE_OK;
}

int drfe_orb_max_keypoints(const drfe_ctx* c) { return c ? c->maxKp : DRFE_ERR_INVALID; }

static int check_status(drfe_ctx* c)
{
    int st = 0;
    HIPCHK(c, hipMemcpy(&st, c->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st & 1) { c->err = "FAST candidate arena overflow"; return DRFE_ERR_CAPACITY; }
    if (st & 2) { c->err = "quadtree node pool overflow"; return DRFE_ERR_CAPACITY; }
    return DRFE_OK;
}

int drfe_orb_extract_batch(drfe_ctx* c, const uint8_t* d_gray, size_t frame_stride, size_t row_stride, int w, int h,
                           int nframes, void* stream)
{
    if (!c || !d_gray || nframes < 1 || nframes > c->cfg.max_batch || w < 1 || h < 1 || row_stride < (size_t)w) {
        if (c) c->err = "drfe_orb_extract_batch: invalid argument";
        return DRFE_ERR_INVALID;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = upload_geometry(c, w, h);
    if (rc != DRFE_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, drfe_launch_orb(c, d_gray, frame_stride, row_stride, nframes, s));
    c->lastBatch = nframes;
    c->glueValid = false;
    return DRFE_OK;
}

int drfe_stream_sync(drfe_ctx* c)
{
    if (!c) return DRFE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    return DRFE_OK;
}

int drfe_orb_counts(drfe_ctx* c, int nframes, int* counts)
{
    if (!c || !counts || nframes < 1 || nframes > c->lastBatch) return DRFE_ERR_INVALID;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    rc = check_status(c);
    if (rc != DRFE_OK) return rc;
    HIPCHK(c, hipMemcpy(counts, c->d_kpCount, sizeof(int) * nframes, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_orb_download(drfe_ctx* c, int slot, drfe_keypoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!c || !n_out || slot < 0 || slot >= c->lastBatch) {
        if (c) c->err = "drfe_orb_download: invalid slot";
        return c ? DRFE_ERR_STATE : DRFE_ERR_INVALID;
    }
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    rc = check_status(c);
    if (rc != DRFE_OK) return rc;
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_kpCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) { c->err = "keypoint buffer too small"; return DRFE_ERR_CAPACITY; }
    if (n > 0 && kps)
        HIPCHK(c, hipMemcpy(kps, c->d_kps + (size_t)slot * c->maxKp, sizeof(drfe_keypoint) * n, hipMemcpyDeviceToHost));
    if (n > 0 && desc)
        HIPCHK(c, hipMemcpy(desc, c->d_desc + (size_t)slot * c->maxKp * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_orb_fast_partition(drfe_ctx* c, int w, int h, int32_t* ncells, int64_t* pixels)
{
    if (!c || !ncells || !pixels) return DRFE_ERR_INVALID;
    DevGeom g;
    std::memset(&g, 0, sizeof(g));
    std::vector<FastCell> cells;
    const int rc = drfe_build_geometry(c, w, h, &g, &cells, nullptr, nullptr);
    if (rc != DRFE_OK) return rc;
    ncells[0] = ncells[1] = 0;
    pixels[0] = pixels[1] = 0;
    const bool cols = g.fastCols && !c->fastGeneric;
    for (size_t i = 0; i < cells.size(); i++) {
        const int k = (cols && (int)i >= g.fastColsSmall) ? 1 : 0;
        ncells[k]++;
        pixels[k] += (int64_t)(cells[i].ww - 6) * (cells[i].wh - 6);
    }
    return DRFE_OK;
}

int drfe_batch_download_async(drfe_ctx* c, int nframes, drfe_keypoint* kps, uint8_t* desc, int32_t* kp_counts, int32_t* matches,
                              int32_t* match_counts, void* stream)
{
    if (!c) return DRFE_ERR_INVALID;
    if (nframes < 1 || nframes > c->lastBatch || !kps || !desc || !kp_counts) {
        c->err = "drfe_batch_download_async: invalid argument";
        return nframes > c->lastBatch ? DRFE_ERR_STATE : DRFE_ERR_INVALID;
    }
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t F = (size_t)nframes, K = (size_t)c->maxKp;
    HIPCHK(c, hipMemcpyAsync(kps, c->d_kps, F * K * sizeof(drfe_keypoint), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(desc, c->d_desc, F * K * 32, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(kp_counts, c->d_kpCount, F * sizeof(int), hipMemcpyDeviceToHost, s));
    if (matches) HIPCHK(c, hipMemcpyAsync(matches, c->d_match, F * K * sizeof(int), hipMemcpyDeviceToHost, s));
    if (match_counts) HIPCHK(c, hipMemcpyAsync(match_counts, c->d_matchCount, F * sizeof(int), hipMemcpyDeviceToHost, s));
    return DRFE_OK;
}

/* The device status words of the batch path, asynchronously like drfe_batch_download_async: status[0] = extraction (bit 0:
 * FAST candidate arena overflow, bit 1: quadtree node pool overflow), status[1] = drfe_match_consecutive_batch (bit 2: more
 * than DRFE_MATCH_MAX_CAND keypoints in a search window).  Non-zero = the batch's keypoints / matches are truncated. */
int drfe_batch_status_async(drfe_ctx* c, int32_t* status, void* stream)
{
    if (!c || !status) { if (c) c->err = "drfe_batch_status_async: invalid argument"; return DRFE_ERR_INVALID; }
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, hipMemcpyAsync(&status[0], c->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&status[1], c->d_status + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    return DRFE_OK;
}

/* the same, blocking, as an error code: DRFE_ERR_CAPACITY (with the reason in drfe_last_error) if the batch most recently
 * extracted / matched on this context overflowed an arena.  Waits for the context's stream. */
int drfe_batch_check(drfe_ctx* c)
{
    if (!c) return DRFE_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int st[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpy(st, c->d_status, sizeof(st), hipMemcpyDeviceToHost));
    if (st[0] & 1) { c->err = "FAST candidate arena overflow"; return DRFE_ERR_CAPACITY; }
    if (st[0] & 2) { c->err = "quadtree node pool overflow"; return DRFE_ERR_CAPACITY; }
    if (st[2] & 4) { c->err = "match candidate list overflow (DRFE_MATCH_MAX_CAND)"; return DRFE_ERR_CAPACITY; }
    return DRFE_OK;
}

} /* extern "C" */

/* The single-frame entry as ONE graph launch: pinned input -> H2D -> the 20-odd kernels of drfe_launch_orb -> status, count,
 * keypoints and descriptors D2H into pinned memory.  Tracking calls this once per frame; at one frame the kernels are
 * launch-latency bound, so replaying a captured hipGraph replaces ~25 enqueues, four blocking copies and two
 * synchronisations by one launch and one synchronisation.  Captured per (w, h); DRFE_NO_GRAPH=1 keeps the plain path. */
struct OrbOneShot {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int w = 0, h = 0;
    uint8_t* h_in = nullptr;      /* pinned, w * h */
    uint8_t* h_out = nullptr;     /* pinned: int status | int count | kps[maxKp] | desc[maxKp][32] */
    uint8_t* d_out = nullptr;     /* the same layout on the device: the graph ends in one download */
    size_t inBytes = 0;
    bool disabled = false;
};

static void one_shot_release(OrbOneShot* o)
{
    if (o->exec) (void)hipGraphExecDestroy(o->exec);
    if (o->graph) (void)hipGraphDestroy(o->graph);
    o->exec = nullptr; o->graph = nullptr; o->w = o->h = 0;
}

void drfe_one_shot_free(drfe_ctx* c)
{
    if (!c->oneShot) return;
    one_shot_release(c->oneShot);
    if (c->oneShot->h_in) (void)hipHostFree(c->oneShot->h_in);
    if (c->oneShot->h_out) (void)hipHostFree(c->oneShot->h_out);
    if (c->oneShot->d_out) (void)hipFree(c->oneShot->d_out);
    delete c->oneShot;
    c->oneShot = nullptr;
}

static int one_shot_prepare(drfe_ctx* c, int w, int h)
{
    if (!c->oneShot) {
        c->oneShot = new (std::nothrow) OrbOneShot();
        if (!c->oneShot) return DRFE_ERR_INVALID;
        const char* e = std::getenv("DRFE_NO_GRAPH");
        c->oneShot->disabled = e && e[0] == '1';
    }
    OrbOneShot* o = c->oneShot;
    if (o->disabled) return DRFE_OK;
    /* the device tables must be this size's before the graph (whose launches carry this size's arguments) replays: a batch or
     * per-frame call at another size may have re-uploaded them since the capture.  No-op when the geometry is unchanged;
     * table uploads are not capturable, so also before a capture. */
    int rc = upload_geometry(c, w, h);
    if (rc != DRFE_OK) return rc;
    if (o->exec && o->w == w && o->h == h) return DRFE_OK;
    one_shot_release(o);
    const size_t K = (size_t)c->maxKp, outBytes = 8 + K * sizeof(drfe_keypoint) + K * 32;
    if (o->inBytes < (size_t)w * h) {
        if (o->h_in) (void)hipHostFree(o->h_in);
        o->h_in = nullptr;
        HIPCHK(c, hipHostMalloc((void**)&o->h_in, (size_t)w * h, hipHostMallocDefault));
        o->inBytes = (size_t)w * h;
    }
    if (!o->h_out) HIPCHK(c, hipHostMalloc((void**)&o->h_out, outBytes, hipHostMallocDefault));
    if (!o->d_out) HIPCHK(c, hipMalloc((void**)&o->d_out, outBytes));
    hipStream_t s = c->stream;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { o->disabled = true; (void)hipGetLastError(); return DRFE_OK; }
    hipError_t e = hipMemcpyAsync(c->d_stage, o->h_in, (size_t)w * h, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = drfe_launch_orb(c, c->d_stage, (size_t)w * h, (size_t)w, 1, s);
    {   /* results gathered into one staging buffer, one download (see frame_enqueue) */
        DrfePackArgs A;
        std::memset(&A, 0, sizeof(A));
        A.src[0] = reinterpret_cast<const uint32_t*>(c->d_status); A.dwords[0] = 1;
        A.src[1] = reinterpret_cast<const uint32_t*>(c->d_kpCount); A.dwords[1] = 1;
        A.src[2] = reinterpret_cast<const uint32_t*>(c->d_kps); A.dwords[2] = (uint32_t)(K * sizeof(drfe_keypoint) / 4);
        A.src[3] = reinterpret_cast<const uint32_t*>(c->d_desc); A.dwords[3] = (uint32_t)(K * 8);
        A.n = 4;
        if (e == hipSuccess) e = drfe_launch_pack_segments(A, reinterpret_cast<uint32_t*>(o->d_out), s);
        if (e == hipSuccess) e = hipMemcpyAsync(o->h_out, o->d_out, outBytes, hipMemcpyDeviceToHost, s);
    }
    hipGraph_t g = nullptr;
    const hipError_t ee = hipStreamEndCapture(s, &g);
    if (e != hipSuccess || ee != hipSuccess || !g || hipGraphInstantiate(&o->exec, g, nullptr, nullptr, 0) != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        o->exec = nullptr;
        o->disabled = true;                   /* graphs unavailable: the plain path stays correct */
        (void)hipGetLastError();
        return DRFE_OK;
    }
    o->graph = g; o->w = w; o->h = h;
    return DRFE_OK;
}

extern "C" {

int drfe_orb_extract(drfe_ctx* c, const uint8_t* gray, int w, int h, size_t stride, drfe_keypoint* kps, uint8_t* desc,
                     int cap, int* n_out)
{
    if (!c || !n_out) return DRFE_ERR_INVALID;
    *n_out = 0;
    if (!gray || w == 0 || h == 0) return DRFE_OK; /* reference: silent return on empty image */
    if (w < 0 || h < 0 || stride < (size_t)w || (size_t)w * h > c->stageBytes) {
        c->err = "drfe_orb_extract: invalid image";
        return DRFE_ERR_INVALID;
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->profile) {
        const int rcp = one_shot_prepare(c, w, h);
        if (rcp != DRFE_OK) return rcp;
    }
    OrbOneShot* o = c->oneShot;
    if (o && o->exec && !o->disabled && !c->profile && o->w == w && o->h == h) {
        for (int y = 0; y < h; y++) std::memcpy(o->h_in + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
        HIPCHK(c, hipGraphLaunch(o->exec, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->lastBatch = 1;
        c->glueValid = false;
        int st = 0, n = 0;
        std::memcpy(&st, o->h_out, 4);
        std::memcpy(&n, o->h_out + 4, 4);
        if (st & 1) { c->err = "FAST candidate arena overflow"; return DRFE_ERR_CAPACITY; }
        if (st & 2) { c->err = "quadtree node pool overflow"; return DRFE_ERR_CAPACITY; }
        *n_out = n;
        if (n > cap) { c->err = "keypoint buffer too small"; return DRFE_ERR_CAPACITY; }
        if (n > 0 && kps) std::memcpy(kps, o->h_out + 8, sizeof(drfe_keypoint) * (size_t)n);
        if (n > 0 && desc) std::memcpy(desc, o->h_out + 8 + (size_t)c->maxKp * sizeof(drfe_keypoint), (size_t)n * 32);
        return DRFE_OK;
    }
    HIPCHK(c, hipMemcpy2DAsync(c->d_stage, (size_t)w, gray, stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice,
                               c->stream));
    int rc = drfe_orb_extract_batch(c, c->d_stage, (size_t)w * h, (size_t)w, w, h, 1, c->stream);
    if (rc != DRFE_OK) return rc;
    return drfe_orb_download(c, 0, kps, desc, cap, n_out);
}

int drfe_orb_pyramid_level(drfe_ctx* c, int slot, int level, uint8_t* out, int* bw, int* bh)
{
    if (!c || slot < 0 || slot >= c->lastBatch || level < 0 || level >= c->cfg.nlevels) return DRFE_ERR_INVALID;
    const DevLevel& L = c->geom.lv[level];
    const int W = L.w + 2 * DRFE_EDGE, H = L.h + 2 * DRFE_EDGE;
    if (bw) *bw = W;
    if (bh) *bh = H;
    if (!out) return DRFE_OK;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    HIPCHK(c, hipMemcpy2D(out, (size_t)W, c->d_pyr + (size_t)slot * c->geom.pyrSlotBytes + L.pyrOff, (size_t)L.pyrPitch,
                          (size_t)W, (size_t)H, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_orb_blurred_level(drfe_ctx* c, int slot, int level, uint8_t* out, int* w, int* h)
{
    if (!c || slot < 0 || slot >= c->lastBatch || level < 0 || level >= c->cfg.nlevels) return DRFE_ERR_INVALID;
    const DevLevel& L = c->geom.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!out) return DRFE_OK;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    /* the level is tiled on the device (drfe_blur_offset): fetch it whole, untile on the host */
    const size_t bytes = (size_t)L.blurPitch * ((L.h + DRFE_BTILE_H - 1) / DRFE_BTILE_H) * 128;
    std::vector<uint8_t> tiled(bytes);
    HIPCHK(c, hipMemcpy(tiled.data(), c->d_blur + (size_t)slot * c->geom.blurSlotBytes + L.blurOff, bytes, hipMemcpyDeviceToHost));
    for (int y = 0; y < L.h; y++)
        for (int x = 0; x < L.w; x++) out[(size_t)y * L.w + x] = tiled[drfe_blur_offset(x, y, L.blurPitch)];
    return DRFE_OK;
}

int drfe_orb_candidates(drfe_ctx* c, int slot, int level, int32_t* xyr, int cap, int* n_out)
{
    if (!c || !n_out || slot < 0 || slot >= c->lastBatch || level < 0 || level >= c->cfg.nlevels) return DRFE_ERR_INVALID;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    rc = check_status(c);
    if (rc != DRFE_OK) return rc;
    const DevLevel& L = c->geom.lv[level];
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_candCount + DRFE_CC_IDX(slot, level), sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (!xyr) return DRFE_OK;
    if (n > cap) return DRFE_ERR_CAPACITY;
    std::vector<uint32_t> k0(n), k1(n);
    const size_t off = (size_t)slot * c->geom.candSlotElems + L.candOff;
    if (n) {
        HIPCHK(c, hipMemcpy(k0.data(), c->d_cand0 + off, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(k1.data(), c->d_cand1 + off, sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    }
    /* report in emission order (the order key is unique) */
    std::vector<int> perm(n);
    for (int i = 0; i < n; i++) perm[i] = i;
    std::sort(perm.begin(), perm.end(), [&](int a, int b) { return k1[a] < k1[b]; });
    for (int i = 0; i < n; i++) {
        const uint32_t k = k0[perm[i]];
        xyr[3 * i] = (int32_t)(k & 0xFFF);
        xyr[3 * i + 1] = (int32_t)((k >> 12) & 0xFFF);
        xyr[3 * i + 2] = (int32_t)(k >> 24);
    }
    return DRFE_OK;
}

int drfe_profile_enable(drfe_ctx* c, int on)
{
    if (!c) return DRFE_ERR_INVALID;
    c->profile = on != 0;
    std::memset(c->evUsed, 0, sizeof(c->evUsed));
    return DRFE_OK;
}

int drfe_profile_stage_ms(drfe_ctx* c, float* ms)
{
    if (!c || !ms) return DRFE_ERR_INVALID;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    for (int i = 0; i < DRFE_STAGE_COUNT; i++) {
        ms[i] = 0.f;
        if (c->evUsed[i]) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, c->ev[i][0], c->ev[i][1]) == hipSuccess) ms[i] = t;
        }
    }
    return DRFE_OK;
}

} /* extern "C" */


/* ------------------------------------------------------------------------------------------------ */
/* Per-frame pipelined flow ("async variants take a frame slot id for pipelining").
 * Tracking::GrabImageRGBD (src/Tracking.cc:191) builds one Frame at a time and matches it against the previous one, so
 * frame k is extracted into slot k % max_batch while the slot of frame k-1 keeps LastFrame's keypoints, descriptors and
 * grid on the device for the slot-pair matchers.  A submission is the H2D of the frame, the kernels of drfe_launch_orb
 * and (with a depth image) drfe_launch_glue on that ONE slot, and the D2H of its results, replayed as one captured
 * hipGraph per slot; it returns without waiting, so the calling thread can run the host halves of the line / plane
 * extractors meanwhile (the reference starts three threads for that, src/Frame.cc:124-134). */

struct FrameLane {
    uint8_t* h_in = nullptr;      /* pinned: gray w*h, then raw depth w*h*2 */
    uint8_t* h_out = nullptr;     /* pinned: int status | int count | kps[K] | desc[K][32] | uRight[K] | depth[K] */
    uint8_t* d_in = nullptr;      /* device staging, same layout as h_in */
    uint8_t* d_out = nullptr;     /* device staging of the results, same layout as h_out: one download per frame */
    size_t inBytes = 0;
    hipEvent_t done = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int w = 0, h = 0, withDepth = 0;
    drfe_camera cam;
    DrfeDistortion dist;
    bool pending = false, graphOff = false;
    /* drfe_frame_submit_tracked: SearchByProjection(this frame, LastFrame) behind the glue.  The pair record, LastFrame's Twc
     * and (caller-supplied) map points travel in the staging behind the images; everything a kernel takes by value is part
     * of the graph's key */
    int tracked = 0, lastSlot = -1, mpsFromCaller = 0, checkOri = 0;
    float th = 0.f;
    size_t trackOff = 0;          /* byte offset of the tracking block inside h_in / d_in */
};

/* tracking block of a staged frame: MatchPair | Twc of LastFrame | map points */
static size_t track_block_bytes(const drfe_ctx* c) { return 256 + (size_t)c->maxKp * sizeof(drfe_map_point); }

static void frame_lane_release_graph(FrameLane& L)
{
    if (L.exec) (void)hipGraphExecDestroy(L.exec);
    if (L.graph) (void)hipGraphDestroy(L.graph);
    L.exec = nullptr; L.graph = nullptr;
}

void drfe_frame_lanes_free(drfe_ctx* c)
{
    auto* v = static_cast<std::vector<FrameLane>*>(c->frameLanes);
    if (!v) return;
    for (FrameLane& L : *v) {
        frame_lane_release_graph(L);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.h_in) (void)hipHostFree(L.h_in);
        if (L.h_out) (void)hipHostFree(L.h_out);
        if (L.d_out) (void)hipFree(L.d_out);
        if (L.d_in) (void)hipFree(L.d_in);
    }
    delete v;
    c->frameLanes = nullptr;
}

/* the stream work of one submission (capturable: no table upload, no synchronisation) */
static hipError_t frame_enqueue(drfe_ctx* c, FrameLane& L, int slot, int w, int h, hipStream_t s)
{
    const size_t K = (size_t)c->maxKp, px = (size_t)w * h;
    /* one upload: the images and, behind them, the tracking block (pair record, Twc and - caller-supplied only - map points) */
    size_t up = L.withDepth ? px * 3 : px;
    if (L.tracked) up = L.trackOff + (L.mpsFromCaller ? track_block_bytes(c) : 256);
    hipError_t e = hipMemcpyAsync(L.d_in, L.h_in, up, hipMemcpyHostToDevice, s);
    {
        SlotShift shift(c, slot);
        if (e == hipSuccess) e = drfe_launch_orb(c, L.d_in, px, (size_t)w, 1, s);
        if (e == hipSuccess && L.withDepth)
            e = drfe_launch_glue(c, reinterpret_cast<const uint16_t*>(L.d_in + px), px, (size_t)w, L.cam, 1, s);
    }
    if (L.tracked && e == hipSuccess) {
        /* ORBmatcher::SearchByProjection(CurrentFrame = this slot, LastFrame = lastSlot, th, mono), src/Tracking.cc:2181-2202:
         * claims start empty (TrackWithMotionModel fills mvpMapPoints with NULL first) */
        MatchBuffers mb = *c->mb;
        mb.d_pairs = reinterpret_cast<MatchPair*>(L.d_in + L.trackOff);
        const float* d_Twc = reinterpret_cast<const float*>(L.d_in + L.trackOff + 128);
        if (L.mpsFromCaller) mb.d_mps = reinterpret_cast<drfe_map_point*>(L.d_in + L.trackOff + 256);
        else {
            /* LastFrame's map points as Tracking::UpdateLastFrame leaves them on an RGB-D stream: its keypoints with depth,
             * unprojected with its Twc (the launcher works on slot 0 of shifted bases; it writes d_mps[0..]) */
            SlotShift last(c, L.lastSlot);
            e = drfe_launch_mappoints_last(c, mb, L.cam, d_Twc, 1, s);
        }
        if (e == hipSuccess) e = drfe_launch_fill_i32_and_word(c->d_match + (size_t)slot * K, (int)K, -1, c->d_matchCount + slot, 0, s);
        if (e == hipSuccess) e = drfe_launch_window_match(c, mb, L.cam, 1, c->maxKp, 0, L.th, 0.f, L.checkOri, nullptr, s, 3);
    }
    /* results: gathered on the device into the h_out layout (status | count | kps | desc | uRight | depth | matches | match
     * count | matcher status), then ONE download - nine copy nodes cost the graph ~40 us of its 0.26 ms */
    DrfePackArgs A;
    std::memset(&A, 0, sizeof(A));
    auto seg = [&A](const void* src, size_t bytes) { A.src[A.n] = static_cast<const uint32_t*>(src); A.dwords[A.n] = (uint32_t)(bytes / 4); A.n++; };
    seg(c->d_status, 4);
    {
        SlotShift shift(c, slot);                    /* the slot's own arrays */
        seg(c->d_kpCount, 4);
        seg(c->d_kps, K * sizeof(drfe_keypoint));
        seg(c->d_desc, K * 32);
        if (L.withDepth || L.tracked) { seg(L.withDepth ? c->d_uRight : nullptr, K * 4); seg(L.withDepth ? c->d_depth : nullptr, K * 4); }
    }
    if (L.tracked) { seg(c->d_match + (size_t)slot * K, K * 4); seg(c->d_matchCount + slot, 4); seg(c->d_status + 3, 4); }
    size_t outBytes = 0;
    for (int k = 0; k < A.n; k++) outBytes += (size_t)A.dwords[k] * 4;
    if (e == hipSuccess) e = drfe_launch_pack_segments(A, reinterpret_cast<uint32_t*>(L.d_out), s);
    if (e == hipSuccess) e = hipMemcpyAsync(L.h_out, L.d_out, outBytes, hipMemcpyDeviceToHost, s);
    return e;
}

extern "C" {

struct TrackArgs {
    int lastSlot; const float* TcwCur; const float* TcwLast; const float* TwcLast; const drfe_map_point* lastMp; int nLast;
    float th; int mono, checkOri;
};

static int frame_submit_impl(drfe_ctx* c, int slot, const uint8_t* gray, int w, int h, size_t stride, const uint16_t* depth,
                             size_t depth_stride_elems, const drfe_camera* cam, const TrackArgs* tr)
{
    if (!c) return DRFE_ERR_INVALID;
    if (slot < 0 || slot >= c->cfg.max_batch || !gray || w < 1 || h < 1 || stride < (size_t)w || (size_t)w * h > c->stageBytes ||
        (depth && (!cam || depth_stride_elems < (size_t)w))) {
        c->err = "drfe_frame_submit: invalid argument";
        return DRFE_ERR_INVALID;
    }
    if (depth && (!(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y))) { c->err = "drfe_frame_submit: empty image bounds"; return DRFE_ERR_INVALID; }
    if (tr) {
        if (!depth || !tr->TcwCur || !tr->TcwLast || (!tr->lastMp && !tr->TwcLast) || tr->lastSlot < 0 || tr->lastSlot >= c->cfg.max_batch ||
            tr->lastSlot == slot || (tr->lastMp && (tr->nLast < 0 || tr->nLast > c->maxKp))) {
            c->err = "drfe_frame_submit_tracked: invalid argument (needs a depth image, both poses, another slot as LastFrame)";
            return DRFE_ERR_INVALID;
        }
        if (tr->lastSlot >= c->lastBatch || !c->glueValid) { c->err = "drfe_frame_submit_tracked: LastFrame's slot holds no frame with its grid"; return DRFE_ERR_STATE; }
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->frameLanes) {
        auto* v = new (std::nothrow) std::vector<FrameLane>((size_t)c->cfg.max_batch);
        if (!v) return DRFE_ERR_INVALID;
        c->frameLanes = v;
    }
    FrameLane& L = (*static_cast<std::vector<FrameLane>*>(c->frameLanes))[(size_t)slot];
    if (L.pending) { c->err = "drfe_frame_submit: the slot's previous submission has not been collected"; return DRFE_ERR_STATE; }
    /* geometry tables first: an upload synchronises the device and invalidates every slot */
    int rc = upload_geometry(c, w, h);
    if (rc != DRFE_OK) return rc;
    if (tr && !drfe_match_buffers(c)) return DRFE_ERR_HIP;
    const size_t K = (size_t)c->maxKp, px = (size_t)w * h;
    const size_t trackOff = (px * 3 + 255) & ~(size_t)255, need = trackOff + track_block_bytes(c);
    if (L.inBytes < need) {
        if (L.h_in) (void)hipHostFree(L.h_in);
        if (L.d_in) (void)hipFree(L.d_in);
        L.h_in = nullptr; L.d_in = nullptr; L.inBytes = 0;
        frame_lane_release_graph(L);
        HIPCHK(c, hipHostMalloc((void**)&L.h_in, need, hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void**)&L.d_in, need));
        L.inBytes = need;
    }
    if (!L.h_out) HIPCHK(c, hipHostMalloc((void**)&L.h_out, 8 + K * (sizeof(drfe_keypoint) + 32 + 8 + 4) + 8, hipHostMallocDefault));
    if (!L.d_out) HIPCHK(c, hipMalloc((void**)&L.d_out, 8 + K * (sizeof(drfe_keypoint) + 32 + 8 + 4) + 8));
    if (!L.done) HIPCHK(c, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
    for (int y = 0; y < h; y++) std::memcpy(L.h_in + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
    if (depth)
        for (int y = 0; y < h; y++) std::memcpy(L.h_in + px + (size_t)y * w * 2, depth + (size_t)y * depth_stride_elems, (size_t)w * 2);
    const int withDepth = depth ? 1 : 0, tracked = tr ? 1 : 0, fromCaller = tr && tr->lastMp ? 1 : 0;
    if (tr) {
        MatchPair P;
        std::memset(&P, 0, sizeof(P));
        P.curSlot = slot; P.lastSlot = tr->lastSlot; P.queryBase = 0; P.mpBase = 0;
        if (fromCaller) { P.mpSlot = -1; P.nQueries = tr->nLast; }
        else { P.mpSlot = tr->lastSlot; P.nQueries = 0; }
        std::memcpy(P.Tcw, tr->TcwCur, sizeof(float) * 16);
        drfe_motion_flags(tr->TcwCur, tr->TcwLast, cam->bf / cam->fx, tr->mono, &P.forward, &P.backward);
        static_assert(sizeof(MatchPair) <= 128, "tracking block layout");
        std::memcpy(L.h_in + trackOff, &P, sizeof(P));
        if (tr->TwcLast) std::memcpy(L.h_in + trackOff + 128, tr->TwcLast, 64);
        if (fromCaller && tr->nLast) std::memcpy(L.h_in + trackOff + 256, tr->lastMp, sizeof(drfe_map_point) * (size_t)tr->nLast);
    }
    /* a captured graph holds kernel arguments by value: camera, distortion model, sizes and the matcher's slot / thresholds
     * are part of its key */
    const bool sameKey = L.w == w && L.h == h && L.withDepth == withDepth && L.tracked == tracked && L.trackOff == trackOff &&
                         (!withDepth || (std::memcmp(&L.cam, cam, sizeof(drfe_camera)) == 0 && std::memcmp(&L.dist, &c->dist, sizeof(DrfeDistortion)) == 0)) &&
                         (!tracked || (L.lastSlot == tr->lastSlot && L.mpsFromCaller == fromCaller && L.th == tr->th && L.checkOri == tr->checkOri));
    if (!sameKey) frame_lane_release_graph(L);
    L.w = w; L.h = h; L.withDepth = withDepth; L.tracked = tracked; L.trackOff = trackOff;
    if (withDepth) { L.cam = *cam; L.dist = c->dist; }
    if (tracked) { L.lastSlot = tr->lastSlot; L.mpsFromCaller = fromCaller; L.th = tr->th; L.checkOri = tr->checkOri; }
    hipStream_t s = c->stream;
    static const bool noGraph = [] { const char* e = std::getenv("DRFE_NO_GRAPH"); return e && e[0] == '1'; }();
    if (!L.exec && !L.graphOff && !noGraph && !c->profile) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const hipError_t e = frame_enqueue(c, L, slot, w, h, s);
            hipGraph_t g = nullptr;
            const hipError_t ee = hipStreamEndCapture(s, &g);
            if (e == hipSuccess && ee == hipSuccess && g && hipGraphInstantiate(&L.exec, g, nullptr, nullptr, 0) == hipSuccess) L.graph = g;
            else { if (g) (void)hipGraphDestroy(g); L.exec = nullptr; L.graphOff = true; (void)hipGetLastError(); }
        } else { L.graphOff = true; (void)hipGetLastError(); }
    }
    if (L.exec && !c->profile) HIPCHK(c, hipGraphLaunch(L.exec, s));
    else HIPCHK(c, frame_enqueue(c, L, slot, w, h, s));
    HIPCHK(c, hipEventRecord(L.done, s));
    L.pending = true;
    c->lastBatch = std::max(c->lastBatch, slot + 1);
    if (withDepth) { c->glueValid = true; c->cam = *cam; }
    return DRFE_OK;
}

int drfe_frame_submit(drfe_ctx* c, int slot, const uint8_t* gray, int w, int h, size_t stride, const uint16_t* depth,
                      size_t depth_stride_elems, const drfe_camera* cam)
{
    return frame_submit_impl(c, slot, gray, w, h, stride, depth, depth_stride_elems, cam, nullptr);
}

/* drfe_frame_submit followed, in the same captured graph, by ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono)
 * of TrackWithMotionModel (src/Tracking.cc:2181-2202): ONE submission per tracked frame.  LastFrame lives in last_slot (an
 * earlier submission).  last_mp != NULL: LastFrame.mvpMapPoints as the caller's map holds them (n_last = its keypoint count);
 * NULL: its keypoints with depth unprojected with Twc_last - what Tracking::UpdateLastFrame leaves on an RGB-D stream.
 * Tcw_cur = the predicted pose (mVelocity * LastFrame.mTcw). */
int drfe_frame_submit_tracked(drfe_ctx* c, int slot, const uint8_t* gray, int w, int h, size_t stride, const uint16_t* depth,
                              size_t depth_stride_elems, const drfe_camera* cam, int last_slot, const float* Tcw_cur,
                              const float* Tcw_last, const float* Twc_last, const drfe_map_point* last_mp, int n_last, float th,
                              int mono, int check_ori)
{
    const TrackArgs tr = {last_slot, Tcw_cur, Tcw_last, Twc_last, last_mp, n_last, th, mono, check_ori};
    return frame_submit_impl(c, slot, gray, w, h, stride, depth, depth_stride_elems, cam, &tr);
}

int drfe_frame_collect(drfe_ctx* c, int slot, drfe_keypoint* kps, uint8_t* desc, float* u_right, float* depth_m, int cap, int* n_out)
{
    if (!c || !n_out) return DRFE_ERR_INVALID;
    *n_out = 0;
    auto* v = static_cast<std::vector<FrameLane>*>(c->frameLanes);
    if (slot < 0 || slot >= c->cfg.max_batch || !v || !(*v)[(size_t)slot].pending) {
        c->err = "drfe_frame_collect: nothing submitted to this slot";
        return DRFE_ERR_STATE;
    }
    FrameLane& L = (*v)[(size_t)slot];
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(L.done));
    L.pending = false;
    const size_t K = (size_t)c->maxKp;
    int st = 0, n = 0;
    std::memcpy(&st, L.h_out, 4);
    std::memcpy(&n, L.h_out + 4, 4);
    if (st & 1) { c->err = "FAST candidate arena overflow"; return DRFE_ERR_CAPACITY; }
    if (st & 2) { c->err = "quadtree node pool overflow"; return DRFE_ERR_CAPACITY; }
    *n_out = n;
    if (n > cap) { c->err = "keypoint buffer too small"; return DRFE_ERR_CAPACITY; }
    if ((u_right || depth_m) && !L.withDepth) { c->err = "drfe_frame_collect: the frame was submitted without a depth image"; return DRFE_ERR_STATE; }
    const uint8_t* o = L.h_out + 8;
    if (n > 0 && kps) std::memcpy(kps, o, sizeof(drfe_keypoint) * (size_t)n);
    o += K * sizeof(drfe_keypoint);
    if (n > 0 && desc) std::memcpy(desc, o, (size_t)n * 32);
    o += K * 32;
    if (n > 0 && u_right) std::memcpy(u_right, o, (size_t)n * 4);
    if (n > 0 && depth_m) std::memcpy(depth_m, o + K * 4, (size_t)n * 4);
    return DRFE_OK;
}

/* drfe_frame_collect of a drfe_frame_submit_tracked submission, plus the matcher's result: cur_to_last[i] = index of the
 * LastFrame keypoint / map point matched to current keypoint i, or -1 (CurrentFrame.mvpMapPoints), *n_matches = nmatches. */
int drfe_frame_collect_tracked(drfe_ctx* c, int slot, drfe_keypoint* kps, uint8_t* desc, float* u_right, float* depth_m, int cap,
                               int* n_out, int32_t* cur_to_last, int* n_matches)
{
    if (!c || !n_out || !n_matches) return DRFE_ERR_INVALID;
    auto* v = static_cast<std::vector<FrameLane>*>(c->frameLanes);
    if (slot < 0 || slot >= c->cfg.max_batch || !v || !(*v)[(size_t)slot].pending || !(*v)[(size_t)slot].tracked) {
        c->err = "drfe_frame_collect_tracked: no tracked submission in this slot";
        return DRFE_ERR_STATE;
    }
    const int rc = drfe_frame_collect(c, slot, kps, desc, u_right, depth_m, cap, n_out);
    if (rc != DRFE_OK) return rc;
    const FrameLane& L = (*v)[(size_t)slot];
    const size_t K = (size_t)c->maxKp;
    const uint8_t* o = L.h_out + 8 + K * (sizeof(drfe_keypoint) + 32 + 8);
    int st = 0;
    std::memcpy(n_matches, o + K * 4, 4);
    std::memcpy(&st, o + K * 4 + 4, 4);
    if (st & 4) { c->err = "match candidate list overflow (DRFE_MATCH_MAX_CAND)"; return DRFE_ERR_CAPACITY; }
    if (cur_to_last && *n_out > 0) std::memcpy(cur_to_last, o, (size_t)*n_out * 4);
    return DRFE_OK;
}

} /* extern "C" */
