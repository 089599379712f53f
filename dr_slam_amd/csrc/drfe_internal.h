/* drfe_internal.h — host-side context and the HBM layout of a batch.
 *
 * One context owns, for `max_batch` frame slots, frame-major arenas:
 *   pyr    bordered pyramid, all levels of a slot contiguous (level l at lv[l].pyrOff, row pitch
 *          lv[l].pyrPitch = align64(w_l + 38))
 *   blur   blurred interior levels (pitch align64(w_l))
 *   cand   FAST candidates per (slot, level): key0 = x | y<<12 | response<<24, key1 = emission-order key
 *   node   quadtree scratch: node id per candidate (u16)
 *   sel    keypoints chosen by the quadtree per (slot, level), list order
 *   kps / desc / counts   final outputs, level-major per slot
 *   glue   depth, uRight, 64x48 grid CSR per slot; match results per slot
 */
#ifndef DRFE_INTERNAL_H
#define DRFE_INTERNAL_H

#include <hip/hip_runtime.h>
#ifndef DRFE_NO_ROCTX
#include <roctracer/roctx.h>
#else                                   /* a build host without the roctracer headers: make DEFS=-DDRFE_NO_ROCTX ROCTX_LIB= ; the stage ranges become no-ops */
static inline int roctxRangePushA(const char*) { return 0; }
static inline int roctxRangePop() { return 0; }
#endif
#include <stdint.h>
#include <time.h>
#include <string>
#include <vector>

#include "../../include/drfe.h"
#include "../../include/drfe_debug.h"

#define DRFE_MAX_LEVELS 16
#define DRFE_EDGE 19          /* EDGE_THRESHOLD, reference src/ORBextractor.cc:72 */
#define DRFE_HALF_PATCH 15    /* HALF_PATCH_SIZE, :71 */
#define DRFE_GRID_COLS 64     /* FRAME_GRID_COLS, reference include/Frame.h:40 */
#define DRFE_GRID_ROWS 48     /* FRAME_GRID_ROWS, :39 */
#define DRFE_GRID_CELLS (DRFE_GRID_COLS * DRFE_GRID_ROWS)
#define DRFE_FASTC_MAX_EW 38       /* k_fast_cells_cols: widest evaluated area (window + alignment <= 48 pixels of a 52-pixel LDS row) */
#define DRFE_FASTC_MAX_RPL 12      /* ... and most rows per lane (its score registers) */
#define DRFE_FAST_MAX_WIN 60       /* largest FAST cell window (wCell+6): fits 64-byte LDS rows at any alignment */
#define DRFE_QT_MAX_NODES 1024    /* quadtree list capacity per level (>= quota + 4) */
#define DRFE_MATCH_MAX_CAND 256   /* candidates kept per query by the window gather */

/* per-level constants shared by host and kernels (passed by value inside DevGeom) */
struct DevLevel {
    int w, h;                 /* interior */
    int pyrPitch, pyrOff;     /* bordered image: pitch, byte offset inside the slot's pyramid block */
    int blurPitch, blurOff;
    int quota;
    int minBX, minBY, maxBX, maxBY;
    int nCols, nRows, wCell, hCell;
    int candCap, candOff;     /* element offset inside the slot's candidate arrays */
    int kpCap, kpOff;         /* element offset inside the slot's sel/kps arrays */
    int nIni;
    float hX;
    float scale;              /* mvScaleFactor[l] */
    float kpSize;             /* (float)(int)(31*scale) */
    int xtabOff, ytabOff;     /* resize coefficient tables (level >= 1), element offsets */
    int xwinOff, ywinOff;     /* per 256-column / 16-row block of the bordered level: (lowest, highest) source index its
                                 taps touch, stored as ResizeTap{s0 = lo, s1 = hi}; resizeLds = 0 if a window exceeds the
                                 LDS tile of k_pyr_resize_lds */
    int resizeLds;
    int cellBegin, cellEnd;   /* range in the FAST cell table */
    int tileBegin, tileEnd;   /* range in the blur tile table */
};

struct DevGeom {
    int nlevels;
    int iniTh, minTh;
    int imgW, imgH;
    int pyrSlotBytes, blurSlotBytes;
    int candSlotElems, kpSlotElems; /* per-slot element counts */
    int totalCells, totalTiles;
    int fastMaxWh;                  /* tallest FAST cell window of this geometry: sizes the kernel's dynamic LDS */
    int fastCols;                   /* 1: every cell fits k_fast_cells_cols (ew <= 38, rows per lane <= 12) */
    int fastColsSmall;              /* cells [0, fastColsSmall) of the table have rpl <= 8 */
    int fastColsRows;               /* LDS tile rows k_fast_cells_cols touches: max over cells of nrb * rpl + 6 */
    DevLevel lv[DRFE_MAX_LEVELS];
};

/* FAST's per-(slot, level) candidate counters are bumped by one returning atomic per cell, and every wavefront waits for its
 * answer: packed as [slot][level] ints, four slots share a 128-byte line and the kernel's time depended on where the 16 KB array
 * happened to lie (0.59 / 0.66 / 0.69 / 0.71 ms from context to context, tools/fast_mode_probe2.py).  One counter per line. */
#ifndef DRFE_CC_LINE
#define DRFE_CC_LINE 32           /* ints between two counters */
#endif
#define DRFE_CC_SLOT (DRFE_MAX_LEVELS * DRFE_CC_LINE)
#define DRFE_CC_IDX(slot, level) ((size_t)(slot) * DRFE_CC_SLOT + (size_t)(level) * DRFE_CC_LINE)

struct FastCell {   /* one cv::FAST call of reference src/ORBextractor.cc:789-816 */
    uint16_t x0, y0;        /* window origin in interior coordinates */
    uint8_t ww, wh;         /* window size (<= 68) */
    uint8_t level, off;     /* off = (x0 + EDGE) & 3: window start inside its first aligned dword */
    uint16_t offX, offY;    /* j*wCell, i*hCell added to the keypoint (:822-823) */
    uint32_t cellIdx;       /* i*nCols + j: emission order of the cell */
    /* copies of the level's layout so that the kernel's loads depend on this record only */
    uint32_t srcOff;        /* byte offset of the window's first aligned dword inside the slot's pyramid block */
    uint32_t pitch;         /* bordered row pitch of the level */
    uint32_t candOff, candCap;
    /* column-pair layout of k_fast_cells_cols: lane = (column pair cp, row block rb); ncp = ceil(ew/2) column pairs,
     * nrb = 64 / ncp row blocks of rpl = ceil(eh/nrb) rows each; ncpMagic = ceil(2^16/ncp): lane / ncp as a multiply */
    uint8_t ncp, nrb, rpl, pad;
    uint32_t ncpMagic;
};

struct BlurTile { uint16_t tx, ty; uint16_t level, pad; };

struct ResizeTap { uint16_t s0, s1; int16_t w0, w1; };

#include "undistort_math.h"

struct drfe_ctx {
    drfe_config cfg;
    int device;
    hipStream_t stream;       /* context-owned stream (used when caller passes NULL) */
    std::string err;

    /* host tables */
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    int umax[DRFE_HALF_PATCH + 1];
    DevGeom geom;             /* current geometry (imgW/imgH == 0 until first frame) */
    int maxKp;                /* per-slot keypoint capacity (sum of level caps) for max geometry */

    /* device tables */
    DevGeom* d_geom;
    FastCell* d_cells; int cellsCap;
    BlurTile* d_tiles; int tilesCap;
    ResizeTap* d_taps; int tapsCap;
    int8_t* d_pattern;        /* 1024 */
    int16_t* d_disc;          /* 749 x (u,v) */
    int discCount;

    /* arenas (sized for max geometry x max_batch) */
    uint8_t* d_pyr; size_t pyrSlotBytesMax;
    uint8_t* d_blur; size_t blurSlotBytesMax;
    uint32_t* d_cand0; uint32_t* d_cand1; uint16_t* d_node; size_t candSlotElemsMax;
    int* d_candCount;         /* [slot][level], one counter per DRFE_CC_LINE ints: DRFE_CC_IDX */
    uint32_t* d_sel;          /* [slot][kpSlotElems] packed x|y<<12|resp<<24 */
    int* d_selCount;          /* [slot][level] */
    drfe_keypoint* d_kps;     /* [slot][maxKp] mvKeys */
    drfe_keypoint* d_kpsUn;   /* [slot][maxKp] mvKeysUn; allocated only when a distortion model is set */
    DrfeDistortion dist;      /* Frame::UndistortKeyPoints model (enabled == 0: mvKeysUn = mvKeys) */
    uint8_t* d_desc;          /* [slot][maxKp][32] */
    int* d_kpCount;           /* [slot] */
    int* d_status;            /* device-side error flags (overflow) */
    int fastScreen;           /* k_fast_cells_cols' screened paths; 2 (default): at iniThFAST first (textured cells), then at minThFAST (low-texture cells); 1: at
                               * minThFAST only; 0: never.  DRFE_FAST_SCREEN in the environment sets it (A/B, tests) */
    int fastGeneric;          /* DRFE_FAST_GENERIC=1 in the environment: run k_fast_cells even where k_fast_cells_cols fits (A/B, tests) */

    /* frame glue + match */
    float* d_uRight; float* d_depth;      /* [slot][maxKp] */
    int* d_gridOff; int* d_gridIdx;       /* [slot][3073], [slot][maxKp] */
    uint4* d_cellKp; uint4* d_cellDesc;   /* keypoints in grid-cell order: [slot][maxKp] records, [slot][maxKp][2] descriptors */
    int* d_match; int* d_matchCount;      /* [slot][maxKp], [slot] */
    float* d_poses;                       /* per-batch Tcw/Twc staging: [2][max_batch][16] */
    uint8_t* d_stage;                     /* staging for single-frame host API */
    size_t stageBytes;
    uint32_t* d_kpUV;                     /* [max_batch][maxKp] depth pixel of every keypoint (drfe_orb_keypoint_pixels_async) */
    uint16_t* d_kpDepth;                  /* [max_batch][maxKp] raw depth per keypoint (drfe_frame_stereo_grid_batch_kpdepth) */
    uint8_t* d_callScratch;               /* grow-only device scratch of the host-buffer matcher calls (one call at a time) */
    size_t callScratchBytes;

    int lastBatch;            /* frames in the most recent batch */
    bool glueValid;
    drfe_camera cam;          /* camera of the most recent glue call */

    struct MatchBuffers* mb;  /* lazily allocated matcher scratch (match_internal.h) */
    struct PlanesScratch* ps; /* lazily allocated plane-path scratch (planes_internal.h) */
    void* planeLanes;         /* std::vector<PlaneLane>*: lanes of drfe_planes_ahc_batch (planes_ahc.cpp) */
    struct BowState* bow;     /* vocabulary + BoW scratch (bow_internal.h), set by drfe_voc_upload */
    struct LinesScratch* ls;  /* line-path scratch (lines_internal.h) */
    void* lineHost;           /* LineHost*: host buffers of the single-frame line entry */
    struct OrbOneShot* oneShot; /* captured hipGraph of the single-frame ORB entry (capi.cpp) */
    void* cape;               /* CapeScratch*: device buffers of drfe_planes_cape (planes_internal.h) */
    void* capeBatch;          /* CapeBatchArena*: device path of drfe_planes_cape_batch (planes_cape.cpp) */
    int planesDeviceCape;     /* drfe_planes_configure_cape: 1 = CAPE::process on the device in the batch entry (default) */
    void* capeLanes;          /* std::vector<CapeLane>*: lanes of drfe_planes_cape_batch (planes_cape.cpp) */
    void* sn;                 /* SnBuffers*: surface-normal scratch (post_internal.h) */
    void* lineWorkers;        /* std::vector<LineWorker>*: lanes of drfe_lsd_extract_batch (lines_lsd.cpp) */
    struct LinesScratch* lsBatch; /* frame slots of drfe_lsd_extract_batch's device region growing (lines_lsd.cpp) */
    int lsdDeviceGrow;        /* drfe_lsd_configure: 1 = the batch entry grows regions on the device (default) */
    int lsdDeviceNfa;         /* drfe_lsd_configure_nfa: 1 = rect_improve's decisions on the device in the batch entry (default) */
    long long lsdStats[4];    /* drfe_lsd_stats */
    int longClock = 0;        /* drfe_long_kernel_clock: the batch entries of the line / plane paths bracket every kernel of their first chunk with events */
    float longMs[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   /* drfe_long_kernel_ms: [0..6] lines, [8..14] planes */
    long long capeStats[2];   /* drfe_planes_cape_stats: frames through the device path, of those finished by the host (kept here: the arena is rebuilt when a batch grows) */
    long long ahcStats[4];    /* drfe_planes_ahc_stats: frames through the device extractor, of those redone on the host, plane voxel grids on the device, of those redone on the host */
    int lsdRectMode;          /* drfe_lsd_configure_rect: rect_nfa's reading, 0 = literal OpenCV 3.4 (default), 1 = real-valued */
    int planesDeviceAhc;      /* drfe_planes_configure_extractor: 1 = drfe_planes_ahc_post_batch runs the extractor on the device (default) */
    void* ahcArena;           /* AhcArena*: frame slots of the device extractor (planes_ahc.cpp) */
    int planesDeviceRefit;    /* drfe_planes_configure_refit: 1 = gates + RANSAC refit on the device behind the device voxel grids (default) */
    long long ahcRefitStats[2];
    int planesDeviceVoxel;    /* drfe_planes_configure: where drfe_planes_ahc_post_batch runs the voxel grids (default 1: device, behind the device extractor) */
    void* frameLanes;         /* std::vector<FrameLane>*: per-slot staging of drfe_frame_submit / drfe_frame_collect (capi.cpp) */

    /* profiling */
    bool profile;
    hipEvent_t ev[DRFE_STAGE_COUNT][2];
    bool evUsed[DRFE_STAGE_COUNT];
};

/* The kernels address a slot as base + slot * stride.  One slot of a larger arena is therefore the same launches on
 * shifted bases: this shifts every per-slot base the ORB and glue launchers read, for the duration of the enqueue. */
struct SlotShift {
    drfe_ctx* c; int slot;
    SlotShift(drfe_ctx* c_, int slot_) : c(c_), slot(slot_) { apply(1); }
    ~SlotShift() { apply(-1); }
    void apply(int sgn)
    {
        const ptrdiff_t s = (ptrdiff_t)sgn * slot, K = c->maxKp;
        const DevGeom& g = c->geom;
        c->d_pyr += s * g.pyrSlotBytes; c->d_blur += s * g.blurSlotBytes;
        c->d_cand0 += s * g.candSlotElems; c->d_cand1 += s * g.candSlotElems; c->d_node += s * g.candSlotElems;
        c->d_candCount += s * DRFE_CC_SLOT; c->d_selCount += s * g.nlevels; c->d_sel += s * g.kpSlotElems;
        c->d_kps += s * K; if (c->d_kpsUn) c->d_kpsUn += s * K;
        c->d_desc += s * K * 32; c->d_kpCount += s;
        c->d_uRight += s * K; c->d_depth += s * K;
        c->d_gridOff += s * (DRFE_GRID_CELLS + 1); c->d_gridIdx += s * K;
        c->d_cellKp += s * K; c->d_cellDesc += s * K * 2;
    }
};

/* what the matchers and the grid read: mvKeysUn (== mvKeys without distortion) */
static inline drfe_keypoint* drfe_kps_un(const drfe_ctx* c) { return c->dist.enabled ? c->d_kpsUn : c->d_kps; }

/* Waiting for a stream on a POOL thread: hipStreamSynchronize spins, and a pool that runs more threads than CPUs (the batch
 * entries do, to cover exactly these waits) then burns the cycles its compute threads need.  Poll an event and sleep in
 * between: the wait costs microseconds of CPU instead of its whole duration.  (hipEventBlockingSync did not help: measured
 * more CPU time, not less.)  The single-frame entries keep spinning - there latency is the point. */
static inline hipError_t drfe_pool_sync(hipStream_t s, hipEvent_t ev)
{
    hipError_t e = hipEventRecord(ev, s);
    if (e != hipSuccess) return e;
    for (int spins = 0;; spins++) {
        e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (spins < 4) continue;                             /* a copy of a few KB is done before the first sleep */
        struct timespec ts = {0, spins < 40 ? 20000 : 100000};
        nanosleep(&ts, nullptr);
    }
}

/* 1 if [p, p + bytes) is pinned (hipHostMalloc / hipHostRegister) host memory: the batch entries then upload straight from the
 * caller's buffer instead of copying it into their own pinned staging first */
static inline bool drfe_host_is_pinned(const void* p, size_t bytes)
{
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (a.type != hipMemoryTypeHost) return false;
    hipPointerAttribute_t b;
    const char* last = static_cast<const char*>(p) + (bytes ? bytes - 1 : 0);
    if (hipPointerGetAttributes(&b, last) != hipSuccess) { (void)hipGetLastError(); return false; }
    return b.type == hipMemoryTypeHost;
}

/* Stage brackets of the headline path: HIP events on the launch stream when drfe_profile_enable is on (drfe_profile_stage_ms),
 * and a roctx range around the stage's launches always (a rocprofv3 --marker-trace / --kernel-trace run shows
 * "drfe:pyramid", "drfe:fast" ... around the kernels instead of bare kernel names; without a tool attached roctx is a no-op). */
static inline const char* drfe_stage_name(int stage)
{
    static const char* const names[DRFE_STAGE_COUNT] = {"drfe:pyramid", "drfe:fast", "drfe:quadtree", "drfe:blur", "drfe:orient+desc", "drfe:glue", "drfe:match", "drfe:fast_b"};
    return stage >= 0 && stage < DRFE_STAGE_COUNT ? names[stage] : "drfe:?";
}
static inline void prof_begin(drfe_ctx* c, int stage, hipStream_t s)
{
    (void)roctxRangePushA(drfe_stage_name(stage));
    if (c->profile) { (void)hipEventRecord(c->ev[stage][0], s); c->evUsed[stage] = true; }
}
static inline void prof_end(drfe_ctx* c, int stage, hipStream_t s)
{
    if (c->profile) (void)hipEventRecord(c->ev[stage][1], s);
    (void)roctxRangePop();
}
/* a named range around a phase of the batch entries (lines / planes / CAPE) */
struct DrfeRange {
    explicit DrfeRange(const char* name) { (void)roctxRangePushA(name); }
    ~DrfeRange() { (void)roctxRangePop(); }
};

/* wait for an already recorded event the same way: poll + sleep instead of the runtime's busy wait */
static inline hipError_t drfe_event_wait_sleeping(hipEvent_t ev)
{
    for (int spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (spins < 4) continue;
        struct timespec ts = {0, spins < 40 ? 20000 : 100000};
        nanosleep(&ts, nullptr);
    }
}

/* CPU time of the batch entries' pool threads, summed per pool (0 lines, 1 AHC planes, 2 CAPE): a worker adds its thread's CPU time when
 * it ends (drfe_debug_pool_cpu_ns reads and clears).  Measurement only. */
void drfe_pool_cpu_add(int pool, long long ns);
struct DrfePoolCpuScope {
    int pool; struct timespec t0;
    explicit DrfePoolCpuScope(int p) : pool(p) { clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0); }
    ~DrfePoolCpuScope() { struct timespec t1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1); drfe_pool_cpu_add(pool, (t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec)); }
};

/* The stream a batch entry runs its long one-wavefront-per-frame kernels on: low priority (the pools' short kernels must not
 * queue behind them) or, with DRFE_CU_SPLIT=<lines share in percent>, restricted to one part of the CUs - part 0 for the line
 * path, part 1 for the plane path (capi.cpp) */
hipError_t drfe_long_kernel_stream(hipStream_t* s, int part);

/* capi.cpp: host threads a batch entry point may start by default - the affinity mask clipped by the cgroup CPU quota
 * (std::thread::hardware_concurrency() reports the machine, which oversubscribes a quota-limited container) */
int drfe_default_host_threads();
void drfe_cape_lanes_free(drfe_ctx* c);                   /* planes_cape.cpp */
void drfe_bow_slot_invalidate(drfe_ctx* c, int slot);     /* capi_bow.cpp: the slot's descriptors changed (drfe_frame_load) */
void drfe_one_shot_free(drfe_ctx* c);                    /* capi.cpp: the captured single-frame ORB graph */
void drfe_frame_lanes_free(drfe_ctx* c);                 /* capi.cpp: staging + graphs of the per-frame pipelined flow */

/* orb_geometry.cpp */
int drfe_build_tables(drfe_ctx* c);                       /* scale tables, quotas, umax */
int drfe_build_geometry(drfe_ctx* c, int w, int h, DevGeom* g, std::vector<FastCell>* cells,
                        std::vector<BlurTile>* tiles, std::vector<ResizeTap>* taps);

/* orb_kernels.hip */
hipError_t drfe_launch_orb(drfe_ctx* c, const uint8_t* d_gray, size_t frameStride, size_t rowStride, int nframes,
                           hipStream_t s);
/* match_kernels.hip */
hipError_t drfe_launch_glue(drfe_ctx* c, const uint16_t* d_depth, size_t frameStride, size_t rowStride,
                            const drfe_camera& cam, int nframes, hipStream_t s);
hipError_t drfe_launch_grid(drfe_ctx* c, const drfe_camera& cam, int nframes, hipStream_t s);
hipError_t drfe_launch_kp_pixels(drfe_ctx* c, int nframes, uint32_t* d_uv, hipStream_t s);
hipError_t drfe_launch_match_consecutive(drfe_ctx* c, const drfe_camera& cam, float th, int mono, int checkOri,
                                         int nframes, hipStream_t s);

#define DRFE_RESIZE_LDS_WD 88     /* dwords per source row of the k_pyr_resize_lds tile (256 output columns * 1.25 + slack) */
#define DRFE_RESIZE_LDS_ROWS 24   /* source rows of the tile (16 output rows * 1.25 + slack) */
/* XCD-aware block numbering for the batch-wide 2-D grids (x = item inside a frame, y = frame slot).  Workgroups are dealt
 * round-robin over the 8 XCDs in dispatch order (x fastest), so consecutive cells / tiles / keypoint groups of one frame -
 * which share pyramid lines - would land on eight different private L2s and every line would cross the fabric several
 * times.  The swizzle gives each XCD a contiguous eighth of the flattened grid, i.e. whole frames: the blocks that share
 * lines share an L2.  Bijective for any grid size (cdna_hip_programming.md, T1); placement is a speed matter only.
 * `magic` = drfe_div_magic(gridDim.x) from the host: swz / gx as one multiply-high, which is the quotient or the quotient
 * plus one for any 32-bit swz (M * gx = 2^32 + e, e < gx), hence the one-step correction in drfe_div_by. */
static inline uint32_t drfe_div_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); }   /* 0 = divide by one */
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t drfe_div_by(uint32_t n, uint32_t d, uint32_t magic)
{
    if (!magic) return n;                         /* d == 1 */
    uint32_t q = __umulhi(n, magic);
    if (q * d > n) q--;
    return q;
}
/* Inclusive prefix sum over the wavefront without a trip through the LDS crossbar: four DPP row shifts inside each row of sixteen
 * lanes (0 shifted in), the three row totals through v_readlane.  ~12 VALU instructions against six dependent ds_bpermute round
 * trips of the __shfl_up form (round 6; k_quadtree's three block scans per round sit on its barrier-to-barrier path).  Every
 * lane of the wavefront must be active. */
__device__ __forceinline__ int drfe_wave_incl_scan(int v, int lane)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);      /* row_shr:1 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);      /* row_shr:2 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);      /* row_shr:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);      /* row_shr:8 */
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}

/* Sum / minimum over the wavefront, to every lane (wave-uniform): two quad permutes and two row rotations as DPP operands leave every
 * lane of a row with the row's result; the four rows meet through v_readlane.  Integers only (the order of the additions differs from
 * the butterfly's).  Every lane must be active. */
__device__ __forceinline__ int drfe_wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);       /* quad_perm [1, 0, 3, 2] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);       /* quad_perm [2, 3, 0, 1] */
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);      /* row_ror:4 */
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);      /* row_ror:8 */
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ uint32_t drfe_wave_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

__device__ __forceinline__ void drfe_xcd_swizzle_2d(uint32_t magic, int& lx, int& ly)
{
    const uint32_t gx = gridDim.x, nwg = gx * gridDim.y;
    const uint32_t orig = blockIdx.y * gx + blockIdx.x;
    const uint32_t q = nwg >> 3, r = nwg & 7, xcd = orig & 7, j = orig >> 3;
    const uint32_t swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const uint32_t y = drfe_div_by(swz, gx, magic);
    lx = (int)(swz - y * gx);
    ly = (int)y;
}
/* the same for 3-D grids (x, y = tile, z = frame slot): magicXY = drfe_div_magic(gridDim.x * gridDim.y), magicX = drfe_div_magic(gridDim.x) */
__device__ __forceinline__ void drfe_xcd_swizzle_3d(uint32_t magicXY, uint32_t magicX, int& lx, int& ly, int& lz)
{
    const uint32_t gx = gridDim.x, gxy = gx * gridDim.y, nwg = gxy * gridDim.z;
    const uint32_t orig = blockIdx.z * gxy + blockIdx.y * gx + blockIdx.x;
    const uint32_t q = nwg >> 3, r = nwg & 7, xcd = orig & 7, j = orig >> 3;
    const uint32_t swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const uint32_t z = drfe_div_by(swz, gxy, magicXY), rem = swz - z * gxy;
    const uint32_t y = drfe_div_by(rem, gx, magicX);
    lx = (int)(rem - y * gx); ly = (int)y; lz = (int)z;
}
#endif

/* layout of a blurred level in HBM: tiles of DRFE_BTILE_W x DRFE_BTILE_H pixels = one 128-byte line each, row-major inside
 * the tile, tiles row-major over the level; DevLevel::blurPitch = tiles per tile row */
#ifndef DRFE_BTILE_W
#define DRFE_BTILE_W 32
#define DRFE_BTILE_H 4
#endif
#if defined(__HIPCC__) || defined(__cplusplus)
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
unsigned drfe_blur_offset(int x, int y, int tilesX)
{
    const unsigned ux = (unsigned)x, uy = (unsigned)y;      /* non-negative by construction: powers of two become shifts */
    return ((uy / DRFE_BTILE_H) * (unsigned)tilesX + ux / DRFE_BTILE_W) * 128u + (uy % DRFE_BTILE_H) * DRFE_BTILE_W + ux % DRFE_BTILE_W;
}
#endif
#ifndef DRFE_BLUR_TW
#define DRFE_BLUR_TW 128      /* blur tile: 128 x 64 output pixels (full 128-byte lines per stored row, 9 % halo rows) */
#define DRFE_BLUR_TH 64
#endif

#endif
