/* capi_match.cpp — C-ABI entry points of the Frame glue and the matchers (include/drfe.h). */
#include "drfe_internal.h"
#include "match_internal.h"
#include "../../include/drfe_math.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

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

void drfe_match_buffers_free(drfe_ctx* c)
{
    MatchBuffers* m = c->mb;
    if (!m) return;
    void* ptrs[] = {m->d_pairs, m->d_queries, m->d_mps, m->d_scale, m->d_candIdx, m->d_candKey, m->d_candCnt, m->d_candBest,
                    m->d_hist, m->d_initObs, m->d_bfIdx, m->d_bfDist, m->d_bfQ, m->d_bfT};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (int k = 0; k < 2; k++) {
        if (m->h_stage[k]) (void)hipHostFree(m->h_stage[k]);
        if (m->stageEv[k]) (void)hipEventDestroy(m->stageEv[k]);
    }
    delete m;
    c->mb = nullptr;
}

MatchBuffers* drfe_match_buffers(drfe_ctx* c)
{
    if (c->mb) return c->mb;
    MatchBuffers* m = new (std::nothrow) MatchBuffers();
    if (!m) return nullptr;
    std::memset(m, 0, sizeof(*m));
    c->mb = m;
    const size_t B = (size_t)c->cfg.max_batch, Q = B * (size_t)c->maxKp;
    m->queryCap = Q;
    m->bfCap = std::max<size_t>(Q, 4096);
    bool ok = dalloc(&m->d_pairs, B) == hipSuccess && dalloc(&m->d_queries, Q) == hipSuccess &&
              dalloc(&m->d_mps, Q) == hipSuccess && dalloc(&m->d_scale, DRFE_MAX_LEVELS) == hipSuccess &&
              dalloc(&m->d_candIdx, Q * DRFE_MATCH_MAX_CAND) == hipSuccess &&
              dalloc(&m->d_candKey, Q * DRFE_MATCH_MAX_CAND) == hipSuccess && dalloc(&m->d_candCnt, Q) == hipSuccess && dalloc(&m->d_candBest, Q) == hipSuccess &&
              dalloc(&m->d_hist, B * 2 * (size_t)c->maxKp) == hipSuccess &&
              dalloc(&m->d_initObs, (size_t)c->maxKp) == hipSuccess && dalloc(&m->d_bfIdx, m->bfCap * 2) == hipSuccess &&
              dalloc(&m->d_bfDist, m->bfCap * 2) == hipSuccess && dalloc(&m->d_bfQ, m->bfCap * 32) == hipSuccess &&
              dalloc(&m->d_bfT, m->bfCap * 32) == hipSuccess;
    if (ok)
        ok = hipMemcpy(m->d_scale, c->scale.data(), sizeof(float) * c->cfg.nlevels, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        c->err = "matcher scratch allocation failed";
        drfe_match_buffers_free(c);
        return nullptr;
    }
    return m;
}

/* bForward / bBackward of reference src/ORBmatcher.cc:1406-1414.  `-Rcw.t()*tcw` takes OpenCV's general
 * gemm path (double accumulation, alpha = -1); `Rlw*twc+tlw` the float small-matrix path. */
void drfe_motion_flags(const float* TcwCur, const float* TcwLast, float mb, int mono, int* fwd, int* bwd)
{
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)TcwCur[k * 4 + i] * (double)TcwCur[k * 4 + 3];
        twc[i] = (float)(s * -1.0);
    }
    for (int r = 0; r < 3; r++) {
        const float d = TcwLast[r * 4 + 0] * twc[0] + TcwLast[r * 4 + 1] * twc[1] + TcwLast[r * 4 + 2] * twc[2];
        tlc[r] = d + TcwLast[r * 4 + 3];
    }
    *fwd = (tlc[2] > mb && !mono) ? 1 : 0;
    *bwd = (-tlc[2] > mb && !mono) ? 1 : 0;
}

static int match_status(drfe_ctx* c, int word = 1)
{
    int st = 0;
    HIPCHK(c, hipMemcpy(&st, c->d_status + word, sizeof(int), hipMemcpyDeviceToHost));
    if (st & 4) { c->err = "match candidate list overflow (DRFE_MATCH_MAX_CAND)"; return DRFE_ERR_CAPACITY; }
    return DRFE_OK;
}

static void frustum_pose(drfe_ctx* c, const float* Tcw, const drfe_camera* cam, float limit, FrustumPose* P)
{
    std::memcpy(P->T, Tcw, 64);
    for (int i = 0; i < 3; i++) {   /* mOw = -mRcw.t()*mtcw: general gemm path, double accumulation (src/Frame.cc:592-600) */
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Tcw[k * 4 + i] * (double)Tcw[k * 4 + 3];
        P->Ow[i] = (float)(s * -1.0);
    }
    P->bf = cam->bf;
    P->logScale = drfe_logf(c->cfg.scale_factor);     /* Frame::mfLogScaleFactor = log(mfScaleFactor) */
    P->nLevels = c->cfg.nlevels;
    P->limit = limit;
}

/* Device scratch of a host-buffer matcher call: one grow-only block per context instead of hipMalloc + hipFree around
 * every call (both synchronise the device and cost tens of microseconds, most of such a call's latency).  Calls on a
 * context are sequential, and each one ends with a stream synchronisation before it returns. */
static hipError_t call_scratch(drfe_ctx* c, size_t bytes, uint8_t** out)
{
    if (bytes > c->callScratchBytes) {
        if (c->d_callScratch) (void)hipFree(c->d_callScratch);
        c->d_callScratch = nullptr; c->callScratchBytes = 0;
        const size_t want = (bytes + (bytes >> 1) + 65535) & ~(size_t)65535;
        const hipError_t e = hipMalloc((void**)&c->d_callScratch, want);
        if (e != hipSuccess) return e;
        c->callScratchBytes = want;
    }
    *out = c->d_callScratch;
    return hipSuccess;
}

/* KeyFrame::GetCameraCenter(): Ow = -Rwc*tcw with Rwc = Rcw.t() stored first (KeyFrame::SetPose, src/KeyFrame.cc:153-154), so
 * the product has no transpose flag and cv::gemm takes its small-matrix float path: float dot, then * alpha = -1 */
static void camera_centre_kf(const float* Tcw, float Ow[3])
{
    for (int i = 0; i < 3; i++) {
        const float d = Tcw[0 * 4 + i] * Tcw[3] + Tcw[1 * 4 + i] * Tcw[7] + Tcw[2 * 4 + i] * Tcw[11];
        Ow[i] = (float)((double)d * -1.0);
    }
}

template <class In, class Out, class Launch>
static int frustum_run(drfe_ctx* c, const In* in, int n, Out* out, Launch launch)
{
    if (n == 0) return DRFE_OK;
    HIPCHK(c, hipSetDevice(c->device));
    uint8_t* d = nullptr;
    const size_t oIn = 0, oOut = (sizeof(In) * (size_t)n + 63) & ~(size_t)63, total = oOut + sizeof(Out) * (size_t)n;
    HIPCHK(c, call_scratch(c, total, &d));
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(d + oIn, in, sizeof(In) * (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oOut, out, sizeof(Out) * (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch(reinterpret_cast<const In*>(d + oIn), reinterpret_cast<Out*>(d + oOut), s);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d + oOut, sizeof(Out) * (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { c->err = std::string("is_in_frustum: ") + hipGetErrorString(e); return DRFE_ERR_HIP; }
    return DRFE_OK;
}


extern "C" {

int drfe_frame_stereo_grid_batch(drfe_ctx* c, const uint16_t* d_depth, size_t frame_stride, size_t row_stride,
                                 const drfe_camera* cam, int nframes, void* stream)
{
    if (!c || !d_depth || !cam) return DRFE_ERR_INVALID;
    if (nframes < 1 || nframes > c->lastBatch) { c->err = "glue: extract the batch first"; return DRFE_ERR_STATE; }
    if (!(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y)) { c->err = "glue: empty image bounds"; return DRFE_ERR_INVALID; }
    /* a depth IMAGE: rows at least a frame wide (row stride 0 is the internal per-keypoint addressing of
     * drfe_frame_stereo_grid_batch_kpdepth and must not be reachable from here) */
    if (row_stride < (size_t)c->geom.imgW || frame_stride < row_stride * (size_t)c->geom.imgH) { c->err = "glue: depth row / frame stride smaller than the image"; return DRFE_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, drfe_launch_glue(c, d_depth, frame_stride, row_stride, *cam, nframes, s));
    c->glueValid = true;
    c->cam = *cam;
    return DRFE_OK;
}

/* Frame::UndistortKeyPoints model: (k1, k2, p1, p2[, k3]) with mK = (fx, fy, cx, cy); k1 == 0 switches it off
 * exactly as src/Frame.cc:836 does */
int drfe_frame_set_distortion(drfe_ctx* c, const drfe_camera* cam, const float* dist, int n)
{
    if (!c || (n > 0 && (!dist || !cam)) || n < 0 || n > 5) return DRFE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    std::memset(&c->dist, 0, sizeof(c->dist));
    c->glueValid = false;
    if (n == 0 || dist[0] == 0.0f) return DRFE_OK;
    if (n < 4) { c->err = "set_distortion: need k1, k2, p1, p2[, k3]"; return DRFE_ERR_INVALID; }
    if (!c->d_kpsUn) {
        void* p = nullptr;
        HIPCHK(c, hipMalloc(&p, sizeof(drfe_keypoint) * (size_t)c->cfg.max_batch * c->maxKp));
        c->d_kpsUn = (drfe_keypoint*)p;
    }
    c->dist.fx = cam->fx; c->dist.fy = cam->fy; c->dist.cx = cam->cx; c->dist.cy = cam->cy;
    for (int i = 0; i < n; i++) c->dist.k[i] = (double)dist[i];
    c->dist.enabled = 1;
    return DRFE_OK;
}

/* Frame::ComputeImageBounds, src/Frame.cc:862-891 (four corners through the same undistortPoints) */
int drfe_frame_image_bounds(const drfe_camera* cam, const float* dist, int n, int cols, int rows, float* bounds)
{
    if (!cam || !bounds || n < 0 || n > 5 || (n > 0 && !dist)) return DRFE_ERR_INVALID;
    if (n >= 4 && dist[0] != 0.0f) {
        DrfeDistortion D;
        std::memset(&D, 0, sizeof(D));
        D.fx = cam->fx; D.fy = cam->fy; D.cx = cam->cx; D.cy = cam->cy; D.enabled = 1;
        for (int i = 0; i < n; i++) D.k[i] = (double)dist[i];
        const float cx[4] = {0.f, (float)cols, 0.f, (float)cols}, cy[4] = {0.f, 0.f, (float)rows, (float)rows};
        float ux[4], uy[4];
        for (int i = 0; i < 4; i++) drfe_undistort_point(D, cx[i], cy[i], &ux[i], &uy[i]);
        bounds[0] = std::min(ux[0], ux[2]); bounds[1] = std::max(ux[1], ux[3]);
        bounds[2] = std::min(uy[0], uy[1]); bounds[3] = std::max(uy[2], uy[3]);
    } else {
        bounds[0] = 0.f; bounds[1] = (float)cols; bounds[2] = 0.f; bounds[3] = (float)rows;
    }
    return DRFE_OK;
}

int drfe_frame_download_keys_un(drfe_ctx* c, int slot, drfe_keypoint* kps, int cap)
{
    if (!c || !kps) return DRFE_ERR_INVALID;
    if (slot < 0 || slot >= c->lastBatch || !c->glueValid) { c->err = "download_keys_un: run the glue first"; return DRFE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_kpCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n > cap) { c->err = "keypoint buffer too small"; return DRFE_ERR_CAPACITY; }
    if (n) HIPCHK(c, hipMemcpy(kps, drfe_kps_un(c) + (size_t)slot * c->maxKp, sizeof(drfe_keypoint) * n, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

/* ---- sparse depth: the host keeps the depth images and ships one raw value per keypoint -------------------------------- */

int drfe_orb_keypoint_pixels_async(drfe_ctx* c, int nframes, uint32_t* uv, int32_t* counts, void* stream)
{
    if (!c || !uv || !counts) return DRFE_ERR_INVALID;
    if (nframes < 1 || nframes > c->lastBatch) { c->err = "keypoint_pixels: extract the batch first"; return DRFE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_kpUV) HIPCHK(c, hipMalloc((void**)&c->d_kpUV, sizeof(uint32_t) * (size_t)c->cfg.max_batch * c->maxKp));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, drfe_launch_kp_pixels(c, nframes, c->d_kpUV, s));
    HIPCHK(c, hipMemcpyAsync(uv, c->d_kpUV, sizeof(uint32_t) * (size_t)nframes * c->maxKp, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(counts, c->d_kpCount, sizeof(int) * (size_t)nframes, hipMemcpyDeviceToHost, s));
    return DRFE_OK;
}

int drfe_gather_keypoint_depth(const uint16_t* depth, size_t frame_stride, size_t row_stride, int nframes, const uint32_t* uv,
                               const int32_t* counts, int max_kp, uint16_t* out, int n_threads)
{
    if (!depth || !uv || !counts || !out || nframes < 0 || max_kp < 1) return DRFE_ERR_INVALID;
    int T = n_threads > 0 ? n_threads : drfe_default_host_threads();
    T = std::max(1, std::min(T, nframes));
    auto work = [&](int f0, int f1) {
        for (int f = f0; f < f1; f++) {
            const uint16_t* img = depth + (size_t)f * frame_stride;
            const uint32_t* p = uv + (size_t)f * max_kp;
            uint16_t* o = out + (size_t)f * max_kp;
            const int n = std::min(counts[f], max_kp);
            /* ~1000 reads scattered over a 600 KB image: every one a cache miss.  Prefetching a window ahead keeps a dozen
             * misses in flight per core instead of one (6 ms -> under 1 ms per 512 frames on 8 threads) */
            const int ahead = 16;
            for (int i = 0; i < std::min(ahead, n); i++)
                if (p[i] != 0xFFFFFFFFu) __builtin_prefetch(&img[(size_t)(p[i] >> 16) * row_stride + (p[i] & 0xFFFFu)]);
            for (int i = 0; i < n; i++) {
                if (i + ahead < n && p[i + ahead] != 0xFFFFFFFFu)
                    __builtin_prefetch(&img[(size_t)(p[i + ahead] >> 16) * row_stride + (p[i + ahead] & 0xFFFFu)]);
                o[i] = p[i] == 0xFFFFFFFFu ? (uint16_t)0 : img[(size_t)(p[i] >> 16) * row_stride + (p[i] & 0xFFFFu)];
            }
        }
    };
    if (T == 1) { work(0, nframes); return DRFE_OK; }
    std::vector<std::thread> th;
    for (int k = 0; k < T; k++) th.emplace_back(work, (int)((int64_t)nframes * k / T), (int)((int64_t)nframes * (k + 1) / T));
    for (std::thread& t : th) t.join();
    return DRFE_OK;
}

int drfe_frame_stereo_grid_batch_kpdepth(drfe_ctx* c, const uint16_t* kp_depth, int kp_depth_on_host, const drfe_camera* cam,
                                         int nframes, void* stream)
{
    if (!c || !kp_depth || !cam) return DRFE_ERR_INVALID;
    if (nframes < 1 || nframes > c->lastBatch) { c->err = "glue: extract the batch first"; return DRFE_ERR_STATE; }
    if (!(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y)) { c->err = "glue: empty image bounds"; return DRFE_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const uint16_t* d = kp_depth;
    if (kp_depth_on_host) {
        if (!c->d_kpDepth) HIPCHK(c, hipMalloc((void**)&c->d_kpDepth, sizeof(uint16_t) * (size_t)c->cfg.max_batch * c->maxKp));
        HIPCHK(c, hipMemcpyAsync(c->d_kpDepth, kp_depth, sizeof(uint16_t) * (size_t)nframes * c->maxKp, hipMemcpyHostToDevice, s));
        d = c->d_kpDepth;
    }
    HIPCHK(c, drfe_launch_glue(c, d, (size_t)c->maxKp, 0, *cam, nframes, s));
    c->glueValid = true;
    c->cam = *cam;
    return DRFE_OK;
}

/* A frame that lives on the HOST - a KeyFrame of the map, a Frame built elsewhere - put into a slot for the slot-based matchers:
 * what Frame::Frame left in mvKeys / mvKeysUn / mDescriptors / mvuRight / mvDepth is uploaded as it is and only
 * AssignFeaturesToGrid runs (on the device), so every matcher sees the slot exactly as if the frame had been extracted there. */
int drfe_frame_load(drfe_ctx* c, int slot, const drfe_keypoint* kps, const drfe_keypoint* kps_un, const uint8_t* desc, const float* u_right,
                    const float* depth_m, int n, const drfe_camera* cam)
{
    if (!c) return DRFE_ERR_INVALID;
    if (slot < 0 || slot >= c->cfg.max_batch || n < 0 || n > c->maxKp || !cam || (n > 0 && (!kps || !desc))) {
        c->err = "drfe_frame_load: invalid argument (slot, keypoint count beyond drfe_orb_max_keypoints, or a missing array)";
        return DRFE_ERR_INVALID;
    }
    if (!(cam->max_x > cam->min_x) || !(cam->max_y > cam->min_y)) { c->err = "drfe_frame_load: empty image bounds"; return DRFE_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    const size_t K = (size_t)c->maxKp, o = (size_t)slot * K;
    hipStream_t s = c->stream;
    if (n > 0) {
        /* without a distortion model the context keeps ONE keypoint array (mvKeysUn == mvKeys): the matchers read mvKeysUn */
        if (c->dist.enabled) {
            HIPCHK(c, hipMemcpyAsync(c->d_kps + o, kps, sizeof(drfe_keypoint) * (size_t)n, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync(c->d_kpsUn + o, kps_un ? kps_un : kps, sizeof(drfe_keypoint) * (size_t)n, hipMemcpyHostToDevice, s));
        } else
            HIPCHK(c, hipMemcpyAsync(c->d_kps + o, kps_un ? kps_un : kps, sizeof(drfe_keypoint) * (size_t)n, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->d_desc + o * 32, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
        if (u_right) HIPCHK(c, hipMemcpyAsync(c->d_uRight + o, u_right, (size_t)n * 4, hipMemcpyHostToDevice, s));
        else HIPCHK(c, drfe_launch_fill_i32(reinterpret_cast<int*>(c->d_uRight + o), n, (int)0xBF800000u /* -1.0f: no right coordinate */, s));
        if (depth_m) HIPCHK(c, hipMemcpyAsync(c->d_depth + o, depth_m, (size_t)n * 4, hipMemcpyHostToDevice, s));
        else HIPCHK(c, drfe_launch_fill_i32(reinterpret_cast<int*>(c->d_depth + o), n, (int)0xBF800000u, s));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_kpCount + slot, &n, sizeof(int), hipMemcpyHostToDevice, s));
    {
        SlotShift shift(c, slot);
        HIPCHK(c, drfe_launch_grid(c, *cam, 1, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));               /* `n` and the caller's arrays are free to go */
    c->lastBatch = std::max(c->lastBatch, slot + 1);
    c->glueValid = true;
    c->cam = *cam;
    if (c->bow) drfe_bow_slot_invalidate(c, slot);
    return DRFE_OK;
}

int drfe_frame_download_stereo(drfe_ctx* c, int slot, float* u_right, float* depth, int cap)
{
    if (!c || slot < 0 || slot >= c->lastBatch || !c->glueValid) return c ? DRFE_ERR_STATE : DRFE_ERR_INVALID;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_kpCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n > cap) return DRFE_ERR_CAPACITY;
    if (n && u_right) HIPCHK(c, hipMemcpy(u_right, c->d_uRight + (size_t)slot * c->maxKp, sizeof(float) * n, hipMemcpyDeviceToHost));
    if (n && depth) HIPCHK(c, hipMemcpy(depth, c->d_depth + (size_t)slot * c->maxKp, sizeof(float) * n, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_frame_download_grid(drfe_ctx* c, int slot, int32_t* offsets, int32_t* indices, int cap)
{
    if (!c || !offsets || slot < 0 || slot >= c->lastBatch || !c->glueValid) return c ? DRFE_ERR_STATE : DRFE_ERR_INVALID;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    HIPCHK(c, hipMemcpy(offsets, c->d_gridOff + (size_t)slot * (DRFE_GRID_CELLS + 1), sizeof(int) * (DRFE_GRID_CELLS + 1),
                        hipMemcpyDeviceToHost));
    const int n = offsets[DRFE_GRID_CELLS];
    if (n > cap) return DRFE_ERR_CAPACITY;
    if (n && indices) HIPCHK(c, hipMemcpy(indices, c->d_gridIdx + (size_t)slot * c->maxKp, sizeof(int) * n, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

} /* extern "C": the two helpers below have C++ linkage (match_internal.h) */

/* drfe_match_consecutive_batch in two halves (drfe_pipeline_submit replays the second one from a captured graph):
 *   stage    the per-pair records and the poses into pinned memory the context owns, two copies onto the stream (returns without waiting)
 *   enqueue  the memsets and kernels: everything a hipGraph can hold - kernel arguments by value, no host data */
int drfe_match_consecutive_stage(drfe_ctx* c, const float* Tcw, const float* Twc, const drfe_camera* cam, int mono, int nframes, hipStream_t s)
{
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    const int np = nframes - 1;
    const size_t pairBytes = sizeof(MatchPair) * (size_t)c->cfg.max_batch, poseBytes = sizeof(float) * 16 * (size_t)c->cfg.max_batch;
    if (!m->h_stage[0]) {
        m->stageBytes = pairBytes + poseBytes;
        for (int k = 0; k < 2; k++) {
            HIPCHK(c, hipHostMalloc(&m->h_stage[k], m->stageBytes, hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&m->stageEv[k], hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(m->stageEv[k], s));
        }
    }
    const int slotK = m->stageNext;
    m->stageNext ^= 1;
    HIPCHK(c, hipEventSynchronize(m->stageEv[slotK]));          /* the copies of two calls ago have left this buffer */
    MatchPair* pairs = static_cast<MatchPair*>(m->h_stage[slotK]);
    float* poses = reinterpret_cast<float*>(static_cast<uint8_t*>(m->h_stage[slotK]) + pairBytes);
    std::memcpy(poses, Twc, sizeof(float) * 16 * (size_t)nframes);
    const float mb = cam->bf / cam->fx;
    for (int p = 0; p < np; p++) {
        MatchPair& P = pairs[p];
        P.curSlot = p + 1; P.lastSlot = p; P.mpSlot = p; P.nQueries = 0;
        P.queryBase = p * c->maxKp;
        P.mpBase = p * c->maxKp;   /* k_mappoints_last writes slot-major */
        std::memcpy(P.Tcw, Tcw + (size_t)(p + 1) * 16, sizeof(float) * 16);
        drfe_motion_flags(Tcw + (size_t)(p + 1) * 16, Tcw + (size_t)p * 16, mb, mono, &P.forward, &P.backward);
    }
    HIPCHK(c, hipMemcpyAsync(m->d_pairs, pairs, sizeof(MatchPair) * np, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->d_poses, poses, sizeof(float) * 16 * nframes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(m->stageEv[slotK], s));
    return DRFE_OK;
}

hipError_t drfe_match_consecutive_enqueue(drfe_ctx* c, const drfe_camera* cam, float th, int check_ori, int nframes, hipStream_t s)
{
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return hipErrorOutOfMemory;
    hipError_t e = hipMemsetAsync(c->d_match, 0xFF, sizeof(int) * (size_t)nframes * c->maxKp, s);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_matchCount, 0, sizeof(int) * nframes, s);
    if (e == hipSuccess) e = drfe_launch_mappoints_last(c, *m, *cam, c->d_poses, nframes, s);
    if (e == hipSuccess) e = drfe_launch_window_match(c, *m, *cam, nframes - 1, c->maxKp, 0, th, 0.f, check_ori, nullptr, s, 2);      /* its own overflow word */
    return e;
}

extern "C" {

int drfe_match_consecutive_batch(drfe_ctx* c, const float* Tcw, const float* Twc, const drfe_camera* cam, float th,
                                 int mono, int check_ori, int nframes, void* stream)
{
    if (!c || !Tcw || !Twc || !cam) return DRFE_ERR_INVALID;
    if (nframes < 2 || nframes > c->lastBatch || !c->glueValid) {
        c->err = "match: needs an extracted batch of >= 2 frames with stereo/grid computed";
        return DRFE_ERR_STATE;
    }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    /* pairs + poses staged in pinned memory the context owns: the call returns without waiting for the stream */
    const int rc = drfe_match_consecutive_stage(c, Tcw, Twc, cam, mono, nframes, s);
    if (rc != DRFE_OK) return rc;
    if (c->profile) { (void)hipEventRecord(c->ev[DRFE_STAGE_MATCH][0], s); c->evUsed[DRFE_STAGE_MATCH] = true; }
    HIPCHK(c, drfe_match_consecutive_enqueue(c, cam, th, check_ori, nframes, s));
    if (c->profile) (void)hipEventRecord(c->ev[DRFE_STAGE_MATCH][1], s);
    return DRFE_OK;
}

int drfe_match_download(drfe_ctx* c, int slot, int32_t* cur_to_last, int cap, int* nmatches)
{
    if (!c || slot < 0 || slot >= c->lastBatch) return c ? DRFE_ERR_STATE : DRFE_ERR_INVALID;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    rc = match_status(c, 2);            /* the batch matcher's own word: later searches do not clear it */
    if (rc != DRFE_OK) return rc;
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_kpCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n > cap) return DRFE_ERR_CAPACITY;
    if (n && cur_to_last)
        HIPCHK(c, hipMemcpy(cur_to_last, c->d_match + (size_t)slot * c->maxKp, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (nmatches) HIPCHK(c, hipMemcpy(nmatches, c->d_matchCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_search_by_projection_last(drfe_ctx* c, int cur_slot, int last_slot, const float* Tcw_cur, const float* Tcw_last,
                                   const drfe_camera* cam, const drfe_map_point* last_mp, int n_last, float th, int mono,
                                   int check_ori, const uint8_t* cur_obs, int32_t* cur_mp, int n_cur, int* nmatches)
{
    if (!c || !Tcw_cur || !Tcw_last || !cam || !last_mp || !cur_mp || !nmatches) return DRFE_ERR_INVALID;
    if (cur_slot < 0 || cur_slot >= c->lastBatch || last_slot < 0 || last_slot >= c->lastBatch || !c->glueValid) {
        c->err = "search_by_projection_last: slots not ready";
        return DRFE_ERR_STATE;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int counts[2];
    HIPCHK(c, hipMemcpy(&counts[0], c->d_kpCount + cur_slot, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&counts[1], c->d_kpCount + last_slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n_cur != counts[0] || n_last != counts[1]) { c->err = "search_by_projection_last: N mismatch"; return DRFE_ERR_INVALID; }
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    MatchPair P;
    P.curSlot = cur_slot; P.lastSlot = last_slot; P.mpSlot = -1; P.nQueries = n_last; P.queryBase = 0; P.mpBase = 0;
    std::memcpy(P.Tcw, Tcw_cur, sizeof(float) * 16);
    drfe_motion_flags(Tcw_cur, Tcw_last, cam->bf / cam->fx, mono, &P.forward, &P.backward);
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpy(m->d_pairs, &P, sizeof(P), hipMemcpyHostToDevice));
    if (n_last) HIPCHK(c, hipMemcpy(m->d_mps, last_mp, sizeof(drfe_map_point) * n_last, hipMemcpyHostToDevice));
    if (n_cur) HIPCHK(c, hipMemcpy(c->d_match + (size_t)cur_slot * c->maxKp, cur_mp, sizeof(int) * n_cur, hipMemcpyHostToDevice));
    const uint8_t* d_obs = nullptr;
    if (cur_obs && n_cur) {
        HIPCHK(c, hipMemcpy(m->d_initObs, cur_obs, n_cur, hipMemcpyHostToDevice));
        d_obs = m->d_initObs;
    }
    *nmatches = 0;
    if (n_last == 0 || n_cur == 0) return DRFE_OK;
    HIPCHK(c, drfe_launch_window_match(c, *m, *cam, 1, n_last, 0, th, 0.f, check_ori, d_obs, s));
    HIPCHK(c, hipStreamSynchronize(s));
    rc = match_status(c);
    if (rc != DRFE_OK) return rc;
    HIPCHK(c, hipMemcpy(cur_mp, c->d_match + (size_t)cur_slot * c->maxKp, sizeof(int) * n_cur, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nmatches, c->d_matchCount + cur_slot, sizeof(int), hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_search_by_projection_map(drfe_ctx* c, int slot, const drfe_tracked_point* mps, int mcount, float th,
                                  float nnratio, const uint8_t* claim_obs, int32_t* frame_mp, int n, int* nmatches)
{
    if (!c || !mps || !frame_mp || !nmatches || mcount < 0) return DRFE_ERR_INVALID;
    if (slot < 0 || slot >= c->lastBatch || !c->glueValid) { c->err = "search_by_projection_map: slot not ready"; return DRFE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int cnt = 0;
    HIPCHK(c, hipMemcpy(&cnt, c->d_kpCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n != cnt) { c->err = "search_by_projection_map: N mismatch"; return DRFE_ERR_INVALID; }
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    if ((size_t)mcount > m->queryCap) { c->err = "search_by_projection_map: too many map points for the scratch"; return DRFE_ERR_CAPACITY; }
    *nmatches = 0;
    if (mcount == 0 || n == 0) return DRFE_OK;
    /* window parameters per map point, src/ORBmatcher.cc:62-70 (RadiusByViewingCos, r*=th) */
    std::vector<MatchQuery> q(mcount);
    const bool bFactor = th != 1.0;
    for (int i = 0; i < mcount; i++) {
        const drfe_tracked_point& t = mps[i];
        MatchQuery& Q = q[i];
        std::memset(&Q, 0, sizeof(Q));
        Q.valid = (t.track_in_view && !t.bad) ? 1 : 0;
        if (Q.valid && (t.level < 0 || t.level >= c->cfg.nlevels)) { c->err = "tracked point level out of range"; return DRFE_ERR_INVALID; }
        float r = ((double)t.view_cos > 0.998) ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        const float rs = Q.valid ? r * c->scale[t.level] : 0.f;
        Q.u = t.proj_x; Q.v = t.proj_y; Q.radius = rs; Q.ur = t.proj_xr; Q.thrR = rs;
        Q.minLevel = t.level - 1; Q.maxLevel = t.level;
        Q.obs = t.obs_positive;
        std::memcpy(Q.desc, t.desc, 32);
    }
    MatchPair P;
    std::memset(&P, 0, sizeof(P));
    P.curSlot = slot; P.lastSlot = slot; P.mpSlot = -1; P.nQueries = mcount; P.queryBase = 0; P.mpBase = 0;
    const drfe_camera cam = c->cam;   /* frame bounds / grid of the glue call that built this slot */
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpy(m->d_pairs, &P, sizeof(P), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(m->d_queries, q.data(), sizeof(MatchQuery) * mcount, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_match + (size_t)slot * c->maxKp, frame_mp, sizeof(int) * n, hipMemcpyHostToDevice));
    const uint8_t* d_obs = nullptr;
    if (claim_obs) {
        HIPCHK(c, hipMemcpy(m->d_initObs, claim_obs, n, hipMemcpyHostToDevice));
        d_obs = m->d_initObs;
    }
    HIPCHK(c, drfe_launch_window_match(c, *m, cam, 1, mcount, 1, th, nnratio, 0, d_obs, s));
    HIPCHK(c, hipStreamSynchronize(s));
    rc = match_status(c);
    if (rc != DRFE_OK) return rc;
    HIPCHK(c, hipMemcpy(frame_mp, c->d_match + (size_t)slot * c->maxKp, sizeof(int) * n, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nmatches, c->d_matchCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_match_orb_points(drfe_ctx* c, int cur_slot, int last_slot, const int32_t* last_mp, const uint8_t* last_outlier,
                          int n_last, int32_t* cur_mp, int n_cur, int* n_pairs)
{
    if (!c || !last_mp || !last_outlier || !cur_mp || !n_pairs) return DRFE_ERR_INVALID;
    if (cur_slot < 0 || cur_slot >= c->lastBatch || last_slot < 0 || last_slot >= c->lastBatch) {
        c->err = "match_orb_points: slots not ready";
        return DRFE_ERR_STATE;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int counts[2];
    HIPCHK(c, hipMemcpy(&counts[0], c->d_kpCount + cur_slot, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&counts[1], c->d_kpCount + last_slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n_cur != counts[0] || n_last != counts[1]) { c->err = "match_orb_points: N mismatch"; return DRFE_ERR_INVALID; }
    *n_pairs = 0;
    if (n_cur == 0 || n_last == 0) return DRFE_OK;
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    /* matcher.match(descriptor1 = Current, descriptor2 = Last): 1-NN on the device-resident rows */
    hipStream_t s = c->stream;
    HIPCHK(c, drfe_launch_bf_knn(c->d_desc + (size_t)cur_slot * c->maxKp * 32, n_cur,
                                 c->d_desc + (size_t)last_slot * c->maxKp * 32, n_last, 1, m->d_bfIdx, m->d_bfDist, s));
    std::vector<int32_t> idx(n_cur), dist(n_cur);
    HIPCHK(c, hipMemcpyAsync(idx.data(), m->d_bfIdx, sizeof(int) * n_cur, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(dist.data(), m->d_bfDist, sizeof(int) * n_cur, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    /* min distance, `dist < max(2*min_dist, 15)` filter, and the pointer copy with the reference's
     * mvbOutlier[i] indexing by match counter (src/ORBmatcher.cc:1350-1389) */
    double min_dist = 1000;
    for (int i = 0; i < n_cur; i++)
        if ((float)dist[i] < min_dist) min_dist = (float)dist[i];
    int npair = 0;
    for (int i = 0; i < n_cur; i++) {
        if (!((float)dist[i] < std::max(2 * min_dist, 15.0))) continue;
        const int mp = last_mp[idx[i]];
        if (mp >= 0 && npair < n_last && !last_outlier[npair]) cur_mp[i] = mp;
        npair++;
    }
    *n_pairs = npair;
    return DRFE_OK;
}

/* 1.4826 * median absolute deviation of the NN1 distance and of the NN2-NN1 gap (Frame::lineDescriptorMAD,
 * reference src/Frame.cc:560-584); DMatch::distance is float, the medians are widened to double. */
static void line_mad(const int32_t* dist, int nq, double* nnMad, double* nn12Mad)
{
    std::vector<float> d0(nq), gap(nq), s(nq);
    for (int i = 0; i < nq; i++) { d0[i] = (float)dist[2 * i]; gap[i] = (float)dist[2 * i + 1] - (float)dist[2 * i]; }
    s = d0;
    std::nth_element(s.begin(), s.begin() + nq / 2, s.end());
    const double med = s[nq / 2];
    for (int i = 0; i < nq; i++) s[i] = std::fabs((float)(d0[i] - med));
    std::nth_element(s.begin(), s.begin() + nq / 2, s.end());
    *nnMad = 1.4826 * s[nq / 2];
    s = gap;
    std::nth_element(s.begin(), s.begin() + nq / 2, s.end(), [](float a, float b) { return a > b; });
    const double med12 = s[nq / 2];
    for (int i = 0; i < nq; i++) s[i] = std::fabs((float)(gap[i] - med12));
    std::nth_element(s.begin(), s.begin() + nq / 2, s.end());
    *nn12Mad = 1.4826 * s[nq / 2];
}

int drfe_lsd_search_by_descriptor(drfe_ctx* c, const uint8_t* desc_q, int n_q, const uint8_t* desc_t, int n_t,
                                  const uint8_t* has_line, int mode, int32_t* out, int* nmatches)
{
    if (!c || !desc_q || !desc_t || !out || !nmatches || n_q < 0 || n_t < 0 || mode < 0 || mode > 1) return DRFE_ERR_INVALID;
    *nmatches = 0;
    const int n_out = mode == 0 ? n_t : n_q;
    for (int i = 0; i < n_out; i++) out[i] = -1;
    if (n_q == 0 || n_t < 2) return DRFE_OK;     /* knnMatch(k=2) needs two train rows */
    std::vector<int32_t> idx((size_t)n_q * 2), dist((size_t)n_q * 2);
    int rc = drfe_match_bf_knn(c, desc_q, n_q, desc_t, n_t, 2, idx.data(), dist.data());
    if (rc != DRFE_OK) return rc;
    double nnTh, nn12Th;
    line_mad(dist.data(), n_q, &nnTh, &nn12Th);
    int n = 0;
    if (mode == 0) {          /* SearchByDescriptor(KeyFrame*, Frame&): ratio d0/d1 < 1/1.5, src/LSDmatcher.cpp:255-277 */
        const float minRatio = 1.0f / 1.5f;
        for (int q = 0; q < n_q; q++) {
            const double r = (float)dist[2 * q] / (float)dist[2 * q + 1];
            if (r < minRatio && (!has_line || has_line[q])) { out[idx[2 * q]] = q; n++; }
        }
    } else {                  /* (KeyFrame*, KeyFrame*) / SerachForInitialize: gap > MAD12/2, :225-238, :294-311 */
        const double th = nn12Th * 0.5;
        for (int q = 0; q < n_q; q++) {
            const double gap = (float)dist[2 * q + 1] - (float)dist[2 * q];
            if (gap > th && (!has_line || has_line[idx[2 * q]])) { out[q] = idx[2 * q]; n++; }
        }
    }
    *nmatches = n;
    return DRFE_OK;
}

/* LSDmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs), src/LSDmatcher.cpp:334-367 */
int drfe_lsd_search_for_triangulation(drfe_ctx* c, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2,
                                      const uint8_t* has1, const uint8_t* has2, int32_t* out12, int* nmatches)
{
    if (!c || !desc1 || !desc2 || !has1 || !has2 || !out12 || !nmatches || n1 < 0 || n2 < 0) return DRFE_ERR_INVALID;
    *nmatches = 0;
    for (int i = 0; i < n1; i++) out12[i] = -1;
    if (n1 == 0 || n2 < 2) return DRFE_OK;
    std::vector<int32_t> idx((size_t)n1 * 2), dist((size_t)n1 * 2);
    int rc = drfe_match_bf_knn(c, desc1, n1, desc2, n2, 2, idx.data(), dist.data());
    if (rc != DRFE_OK) return rc;
    double nnTh, nn12Th;
    line_mad(dist.data(), n1, &nnTh, &nn12Th);
    const double th = nn12Th * 0.1;
    int n = 0;
    for (int q = 0; q < n1; q++) {
        const int t = idx[2 * q];
        if (has1[q] || has2[t]) continue;
        const double gap = (float)dist[2 * q + 1] - (float)dist[2 * q];
        if (gap > th) { out12[q] = t; n++; }
    }
    *nmatches = n;
    return DRFE_OK;
}

int drfe_match_bf_knn(drfe_ctx* c, const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, int32_t* dist)
{
    if (!c || !q || !t || !idx || !dist || nq < 0 || nt < 0 || k < 1 || k > 2) return DRFE_ERR_INVALID;
    if (nq == 0) return DRFE_OK;
    HIPCHK(c, hipSetDevice(c->device));
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    if ((size_t)nq > m->bfCap || (size_t)nt > m->bfCap) { c->err = "bf_knn: descriptor set larger than the scratch"; return DRFE_ERR_CAPACITY; }
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpyAsync(m->d_bfQ, q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
    if (nt) HIPCHK(c, hipMemcpyAsync(m->d_bfT, t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
    HIPCHK(c, drfe_launch_bf_knn(m->d_bfQ, nq, m->d_bfT, nt, k, m->d_bfIdx, m->d_bfDist, s));
    HIPCHK(c, hipMemcpyAsync(idx, m->d_bfIdx, sizeof(int) * (size_t)nq * k, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(dist, m->d_bfDist, sizeof(int) * (size_t)nq * k, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return DRFE_OK;
}


/* ---- LSDmatcher::SearchByProjection (row a-15) ---------------------------------------------------- */

static int line_search_run(drfe_ctx* c, const std::vector<LineQuery>* hostQ, const drfe_map_line* last, int n,
                           const float* TcwCur, const drfe_camera* cam, int fwd, int bwd, float th,
                           const drfe_keyline* cur_lines, const uint8_t* cur_desc, int n_cur, float nnratio,
                           const uint8_t* cur_obs, int32_t* cur_ml, int* nmatches)
{
    *nmatches = 0;
    if (n == 0 || n_cur == 0) return DRFE_OK;
    if (n_cur > 4096 || n > 65536) { c->err = "lsd_search_by_projection: too many lines"; return DRFE_ERR_CAPACITY; }
    HIPCHK(c, hipSetDevice(c->device));
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    std::vector<LineCur> lc(n_cur);
    std::vector<uint8_t> claim(n_cur);
    for (int i = 0; i < n_cur; i++) {
        lc[i].ptX = cur_lines[i].pt_x; lc[i].ptY = cur_lines[i].pt_y; lc[i].angle = cur_lines[i].angle;
        lc[i].octave = cur_lines[i].octave;
        claim[i] = cur_ml[i] >= 0 ? (uint8_t)(1 | ((cur_obs ? cur_obs[i] : 1) ? 2 : 0)) : 0;
    }
    /* one scratch block: queries | map lines | Tcw | current lines | descriptors | claims | cur_ml | count */
    const size_t oQ = 0, oL = oQ + sizeof(LineQuery) * n, oT = oL + (last ? sizeof(drfe_map_line) * n : 0),
                 oC = oT + 64, oD = oC + sizeof(LineCur) * n_cur, oK = oD + (size_t)n_cur * 32,
                 oM = (oK + n_cur + 15) & ~(size_t)15, oN = oM + sizeof(int) * n_cur, total = oN + 16;
    uint8_t* d = nullptr;
    HIPCHK(c, call_scratch(c, total, &d));
    hipStream_t s = c->stream;
    hipError_t e = hipSuccess;
    auto up = [&](size_t off, const void* src, size_t bytes) { if (e == hipSuccess && bytes) e = hipMemcpyAsync(d + off, src, bytes, hipMemcpyHostToDevice, s); };
    if (hostQ) up(oQ, hostQ->data(), sizeof(LineQuery) * n);
    else { up(oL, last, sizeof(drfe_map_line) * n); up(oT, TcwCur, 64); }
    up(oC, lc.data(), sizeof(LineCur) * n_cur);
    up(oD, cur_desc, (size_t)n_cur * 32);
    up(oK, claim.data(), n_cur);
    up(oM, cur_ml, sizeof(int) * n_cur);
    if (e == hipSuccess && !hostQ)
        e = drfe_launch_line_projection(reinterpret_cast<const drfe_map_line*>(d + oL), n, reinterpret_cast<const float*>(d + oT),
                                        *cam, fwd, bwd, m->d_scale, th, reinterpret_cast<LineQuery*>(d + oQ), s);
    if (e == hipSuccess)
        e = drfe_launch_line_search(reinterpret_cast<const LineQuery*>(d + oQ), n, reinterpret_cast<const LineCur*>(d + oC), d + oD,
                                    n_cur, nnratio, d + oK, reinterpret_cast<int*>(d + oM), reinterpret_cast<int*>(d + oN), s);
    if (e == hipSuccess) e = hipMemcpyAsync(cur_ml, d + oM, sizeof(int) * n_cur, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(nmatches, d + oN, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { c->err = std::string("lsd_search_by_projection: ") + hipGetErrorString(e); return DRFE_ERR_HIP; }
    return DRFE_OK;
}

int drfe_lsd_search_by_projection_last(drfe_ctx* c, const float* Tcw_cur, const float* Tcw_last, const drfe_camera* cam,
                                       const drfe_map_line* last_lines, int n_last, const drfe_keyline* cur_lines,
                                       const uint8_t* cur_desc, int n_cur, float th, int mono, float nnratio,
                                       const uint8_t* cur_obs, int32_t* cur_ml, int* nmatches)
{
    if (!c || !Tcw_cur || !Tcw_last || !cam || !nmatches || n_last < 0 || n_cur < 0) return DRFE_ERR_INVALID;
    if ((n_last && !last_lines) || (n_cur && (!cur_lines || !cur_desc || !cur_ml))) return DRFE_ERR_INVALID;
    for (int i = 0; i < n_last; i++)
        if (last_lines[i].valid && (last_lines[i].octave < 0 || last_lines[i].octave >= c->cfg.nlevels)) {
            c->err = "lsd_search_by_projection_last: key line octave outside the scale table";
            return DRFE_ERR_INVALID;
        }
    int fwd = 0, bwd = 0;
    drfe_motion_flags(Tcw_cur, Tcw_last, cam->bf / cam->fx, mono, &fwd, &bwd);
    return line_search_run(c, nullptr, last_lines, n_last, Tcw_cur, cam, fwd, bwd, th, cur_lines, cur_desc, n_cur, nnratio,
                           cur_obs, cur_ml, nmatches);
}

int drfe_lsd_search_by_projection_map(drfe_ctx* c, const drfe_tracked_line* lines, int n, const drfe_keyline* cur_lines,
                                      const uint8_t* cur_desc, int n_cur, float th, float nnratio, const uint8_t* cur_obs,
                                      int32_t* cur_ml, int* nmatches)
{
    if (!c || !nmatches || n < 0 || n_cur < 0) return DRFE_ERR_INVALID;
    if ((n && !lines) || (n_cur && (!cur_lines || !cur_desc || !cur_ml))) return DRFE_ERR_INVALID;
    /* window per map line, src/LSDmatcher.cpp:152-161 (RadiusByViewingCos 5 / 8, r *= th, levels l-1..l) */
    std::vector<LineQuery> q(n);
    const bool bFactor = th != 1.0;
    for (int i = 0; i < n; i++) {
        const drfe_tracked_line& t = lines[i];
        LineQuery& Q = q[i];
        std::memset(&Q, 0, sizeof(Q));
        Q.valid = t.in_view ? 1 : 0;
        if (Q.valid && (t.level < 0 || t.level >= c->cfg.nlevels)) { c->err = "tracked line level out of range"; return DRFE_ERR_INVALID; }
        float r = ((double)t.view_cos > 0.998) ? 5.0f : 8.0f;
        if (bFactor) r *= th;
        Q.obs = t.obs_positive ? 1 : 0;
        Q.minLevel = t.level - 1; Q.maxLevel = t.level;
        Q.x1 = t.x1; Q.y1 = t.y1; Q.x2 = t.x2; Q.y2 = t.y2;
        Q.r = Q.valid ? r * c->scale[t.level] : 0.f;
        std::memcpy(Q.desc, t.desc, 32);
    }
    return line_search_run(c, &q, nullptr, n, nullptr, nullptr, 0, 0, th, cur_lines, cur_desc, n_cur, nnratio, cur_obs,
                           cur_ml, nmatches);
}


/* ---- Frame::isInFrustum ---------------------------------------------------------------------------- */

int drfe_frame_is_in_frustum(drfe_ctx* c, const float* Tcw, const drfe_camera* cam, const drfe_frustum_point* pts, int n,
                             float viewing_cos_limit, drfe_tracked_point* out)
{
    if (!c || !Tcw || !cam || n < 0 || (n && (!pts || !out))) return DRFE_ERR_INVALID;
    FrustumPose P;
    frustum_pose(c, Tcw, cam, viewing_cos_limit, &P);
    const drfe_camera cm = *cam;
    return frustum_run(c, pts, n, out, [&](const drfe_frustum_point* di, drfe_tracked_point* dout, hipStream_t s) {
        return drfe_launch_frustum_points(di, n, P, cm, dout, s);
    });
}

int drfe_frame_is_in_frustum_lines(drfe_ctx* c, const float* Tcw, const drfe_camera* cam, const drfe_frustum_line* lines, int n,
                                   float viewing_cos_limit, drfe_tracked_line* out)
{
    if (!c || !Tcw || !cam || n < 0 || (n && (!lines || !out))) return DRFE_ERR_INVALID;
    FrustumPose P;
    frustum_pose(c, Tcw, cam, viewing_cos_limit, &P);
    const drfe_camera cm = *cam;
    return frustum_run(c, lines, n, out, [&](const drfe_frustum_line* di, drfe_tracked_line* dout, hipStream_t s) {
        return drfe_launch_frustum_lines(di, n, P, cm, dout, s);
    });
}


/* ---- ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th): search part -------------------------------- */
static int fuse_search_impl(drfe_ctx* c, int slot, const float* Tcw, int sim3, const drfe_frustum_point* pts,
                            const uint8_t* descs, const uint8_t* skip, int n, float th, int32_t* best_idx, int32_t* best_dist,
                            const float* sR2 = nullptr, const float* t2 = nullptr)
{
    if (!c || !Tcw || n < 0 || (n && (!pts || !descs || !best_idx || !best_dist))) return DRFE_ERR_INVALID;
    if (slot < 0 || slot >= c->lastBatch || !c->glueValid) { c->err = "fuse_search: slot needs extract + glue first"; return DRFE_ERR_STATE; }
    if (c->cfg.nlevels > 16) { c->err = "fuse_search: more than 16 pyramid levels"; return DRFE_ERR_INVALID; }
    if (n == 0) return DRFE_OK;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    const drfe_camera cam = c->cam;                       /* the glue call's camera: bounds and grid of this slot */
    FrustumPose fp;
    frustum_pose(c, Tcw, &cam, 0.f, &fp);
    FuseParams P;
    std::memset(&P, 0, sizeof(P));
    std::memcpy(P.T, fp.T, 64);
    std::memcpy(P.Ow, fp.Ow, 12);                         /* `-Rcw.t()*tcw` of the Scw overloads: double accumulation */
    if (sim3 == 0) camera_centre_kf(Tcw, P.Ow);           /* pKF->GetCameraCenter(), src/ORBmatcher.cc:840 */
    P.bf = cam.bf; P.logScale = fp.logScale; P.th = th; P.nLevels = c->cfg.nlevels;
    P.sim3 = sim3;
    if (sR2) { std::memcpy(P.sR2, sR2, 36); std::memcpy(P.t2, t2, 12); }
    for (int l = 0; l < c->cfg.nlevels; l++) { P.scale[l] = c->scale[l]; P.invSigma2[l] = c->invSigma2[l]; }
    uint8_t* d = nullptr;
    const size_t oP = 0, oD = (sizeof(drfe_frustum_point) * (size_t)n + 63) & ~(size_t)63, oS = oD + (((size_t)n * 32 + 63) & ~(size_t)63),
                 oI = oS + (((size_t)n + 63) & ~(size_t)63), oB = oI + sizeof(int) * (size_t)n, total = oB + sizeof(int) * (size_t)n;
    HIPCHK(c, call_scratch(c, total, &d));
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(d + oP, pts, sizeof(drfe_frustum_point) * (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oD, descs, (size_t)n * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && skip) e = hipMemcpyAsync(d + oS, skip, (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = drfe_launch_fuse_search(c, slot, reinterpret_cast<const drfe_frustum_point*>(d + oP), d + oD, skip ? d + oS : nullptr, n, P,
                                    cam, reinterpret_cast<int*>(d + oI), reinterpret_cast<int*>(d + oB), s);
    if (e == hipSuccess) e = hipMemcpyAsync(best_idx, d + oI, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(best_dist, d + oB, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { c->err = std::string("fuse_search: ") + hipGetErrorString(e); return DRFE_ERR_HIP; }
    return DRFE_OK;
}

int drfe_fuse_search(drfe_ctx* c, int slot, const float* Tcw, const drfe_frustum_point* pts, const uint8_t* descs,
                     const uint8_t* skip, int n, float th, int32_t* best_idx, int32_t* best_dist)
{
    return fuse_search_impl(c, slot, Tcw, 0, pts, descs, skip, n, th, best_idx, best_dist);
}

/* ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, points, th, vpReplacePoint), src/ORBmatcher.cc:981-1107: Scw decomposed as
 * at :989-993 (double-accumulated row norm, float scale 1/s through convertTo) */
int drfe_fuse_search_sim3(drfe_ctx* c, int slot, const float* Scw, const drfe_frustum_point* pts, const uint8_t* descs,
                          const uint8_t* skip, int n, float th, int32_t* best_idx, int32_t* best_dist)
{
    if (!Scw) return DRFE_ERR_INVALID;
    const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    float T[16];
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) T[r * 4 + k] = Scw[r * 4 + k] * inv;
        T[r * 4 + 3] = Scw[r * 4 + 3] * inv;
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
    return fuse_search_impl(c, slot, T, 1, pts, descs, skip, n, th, best_idx, best_dist);
}

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1106-1330 */
int drfe_search_by_sim3(drfe_ctx* c, int slot1, int slot2, const float* T1w, const float* T2w, float s12, const float* R12,
                        const float* t12, const drfe_frustum_point* pts1, const uint8_t* descs1, const uint8_t* skip1, int n1,
                        const drfe_frustum_point* pts2, const uint8_t* descs2, const uint8_t* skip2, int n2, float th,
                        int32_t* matches12, int* n_found)
{
    if (!c || !T1w || !T2w || !R12 || !t12 || !skip1 || !skip2 || !matches12 || !n_found || n1 < 0 || n2 < 0) return DRFE_ERR_INVALID;
    *n_found = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return DRFE_OK;
    /* sR12 = s12*R12, sR21 = (1.0/s12)*R12.t(), t21 = -sR21*t12 (:1121-1123) as cv::Mat evaluates them */
    float sR12[9], sR21[9], t21[3];
    const float a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) { sR12[r * 3 + k] = R12[r * 3 + k] * s12; sR21[r * 3 + k] = R12[k * 3 + r] * a21; }
    for (int r = 0; r < 3; r++) {
        const float d = sR21[r * 3] * t12[0] + sR21[r * 3 + 1] * t12[1] + sR21[r * 3 + 2] * t12[2];
        t21[r] = (float)((double)d * -1.0);
    }
    std::vector<int32_t> m1(n1), d1(n1), m2(n2), d2(n2);
    int rc = fuse_search_impl(c, slot2, T1w, 2, pts1, descs1, skip1, n1, th, m1.data(), d1.data(), sR21, t21);
    if (rc != DRFE_OK) return rc;
    rc = fuse_search_impl(c, slot1, T2w, 2, pts2, descs2, skip2, n2, th, m2.data(), d2.data(), sR12, t12);
    if (rc != DRFE_OK) return rc;
    int found = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        const int idx2 = (m1[i1] >= 0 && d1[i1] <= 100) ? m1[i1] : -1;                /* TH_HIGH */
        if (idx2 < 0 || idx2 >= n2) continue;
        const int idx1 = (m2[idx2] >= 0 && d2[idx2] <= 100) ? m2[idx2] : -1;
        if (idx1 == i1) { matches12[i1] = idx2; found++; }
    }
    *n_found = found;
    return DRFE_OK;
}

/* The two matchers whose loop over the map points is first come, first served in the reference (a keypoint claimed by an
 * earlier point stops being a candidate): SearchByProjection(KeyFrame*, Scw, ...) (mode 3) and the relocalisation
 * SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (mode 4).  The device lists, per point, its
 * FUSE_LIST_K best candidates within listTh among the keypoints free on entry, in the reference's comparison order; the
 * host then walks the points in order and takes the first listed candidate that is still free.  A point whose list ran
 * out while the device counted more candidates than it listed is asked again, alone, against the current claims (the
 * device search, not a host one).  pick[i] = the keypoint point i claimed or -1; taken[] is updated. */
static int first_come_search(drfe_ctx* c, const char* who, int slot, const float* T, int mode, int listTh,
                             const drfe_frustum_point* pts, const uint8_t* descs, const uint8_t* skip, int n,
                             std::vector<uint8_t>& taken, float th, std::vector<int>& pick)
{
    const int n_kp = (int)taken.size();
    pick.assign(n, -1);
    if (slot < 0 || slot >= c->lastBatch || !c->glueValid) { c->err = std::string(who) + ": slot needs extract + glue first"; return DRFE_ERR_STATE; }
    if (c->cfg.nlevels > 16) { c->err = std::string(who) + ": more than 16 pyramid levels"; return DRFE_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int slotCount = 0;
    HIPCHK(c, hipMemcpy(&slotCount, c->d_kpCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n_kp != slotCount) { c->err = std::string(who) + ": n_kp is not the slot's keypoint count"; return DRFE_ERR_INVALID; }
    if (n == 0 || n_kp == 0) return DRFE_OK;
    const drfe_camera cam = c->cam;
    FrustumPose fp;
    frustum_pose(c, T, &cam, 0.f, &fp);                                                 /* Ow = -Rcw.t()*tcw in both callers */
    FuseParams P;
    std::memset(&P, 0, sizeof(P));
    std::memcpy(P.T, fp.T, 64);
    std::memcpy(P.Ow, fp.Ow, 12);
    P.bf = cam.bf; P.logScale = fp.logScale; P.th = th; P.nLevels = c->cfg.nlevels;
    P.sim3 = mode; P.listTh = listTh;
    for (int l = 0; l < c->cfg.nlevels; l++) { P.scale[l] = c->scale[l]; P.invSigma2[l] = c->invSigma2[l]; }
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t oP = 0, oD = up(sizeof(drfe_frustum_point) * (size_t)n), oS = oD + up((size_t)n * 32), oT = oS + up((size_t)n),
                 oI = oT + up((size_t)n_kp), oB = oI + up(sizeof(int) * (size_t)n), oC = oB + up(sizeof(int) * (size_t)n),
                 oL = oC + up(sizeof(int) * (size_t)n), total = oL + sizeof(int2) * (size_t)n * FUSE_LIST_K;
    uint8_t* d = nullptr;
    HIPCHK(c, call_scratch(c, total, &d));
    hipStream_t s = c->stream;
    std::vector<int2> list((size_t)n * FUSE_LIST_K);
    std::vector<int> count(n);
    auto P_ = [&](size_t o) { return d + o; };
    hipError_t e = hipMemcpyAsync(P_(oP), pts, sizeof(drfe_frustum_point) * (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(P_(oD), descs, (size_t)n * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && skip) e = hipMemcpyAsync(P_(oS), skip, (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(P_(oT), taken.data(), (size_t)n_kp, hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = drfe_launch_fuse_search(c, slot, reinterpret_cast<const drfe_frustum_point*>(P_(oP)), P_(oD), skip ? P_(oS) : nullptr, n, P, cam,
                                    reinterpret_cast<int*>(P_(oI)), reinterpret_cast<int*>(P_(oB)), s, P_(oT),
                                    reinterpret_cast<int2*>(P_(oL)), reinterpret_cast<int*>(P_(oC)));
    if (e == hipSuccess) e = hipMemcpyAsync(list.data(), P_(oL), sizeof(int2) * list.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(count.data(), P_(oC), sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    /* DRFE_TEST_LIST_K=1..FUSE_LIST_K shortens the lists the host trusts so that tests reach the ask-again path */
    int listK = FUSE_LIST_K;
    if (const char* ev = std::getenv("DRFE_TEST_LIST_K")) { const int v = std::atoi(ev); if (v >= 1 && v < FUSE_LIST_K) listK = v; }
    for (int i = 0; i < n && e == hipSuccess; i++) {
        const int2* L = &list[(size_t)i * FUSE_LIST_K];
        /* entries the list really holds in order: all of them up to listK unless more than 64 candidates folded */
        const int held = count[i] > 64 ? 1 : (count[i] < listK ? count[i] : listK);
        int pk = -1;
        for (int r = 0; r < held; r++)
            if (!taken[L[r].x]) { pk = L[r].x; break; }
        if (pk < 0 && count[i] > held) {
            /* every listed candidate was claimed and there were more: search this point again with today's claims */
            int one[2] = {-1, 256};
            e = hipMemcpyAsync(P_(oT), taken.data(), (size_t)n_kp, hipMemcpyHostToDevice, s);
            if (e == hipSuccess)
                e = drfe_launch_fuse_search(c, slot, reinterpret_cast<const drfe_frustum_point*>(P_(oP)) + i, P_(oD) + (size_t)i * 32,
                                            skip ? P_(oS) + i : nullptr, 1, P, cam, reinterpret_cast<int*>(P_(oI)),
                                            reinterpret_cast<int*>(P_(oB)), s, P_(oT), nullptr, nullptr);
            if (e == hipSuccess) e = hipMemcpyAsync(&one[0], P_(oI), sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipMemcpyAsync(&one[1], P_(oB), sizeof(int), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e == hipSuccess && one[0] >= 0 && one[1] <= listTh) pk = one[0];
        }
        if (pk >= 0) { taken[pk] = 1; pick[i] = pk; }
    }
    if (e != hipSuccess) { c->err = std::string(who) + ": " + hipGetErrorString(e); return DRFE_ERR_HIP; }
    return DRFE_OK;
}

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:294-407 */
int drfe_search_by_projection_kf(drfe_ctx* c, int slot, const float* Scw, const drfe_frustum_point* pts, const uint8_t* descs,
                                 const uint8_t* skip, int n, const uint8_t* matched, int n_kp, float th, int32_t* new_match,
                                 int* n_matches)
{
    if (!c || !Scw || !matched || !new_match || !n_matches || n < 0 || n_kp < 0 || (n && (!pts || !descs))) return DRFE_ERR_INVALID;
    *n_matches = 0;
    for (int k = 0; k < n_kp; k++) new_match[k] = -1;
    const double dd = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
    const float scw = (float)std::sqrt(dd);
    const float inv = (float)(1.0 / (double)scw);
    float T[16];
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) T[r * 4 + k] = Scw[r * 4 + k] * inv;
        T[r * 4 + 3] = Scw[r * 4 + 3] * inv;
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
    std::vector<uint8_t> taken(matched, matched + n_kp);
    std::vector<int> pick;
    const int rc = first_come_search(c, "search_by_projection_kf", slot, T, 3, 50 /* TH_LOW */, pts, descs, skip, n, taken, th, pick);
    if (rc != DRFE_OK) return rc;
    int nm = 0;
    for (int i = 0; i < n; i++)
        if (pick[i] >= 0) { new_match[pick[i]] = i; nm++; }
    *n_matches = nm;
    return DRFE_OK;
}

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist),
 * src/ORBmatcher.cc:1537-1664 (Tracking::Relocalization) */
int drfe_search_by_projection_reloc(drfe_ctx* c, int slot, const float* Tcw, const drfe_frustum_point* pts, const uint8_t* descs,
                                    const float* kf_angles, const uint8_t* skip, int n, const uint8_t* matched, int n_kp, float th,
                                    int orb_dist, int check_orientation, int32_t* new_match, int* n_matches)
{
    if (!c || !Tcw || !matched || !new_match || !n_matches || n < 0 || n_kp < 0 || (n && (!pts || !descs)) || orb_dist < 0 || orb_dist > 256)
        return DRFE_ERR_INVALID;
    if (check_orientation && n && !kf_angles) return DRFE_ERR_INVALID;
    *n_matches = 0;
    for (int k = 0; k < n_kp; k++) new_match[k] = -1;
    std::vector<uint8_t> taken(matched, matched + n_kp);
    std::vector<int> pick;
    const int rc = first_come_search(c, "search_by_projection_reloc", slot, Tcw, 4, orb_dist, pts, descs, skip, n, taken, th, pick);
    if (rc != DRFE_OK) return rc;
    std::vector<drfe_keypoint> kp;
    if (check_orientation && n && n_kp) {
        kp.resize(n_kp);
        HIPCHK(c, hipMemcpy(kp.data(), drfe_kps_un(c) + (size_t)slot * c->maxKp, sizeof(drfe_keypoint) * (size_t)n_kp, hipMemcpyDeviceToHost));
    }
    int nm = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / 30;
    for (int i = 0; i < n; i++) {
        if (pick[i] < 0) continue;
        new_match[pick[i]] = i;
        nm++;
        if (check_orientation) {
            float rot = kf_angles[i] - kp[pick[i]].angle;                               /* :1623-1630 */
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == 30) bin = 0;
            if (bin < 0 || bin >= 30) { c->err = "search_by_projection_reloc: keypoint angle outside [0, 360)"; return DRFE_ERR_INVALID; }
            rotHist[bin].push_back(pick[i]);
        }
    }
    if (check_orientation) {                                                            /* ComputeThreeMaxima, :1666-1707 */
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < 30; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int k : rotHist[i]) { new_match[k] = -1; nm--; }
    }
    *n_matches = nm;
    return DRFE_OK;
}

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:409-524 (monocular
 * initialisation).  The loop over F1's level-0 keypoints is first come, first served with replacement: a candidate whose
 * recorded match is at least as close is skipped (:448), a new match evicts the older one (:467-471) - so the device gathers
 * each keypoint's window (GetFeaturesInArea order, level 0 only, Hamming distances) and the host replays the loop on the lists. */
int drfe_search_for_initialization(drfe_ctx* c, int slot1, int slot2, float* prev_matched, int n1, int window_size, float nnratio,
                                   int check_orientation, int32_t* matches12, int* n_matches)
{
    if (!c || !prev_matched || !matches12 || !n_matches || n1 < 0 || window_size < 0) return DRFE_ERR_INVALID;
    if (slot1 < 0 || slot1 >= c->lastBatch || slot2 < 0 || slot2 >= c->lastBatch || !c->glueValid) {
        c->err = "search_for_initialization: slots not ready";
        return DRFE_ERR_STATE;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int cnt[2] = {0, 0};
    HIPCHK(c, hipMemcpy(&cnt[0], c->d_kpCount + slot1, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&cnt[1], c->d_kpCount + slot2, sizeof(int), hipMemcpyDeviceToHost));
    if (n1 != cnt[0]) { c->err = "search_for_initialization: N mismatch"; return DRFE_ERR_INVALID; }
    const int n2 = cnt[1];
    *n_matches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                                   /* :412 */
    if (n1 == 0 || n2 == 0) return DRFE_OK;
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    std::vector<drfe_keypoint> k1(n1), k2(n2);
    std::vector<uint8_t> d1((size_t)n1 * 32);
    HIPCHK(c, hipMemcpy(k1.data(), drfe_kps_un(c) + (size_t)slot1 * c->maxKp, sizeof(drfe_keypoint) * (size_t)n1, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(k2.data(), drfe_kps_un(c) + (size_t)slot2 * c->maxKp, sizeof(drfe_keypoint) * (size_t)n2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(d1.data(), c->d_desc + (size_t)slot1 * c->maxKp * 32, d1.size(), hipMemcpyDeviceToHost));
    /* one window query per level-0 keypoint of F1 (:425-429): GetFeaturesInArea(prev.x, prev.y, windowSize, 0, 0), no stereo gate */
    std::vector<int> owner;
    std::vector<MatchQuery> q;
    for (int i1 = 0; i1 < n1; i1++) {
        if (k1[i1].octave > 0) continue;
        MatchQuery Q;
        std::memset(&Q, 0, sizeof(Q));
        Q.valid = 1;
        Q.u = prev_matched[2 * i1]; Q.v = prev_matched[2 * i1 + 1]; Q.radius = (float)window_size;
        Q.ur = 0.f; Q.thrR = std::numeric_limits<float>::infinity();
        Q.minLevel = k1[i1].octave; Q.maxLevel = k1[i1].octave;
        std::memcpy(Q.desc, &d1[(size_t)i1 * 32], 32);
        q.push_back(Q);
        owner.push_back(i1);
    }
    const int nq = (int)q.size();
    if (nq == 0) return DRFE_OK;
    if ((size_t)nq > m->queryCap) { c->err = "search_for_initialization: too many level-0 keypoints for the scratch"; return DRFE_ERR_CAPACITY; }
    MatchPair P;
    std::memset(&P, 0, sizeof(P));
    P.curSlot = slot2; P.lastSlot = slot2; P.mpSlot = -1; P.nQueries = nq;
    hipStream_t s = c->stream;
    HIPCHK(c, hipMemcpy(m->d_pairs, &P, sizeof(P), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(m->d_queries, q.data(), sizeof(MatchQuery) * (size_t)nq, hipMemcpyHostToDevice));
    HIPCHK(c, drfe_launch_window_candidates(c, *m, c->cam, 1, nq, s));
    std::vector<uint32_t> cIdx((size_t)nq * DRFE_MATCH_MAX_CAND), cKey((size_t)nq * DRFE_MATCH_MAX_CAND);
    std::vector<int> cCnt(nq);
    HIPCHK(c, hipMemcpyAsync(cIdx.data(), m->d_candIdx, cIdx.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(cKey.data(), m->d_candKey, cKey.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(cCnt.data(), m->d_candCnt, sizeof(int) * (size_t)nq, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    rc = match_status(c);
    if (rc != DRFE_OK) return rc;
    /* the loop of :422-491 on the gathered lists */
    const int TH_LOW = 50;
    std::vector<int> matchedDistance(n2, std::numeric_limits<int>::max()), matches21(n2, -1);
    std::vector<int> rotHist[30];
    const float factor = 1.0f / 30;
    int nm = 0;
    for (int k = 0; k < nq; k++) {
        const int i1 = owner[k];
        int bestDist = std::numeric_limits<int>::max(), bestDist2 = bestDist, bestIdx2 = -1;
        for (int t = 0; t < cCnt[k]; t++) {
            const int i2 = (int)(cIdx[(size_t)k * DRFE_MATCH_MAX_CAND + t] & 0xFFFFFFu);
            const int dist = (int)(cKey[(size_t)k * DRFE_MATCH_MAX_CAND + t] >> 22);
            if (matchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nm--; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDistance[bestIdx2] = bestDist;
            nm++;
            if (check_orientation) {
                float rot = k1[i1].angle - k2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == 30) bin = 0;
                if (bin < 0 || bin >= 30) { c->err = "search_for_initialization: keypoint angle outside [0, 360)"; return DRFE_ERR_INVALID; }
                rotHist[bin].push_back(i1);
            }
        }
    }
    if (check_orientation) {                                                          /* ComputeThreeMaxima, :1666-1707 */
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < 30; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx1 : rotHist[i])
                    if (matches12[idx1] >= 0) { matches12[idx1] = -1; nm--; }
    }
    for (int i1 = 0; i1 < n1; i1++)                                                   /* :519-521 */
        if (matches12[i1] >= 0) { prev_matched[2 * i1] = k2[matches12[i1]].x; prev_matched[2 * i1 + 1] = k2[matches12[i1]].y; }
    *n_matches = nm;
    return DRFE_OK;
}

/* the device search shared by LSDmatcher::Fuse x2, SearchByProjection(KF, Scw) and both directions of SearchBySim3: pose P
 * (and, for SearchBySim3, the similarity chained onto it), one wavefront per map line over the keyframe's key lines.
 * dist_rows != NULL: [n][n_kf] distances of the key lines that passed every gate (-1 otherwise) for a first-come replay. */
static int lsd_fuse_impl(drfe_ctx* c, const char* who, const FrustumPose& P, const drfe_camera* cam, const drfe_frustum_line* lines,
                         const uint8_t* descs, const uint8_t* skip, int n, const drfe_keyline* kf_lines, const uint8_t* kf_desc, int n_kf,
                         float th, int32_t* best_idx, int32_t* best_dist, const LineSim3* sim3, std::vector<int32_t>* dist_rows)
{
    if (n_kf > 65535) { c->err = std::string(who) + ": too many key lines"; return DRFE_ERR_CAPACITY; }
    HIPCHK(c, hipSetDevice(c->device));
    MatchBuffers* m = drfe_match_buffers(c);
    if (!m) return DRFE_ERR_HIP;
    std::vector<LineCur> lc(n_kf);
    for (int i = 0; i < n_kf; i++) {
        lc[i].ptX = kf_lines[i].pt_x; lc[i].ptY = kf_lines[i].pt_y; lc[i].angle = kf_lines[i].angle; lc[i].octave = kf_lines[i].octave;
    }
    auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
    const size_t rows = dist_rows ? sizeof(int) * (size_t)n * (size_t)n_kf : 0;
    const size_t oL = 0, oD = up(sizeof(drfe_frustum_line) * (size_t)n), oS = oD + up((size_t)n * 32), oK = oS + up((size_t)n),
                 oKD = oK + up(sizeof(LineCur) * (size_t)n_kf), oI = oKD + up((size_t)n_kf * 32), oB = oI + up(sizeof(int) * (size_t)n),
                 oR = oB + up(sizeof(int) * (size_t)n), total = oR + rows;
    uint8_t* d = nullptr;
    HIPCHK(c, call_scratch(c, total, &d));
    hipStream_t s = c->stream;
    hipError_t e = hipMemcpyAsync(d + oL, lines, sizeof(drfe_frustum_line) * (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d + oD, descs, (size_t)n * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && skip) e = hipMemcpyAsync(d + oS, skip, (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && n_kf) e = hipMemcpyAsync(d + oK, lc.data(), sizeof(LineCur) * (size_t)n_kf, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && n_kf) e = hipMemcpyAsync(d + oKD, kf_desc, (size_t)n_kf * 32, hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = drfe_launch_line_fuse_search(reinterpret_cast<const drfe_frustum_line*>(d + oL), d + oD, skip ? d + oS : nullptr, n, P, *cam,
                                         m->d_scale, th, reinterpret_cast<const LineCur*>(d + oK), d + oKD, n_kf,
                                         reinterpret_cast<int*>(d + oI), reinterpret_cast<int*>(d + oB), s, sim3,
                                         rows ? reinterpret_cast<int*>(d + oR) : nullptr);
    if (e == hipSuccess) e = hipMemcpyAsync(best_idx, d + oI, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(best_dist, d + oB, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rows) {
        dist_rows->resize((size_t)n * (size_t)n_kf);
        e = hipMemcpyAsync(dist_rows->data(), d + oR, rows, hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { c->err = std::string(who) + ": " + hipGetErrorString(e); return DRFE_ERR_HIP; }
    return DRFE_OK;
}

/* Scw -> [Rcw | tcw] as LSDmatcher.cpp:386-390 / :759-763 evaluate it (Mat::dot in double, Mat / float through convertTo) */
static void decompose_scw(const float* Scw, float T[16])
{
    const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) T[r * 4 + k] = Scw[r * 4 + k] * inv;
        T[r * 4 + 3] = Scw[r * 4 + 3] * inv;
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
}

/* LSDmatcher::Fuse(KeyFrame* pKF, const vector<MapLine*>& vpMapLines, th), src/LSDmatcher.cpp:884-1010: the search */
int drfe_lsd_fuse_search(drfe_ctx* c, const float* Tcw, const drfe_camera* cam, const drfe_frustum_line* lines, const uint8_t* descs,
                         const uint8_t* skip, int n, const drfe_keyline* kf_lines, const uint8_t* kf_desc, int n_kf, float th,
                         int32_t* best_idx, int32_t* best_dist)
{
    if (!c || !Tcw || !cam || n < 0 || n_kf < 0 || (n && (!lines || !descs || !best_idx || !best_dist)) || (n_kf && (!kf_lines || !kf_desc)))
        return DRFE_ERR_INVALID;
    if (n == 0) return DRFE_OK;
    FrustumPose P;
    frustum_pose(c, Tcw, cam, 0.f, &P);
    camera_centre_kf(Tcw, P.Ow);
    return lsd_fuse_impl(c, "lsd_fuse_search", P, cam, lines, descs, skip, n, kf_lines, kf_desc, n_kf, th, best_idx, best_dist, nullptr, nullptr);
}

/* LSDmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, vpLines, th, vpReplaceLine), src/LSDmatcher.cpp:750-882: the search.  The pose
 * is the decomposed similarity, the camera centre -Rcw.t()*tcw (:763) */
int drfe_lsd_fuse_search_sim3(drfe_ctx* c, const float* Scw, const drfe_camera* cam, const drfe_frustum_line* lines, const uint8_t* descs,
                              const uint8_t* skip, int n, const drfe_keyline* kf_lines, const uint8_t* kf_desc, int n_kf, float th,
                              int32_t* best_idx, int32_t* best_dist)
{
    if (!c || !Scw || !cam || n < 0 || n_kf < 0 || (n && (!lines || !descs || !best_idx || !best_dist)) || (n_kf && (!kf_lines || !kf_desc)))
        return DRFE_ERR_INVALID;
    if (n == 0) return DRFE_OK;
    float T[16];
    decompose_scw(Scw, T);
    FrustumPose P;
    frustum_pose(c, T, cam, 0.f, &P);                 /* Ow = -Rcw.t()*tcw */
    return lsd_fuse_impl(c, "lsd_fuse_search_sim3", P, cam, lines, descs, skip, n, kf_lines, kf_desc, n_kf, th, best_idx, best_dist, nullptr,
                         nullptr);
}

/* LSDmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpLines, vpMatched, th), src/LSDmatcher.cpp:377-502.  The loop over
 * the map lines is first come, first served (a key line taken by an earlier line is no candidate, :476): the device returns every
 * line's gated distances, the host replays the claims in order. */
int drfe_lsd_search_by_projection_kf(drfe_ctx* c, const float* Scw, const drfe_camera* cam, const drfe_frustum_line* lines,
                                     const uint8_t* descs, const uint8_t* skip, int n, const drfe_keyline* kf_lines, const uint8_t* kf_desc,
                                     int n_kf, const uint8_t* matched, int th, int32_t* new_match, int* n_matches)
{
    if (!c || !Scw || !cam || !n_matches || n < 0 || n_kf < 0 || (n && (!lines || !descs)) || (n_kf && (!kf_lines || !kf_desc || !matched || !new_match)))
        return DRFE_ERR_INVALID;
    *n_matches = 0;
    for (int k = 0; k < n_kf; k++) new_match[k] = -1;
    if (n == 0 || n_kf == 0) return DRFE_OK;
    float T[16];
    decompose_scw(Scw, T);
    FrustumPose P;
    frustum_pose(c, T, cam, 0.f, &P);
    std::vector<int32_t> bi(n), bd(n), rows;
    const int rc = lsd_fuse_impl(c, "lsd_search_by_projection_kf", P, cam, lines, descs, skip, n, kf_lines, kf_desc, n_kf, (float)th, bi.data(),
                                 bd.data(), nullptr, &rows);
    if (rc != DRFE_OK) return rc;
    std::vector<uint8_t> taken(matched, matched + n_kf);
    int nm = 0;
    for (int i = 0; i < n; i++) {
        if (bi[i] < 0) continue;                         /* dropped by a gate, no candidate, or level outside the pyramid (-2) */
        int bestDist = 256, bestIdx = -1;
        const int32_t* row = &rows[(size_t)i * n_kf];
        for (int idx = 0; idx < n_kf; idx++) {
            if (row[idx] < 0 || taken[idx]) continue;
            if (row[idx] < bestDist) { bestDist = row[idx]; bestIdx = idx; }
        }
        if (bestDist <= 50) { taken[bestIdx] = 1; new_match[bestIdx] = i; nm++; }        /* TH_LOW */
    }
    *n_matches = nm;
    return DRFE_OK;
}

/* LSDmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/LSDmatcher.cpp:504-748: both directions on the
 * device, the agreement check on the host.  lines / descs / skip are per key line of the keyframe (skip = no map line, bad, or
 * already matched); kf1_lines / kf2_lines the key lines themselves. */
int drfe_lsd_search_by_sim3(drfe_ctx* c, const drfe_camera* cam, const float* T1w, const float* T2w, float s12, const float* R12,
                            const float* t12, const drfe_frustum_line* lines1, const uint8_t* descs1, const uint8_t* skip1,
                            const drfe_keyline* kf1_lines, const uint8_t* kf1_desc, int n1, const drfe_frustum_line* lines2,
                            const uint8_t* descs2, const uint8_t* skip2, const drfe_keyline* kf2_lines, const uint8_t* kf2_desc, int n2,
                            float th, int32_t* matches12, int* n_found)
{
    if (!c || !cam || !T1w || !T2w || !R12 || !t12 || !matches12 || !n_found || n1 < 0 || n2 < 0) return DRFE_ERR_INVALID;
    if ((n1 && (!lines1 || !descs1 || !skip1 || !kf1_lines || !kf1_desc)) || (n2 && (!lines2 || !descs2 || !skip2 || !kf2_lines || !kf2_desc)))
        return DRFE_ERR_INVALID;
    *n_found = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return DRFE_OK;
    /* sR12 = s12*R12, sR21 = (1.0/s12)*R12.t(), t21 = -sR21*t12 (:521-523) as cv::Mat evaluates them */
    LineSim3 c21, c12;
    const float a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) { c12.sR[r * 3 + k] = R12[r * 3 + k] * s12; c21.sR[r * 3 + k] = R12[k * 3 + r] * a21; }
    for (int r = 0; r < 3; r++) {
        const float d = c21.sR[r * 3] * t12[0] + c21.sR[r * 3 + 1] * t12[1] + c21.sR[r * 3 + 2] * t12[2];
        c21.t[r] = (float)((double)d * -1.0);
        c12.t[r] = t12[r];
    }
    std::vector<int32_t> m1(n1), d1(n1), m2(n2), d2(n2);
    FrustumPose P;
    frustum_pose(c, T1w, cam, 0.f, &P);
    int rc = lsd_fuse_impl(c, "lsd_search_by_sim3", P, cam, lines1, descs1, skip1, n1, kf2_lines, kf2_desc, n2, th, m1.data(), d1.data(), &c21,
                           nullptr);
    if (rc != DRFE_OK) return rc;
    frustum_pose(c, T2w, cam, 0.f, &P);
    rc = lsd_fuse_impl(c, "lsd_search_by_sim3", P, cam, lines2, descs2, skip2, n2, kf1_lines, kf1_desc, n1, th, m2.data(), d2.data(), &c12,
                       nullptr);
    if (rc != DRFE_OK) return rc;
    int found = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        const int idx2 = (m1[i1] >= 0 && d1[i1] <= 100) ? m1[i1] : -1;                /* TH_HIGH */
        if (idx2 < 0) continue;
        const int idx1 = (m2[idx2] >= 0 && d2[idx2] <= 100) ? m2[idx2] : -1;
        if (idx1 == i1) { matches12[i1] = idx2; found++; }
    }
    *n_found = found;
    return DRFE_OK;
}

} /* extern "C" */
