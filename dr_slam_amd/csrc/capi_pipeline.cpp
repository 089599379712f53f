/* capi_pipeline.cpp - batches in flight as a C-ABI object (include/drfe.h: drfe_pipeline_*).
 *
 * The device rate of the batch path is bound by VALU instruction issue, but a third of a batch's kernels are latency-bound
 * (quadtree, claim resolution, the glue kernels: 10-40 % VALU-busy).  Running `depth` independent batches at once - each in its
 * own context, on the stream that context owns, which sits on its own hardware queue - lets those run beside the VALU-bound
 * kernels of the other batches: +10-13 % frames/s at depth 3.  A pipeline is nothing but `depth` contexts
 * used round robin; everything per batch is the public batch API. */
#include "drfe_internal.h"

#include <new>
#include <vector>

struct drfe_pipeline {
    std::vector<drfe_ctx*> ctx;
    unsigned long long submitted = 0;
    std::string err;
};

extern "C" {

int drfe_pipeline_create(const drfe_config* cfg, int depth, drfe_pipeline** out)
{
    if (!cfg || !out || depth < 1 || depth > 16) return DRFE_ERR_INVALID;
    *out = nullptr;
    drfe_pipeline* p = new (std::nothrow) drfe_pipeline();
    if (!p) return DRFE_ERR_INVALID;
    for (int k = 0; k < depth; k++) {
        drfe_ctx* c = nullptr;
        const int rc = drfe_create(cfg, &c);
        if (rc != DRFE_OK) {                       /* drfe_last_error(NULL) holds the reason */
            for (drfe_ctx* d : p->ctx) drfe_destroy(d);
            delete p;
            return rc;
        }
        p->ctx.push_back(c);
    }
    *out = p;
    return DRFE_OK;
}

void drfe_pipeline_destroy(drfe_pipeline* p)
{
    if (!p) return;
    for (drfe_ctx* c : p->ctx) drfe_destroy(c);
    delete p;
}

int drfe_pipeline_depth(const drfe_pipeline* p) { return p ? (int)p->ctx.size() : 0; }

drfe_ctx* drfe_pipeline_context(drfe_pipeline* p, int k)
{
    return (p && k >= 0 && k < (int)p->ctx.size()) ? p->ctx[(size_t)k] : nullptr;
}

const char* drfe_pipeline_last_error(const drfe_pipeline* p) { return p ? p->err.c_str() : ""; }

int drfe_pipeline_submit(drfe_pipeline* p, const uint8_t* d_gray, const uint16_t* d_depth, size_t frame_stride, size_t row_stride,
                         int w, int h, const float* Tcw, const float* Twc, const drfe_camera* cam, float th, int mono, int check_ori,
                         int nframes)
{
    if (!p || !d_gray || p->ctx.empty()) return DRFE_ERR_INVALID;
    const int k = (int)(p->submitted % p->ctx.size());
    drfe_ctx* c = p->ctx[(size_t)k];
    int rc = drfe_orb_extract_batch(c, d_gray, frame_stride, row_stride, w, h, nframes, nullptr);      /* NULL: the context's own stream */
    if (rc == DRFE_OK && d_depth) {
        if (!cam) { p->err = "drfe_pipeline_submit: a depth batch needs the camera"; return DRFE_ERR_INVALID; }
        rc = drfe_frame_stereo_grid_batch(c, d_depth, frame_stride, row_stride, cam, nframes, nullptr);
        if (rc == DRFE_OK && Tcw && Twc && nframes >= 2)
            rc = drfe_match_consecutive_batch(c, Tcw, Twc, cam, th, mono, check_ori, nframes, nullptr);
    }
    if (rc != DRFE_OK) { p->err = drfe_last_error(c); return rc; }
    p->submitted++;
    return k;
}

int drfe_pipeline_sync(drfe_pipeline* p, int k)
{
    if (!p || k >= (int)p->ctx.size()) return DRFE_ERR_INVALID;
    for (int i = 0; i < (int)p->ctx.size(); i++) {
        if (k >= 0 && i != k) continue;
        drfe_ctx* c = p->ctx[(size_t)i];
        if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            p->err = "drfe_pipeline_sync: stream synchronisation failed";
            return DRFE_ERR_HIP;
        }
        /* the throughput path reports arena overflows here: a truncated batch must not pass silently */
        const int rc = drfe_batch_check(c);
        if (rc != DRFE_OK) { p->err = drfe_last_error(c); return rc; }
    }
    return DRFE_OK;
}

} /* extern "C" */
