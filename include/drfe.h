/* drfe.h — C-ABI of the MI355X-native DR-SLAM feature front-end (libdrfe.so).
 *
 * The reference (WangWen-Believer/DR-SLAM) has no FFI: its per-frame feature path is reached through
 * C++ classes.  Each entry point below names the reference interface it replaces; INTEGRATION.md
 * shows the adaptor a maintainer would drop into DR-SLAM (`ORBextractor::operator()` etc. calling
 * these functions).  Plain C: opaque context, caller-owned buffers with capacities, int status
 * (0 = ok, <0 = drfe_status), no exceptions and no OpenCV/torch types cross this boundary.
 *
 * Threading: a drfe_ctx is bound to one HIP device and is NOT re-entrant (like the reference's
 * ORBextractor instance, include/ORBextractor.h:85 public pyramid member); use one context per calling
 * thread (Tracking / LocalMapping / LoopClosing).
 *
 * Pointers named d_* are device (HBM) pointers; all others are host pointers.
 */
#ifndef DRFE_H
#define DRFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum drfe_status {
    DRFE_OK = 0,
    DRFE_ERR_INVALID = -1,    /* bad argument (null pointer, size mismatch, unsupported geometry) */
    DRFE_ERR_HIP = -2,        /* HIP runtime failure; see drfe_last_error */
    DRFE_ERR_CAPACITY = -3,   /* caller buffer or internal pool too small; nothing partial is returned */
    DRFE_ERR_STATE = -4       /* call order violated (e.g. match before extract) */
} drfe_status;

/* cv::KeyPoint binary layout (7 x 4 bytes) so an adaptor can memcpy into std::vector<cv::KeyPoint>. */
typedef struct drfe_keypoint {
    float x, y;       /* pt, level-0 pixel units (already multiplied by the level scale) */
    float size;       /* 31 * scale[octave] truncated to int, src/ORBextractor.cc:837,846 */
    float angle;      /* degrees [0,360), IC_Angle */
    float response;   /* FAST score */
    int32_t octave;
    int32_t class_id; /* -1 */
} drfe_keypoint;

/* ORBextractor constructor arguments (include/ORBextractor.h:56-57; yaml keys ORBextractor.*,
 * src/Tracking.cc:120-126) plus the capacities the device buffers are sized for. */
typedef struct drfe_config {
    int32_t device;        /* HIP device ordinal */
    int32_t max_width;     /* largest frame the context will see (e.g. 640) */
    int32_t max_height;    /* (e.g. 480) */
    int32_t max_batch;     /* frames processed per batched call (>=1) */
    int32_t nfeatures;     /* 1000 */
    float scale_factor;    /* 1.2f */
    int32_t nlevels;       /* 8 */
    int32_t ini_th_fast;   /* 20 */
    int32_t min_th_fast;   /* 7 */
} drfe_config;

typedef struct drfe_ctx drfe_ctx;

/* Camera / Frame constants the glue and matchers read (Frame::Frame, src/Frame.cc:104-196). */
typedef struct drfe_camera {
    float fx, fy, cx, cy;
    float bf;            /* Camera.bf */
    float depth_factor;  /* 1/DepthMapFactor: metres per raw depth unit (src/Tracking.cc:144-148) */
    float min_x, max_x, min_y, max_y; /* mnMinX.. (image bounds when k1 == 0, src/Frame.cc:884-889) */
} drfe_camera;

/* ------------------------------------------------------------------------------------------------ */
/* lifetime                                                                                          */
int drfe_create(const drfe_config* cfg, drfe_ctx** out);
void drfe_destroy(drfe_ctx* ctx);
const char* drfe_last_error(const drfe_ctx* ctx); /* ctx may be NULL: last create() failure */
const char* drfe_version(void);

/* ------------------------------------------------------------------------------------------------ */
/* ORBextractor (replaces src/ORBextractor.cc)                                                       */

/* Getters of include/ORBextractor.h:63-83: arrays of nlevels floats. */
int drfe_orb_scale_tables(const drfe_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);
/* Capacity a caller needs for keypoint/descriptor buffers of one frame (>= any N the path can emit). */
int drfe_orb_max_keypoints(const drfe_ctx* ctx);

/* ORBextractor::operator()(image, mask, keypoints, descriptors), src/ORBextractor.cc:1043-1105.
 * gray: CV_8UC1 host image with row stride `stride` bytes.  Writes N keypoints and N x 32 descriptor
 * bytes (level-major order exactly as the reference concatenates them).  Empty image (w==0||h==0 or
 * gray NULL) -> *n_out = 0, DRFE_OK (reference returns silently, :1046-1047). */
int drfe_orb_extract(drfe_ctx* ctx, const uint8_t* gray, int w, int h, size_t stride, drfe_keypoint* kps,
                     uint8_t* desc, int cap, int* n_out);

/* Batched, device-resident form: nframes (<= max_batch) CV_8UC1 frames already in HBM, frame f at
 * d_gray + f*frame_stride, rows `row_stride` bytes apart.  Asynchronous on `stream` (a hipStream_t;
 * NULL = the context's own stream).  Results stay in the context's frame slots 0..nframes-1.
 * Several contexts working at once (batches in flight): pass NULL - the stream every drfe_create makes sits on its own
 * hardware queue, so the contexts' kernels really overlap (+10 % device rate at three batches in flight); streams from a
 * framework's pool may share a queue, and the batches then run one behind the other. */
int drfe_orb_extract_batch(drfe_ctx* ctx, const uint8_t* d_gray, size_t frame_stride, size_t row_stride, int w,
                           int h, int nframes, void* stream);
/* Copy slot results to host (synchronises the batch stream). */
int drfe_orb_download(drfe_ctx* ctx, int slot, drfe_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* The whole batch to host memory in five copies, asynchronous on `stream` (hipStream_t; NULL = the context stream): for a
 * pipeline that overlaps the results of batch i with the kernels of batch i+1 (pinned host buffers).  kps
 * [nframes][max_keypoints], desc [nframes][max_keypoints][32], kp_counts [nframes]; matches [nframes][max_keypoints] and
 * match_counts [nframes] (both may be NULL) as drfe_match_download returns them.  Rows past a frame's count are
 * unspecified.  The caller synchronises the stream before reading. */
int drfe_batch_download_async(drfe_ctx* ctx, int nframes, drfe_keypoint* kps, uint8_t* desc, int32_t* kp_counts,
                              int32_t* matches, int32_t* match_counts, void* stream);
/* The device status words of the batch most recently extracted / matched on ctx, copied asynchronously on `stream` (NULL: the
 * context's): status[0] = extraction (bit 0: FAST candidate arena overflow, bit 1: quadtree node pool overflow), status[1] =
 * drfe_match_consecutive_batch (bit 2: more than 256 keypoints in one search window).  Non-zero = keypoints / matches of that
 * batch are truncated.  The batch matcher's word is its own: no other search clears it. */
int drfe_batch_status_async(drfe_ctx* ctx, int32_t* status, void* stream);
/* The same check, blocking (waits for the context's stream): DRFE_ERR_CAPACITY with the reason in drfe_last_error if an arena
 * overflowed.  drfe_pipeline_sync calls it for every context it waits for. */
int drfe_batch_check(drfe_ctx* ctx);
/* Per-slot keypoint counts (host array of nframes ints; synchronises). */
int drfe_orb_counts(drfe_ctx* ctx, int nframes, int* counts);

/* mvImagePyramid parity (public member, include/ORBextractor.h:85): copy the bordered level
 * ((w_l+38) x (h_l+38), tightly packed) of a slot to host.  out may be NULL to query sizes. */
int drfe_orb_pyramid_level(drfe_ctx* ctx, int slot, int level, uint8_t* out, int* bordered_w, int* bordered_h);
/* Debug/parity taps (tests): blurred interior level, FAST candidates (x,y,response relative to the
 * 16-px detection border, unordered). */
int drfe_orb_blurred_level(drfe_ctx* ctx, int slot, int level, uint8_t* out, int* w, int* h);
int drfe_orb_candidates(drfe_ctx* ctx, int slot, int level, int32_t* xyr, int cap, int* n_out);

/* ------------------------------------------------------------------------------------------------ */
/* Frame glue (replaces Frame::ComputeStereoFromRGBD + AssignFeaturesToGrid, src/Frame.cc:224-237,
 * 893-911; no-distortion path of UndistortKeyPoints :836-838).                                      */

/* d_depth: raw CV_16U depth frames in HBM (same indexing as d_gray).  For every slot computes
 * mvDepth/mvuRight and the 64x48 feature grid, all kept on device for the matchers. */
int drfe_frame_stereo_grid_batch(drfe_ctx* ctx, const uint16_t* d_depth, size_t frame_stride_elems,
                                 size_t row_stride_elems, const drfe_camera* cam, int nframes, void* stream);
/* Sparse-depth form of the glue for a pipeline fed from HOST memory: ComputeStereoFromRGBD reads the depth image at <= max_keypoints
 * pixels per frame, so a host that keeps its depth images can ship one raw value per keypoint (2 KB per frame) instead of the
 * frame (614 KB).  Three steps after drfe_orb_extract_batch:
 *   drfe_orb_keypoint_pixels_async   the pixel of every keypoint (u | v << 16, truncated as the reference truncates kp.pt;
 *                                    0xFFFFFFFF = outside the image) and the per-frame counts, D2H on `stream`
 *   drfe_gather_keypoint_depth       HOST code: out[f][i] = depth_f(v, u) on n_threads threads (<= 0: the CPUs this process may use)
 *   drfe_frame_stereo_grid_batch_kpdepth   the glue from those values ([nframes][max_keypoints], host or device memory)
 * Results are identical to drfe_frame_stereo_grid_batch on the same depth images (tests/test_gpu_match.py). */
int drfe_orb_keypoint_pixels_async(drfe_ctx* ctx, int nframes, uint32_t* uv, int32_t* counts, void* stream);
int drfe_gather_keypoint_depth(const uint16_t* depth, size_t frame_stride_elems, size_t row_stride_elems, int nframes,
                               const uint32_t* uv, const int32_t* counts, int max_keypoints, uint16_t* out, int n_threads);
int drfe_frame_stereo_grid_batch_kpdepth(drfe_ctx* ctx, const uint16_t* kp_depth, int kp_depth_on_host, const drfe_camera* cam,
                                         int nframes, void* stream);
/* Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:835-891) for cameras with k1 != 0 (TUM1/TUM2
 * settings): dist = (k1, k2, p1, p2[, k3]) as Tracking reads Camera.k1.. into mDistCoef, cam supplies mK.  With a
 * model set, drfe_frame_stereo_grid_batch first builds mvKeysUn on the device (cv::undistortPoints(pts, K, dist,
 * Mat(), K): five fixed-point iterations in double) and the depth association, the grid and every matcher read
 * mvKeysUn, exactly as the reference does; n = 0 or k1 == 0 restores mvKeysUn = mvKeys.  The image bounds
 * (min_x, max_x, min_y, max_y of drfe_camera) come from drfe_frame_image_bounds. */
int drfe_frame_set_distortion(drfe_ctx* ctx, const drfe_camera* cam, const float* dist, int n);
int drfe_frame_image_bounds(const drfe_camera* cam, const float* dist, int n, int cols, int rows,
                            float* bounds /* min_x, max_x, min_y, max_y */);
int drfe_frame_download_keys_un(drfe_ctx* ctx, int slot, drfe_keypoint* kps, int cap);
int drfe_frame_download_stereo(drfe_ctx* ctx, int slot, float* u_right, float* depth, int cap);
/* grid as CSR in the reference's iteration order (cell = ix*48 + iy): offsets[64*48+1], indices[N] */
int drfe_frame_download_grid(drfe_ctx* ctx, int slot, int32_t* offsets, int32_t* indices, int cap);

/* ------------------------------------------------------------------------------------------------ */
/* Batches in flight: `depth` contexts used round robin, each on the stream it owns (its own hardware queue), so that the
 * latency-bound kernels of one batch (quadtree, claim resolution, glue) run beside the VALU-bound ones of the others:
 * +10-13 % device rate at depth 3, the measured optimum (bench.py uses this object).
 *   drfe_pipeline_submit   one batch through drfe_orb_extract_batch -> [d_depth != NULL: drfe_frame_stereo_grid_batch ->
 *                          Tcw && Twc: drfe_match_consecutive_batch] on the next context; asynchronous; returns the index k
 *                          (>= 0) of the context that holds the batch's results, or a negative DRFE_ERR_* code.  The caller
 *                          must have consumed the previous results of that context (depth submissions earlier).
 *                          frame_stride / row_stride count ELEMENTS of each image (bytes of d_gray, 16-bit words of d_depth).
 *   drfe_pipeline_context  context k: result access (drfe_orb_download, drfe_match_download, drfe_batch_download_async, ...)
 *                          and per-context setup (drfe_frame_set_distortion, drfe_voc_upload - once per context)
 *   drfe_pipeline_sync     waits for context k's stream (k < 0: all of them) */
typedef struct drfe_pipeline drfe_pipeline;
int drfe_pipeline_create(const drfe_config* cfg, int depth, drfe_pipeline** out);
void drfe_pipeline_destroy(drfe_pipeline* pipe);
int drfe_pipeline_depth(const drfe_pipeline* pipe);
drfe_ctx* drfe_pipeline_context(drfe_pipeline* pipe, int k);
const char* drfe_pipeline_last_error(const drfe_pipeline* pipe);
int drfe_pipeline_submit(drfe_pipeline* pipe, const uint8_t* d_gray, const uint16_t* d_depth, size_t frame_stride, size_t row_stride,
                         int w, int h, const float* Tcw, const float* Twc, const drfe_camera* cam, float th, int mono, int check_ori,
                         int nframes);
int drfe_pipeline_sync(drfe_pipeline* pipe, int k);

/* ------------------------------------------------------------------------------------------------ */
/* Per-frame pipelined flow: what Frame::Frame (src/Frame.cc:74-160) does for ONE frame, without waiting for it.
 * Tracking::GrabImageRGBD (src/Tracking.cc:191) builds one Frame at a time and matches it against the previous one:
 * frame k goes to slot k % max_batch, the slot of frame k-1 keeps LastFrame's keypoints, descriptors and grid on the
 * device for the slot-pair matchers (drfe_search_by_projection_last(cur_slot, last_slot, ...), drfe_match_orb_points,
 * drfe_search_by_bow, ...).  Other slots are not touched.
 *   drfe_frame_submit   copies the frame into pinned staging and enqueues, as one captured hipGraph per slot: H2D ->
 *                       ORBextractor::operator() -> [depth != NULL: UndistortKeyPoints / ComputeStereoFromRGBD /
 *                       AssignFeaturesToGrid with `cam` and the model of drfe_frame_set_distortion] -> D2H of mvKeys,
 *                       mDescriptors, mvuRight, mvDepth.  Returns without waiting: the calling thread is free for the host
 *                       halves of the line / plane extractors (the reference starts threads for ExtractORB / ExtractLSD /
 *                       ComputePlanes, src/Frame.cc:124-134).  depth: raw CV_16U image, rows depth_stride_elems apart,
 *                       or NULL (ORB only).  One submission per slot may be outstanding (DRFE_ERR_STATE otherwise).
 *   drfe_frame_collect  waits for that slot's submission and copies its results out (u_right / depth_m may be NULL; they
 *                       need a submission with a depth image).  Results are identical to drfe_orb_extract +
 *                       drfe_frame_stereo_grid_batch on the same frame (tests/test_gpu_match.py).
 * The batch entry points renumber the slots (a batch fills 0..nframes-1): use one context per flow. */
int drfe_frame_submit(drfe_ctx* ctx, int slot, const uint8_t* gray, int w, int h, size_t stride, const uint16_t* depth,
                      size_t depth_stride_elems, const drfe_camera* cam);
int drfe_frame_collect(drfe_ctx* ctx, int slot, drfe_keypoint* kps, uint8_t* desc, float* u_right, float* depth_m, int cap,
                       int* n_out);

/* A frame that lives on the HOST into a slot: the KeyFrame* / Frame& arguments of ORBmatcher's methods (include/ORBmatcher.h:41-84)
 * are objects of the map whose keypoints were extracted long ago (KeyFrame copies mvKeys, mvKeysUn, mDescriptors, mvuRight, mvDepth
 * from its Frame, src/KeyFrame.cc:36-66).  kps = mvKeys, kps_un = mvKeysUn (NULL: the same array), desc = mDescriptors (n x 32),
 * u_right = mvuRight, depth_m = mvDepth (either may be NULL: monocular, all -1), cam = intrinsics + the image bounds
 * mnMinX .. mnMaxY.  The arrays are uploaded as they are and Frame::AssignFeaturesToGrid (src/Frame.cc:224-237) runs on the
 * device, so the slot is indistinguishable from one the frame was extracted in; every slot-based matcher below then takes it.
 * n <= drfe_orb_max_keypoints(ctx).  Without a distortion model the context keeps one keypoint array (mvKeysUn == mvKeys) and
 * stores kps_un.  Synchronous; the slot must have no drfe_frame_submit outstanding.  The slot's BoW transform, if any, is
 * forgotten (drfe_bow_transform_slot). */
int drfe_frame_load(drfe_ctx* ctx, int slot, const drfe_keypoint* kps, const drfe_keypoint* kps_un, const uint8_t* desc,
                    const float* u_right, const float* depth_m, int n, const drfe_camera* cam);

/* ------------------------------------------------------------------------------------------------ */
/* ORBmatcher (replaces src/ORBmatcher.cc hot paths)                                                 */

/* What the matcher reads through LastFrame.mvpMapPoints[i] (MapPoint::GetWorldPos/GetDescriptor/
 * Observations, src/ORBmatcher.cc:1419-1423,1468-1470). */
typedef struct drfe_map_point {
    uint8_t valid;        /* pMP != NULL && !mvbOutlier[i] */
    uint8_t obs_positive; /* Observations() > 0 */
    uint8_t pad[2];
    float world[3];
    uint8_t desc[32];
} drfe_map_point;

/* ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), src/ORBmatcher.cc:1396-1535,
 * for every consecutive slot pair of a batch: cur = slot s, last = slot s-1 (s = 1..nframes-1).
 * Map points of `last` are its own keypoints with valid depth, unprojected with Twc[last]
 * (Frame::UnprojectStereo, src/Frame.cc:913-923) — the state Tracking::UpdateLastFrame leaves for an
 * RGB-D stream.  Tcw/Twc: nframes row-major 4x4 float matrices (host).  d_matches (device, may be
 * NULL) / results kept in ctx: per slot s, int32[max_keypoints]: index of the last-frame keypoint
 * matched to current keypoint i, or -1.  nnratio unused by this overload (no ratio test in the
 * reference); check_ori = mbCheckOrientation. */
int drfe_match_consecutive_batch(drfe_ctx* ctx, const float* Tcw, const float* Twc, const drfe_camera* cam,
                                 float th, int mono, int check_ori, int nframes, void* stream);
int drfe_match_download(drfe_ctx* ctx, int slot, int32_t* cur_to_last, int cap, int* nmatches);

/* General form of the same overload with caller-supplied map points for `last` slot and initial
 * CurrentFrame.mvpMapPoints claims (cur_mp in/out: -1 = NULL; cur_obs: Observations()>0 of the
 * initial claims, may be NULL = all 1). Synchronous. */
int drfe_search_by_projection_last(drfe_ctx* ctx, int cur_slot, int last_slot, const float* Tcw_cur,
                                   const float* Tcw_last, const drfe_camera* cam, const drfe_map_point* last_mp,
                                   int n_last, float th, int mono, int check_ori, const uint8_t* cur_obs,
                                   int32_t* cur_mp, int n_cur, int* nmatches);

/* ONE submission per tracked frame: drfe_frame_submit (above) followed, inside the same captured graph, by
 * ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) as Tracking::TrackWithMotionModel calls it
 * (src/Tracking.cc:2181-2202; src/ORBmatcher.cc:1396-1535) - H2D -> ORB -> glue -> map points -> window search + claim
 * resolution + rotation histogram -> D2H of keypoints, descriptors, mvuRight / mvDepth AND the match array.  CurrentFrame goes
 * to `slot`, LastFrame is the frame an earlier submission left in `last_slot` (with its grid: submitted with a depth image).
 * Tcw_cur = the pose the search projects with (mVelocity * LastFrame.mTcw), Tcw_last = LastFrame.mTcw (forward / backward
 * test).  last_mp != NULL: LastFrame.mvpMapPoints as the caller's map holds them, n_last = LastFrame's keypoint count;
 * last_mp == NULL: every keypoint of LastFrame with depth, unprojected with Twc_last (Frame::UnprojectStereo) - the
 * temporal points Tracking::UpdateLastFrame creates on an RGB-D stream - built on the device, so a frame can be submitted
 * before the previous one has been collected.  CurrentFrame's claims start empty, as TrackWithMotionModel's do.  Needs a depth
 * image.  drfe_frame_collect_tracked returns what drfe_frame_collect does plus cur_to_last[n] (index of the matched
 * LastFrame keypoint per current keypoint, -1 = none) and the number of matches; results are identical to
 * drfe_frame_submit + drfe_frame_collect + drfe_search_by_projection_last (tests/test_gpu_match.py). */
int drfe_frame_submit_tracked(drfe_ctx* ctx, int slot, const uint8_t* gray, int w, int h, size_t stride, const uint16_t* depth,
                              size_t depth_stride_elems, const drfe_camera* cam, int last_slot, const float* Tcw_cur,
                              const float* Tcw_last, const float* Twc_last, const drfe_map_point* last_mp, int n_last, float th,
                              int mono, int check_ori);
int drfe_frame_collect_tracked(drfe_ctx* ctx, int slot, drfe_keypoint* kps, uint8_t* desc, float* u_right, float* depth_m, int cap,
                               int* n_out, int32_t* cur_to_last, int* n_matches);

/* What SearchByProjection(Frame&, vector<MapPoint*>&, th) reads (fields written by
 * Frame::isInFrustum, src/Frame.cc:602-657). */
typedef struct drfe_tracked_point {
    uint8_t track_in_view, bad, obs_positive, pad;
    int32_t level;   /* mnTrackScaleLevel */
    float proj_x, proj_y, proj_xr, view_cos;
    uint8_t desc[32];
} drfe_tracked_point;

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:46-130.
 * frame_mp in/out: per frame keypoint the index into mps or -1. Synchronous. */
int drfe_search_by_projection_map(drfe_ctx* ctx, int slot, const drfe_tracked_point* mps, int m, float th,
                                  float nnratio, const uint8_t* claim_obs, int32_t* frame_mp, int n,
                                  int* nmatches);

/* ORBmatcher::MatchORBPoints(CurrentFrame, LastFrame), src/ORBmatcher.cc:1332-1394 — the fallback
 * Tracking uses when SearchByProjection finds < 40 matches (src/Tracking.cc:2195-2200): brute-force
 * 1-NN of the current slot's descriptors against the last slot's (both device-resident), keep
 * dist < max(2*min_dist, 15), copy map points with the reference's mvbOutlier[match counter] quirk.
 * last_mp: per last-frame keypoint the map point id or -1; cur_mp (in/out): ids written for matched
 * current keypoints; *n_pairs = the function's return value (NPair). */
int drfe_match_orb_points(drfe_ctx* ctx, int cur_slot, int last_slot, const int32_t* last_mp,
                          const uint8_t* last_outlier, int n_last, int32_t* cur_mp, int n_cur, int* n_pairs);

/* LSDmatcher descriptor matching (src/LSDmatcher.cpp): BFMatcher::knnMatch(k=2) of LBD descriptors on the
 * device + Frame::lineDescriptorMAD (src/Frame.cc:560-584) + the accept rule.
 *   mode 0  SearchByDescriptor(KeyFrame*, Frame&, ...), :242-279: query = keyframe lines, train = frame
 *           lines, accept d0/d1 < 1/1.5 when the keyframe line has a MapLine (has_line[q]);
 *           out[train idx] = query idx (later queries overwrite earlier ones), out has n_t entries.
 *   mode 1  SearchByDescriptor(KeyFrame*, KeyFrame*, ...), :281-314 and SerachForInitialize, :213-240:
 *           accept when d1 - d0 > 0.5 * MAD of that gap and has_line[train idx] (NULL = all);
 *           out[query idx] = train idx, out has n_q entries. */
int drfe_lsd_search_by_descriptor(drfe_ctx* ctx, const uint8_t* desc_q, int n_q, const uint8_t* desc_t, int n_t,
                                  const uint8_t* has_line, int mode, int32_t* out, int* nmatches);
/* LSDmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs), src/LSDmatcher.cpp:334-367 (LocalMapping::
 * CreateNewMapLines): knnMatch(k = 2) on the device, accept the nearest neighbour when the NN2-NN1 gap exceeds a tenth
 * of its MAD and neither line has a MapLine (has1 / has2).  out12[line of KF1] = line of KF2 or -1. */
int drfe_lsd_search_for_triangulation(drfe_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2,
                                      const uint8_t* has1, const uint8_t* has2, int32_t* out12, int* nmatches);

/* cv::BFMatcher(NORM_HAMMING).match / knnMatch(k<=2) on 256-bit descriptors (src/ORBmatcher.cc:1346,
 * src/LSDmatcher.cpp:222,254): ascending distance, ties -> lower train index. idx/dist: nq x k. */
int drfe_match_bf_knn(drfe_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx,
                      int32_t* dist);

/* ------------------------------------------------------------------------------------------------ */
/* LineSegment (replaces src/LSDextractor.cpp: LSDDetector::detect + BinaryDescriptor::compute)      */

/* cv::line_descriptor::KeyLine fields (angle, class_id, octave, pt, response, size, start/end point,
 * start/end point in octave, lineLength, numOfPixels). */
typedef struct drfe_keyline {
    float angle;
    int32_t class_id, octave;
    float pt_x, pt_y, response, size;
    float start_point_x, start_point_y, end_point_x, end_point_y;
    float s_point_in_octave_x, s_point_in_octave_y, e_point_in_octave_x, e_point_in_octave_y;
    float line_length;
    int32_t num_of_pixels;
} drfe_keyline;

/* LineSegment::ExtractLineSegment(img, keylines, ldesc, keylineFunctions, scale = 1.2f, numOctaves = 1),
 * src/LSDextractor.cpp:12-43.  gray: host CV_8UC1.  max_lines = lsdNFeatures (40).  Outputs: up to
 * max_lines key lines (the highest-response ones, class_id renumbered, :23-28), ldesc = NL x 32 LBD
 * bytes, line_f = NL x 3 normalised line equations; *n_detected = lines found before the cut. */
int drfe_lsd_extract(drfe_ctx* ctx, const uint8_t* gray, int w, int h, size_t stride, int max_lines,
                     drfe_keyline* lines, uint8_t* ldesc, double* line_f, int cap, int* n_lines, int* n_detected);
/* The sequential half of cv::LineSegmentDetector (pixel ordering, region growing, rectangle fit + refinement, NFA validation)
 * on caller-supplied level-line fields, without a device: what drfe_lsd_extract runs on the host between its device passes,
 * with the rectangle pixel counts taken on the host too.  modgrad / angles: W x H doubles of the 0.8-scaled image (angle
 * -1024 = undefined); cs: W x H x 2 floats, (cos, sin) of float(angle); max_grad: the largest modgrad.  segs: up to cap x 4
 * floats (x1, y1, x2, y2 in input-image coordinates).  Host code: CPU tests and profiling. */
int drfe_lsd_segments_host(const double* modgrad, const double* angles, const float* cs, int W, int H, double max_grad, float* segs,
                           int cap, int* n_segs);
/* The same for nframes host images (gray + f * frame_stride).  Default: the whole extractor runs on the DEVICE, frames side by
 * side - the image passes, std::sort's permutation of the pixel ordering (k_lsd_order), the seed loop with region_grow /
 * region2rect / refine (k_lsd_grow: one wavefront per frame executing the reference's order of operations), rect_improve with
 * its NFA decisions certified against the host libm's values (k_rect_improve, drfe_lsd_configure_nfa), the key lines, the
 * response cut's std::sort, the line equations (k_lsd_keylines) and the LBD descriptors (k_lbd); n_threads host threads upload,
 * launch and copy the results out (two are plenty), and finish on the host the rare frame a device stage hands back (an
 * uncertified rounding or comparison, a capacity: drfe_lsd_stats counts them).
 * drfe_lsd_configure(ctx, 0): every frame through drfe_lsd_extract's host path on the pool, one device lane per thread
 * (frames instead of the reference's four extractors across threads, src/Frame.cc:116-126).  Outputs per frame f at
 * lines[f * cap], ldesc[f * cap * 32], line_f[f * cap * 3], n_lines[f], n_detected[f]; results are identical to nframes
 * calls of drfe_lsd_extract either way.  n_threads <= 0: 1.25 x the CPUs this process may use (affinity mask clipped by the
 * cgroup quota). */
int drfe_lsd_extract_batch(drfe_ctx* ctx, const uint8_t* gray, size_t frame_stride, int w, int h, size_t stride, int nframes,
                           int max_lines, drfe_keyline* lines, uint8_t* ldesc, double* line_f, int cap, int* n_lines,
                           int* n_detected, int n_threads);
/* Where drfe_lsd_extract_batch grows regions: 0 on the host threads; on the device 1 (default) with the kernel chosen by the size of
 * the call, 2 with one wavefront per frame (the least device time per frame: what counts when calls of hundreds of frames run side by
 * side), 3 with four wavefronts per frame (seeds speculated against a commit-only `used` map and committed in seed order: 1.6x
 * shorter per frame, what counts when the call cannot fill the device; the default up to 256 frames per call).  Frames whose
 * 0.8-scaled size exceeds the device path's LDS bitmap (about 1.2 M pixels) take the host path regardless.  Results are identical. */
int drfe_lsd_configure(drfe_ctx* ctx, int device_grow);
/* Which reading of cv::LineSegmentDetectorImpl::rect_nfa / nfa (OpenCV 3.4 imgproc/src/lsd.cpp, the detector behind reference
 * src/LSDextractor.cpp:14-17) validates the rectangles of this context's line entries.
 * 0 (default) - the source text.  rect_nfa: `struct edge { cv::Point p; bool taken; }` has integer corners, so the four edge steps
 *   are INTEGER quotients, and the second steps use (y - tailp->p.x) in their guards and in their denominators (the scan lines of
 *   an oblique rectangle are then not the rectangle's own).  nfa: `log1term = (double(n) + 1) - log_gamma(double(k) + 1) -
 *   log_gamma(double(n-k) + 1) + ...` - the first log_gamma of the LSD paper is missing, the binomial tail all but vanishes and
 *   nearly every rectangle with k > n p passes at rect_improve's first test.  Library behaviour, preserved.
 * 1 - the LSD paper's reading of both, as rounds 2-3 shipped (double quotients, (y - tailp->p.y) denominators, a step that would
 *   divide by zero taken as 0; log_gamma(n + 1)).
 * 2 - integer corners with log_gamma(n + 1): round 4's default.
 * 1 and 2 stay until a pin against a real OpenCV 3.4.4 build decides (tools/dump_opencv_reference.py dumps the per-rectangle
 * counts and NFA values such a pin compares). */
int drfe_lsd_configure_rect(drfe_ctx* ctx, int rect_mode);
/* Where drfe_lsd_extract_batch takes the decisions of cv::LineSegmentDetectorImpl::rect_improve (OpenCV 3.4 lsd.cpp: which
 * refinement candidate replaces a rectangle, whether its NFA passes): 1 (default) on the device behind the region growing
 * (k_rect_improve: the host libm's log-gamma / log tables, its own exp / log10 / pow, every comparison certified against a
 * bound on the difference to the host's value - a frame with a comparison too close to call is validated on the host), 0 on
 * the pool threads with the caller's libm (rounds 2-3).  Results are identical. */
int drfe_lsd_configure_nfa(drfe_ctx* ctx, int device_nfa);
/* counters since drfe_create: out4[0] frames through drfe_lsd_extract_batch's device path, [1] frames whose region growing
 * returned to the host, [2] frames whose NFA decisions returned to the host, [3] frames whose key-line stage (KeyLine fields,
 * the response cut's std::sort, LBD direction: k_lsd_keylines) returned to the host */
int drfe_lsd_stats(drfe_ctx* ctx, long long* out4);
/* Kernel durations of the two long batch entries, measured live: with the clock on, drfe_lsd_extract_batch and
 * drfe_planes_ahc_post_batch bracket every kernel of their first chunk with HIP events on the stream it is launched on.
 * drfe_long_kernel_ms: the last clocked call's intervals in ms - lines [0..6]: upload, image passes, k_lsd_keys, k_lsd_order,
 * k_lsd_grow(_mw), k_rect_improve, k_lsd_keylines + k_lbd; planes [8..14]: upload, k_ahc_blocks, k_ahc_cluster, k_ahc_refine,
 * k_ahc_labels_*, k_voxel_grid, k_plane_refit ([7], [15] unused).  An interval contains whatever its kernel waited for: it is the
 * kernel's duration when the call runs alone on the device (bench.py full_frontend's solo steps). */
int drfe_long_kernel_clock(drfe_ctx* ctx, int on);
int drfe_long_kernel_ms(drfe_ctx* ctx, float* out16);
/* drfe_lsd_segments_host with rect_nfa's reading chosen by the caller (drfe_lsd_segments_host: 0). */
int drfe_lsd_segments_host_mode(const double* modgrad, const double* angles, const float* cs, int W, int H, double max_grad,
                                int rect_mode, float* segs, int cap, int* n_segs);
/* Parity taps of the device image passes of the last drfe_lsd_extract call (any pointer may be NULL):
 * 0.8-scaled image, gradient magnitude and level-line angle (sw x sh), Sobel dx/dy of the LBD image. */
int drfe_lsd_stages(drfe_ctx* ctx, uint8_t* scaled, double* modgrad, double* angles, int16_t* gx, int16_t* gy,
                    int* sw, int* sh);

/* Frame::isLineGood, src/Frame.cc:481-558, with the 3-D line lifting of src/LineExtractor.cpp:1157-1470: up to 51
 * nearest-pixel depth samples per key line, per-sample covariance, Mahalanobis RANSAC driven by rand(), refit,
 * accept if inliers / length > 0.4 and |A - B| > 0.02.  Host code (40 lines x 51 samples of double arithmetic), no
 * context needed.  depth: the CV_32F depth image in metres; K: the nine floats of mK exactly as the reference passes
 * them.  k_as_f64 = 0 reproduces the shipped behaviour — compPt3dCov reads that CV_32F matrix with at<double>, the
 * focal length becomes a subnormal, every covariance is NaN and NO line is accepted (depth_line = -1, lines3d = 0
 * for all) — by doing the same arithmetic on the same bytes; k_as_f64 = 1 runs the algorithm with f = fx as it was
 * meant (agrees with an OpenCV build to rounding, endpoint order not pinned; see lines_3d.cpp).  seed: srand() state
 * (1 = a fresh process).  Outputs per line: mvDepthLine, mvLines3D (A xyz, B xyz), optional inlier counts. */
int drfe_lines_is_good(const drfe_keyline* lines, int n, const float* depth, int w, int h, size_t stride,
                       const float* K, int k_as_f64, float cx, float cy, float invfx, float invfy, uint32_t seed,
                       float* depth_line, double* lines3d, int32_t* n_inliers, int* n_good);

/* LSDmatcher::SearchByProjection, src/LSDmatcher.cpp:20-139 (Frame, Frame) and :141-211 (Frame,
 * vector<MapLine*>), with Frame::GetLinesInArea (src/Frame.cc:781-813): project the 3-D end points with
 * Tcw (float cv::Mat path), collect the current key lines whose midpoint lies within the radius and
 * whose slope difference passes, best / second-best LBD Hamming distance in index order, TH_HIGH = 100
 * and the same-octave ratio test, sequential claims (a line already holding a MapLine with
 * Observations() > 0 is skipped).  One wavefront replays the loop on the device. */
typedef struct drfe_map_line {      /* read through LastFrame.mvpMapLines[i] / mvKeylinesUn[i] */
    int32_t valid;                  /* pML && !pML->isBad() && !mvbLineOutlier[i] */
    int32_t octave;                 /* LastFrame.mvKeylinesUn[i].octave */
    int32_t obs_positive;           /* pML->Observations() > 0 */
    int32_t pad;
    double world[6];                /* MapLine::GetWorldPos(): start xyz, end xyz */
    uint8_t desc[32];
} drfe_map_line;
typedef struct drfe_tracked_line {  /* fields Frame::isInFrustum(MapLine*) leaves on the map line */
    int32_t in_view;                /* pML && !isBad() && mbTrackInView */
    int32_t level;                  /* mnTrackScaleLevel */
    int32_t obs_positive;
    float x1, y1, x2, y2;           /* mTrackProjX1 / Y1 / X2 / Y2 */
    float view_cos;
    uint8_t desc[32];
} drfe_tracked_line;
/* cur_lines / cur_desc: the current frame's key lines (pt, angle, octave are read) and LBD rows.
 * cur_ml in/out, n_cur entries: -1 = no MapLine, >= 0 = holds one (cur_obs[i] = its Observations() > 0,
 * NULL = all); a new match writes the index of the matched last/tracked record.  Synchronous. */
int drfe_lsd_search_by_projection_last(drfe_ctx* ctx, const float* Tcw_cur, const float* Tcw_last,
                                       const drfe_camera* cam, const drfe_map_line* last_lines, int n_last,
                                       const drfe_keyline* cur_lines, const uint8_t* cur_desc, int n_cur, float th,
                                       int mono, float nnratio, const uint8_t* cur_obs, int32_t* cur_ml, int* nmatches);
int drfe_lsd_search_by_projection_map(drfe_ctx* ctx, const drfe_tracked_line* lines, int n,
                                      const drfe_keyline* cur_lines, const uint8_t* cur_desc, int n_cur, float th,
                                      float nnratio, const uint8_t* cur_obs, int32_t* cur_ml, int* nmatches);

/* Frame::isInFrustum, src/Frame.cc:602-657 (MapPoint*) and :659-727 (MapLine*), the step Tracking::SearchLocalPoints
 * runs over the local map before SearchByProjection(F, MapPoints): project with Tcw, image bounds, scale-invariance
 * distance band (0.8 * min, 1.2 * max), viewing-angle cosine against the mean normal, MapPoint::PredictScale
 * (ceil(log(max / dist) / log scaleFactor), clamped to the pyramid; MapLine::PredictScale does not clamp).  One thread
 * per map point.  log() is the shared routine drfe_logf (include/drfe_math.h) because glibc's logf is not pinned.
 * out: only the tracking fields are written (track_in_view / in_view, level, projections, view_cos): fill bad /
 * obs_positive / desc yourself and hand the array to drfe_search_by_projection_map / drfe_lsd_search_by_projection_map. */
typedef struct drfe_frustum_point { float world[3], normal[3], min_distance, max_distance; } drfe_frustum_point;
typedef struct drfe_frustum_line { double world[6], normal[3]; float min_distance, max_distance; } drfe_frustum_line;
int drfe_frame_is_in_frustum(drfe_ctx* ctx, const float* Tcw, const drfe_camera* cam, const drfe_frustum_point* pts, int n,
                             float viewing_cos_limit, drfe_tracked_point* out);
int drfe_frame_is_in_frustum_lines(drfe_ctx* ctx, const float* Tcw, const drfe_camera* cam, const drfe_frustum_line* lines,
                                   int n, float viewing_cos_limit, drfe_tracked_line* out);

/* ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th), src/ORBmatcher.cc:829-985 (LocalMapping::SearchInNeighbors): the
 * search.  The keyframe is a slot with extract + glue done; per map point (skip[i] = !pMP || isBad() ||
 * IsInKeyFrame(pKF), may be NULL): projection with Tcw, KeyFrame::IsInImage, distance band, 60-degree viewing cone,
 * PredictScale, KeyFrame::GetFeaturesInArea(u, v, th * scale[level]), octave in [level-1, level], chi-square
 * reprojection gate (7.8 with a right coordinate, 5.99 without), first minimum of the Hamming distance.
 * best_idx[i] = keyframe keypoint or -1, best_dist[i] = its distance (256 if none).  Applying the result
 * (bestDist <= TH_LOW: Replace / AddObservation / AddMapPoint) mutates the map graph and is the caller's loop. */
int drfe_fuse_search(drfe_ctx* ctx, int slot, const float* Tcw, const drfe_frustum_point* pts, const uint8_t* descs,
                     const uint8_t* skip, int n, float th, int32_t* best_idx, int32_t* best_dist);
/* ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, points, th, vpReplacePoint), src/ORBmatcher.cc:981-1107 (LoopClosing::
 * SearchAndFuse): the same search under a similarity pose — Scw is decomposed as the reference does (:989-993), invz is
 * 1.0 / z evaluated in double, and there is no chi-square gate. */
int drfe_fuse_search_sim3(drfe_ctx* ctx, int slot, const float* Scw, const drfe_frustum_point* pts, const uint8_t* descs,
                          const uint8_t* skip, int n, float th, int32_t* best_idx, int32_t* best_dist);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1106-1330 (LoopClosing::ComputeSim3):
 * both keyframes are slots with extract + glue done; pts / descs / skip are per KEYPOINT of the keyframe (skip[i] = no good
 * map point there or already matched, :1131-1142).  Every remaining map point of KF1 is carried into KF2 by
 * (sR21, t21) — and those of KF2 into KF1 by (sR12, t12) — and searched in a th * scale[level] window (octave
 * level-1..level, distance <= TH_HIGH, first minimum) on the device; a pair is kept when both directions agree.
 * matches12[i1] = keypoint of KF2 or -1 (vpMatches12[i1] = vpMapPoints2[matches12[i1]]); *n_found = the return value. */
int drfe_search_by_sim3(drfe_ctx* ctx, int slot1, int slot2, const float* T1w, const float* T2w, float s12, const float* R12,
                        const float* t12, const drfe_frustum_point* pts1, const uint8_t* descs1, const uint8_t* skip1, int n1,
                        const drfe_frustum_point* pts2, const uint8_t* descs2, const uint8_t* skip2, int n2, float th,
                        int32_t* matches12, int* n_found);

/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched,
 * int th), src/ORBmatcher.cc:294-407 (LoopClosing::ComputeSim3, th = 10): the n candidate map points are projected with
 * the similarity and matched, IN ORDER, to the slot's keypoints that have no match yet: matched[k] != 0 means
 * vpMatched[k] != NULL on entry (n_kp = the slot's keypoint count), skip[i] != 0 means the point is bad or already in
 * vpMatched (:321).  new_match[k] = the point this call assigned to keypoint k (vpMatched[k] = vpPoints[new_match[k]]) or
 * -1; *n_matches = the return value.  Windows, octave gate and Hamming distances run on the device, the first-come
 * claim order is replayed on the host from the device's per-point candidate lists. */
int drfe_search_by_projection_kf(drfe_ctx* ctx, int slot, const float* Scw, const drfe_frustum_point* pts, const uint8_t* descs,
                                 const uint8_t* skip, int n, const uint8_t* matched, int n_kp, float th, int32_t* new_match,
                                 int* n_matches);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th,
 * const int ORBdist), src/ORBmatcher.cc:1537-1664 (Tracking::Relocalization, src/Tracking.cc:3638 / :3651): the keyframe's
 * map points (pts / descs per entry of pKF->GetMapPointMatches(), skip[i] = !pMP || isBad() || sAlreadyFound.count(pMP),
 * kf_angles[i] = pKF->mvKeysUn[i].angle) are projected into the current frame (slot) with Tcw = CurrentFrame.mTcw — no
 * depth test, double 1/z, inclusive image bounds, 0.8 / 1.2 distance band, MapPoint::PredictScale — and matched IN ORDER
 * to the frame's keypoints without a map point (matched[k] = CurrentFrame.mvpMapPoints[k] != NULL), octaves
 * level-1..level+1, first minimum, accepted when <= orb_dist; then the rotation-histogram filter when check_orientation.
 * new_match[k] = index i of the point assigned to keypoint k (CurrentFrame.mvpMapPoints[k] = vpMPs[i]) or -1;
 * *n_matches = the return value.  A point with camera-space z == 0 is skipped (the reference projects it to infinity). */
int drfe_search_by_projection_reloc(drfe_ctx* ctx, int slot, const float* Tcw, const drfe_frustum_point* pts, const uint8_t* descs,
                                    const float* kf_angles, const uint8_t* skip, int n, const uint8_t* matched, int n_kp, float th,
                                    int orb_dist, int check_orientation, int32_t* new_match, int* n_matches);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:409-524; monocular
 * initialisation).  F1 / F2 = slots of the last extracted batch (frame glue done).  prev_matched[n1 * 2] = vbPrevMatched (x, y per
 * F1 keypoint), updated in place as the reference does; matches12[n1] = index into F2's keypoints or -1; *n_matches = the return
 * value.  nnratio / check_orientation = the ORBmatcher constructor's arguments.  A window holding more than 256 level-0
 * keypoints fails with DRFE_ERR_CAPACITY. */
int drfe_search_for_initialization(drfe_ctx* ctx, int slot1, int slot2, float* prev_matched, int n1, int window_size, float nnratio,
                                   int check_orientation, int32_t* matches12, int* n_matches);

/* LSDmatcher::Fuse(KeyFrame* pKF, const vector<MapLine*>& vpMapLines, const float th = 3.0), src/LSDmatcher.cpp:884-1010
 * (LocalMapping::SearchInNeighbors, src/LocalMapping.cc:1103 / :1124): the search per map line — both end points projected
 * with the keyframe pose (camera centre as KeyFrame::SetPose builds it), image bounds, distance band, 60-degree cone,
 * MapLine::PredictScale, KeyFrame::GetLinesInArea (src/KeyFrame.cc:749-781) over the keyframe's key lines, octave window
 * level-1..level, first minimum of the LBD Hamming distance.  skip[i] = !pML || pML->isBad().  best_idx[i] = key line
 * or -1; -2 when the predicted level is outside the pyramid (MapLine::PredictScale does not clamp and the reference then
 * reads mvScaleFactors out of bounds: undefined there, reported here).  best_dist[i] = the distance (INT_MAX when none);
 * the caller applies `<= TH_LOW` (50) and the Replace / AddObservation / AddMapLine surgery (:993-1008). */
int drfe_lsd_fuse_search(drfe_ctx* ctx, const float* Tcw, const drfe_camera* cam, const drfe_frustum_line* lines, const uint8_t* descs,
                         const uint8_t* skip, int n, const drfe_keyline* kf_lines, const uint8_t* kf_desc, int n_kf, float th,
                         int32_t* best_idx, int32_t* best_dist);
/* The search of LSDmatcher::Fuse(pKF, Scw, vpLines, th, vpReplaceLine) (src/LSDmatcher.cpp:750-882; public API without a caller in
 * the reference): as drfe_lsd_fuse_search with the pose taken from the similarity Scw (4x4 row-major, decomposed as :759-763).
 * skip[i] = !pML || isBad() || spAlreadyFound.count(pML).  The caller applies TH_LOW and the Replace / AddObservation surgery. */
int drfe_lsd_fuse_search_sim3(drfe_ctx* ctx, const float* Scw, const drfe_camera* cam, const drfe_frustum_line* lines, const uint8_t* descs,
                              const uint8_t* skip, int n, const drfe_keyline* kf_lines, const uint8_t* kf_desc, int n_kf, float th,
                              int32_t* best_idx, int32_t* best_dist);
/* LSDmatcher::SearchByProjection(pKF, Scw, vpLines, vpMatched, th) (src/LSDmatcher.cpp:377-502).  matched[n_kf] = vpMatched[idx] !=
 * NULL on entry; new_match[n_kf] = the map line that claimed key line idx in this call or -1 (the caller sets vpMatched[idx] =
 * vpLines[new_match[idx]]); *n_matches = the return value.  First come, first served over the map lines, as the reference. */
int drfe_lsd_search_by_projection_kf(drfe_ctx* ctx, const float* Scw, const drfe_camera* cam, const drfe_frustum_line* lines,
                                     const uint8_t* descs, const uint8_t* skip, int n, const drfe_keyline* kf_lines, const uint8_t* kf_desc,
                                     int n_kf, const uint8_t* matched, int th, int32_t* new_match, int* n_matches);
/* LSDmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/LSDmatcher.cpp:504-748).  lines / descs / skip: one
 * entry per key line of the keyframe (its map line; skip = none, bad, or already matched: :534-555), kf*_lines / kf*_desc the key
 * lines themselves; T1w / T2w = the keyframe poses (4x4 row-major), R12 3x3 row-major.  matches12[n1] = key line of KF2 or -1
 * (the caller stores vpMapLines2[matches12[i1]]); *n_found = the return value. */
int drfe_lsd_search_by_sim3(drfe_ctx* ctx, const drfe_camera* cam, const float* T1w, const float* T2w, float s12, const float* R12,
                            const float* t12, const drfe_frustum_line* lines1, const uint8_t* descs1, const uint8_t* skip1,
                            const drfe_keyline* kf1_lines, const uint8_t* kf1_desc, int n1, const drfe_frustum_line* lines2,
                            const uint8_t* descs2, const uint8_t* skip2, const drfe_keyline* kf2_lines, const uint8_t* kf2_desc, int n2,
                            float th, int32_t* matches12, int* n_found);

/* ------------------------------------------------------------------------------------------------ */
/* Multi-GPU, batched-sequence mode (no counterpart in the reference, whose only parallelism is the three
 * extractor threads of src/Frame.cc:124-134).  One process (or thread) per GPU, each with its own drfe_ctx / drfe_pipeline;
 * whole sequences are dealt to the ranks and NO collective runs on the data path.  The one exchange is the ORB vocabulary at
 * start-up: rank 0 loads it and broadcasts the node table over RCCL (xGMI inside a node); every rank then calls
 * drfe_voc_upload on its own context.  RCCL is loaded at run time (librccl.so.1): a single-GPU host never needs it.
 *   rank 0:     drfe_shard_unique_id(id)  -> hand the 128 bytes to the other ranks (file, pipe, MPI, environment ...)
 *   every rank: drfe_shard_create(id, nranks, rank, device, &sh)
 *               drfe_shard_broadcast(sh, &header, sizeof header, 0); drfe_shard_broadcast(sh, parent, 4 * n_nodes, 0); ... desc, weight, is_leaf
 *               drfe_voc_upload(ctx, k, L, scoring, weighting, n_nodes, parent, desc, weight, is_leaf)
 *               ... its sequences (drfe_shard_sequences_of_rank) through drfe_pipeline_submit ...
 *               drfe_shard_reduce_report(sh, &seconds, 1, &frames, 1)      MAX of the times, SUM of the frames: whole-job frames/s
 *               drfe_shard_destroy(sh) */
typedef struct drfe_shard drfe_shard;
#define DRFE_SHARD_ID_BYTES 128
int drfe_shard_unique_id(uint8_t* id /* DRFE_SHARD_ID_BYTES */);
int drfe_shard_create(const uint8_t* id, int nranks, int rank, int device, drfe_shard** out);
void drfe_shard_destroy(drfe_shard* shard);
/* a HOST buffer from rank `root` to every rank (staged through device memory, one ncclBroadcast); collective: every rank calls it */
int drfe_shard_broadcast(drfe_shard* shard, void* buf, size_t bytes, int root);
/* in place on every rank: element-wise MAX over the ranks of n_max doubles, SUM of n_sum 64-bit counters; collective */
int drfe_shard_reduce_report(drfe_shard* shard, double* max_inout, int n_max, long long* sum_inout, int n_sum);
/* sequences dealt round robin (sequence i -> rank i % nranks): writes up to cap indices of rank's share, returns their number */
int drfe_shard_sequences_of_rank(int n_sequences, int nranks, int rank, int* out, int cap);
const char* drfe_shard_last_error(const drfe_shard* shard /* NULL: the calling thread's last creation error */);

/* ------------------------------------------------------------------------------------------------ */
/* Bag of words (replaces the DBoW2 tree descent of Frame::ComputeBoW, src/Frame.cc:828-833, and
 * ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...), src/ORBmatcher.cc:160-292)                        */

/* Upload a vocabulary in the node format of TemplatedVocabulary::loadFromTextFile
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424): node 0 is the root; for node i >= 1
 * parent[i], is_leaf[i], 32 descriptor bytes, weight.  Children of a node are ordered by node id and
 * word ids are assigned to leaves in node order, exactly as the loader does.  k <= 20, L <= 10. */
int drfe_voc_upload(drfe_ctx* ctx, int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                    const uint8_t* desc, const double* weight, const uint8_t* is_leaf);
/* Device part of TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup) for the
 * descriptors of slots 0..nframes-1: per feature the word id, the word weight and the node id at level
 * L-levelsup.  Asynchronous on `stream`. */
int drfe_bow_transform_batch(drfe_ctx* ctx, int levelsup, int nframes, void* stream);
/* The same for the frame in ONE slot (drfe_frame_submit / drfe_frame_load): Frame::ComputeBoW / KeyFrame::ComputeBoW.  The transform
 * is a pure function of descriptors and vocabulary: a keyframe loaded from the host gets the mFeatVec it stored at creation. */
int drfe_bow_transform_slot(drfe_ctx* ctx, int levelsup, int slot, void* stream);
/* Per-feature results of a slot.  The caller folds them into BowVector / FeatureVector (std::map) in
 * feature order — addWeight's float64 sums depend on that order (BowVector.cpp). */
int drfe_bow_download(drfe_ctx* ctx, int slot, int32_t* word, double* weight, int32_t* nid, int cap);
/* ORBmatcher(nnratio, checkOri).SearchByBoW(pKF = kf_slot, F = f_slot, vpMapPointMatches).
 * kf_mp[i] >= 0 iff keyframe keypoint i holds a non-bad MapPoint.  f_match[j] = keyframe keypoint index
 * matched to frame keypoint j, or -1; *nmatches = return value. */
int drfe_search_by_bow(drfe_ctx* ctx, int kf_slot, int f_slot, const int32_t* kf_mp, int n_kf, float nnratio,
                       int check_ori, int32_t* f_match, int n_f, int* nmatches);
/* ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12), src/ORBmatcher.cc:526-660 (LoopClosing::ComputeSim3): the keyframe-
 * keyframe overload — keypoints of BOTH keyframes need a good map point (mp >= 0) and the test is bestDist1 < TH_LOW.
 * match2[keypoint of KF2] = keypoint of KF1 or -1, i.e. vpMatches12[match2[i2]] = vpMapPoints2[i2]. */
int drfe_search_by_bow_kf(drfe_ctx* ctx, int slot1, int slot2, const int32_t* mp1, int n1, const int32_t* mp2, int n2,
                          float nnratio, int check_ori, int32_t* match2, int* nmatches);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), src/ORBmatcher.cc:661-827 (LocalMapping::
 * CreateNewMapPoints): for every vocabulary node common to both keyframes, match the keypoints of KF1 without a map
 * point against those of KF2 (distance <= TH_LOW, epipole exclusion, CheckDistEpipolarLine against F12, rotation
 * histogram).  Both slots need the glue (mvKeysUn, mvuRight) and drfe_bow_transform_batch.  mp1 / mp2: >= 0 where the
 * keyframe keypoint already has a map point.  F12: 3x3 row-major; Cw1: KF1 camera centre (GetCameraCenter); T2w: KF2
 * pose (4x4 row-major); cam2: KF2 intrinsics.  matches12[i1] = keypoint of KF2 or -1 (vMatchedPairs = the pairs with
 * a match, ascending i1). */
int drfe_search_for_triangulation(drfe_ctx* ctx, int slot1, int slot2, const int32_t* mp1, int n1, const int32_t* mp2, int n2,
                                  const float* F12, const float* Cw1, const float* T2w, const drfe_camera* cam2,
                                  int only_stereo, int check_ori, int32_t* matches12, int* nmatches);

/* ------------------------------------------------------------------------------------------------ */
/* PlaneDetection (replaces src/PlaneExtractor.cpp:7-63 + include/peac/ : the live AHC extractor)   */

/* One extracted plane: ahc::PlaneSeg fields Frame::ComputePlanes reads (src/Frame.cc:952-979:
 * plane_filter.extractedPlanes[i]->normal / center), plus mse, curvature, N, rid. */
typedef struct drfe_plane {
    double normal[3];   /* unit normal pointing towards the camera (dot(normal, center) <= 0) */
    double center[3];   /* centre of mass, metres */
    double mse, curvature;
    int32_t n_points;   /* PlaneSeg::N (points of the merged init blocks) */
    int32_t rid;        /* root block id */
} drfe_plane;

/* PlaneDetection::readDepthImage(depth16, K, depthfactor) + runPlaneDetection().
 * depth: host CV_16U image, `stride` elements per row; K4 = {fx, fy, cx, cy} (the float K entries the
 * reference promotes to double); depth_factor = the float passed as `depthfactor` (1/DepthMapFactor).
 * Outputs: planes[0..n) sorted by N descending (extractedPlanes); seg (w*h, may be NULL) = seg_output
 * (plane index + 1, 0 = none); member_offsets[n+1] / member_idx (may be NULL) = plane_vertices_ as CSR,
 * pixel indices ascending. */
int drfe_planes_ahc(drfe_ctx* ctx, const uint16_t* depth, int w, int h, size_t stride, const float* K4,
                    float depth_factor, drfe_plane* planes, int cap, int* n_planes, uint8_t* seg,
                    int32_t* member_offsets, int32_t* member_idx);
/* drfe_planes_ahc for nframes host depth images (depth + f * frame_stride elements).  Default (drfe_planes_configure_extractor
 * (ctx, 1), frames up to 640 x 480 class, nframes > 1): clustering / flood fill / re-merge run on the device, one wavefront per
 * frame (ahc_frame_kernels.hip), the n_threads host threads only fetch results; otherwise the per-frame sequential host code
 * (~3.5 ms) runs on n_threads host threads, one device lane each.  Outputs per frame f at
 * planes[f * cap], n_planes[f], seg + f * w * h, member_offsets[f * (cap + 1)], member_idx + f * w * h (the last three
 * may be NULL); identical to nframes single calls.  n_threads <= 0: 1.25 x the CPUs this process may use (affinity mask clipped by
 * the cgroup quota). */
int drfe_planes_ahc_batch(drfe_ctx* ctx, const uint16_t* depth, size_t frame_stride, int w, int h, size_t stride, int nframes,
                          const float* K4, float depth_factor, drfe_plane* planes, int cap, int* n_planes, uint8_t* seg,
                          int32_t* member_offsets, int32_t* member_idx, int n_threads);
/* Parity tap of the device stage: per 10x10 init block 17 doubles (9 sums, center, normal, mse,
 * curvature) and (enters-graph flag, N). cap = number of blocks the buffers hold. */
int drfe_planes_ahc_blocks(drfe_ctx* ctx, const uint16_t* depth, int w, int h, size_t stride, const float* K4,
                           float depth_factor, double* blocks17, int32_t* valid_n, int cap);

/* The host half of drfe_planes_ahc on caller-supplied block fits (the records drfe_planes_ahc_blocks returns), without a device:
 * graph, agglomerative clustering, block membership, flood fill, re-merge and labels.  Host code: CPU tests and profiling. */
int drfe_planes_ahc_from_blocks(const double* blocks17, const int32_t* valid_n, const uint16_t* depth, int w, int h, size_t stride,
                                const float* K4, float depth_factor, drfe_plane* planes, int cap, int* n_planes, uint8_t* seg,
                                int32_t* member_offsets, int32_t* member_idx);

/* PlaneDetection_CAPE (replaces src/PlaneExtractor.cpp:65-191 + src/CAPE/{CAPE,PlaneSeg,Histogram}.cpp;
 * its thread launch is commented out in the reference, src/Frame.cc:129, but the class is public API).
 * One entry of plane_params (CAPE PlaneSeg: normal, d, mean, MSE, score, nr_pts). */
typedef struct drfe_cape_plane {
    double normal[3];
    double mean[3];
    double d;           /* plane equation normal . p + d = 0, d > 0 */
    float mse, score;
    int32_t n_points;
    int32_t pad;
} drfe_cape_plane;

/* PlaneDetection_CAPE::readDepthImage(depth32f, K) + runPlaneDetection().  depth_m: host CV_32F depth in
 * metres (`stride` elements per row); patch = PATCH_SIZE (Plane.PATCH_SIZE: 20 or 10); cos_angle_max =
 * COS_ANGLE_MAX (cos(pi/12)); max_merge_dist = MAX_MERGE_DIST (Plane.MAX_MERGE_DIST: 50).  w and h must
 * be multiples of patch.  Outputs: planes[0..n), seg (w*h) = seg_output (plane number, 0 = none).
 * Parity taps (may be NULL): per cell 16 doubles (9 sums, mean, normal, d), 3 floats (MSE, score,
 * merge tolerance), 2 ints (planar, nr_pts). */
int drfe_planes_cape(drfe_ctx* ctx, const float* depth_m, int w, int h, size_t stride, const float* K4, int patch,
                     float cos_angle_max, float max_merge_dist, drfe_cape_plane* planes, int cap, int* n_planes,
                     uint8_t* seg, double* cells16, float* cells_mst, int32_t* cells_pn);
/* The same for nframes depth images (metres; frame f at depth_m + f * frame_stride floats): planes[f * cap ..], n_planes[f],
 * seg[f * w * h ..] (seg may be NULL).  Default (drfe_planes_configure_cape(ctx, 1), nframes > 1): end to end on the device -
 * cell fits, CAPE::process (one wavefront per frame), per-pixel refinement - from the calling thread, which uploads the frames
 * through two pinned staging buffers and copies the results out; n_threads host threads (one device lane each) only for frames
 * the device hands back and in the host mode.  Identical to nframes calls of drfe_planes_cape.  n_threads <= 0: up to 4. */
int drfe_planes_cape_batch(drfe_ctx* ctx, const float* depth_m, size_t frame_stride, int w, int h, size_t stride, int nframes,
                           const float* K4, int patch, float cos_angle_max, float max_merge_dist, drfe_cape_plane* planes, int cap,
                           int* n_planes, uint8_t* seg, int n_threads);

/* ------------------------------------------------------------------------------------------------ */
/* Plane post-processing and surface normals (replaces the PCL part of Frame::ComputePlanes / ComputePlanes_CAPE,  */
/* src/Frame.cc:949-1094, 1096-1213, and Frame::MaxPointDistanceFromPlane, src/Frame.cc:1222-1307)                 */

/* One plane after the per-plane loop of Frame::ComputePlanes (:952-1011): coef = the 4x1 float Mat pushed into
 * mvPlaneCoefficients when accepted (normal, d AFTER the RANSAC + least-squares refit of MaxPointDistanceFromPlane, sign
 * kept on the side of the extractor's d), else the extractor's (normal, d); n_voxels = coarseCloud->size(). */
typedef struct drfe_plane_post {
    float coef[4];
    int32_t accepted;   /* 1: coefficients + voxel cloud pushed (mvPlaneCoefficients / mvPlanePoints) */
    int32_t n_voxels;
} drfe_plane_post;

/* pcl::VoxelGrid (leaf x leaf x leaf) of a point list in inputCloud order: centroids ordered by leaf index.  Host code. */
int drfe_plane_voxel_grid(const float* xyz, int n, float leaf, float* out_xyz, int cap, int* n_out);
/* Frame::MaxPointDistanceFromPlane(plane, cloud) with Plane.DistanceThreshold = dist_threshold: *valid = its return value;
 * coef4 is overwritten by the refit when valid.  Host code. */
int drfe_plane_refit(float* coef4, const float* xyz, int n, double dist_threshold, int* valid);

/* The per-plane loop of Frame::ComputePlanes for the planes drfe_planes_ahc returned (same depth image, K4, depth_factor;
 * planes / member_offsets / member_idx as that call filled them).  max_point_dist = Point.MaxDistance, dist_threshold =
 * Plane.DistanceThreshold.  Outputs: post[n_planes]; voxel_offsets[n_planes + 1] + voxel_xyz (may be NULL) = mvPlanePoints
 * of the accepted planes as CSR; *n_accepted = mvPlaneCoefficients.size(); *plane_num (may be NULL) = planeDetector.plane_num_
 * after `-= fail_planes`.  Host code: ctx may be NULL (no error text then). */
int drfe_planes_ahc_postprocess(drfe_ctx* ctx, const uint16_t* depth, int w, int h, size_t stride, const float* K4, float depth_factor,
                                const drfe_plane* planes, int n_planes, const int32_t* member_offsets, const int32_t* member_idx,
                                float max_point_dist, double dist_threshold, drfe_plane_post* post, float* voxel_xyz,
                                int32_t* voxel_offsets, int cap_voxels, int* n_accepted, int* plane_num);
/* drfe_planes_ahc_batch and, on the same worker thread, drfe_planes_ahc_postprocess of every frame: the plane path of nframes host
 * depth images end to end.  planes / post: [nframes][cap]; n_planes / n_accepted / plane_num (may be NULL): [nframes]; seg (may be
 * NULL): [nframes][w*h].  The voxel clouds are not returned here.  n_threads <= 0: the CPUs this process may use. */
int drfe_planes_ahc_post_batch(drfe_ctx* ctx, const uint16_t* depth, size_t frame_stride, int w, int h, size_t stride, int nframes,
                               const float* K4, float depth_factor, float max_point_dist, double dist_threshold, drfe_plane* planes,
                               int cap, int* n_planes, uint8_t* seg, drfe_plane_post* post, int* n_accepted, int* plane_num, int n_threads);
/* Where drfe_planes_ahc_post_batch runs pcl::VoxelGrid (leaf 0.05) of a frame's planes (voxel_kernels.hip: leaf indices,
 * std::sort's permutation by the device introsort, centroid sums in that order - one workgroup per plane).  1 (default): on the
 * device behind the device extractor, which also gathers each plane's cloud (one launch for all planes of the batch; the pool's
 * threads fetch centroids and run gates + refit); the host-extractor mode keeps the host grid.  2: on the device in the
 * host-extractor mode too (one launch per frame from each pool thread).  0: always on the pool's host threads.  Results are
 * identical (tests/test_gpu_post.py). */
int drfe_planes_configure(drfe_ctx* ctx, int device_voxel_grid);
/* Where drfe_planes_ahc_post_batch runs the gates and Frame::MaxPointDistanceFromPlane (src/Frame.cc:1003-1011, 1222-1307: RANSAC with
 * pcl's mt19937 sample sequence, adaptive iteration bound, least-squares refit, sign) of every plane: 1 (default) on the device behind
 * the device voxel grids, one wavefront per plane (refit_kernels.hip) - the pool's threads then receive 24-byte post records instead of
 * voxel clouds; 0 on the pool's host threads.  A plane whose libm-dependent roundings the device cannot certify sends its frame to the
 * host's refit (drfe_planes_refit_stats counts them).  Results are identical (tests/test_gpu_post.py). */
int drfe_planes_configure_refit(drfe_ctx* ctx, int on_device);
/* out2[0] = frames whose gates + refit ran on the device since drfe_create, out2[1] = of those, sent to the host's refit */
int drfe_planes_refit_stats(drfe_ctx* ctx, long long* out2);
/* Where drfe_planes_ahc_post_batch runs PEAC's extractor after the init-block fits (graph, agglomerative clustering, block
 * membership, flood fill, re-merge, labels and member lists): 1 (default) on the device, one wavefront per frame executing the
 * reference's sequence (ahc_frame_kernels.hip), the batch's frames side by side; 0 on the pool's host threads (the path of
 * drfe_planes_ahc).  Results are identical (tests/test_gpu_post.py, tests/test_gpu_planes.py). */
int drfe_planes_configure_extractor(drfe_ctx* ctx, int on_device);
/* out4[0] = frames through the device extractor (drfe_planes_ahc_batch / drfe_planes_ahc_post_batch) since drfe_create, out4[1] = of
 * those, redone on the host (a capacity of the device path ran out, a cosine could not be certified), out4[2] = plane voxel grids
 * run on the device, out4[3] = of those, redone on the host.  The device extractor takes frames of up to 12 800 init blocks and
 * 2^21 pixels (1280 x 960, BASELINE config 5, included); larger frames run on the pool's host threads and are not counted. */
int drfe_planes_ahc_stats(drfe_ctx* ctx, long long* out4);
/* Where drfe_planes_cape_batch runs CAPE::process between the cell fits and the per-pixel refinement (src/CAPE/CAPE.cpp:81-293:
 * normal histogram + seeding, cell growing, segment fits, merging, erode / dilate masks): 1 (default) on the device, one wavefront
 * per frame (cape_frame_kernels.hip; a frame whose histogram bins the device cannot certify - acos / atan2 are the host libm's in
 * the reference - is finished by the host), 0 on the pool's host threads.  Results are identical. */
int drfe_planes_configure_cape(drfe_ctx* ctx, int on_device);
/* out2[0] = frames through drfe_planes_cape_batch's device path since drfe_create, out2[1] = of those, finished by the host */
int drfe_planes_cape_stats(drfe_ctx* ctx, long long* out2);
/* The same loop of Frame::ComputePlanes_CAPE (:1111-1141) for the planes / seg image drfe_planes_cape returned: plane_cloud[i]
 * = the points of the pixels labelled i + 1 in raster order (src/PlaneExtractor.cpp:171-188). */
int drfe_planes_cape_postprocess(drfe_ctx* ctx, const float* depth_m, int w, int h, size_t stride, const float* K4, const uint8_t* seg,
                                 const drfe_cape_plane* planes, int n_planes, float max_point_dist, double dist_threshold,
                                 drfe_plane_post* post, float* voxel_xyz, int32_t* voxel_offsets, int cap_voxels, int* n_accepted,
                                 int* plane_num);

/* SurfaceNormal (include/LSDextractor.h:34-41): cv::Point3f normal, cv::Point3f cameraPosition, cv::Point2i FramePosition */
typedef struct drfe_surface_normal {
    float normal[3];            /* NaN where PCL leaves the normal undefined (border, depth discontinuity) */
    float camera_position[3];
    int32_t frame_x, frame_y;
} drfe_surface_normal;

/* vSurfaceNormal of Frame::ComputePlanes (:1025-1090): depth_m = the CV_32F depth in metres (`stride` elements per row),
 * K4 = {fx, fy, cx, cy} (Frame's static floats).  out[(h3/2) * (w3/2)] with w3 = ceil(w/3), h3 = ceil(h/3), rows of odd m,
 * inside a row odd n, as the reference pushes them.  Parity taps (may be NULL): the organized cloud (w3*h3*3), every normal
 * (w3*h3*3), the distance map (w3*h3). */
int drfe_surface_normals(drfe_ctx* ctx, const float* depth_m, int w, int h, size_t stride, const float* K4, float max_point_dist,
                         drfe_surface_normal* out, int cap, int* n_out, float* cloud_tap, float* normals_tap, float* dist_tap);
/* The same for nframes device-resident raw depth images (CV_16U; depth = raw * depth_factor in float32, as
 * imDepth.convertTo does): asynchronous on `stream` (hipStream_t; NULL = the context stream), results stay on the device. */
int drfe_surface_normals_batch(drfe_ctx* ctx, const uint16_t* d_depth, size_t frame_stride, size_t row_stride, int w, int h,
                               const float* K4, float depth_factor, float max_point_dist, int nframes, void* stream);
/* Records of frame `slot` of the most recent drfe_surface_normals_batch (synchronises). */
int drfe_surface_normals_download(drfe_ctx* ctx, int slot, drfe_surface_normal* out, int cap, int* n_out);

/* ------------------------------------------------------------------------------------------------ */
/* measurement                                                                                       */
/* DRFE_STAGE_FAST = the first FAST launch (k_fast_cells_cols<8>: the cells of at most 8 rows per lane - the four large levels at
 * 640x480 - or the generic k_fast_cells over all cells); DRFE_STAGE_FAST_B = the second one (k_fast_cells_cols<12>, the rest). */
enum { DRFE_STAGE_PYRAMID = 0, DRFE_STAGE_FAST, DRFE_STAGE_QUADTREE, DRFE_STAGE_BLUR, DRFE_STAGE_DESC,
       DRFE_STAGE_GLUE, DRFE_STAGE_MATCH, DRFE_STAGE_FAST_B, DRFE_STAGE_COUNT };
/* How the FAST cells of a w x h frame split over the two launches: cells[2] and evaluated pixels[2] (sum of the cells'
 * (window - 6)^2 areas: every pixel of the detection region belongs to exactly one cell).  Host code. */
int drfe_orb_fast_partition(drfe_ctx* ctx, int w, int h, int32_t* cells, int64_t* pixels);
/* When enabled, batched calls bracket every stage with HIP events on the launch stream. */
int drfe_profile_enable(drfe_ctx* ctx, int on);
/* Milliseconds per stage of the most recent batched calls (synchronises). */
int drfe_profile_stage_ms(drfe_ctx* ctx, float* ms /* DRFE_STAGE_COUNT */);
int drfe_stream_sync(drfe_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* DRFE_H */
