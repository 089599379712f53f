/* match_kernels.hip — gfx950 kernels of the Frame glue and the windowed Hamming matchers
 *  and the brute-force matcher (a-11/a-14).
 *
 *   k_stereo            Frame::ComputeStereoFromRGBD                  (reference src/Frame.cc:893-911)
 *   k_grid              Frame::AssignFeaturesToGrid / PosInGrid       (:224-237, 816-825) -> CSR
 *   k_mappoints_last    Frame::UnprojectStereo of the last frame      (:913-923)
 *   k_queries_last      projection part of SearchByProjection(F,F)    (src/ORBmatcher.cc:1417-1455)
 *   k_window_candidates Frame::GetFeaturesInArea + DescriptorDistance (src/Frame.cc:730-779, ORBmatcher.cc:1712)
 *   k_resolve_last      claim / best / rotation histogram             (src/ORBmatcher.cc:1459-1531)
 *   k_resolve_map       claim / best+second / ratio                   (:75-127)
 *   k_bf_knn            cv::BFMatcher(NORM_HAMMING) 1-NN / 2-NN
 *
 * The reference's matchers are sequential in the map-point index (a keypoint claimed by an earlier
 * map point is skipped by later ones).  The split used here: an order-free, fully parallel kernel
 * computes for every query the Hamming distance to every keypoint of its search window, tagged with
 * the position the reference's cell scan would visit it at; a one-wavefront-per-frame kernel then
 * replays the claims in map-point order, picking min(distance, visit position) among unclaimed
 * candidates — the same winner as the reference's strict `dist < bestDist` scan.
 */
#include "drfe_internal.h"
#include "match_internal.h"
#include "../../include/drfe_math.h"

#define WAVE 64

/* ------------------------------------------------------------------------------------------------ */
/* Frame glue                                                                                        */

/* Frame::UndistortKeyPoints, src/Frame.cc:835-860: mvKeysUn[i] = mvKeys[i] with pt replaced by
 * cv::undistortPoints (undistort_math.h) */
__global__ __launch_bounds__(256) void k_undistort(const drfe_keypoint* __restrict__ kps, const int* __restrict__ kpCount,
                                                   int maxKp, DrfeDistortion D, drfe_keypoint* __restrict__ kpsUn)
{
    const int slot = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kpCount[slot]) return;
    drfe_keypoint kp = kps[(size_t)slot * maxKp + i];
    float ux, uy;
    drfe_undistort_point(D, kp.x, kp.y, &ux, &uy);
    kp.x = ux; kp.y = uy;
    kpsUn[(size_t)slot * maxKp + i] = kp;
}

__global__ __launch_bounds__(256) void k_stereo(const drfe_keypoint* __restrict__ kps, const drfe_keypoint* __restrict__ kpsUn,
                                                const int* __restrict__ kpCount,
                                                int maxKp, const uint16_t* __restrict__ depth, size_t frameStride,
                                                size_t rowStride, int w, int h, drfe_camera cam,
                                                float* __restrict__ uRight, float* __restrict__ zDepth)
{
    const int slot = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kpCount[slot]) return;
    const drfe_keypoint kp = kps[(size_t)slot * maxKp + i];
    /* imDepth.at<float>(v,u) with float indices: truncation; depth = raw * factor in
     * float32 (imDepth.convertTo(CV_32F, factor), src/Frame.cc:113-115) */
    const int v = (int)kp.y, u = (int)kp.x;
    float d = 0.f;
    if (u >= 0 && u < w && v >= 0 && v < h)      /* rowStride == 0: `depth` holds the raw value AT each keypoint ([slot][frameStride]) */
        d = (float)depth[rowStride ? (size_t)slot * frameStride + (size_t)v * rowStride + u : (size_t)slot * frameStride + i] * cam.depth_factor;
    float ur = -1.f, z = -1.f;
    if (d > 0) { z = d; ur = kpsUn[(size_t)slot * maxKp + i].x - cam.bf / d; }   /* kpU.pt.x - mbf/d, :906 */
    uRight[(size_t)slot * maxKp + i] = ur;
    zDepth[(size_t)slot * maxKp + i] = z;
}

__device__ __forceinline__ int grid_cell(const drfe_keypoint& kp, const drfe_camera& cam, float invW, float invH)
{
    const int px = (int)roundf((kp.x - cam.min_x) * invW);
    const int py = (int)roundf((kp.y - cam.min_y) * invH);
    if (px < 0 || px >= DRFE_GRID_COLS || py < 0 || py >= DRFE_GRID_ROWS) return -1;
    return px * DRFE_GRID_ROWS + py;   /* mGrid[px][py] */
}

/* one workgroup per slot: count -> scan -> fill -> per-cell ascending sort (== insertion order) */
__global__ __launch_bounds__(256) void k_grid(const drfe_keypoint* __restrict__ kps, const int* __restrict__ kpCount,
                                              int maxKp, drfe_camera cam, float invW, float invH,
                                              const float* __restrict__ uRight, const uint8_t* __restrict__ desc,
                                              int* __restrict__ gridOff, int* __restrict__ gridIdx,
                                              uint4* __restrict__ cellKp, uint4* __restrict__ cellDesc)
{
    __shared__ int cnt[DRFE_GRID_CELLS];
    __shared__ int off[DRFE_GRID_CELLS + 1];
    __shared__ int part[256];
    const int slot = blockIdx.x, tid = threadIdx.x;
    const int n = kpCount[slot];
    const drfe_keypoint* K = kps + (size_t)slot * maxKp;
    int* gIdx = gridIdx + (size_t)slot * maxKp;
    int* gOff = gridOff + (size_t)slot * (DRFE_GRID_CELLS + 1);
    for (int c = tid; c < DRFE_GRID_CELLS; c += 256) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int c = grid_cell(K[i], cam, invW, invH);
        if (c >= 0) atomicAdd(&cnt[c], 1);
    }
    __syncthreads();
    const int per = DRFE_GRID_CELLS / 256; /* 12 */
    int s = 0;
    for (int k = 0; k < per; k++) s += cnt[tid * per + k];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = (tid >= o) ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int k = 0; k < per; k++) { off[tid * per + k] = run; run += cnt[tid * per + k]; }
    if (tid == 255) off[DRFE_GRID_CELLS] = run;
    __syncthreads();
    for (int c = tid; c <= DRFE_GRID_CELLS; c += 256) gOff[c] = off[c];
    for (int c = tid; c < DRFE_GRID_CELLS; c += 256) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int c = grid_cell(K[i], cam, invW, invH);
        if (c >= 0) gIdx[off[c] + atomicAdd(&cnt[c], 1)] = i;
    }
    __syncthreads();
    for (int c = tid; c < DRFE_GRID_CELLS; c += 256) {
        const int b = off[c], e = off[c + 1];
        for (int a = b + 1; a < e; a++) {
            const int v = gIdx[a];
            int j = a - 1;
            while (j >= b && gIdx[j] > v) { gIdx[j + 1] = gIdx[j]; j--; }
            gIdx[j + 1] = v;
        }
    }
    __syncthreads();
    /* the same keypoints once more in cell order, packed for the window search: a window column is one
     * contiguous run of (x, y, uRight, index | octave << 24) records and of descriptors */
    const int nIn = off[DRFE_GRID_CELLS];
    uint4* cK = cellKp + (size_t)slot * maxKp;
    uint4* cD = cellDesc + (size_t)slot * maxKp * 2;
    const uint4* D4 = reinterpret_cast<const uint4*>(desc + (size_t)slot * maxKp * 32);
    const float* UR = uRight + (size_t)slot * maxKp;
    for (int p = tid; p < nIn; p += 256) {
        const int idx = gIdx[p];
        const drfe_keypoint kp = K[idx];
        cK[p] = make_uint4(__float_as_uint(kp.x), __float_as_uint(kp.y), __float_as_uint(UR[idx]),
                           (uint32_t)idx | ((uint32_t)kp.octave << 24));
        cD[2 * p] = D4[2 * idx];
        cD[2 * p + 1] = D4[2 * idx + 1];
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* queries for SearchByProjection(CurrentFrame, LastFrame)                                           */

/* (3x3)*(3x1)+(3x1) in float32 exactly as OpenCV's small-matrix gemm path evaluates
 * `Rcw*x3Dw+tcw`: float dot product left to right, then one add (SURVEY.md §10 / oracle). */
__device__ __forceinline__ void mat3_mul_add(const float* T /* 4x4 row-major */, const float x[3], float out[3])
{
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float d = T[r * 4 + 0] * x[0] + T[r * 4 + 1] * x[1] + T[r * 4 + 2] * x[2];
        out[r] = d + T[r * 4 + 3];
    }
}

/* map points of the last frame = its keypoints with depth, unprojected with Twc (UnprojectStereo) */
__global__ __launch_bounds__(256) void k_mappoints_last(const drfe_keypoint* __restrict__ kps,
                                                        const uint8_t* __restrict__ desc,
                                                        const int* __restrict__ kpCount, int maxKp,
                                                        const float* __restrict__ zDepth, drfe_camera cam,
                                                        const float* __restrict__ Twc /* [slot][16] */,
                                                        drfe_map_point* __restrict__ mps)
{
    const int slot = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kpCount[slot]) return;
    const size_t o = (size_t)slot * maxKp + i;
    drfe_map_point mp;
    const float z = zDepth[o];
    mp.valid = z > 0 ? 1 : 0;
    mp.obs_positive = 1;
    mp.pad[0] = mp.pad[1] = 0;
    mp.world[0] = mp.world[1] = mp.world[2] = 0.f;
    if (z > 0) {
        const float invfx = 1.0f / cam.fx, invfy = 1.0f / cam.fy;
        const drfe_keypoint kp = kps[o];
        float p[3];
        p[0] = (kp.x - cam.cx) * z * invfx;
        p[1] = (kp.y - cam.cy) * z * invfy;
        p[2] = z;
        mat3_mul_add(Twc + (size_t)slot * 16, p, mp.world);
    }
    const uint32_t* d = reinterpret_cast<const uint32_t*>(desc + o * 32);
    uint32_t* md = reinterpret_cast<uint32_t*>(mp.desc);
#pragma unroll
    for (int k = 0; k < 8; k++) md[k] = d[k];
    mps[o] = mp;
}

/* one thread per last-frame map point: project into the current frame, emit the window query */
__global__ __launch_bounds__(256) void k_queries_last(const MatchPair* __restrict__ pairs,
                                                      const drfe_keypoint* __restrict__ kps,
                                                      const int* __restrict__ kpCount, int maxKp,
                                                      const drfe_map_point* __restrict__ mps, drfe_camera cam,
                                                      const float* __restrict__ scaleFactors, float th,
                                                      MatchQuery* __restrict__ queries)
{
    const MatchPair P = pairs[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nLast = P.mpSlot >= 0 ? kpCount[P.lastSlot] : P.nQueries;
    if (i >= nLast) return;
    MatchQuery q;
    q.valid = 0;
    q.u = q.v = q.radius = q.ur = q.thrR = 0.f;
    q.minLevel = q.maxLevel = -1;
    q.obs = 0;
    const drfe_map_point mp = mps[(size_t)P.mpBase + i];
#pragma unroll
    for (int k = 0; k < 8; k++) q.desc[k] = reinterpret_cast<const uint32_t*>(mp.desc)[k];
    q.obs = mp.obs_positive;
    if (mp.valid) {
        float xc[3];
        mat3_mul_add(P.Tcw, mp.world, xc);
        const float invzc = (float)(1.0 / (double)xc[2]);
        if (!(invzc < 0)) {
            const float u = cam.fx * xc[0] * invzc + cam.cx;
            const float v = cam.fy * xc[1] * invzc + cam.cy;
            if (!(u < cam.min_x || u > cam.max_x) && !(v < cam.min_y || v > cam.max_y)) {
                const int oct = kps[(size_t)P.lastSlot * maxKp + i].octave;
                const float radius = th * scaleFactors[oct];
                q.u = u; q.v = v; q.radius = radius; q.thrR = radius;
                q.ur = u - cam.bf * invzc;
                if (P.forward) { q.minLevel = oct; q.maxLevel = -1; }
                else if (P.backward) { q.minLevel = 0; q.maxLevel = oct; }
                else { q.minLevel = oct - 1; q.maxLevel = oct + 1; }
                q.valid = 1;
            }
        }
    }
    queries[(size_t)P.queryBase + i] = q;
}

/* ------------------------------------------------------------------------------------------------ */
/* window gather + Hamming: a quarter wavefront per query, a whole one for the rare wide window       */

/* Frame::GetFeaturesInArea cell range, src/Frame.cc:735-749; false when the window misses the grid */
__device__ __forceinline__ bool window_cells(const MatchQuery& q, const drfe_camera& cam, float invW, float invH, int& minX,
                                             int& maxX, int& minY, int& maxY)
{
    const float x = q.u, y = q.v, r = q.radius;
    minX = max(0, (int)floorf((x - cam.min_x - r) * invW));
    maxX = min(DRFE_GRID_COLS - 1, (int)ceilf((x - cam.min_x + r) * invW));
    minY = max(0, (int)floorf((y - cam.min_y - r) * invH));
    maxY = min(DRFE_GRID_ROWS - 1, (int)ceilf((y - cam.min_y + r) * invH));
    return !(minX >= DRFE_GRID_COLS || maxX < 0 || minY >= DRFE_GRID_ROWS || maxY < 0);
}

/* one candidate record against one query: the level, window and stereo gates of the reference's inner loop, then
 * the Hamming distance; returns false when the record is skipped */
__device__ __forceinline__ bool window_candidate(const MatchQuery& q, bool bCheckLevels, const uint4* __restrict__ cK,
                                                 const uint4* __restrict__ cD, int p, int sq, uint64_t q0, uint64_t q1,
                                                 uint64_t q2, uint64_t q3, uint32_t& key, uint32_t& id)
{
    const uint4 k4 = cK[p];
    const uint4 da = cD[2 * p], db = cD[2 * p + 1];
    const float kx = __uint_as_float(k4.x), ky = __uint_as_float(k4.y), ur = __uint_as_float(k4.z);
    const int oct = (int)(k4.w >> 24);
    bool ok = true;
    if (bCheckLevels) {
        if (oct < q.minLevel) ok = false;
        if (q.maxLevel >= 0 && oct > q.maxLevel) ok = false;
    }
    const float dx = kx - q.u, dy = ky - q.v;
    if (!(fabsf(dx) < q.radius && fabsf(dy) < q.radius)) ok = false;
    if (ur > 0) {
        const float er = fabsf(q.ur - ur);
        if (er > q.thrR) ok = false;
    }
    if (!ok) return false;
    const int dist = __popcll(q0 ^ ((uint64_t)da.x | ((uint64_t)da.y << 32))) + __popcll(q1 ^ ((uint64_t)da.z | ((uint64_t)da.w << 32))) +
                     __popcll(q2 ^ ((uint64_t)db.x | ((uint64_t)db.y << 32))) + __popcll(q3 ^ ((uint64_t)db.z | ((uint64_t)db.w << 32)));
    key = ((uint32_t)dist << 22) | (uint32_t)min(sq, (1 << 22) - 1);
    id = k4.w;
    return true;
}

/* A whole wavefront on one query (windows wider than 16 grid columns or with more than WQ_TAB records).
 * The reference scans cells ix-outer / iy-inner and each cell in insertion order: in the cell-sorted arrays that is,
 * per window column, ONE contiguous run [gOff[ix][minY], gOff[ix][maxY+1]).  Lane c owns column c (<= 64 columns); a
 * wave scan turns the run lengths into visit positions, and from then on lanes work on candidates, not cells. */
__device__ void window_query_wave(const MatchQuery& q, size_t qo, int lane, int nMinCellX, int nMaxCellX, int nMinCellY, int nMaxCellY,
                                  const int* __restrict__ gOff, const uint4* __restrict__ cK, const uint4* __restrict__ cD,
                                  uint32_t* __restrict__ candIdx, uint32_t* __restrict__ candKey, int* __restrict__ candCnt,
                                  uint2* __restrict__ candBest, int* __restrict__ status)
{
    const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
    uint32_t* oIdx = candIdx + qo * DRFE_MATCH_MAX_CAND;
    uint32_t* oKey = candKey + qo * DRFE_MATCH_MAX_CAND;
    const uint64_t q0 = (uint64_t)q.desc[0] | ((uint64_t)q.desc[1] << 32), q1 = (uint64_t)q.desc[2] | ((uint64_t)q.desc[3] << 32),
                   q2 = (uint64_t)q.desc[4] | ((uint64_t)q.desc[5] << 32), q3 = (uint64_t)q.desc[6] | ((uint64_t)q.desc[7] << 32);
    const int ncol = nMaxCellX - nMinCellX + 1;
    int runB = 0, runN = 0;
    if (lane < ncol) {
        const int cb = (nMinCellX + lane) * DRFE_GRID_ROWS;
        runB = gOff[cb + nMinCellY];
        runN = gOff[cb + nMaxCellY + 1] - runB;
    }
    const int incl = drfe_wave_incl_scan(runN, lane);
    const int T = __builtin_amdgcn_readlane(incl, 63);
    int nOut = 0;
    bool overflow = false;
    uint32_t myKey = 0xFFFFFFFFu, myIdx = 0;
    for (int s0 = 0; s0 < T; s0 += WAVE) {
        const int sq = s0 + lane;                /* visit position of this lane's candidate */
        bool pass = false;
        uint32_t key = 0xFFFFFFFFu, id = 0;
        /* record position = runB[col] + (sq - exclusive[col]); the column walk uses wave-uniform lane reads */
        int base = __builtin_amdgcn_readfirstlane(runB);
        for (int j = 0; j + 1 < ncol; j++) {
            const int inclJ = __builtin_amdgcn_readlane(incl, j), nextB = __builtin_amdgcn_readlane(runB, j + 1);
            if (inclJ <= sq) base = nextB - inclJ;
        }
        if (sq < T) pass = window_candidate(q, bCheckLevels, cK, cD, base + sq, sq, q0, q1, q2, q3, key, id);
        const unsigned long long m = __ballot(pass);
        const int pos = nOut + __popcll(m & ((1ull << lane) - 1ull));
        if (pass) {
            if (pos < DRFE_MATCH_MAX_CAND) {
                oIdx[pos] = id;
                oKey[pos] = key;
                if (key < myKey) { myKey = key; myIdx = id; }
            } else overflow = true;
        }
        nOut += __popcll(m);
    }
    if (__any(overflow) && lane == 0) atomicOr(status, 4);
    /* the query's overall best (min distance, then earliest visit): what the claim replay takes when
     * nobody claimed it first */
    const uint32_t mn = drfe_wave_min_u32(myKey);
    const unsigned long long who = __ballot(myKey == mn);
    const uint32_t bIdx = (uint32_t)__builtin_amdgcn_readlane((int)myIdx, __ffsll((long long)who) - 1);
    if (lane == 0) {
        candCnt[qo] = min(nOut, DRFE_MATCH_MAX_CAND);
        candBest[qo] = make_uint2(mn, bIdx);
    }
}

#define WQ_TAB 64                                /* visit positions a quarter wave handles; longer windows take a whole wave */
/* Typical windows hold 5-40 records in 4-12 grid columns, so a whole wavefront per query idles most of its lanes and the
 * kernel is bound by instruction issue.  Here a wavefront takes FOUR queries, 16 lanes each: lane = column for the
 * run lengths (DPP row prefix), the column owners scatter the record positions of their runs into a 64-entry LDS table
 * in visit order, then lane = candidate, 16 at a time; compaction positions come from the query's 16-bit slice of the
 * ballot, the minimum from a DPP row reduction.  Output is identical to the whole-wave routine, which still handles the
 * queries whose window is wider than 16 columns or longer than WQ_TAB records. */
#ifndef WQ_THREADS
#define WQ_THREADS 64                 /* wavefronts of this kernel are independent of each other: one per workgroup */
#endif
#define WQ_PER_BLOCK (WQ_THREADS / WAVE * 4)
__global__ __launch_bounds__(WQ_THREADS) void k_window_candidates(const MatchPair* __restrict__ pairs,
                                                           const MatchQuery* __restrict__ queries,
                                                           const int* __restrict__ kpCount, int maxKp,
                                                           const int* __restrict__ gridOff,
                                                           const uint4* __restrict__ cellKp,
                                                           const uint4* __restrict__ cellDesc, drfe_camera cam,
                                                           float invW, float invH,
                                                           uint32_t* __restrict__ candIdx,
                                                           uint32_t* __restrict__ candKey,
                                                           int* __restrict__ candCnt, uint2* __restrict__ candBest,
                                                           int* __restrict__ status, uint32_t gxMagic)
{
    __shared__ int sTab[WQ_THREADS / WAVE][4][WQ_TAB];
    int bx, by;
    drfe_xcd_swizzle_2d(gxMagic, bx, by);        /* a frame pair's queries on one XCD: they gather the same cell-sorted records */
    const MatchPair P = pairs[by];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   /* wave-uniform for the compiler */
    const int g = lane >> 4, l = lane & 15;
    const int qbase = (bx * (WQ_THREADS / WAVE) + wv) * 4;
    const int nQ = P.mpSlot >= 0 ? kpCount[P.lastSlot] : P.nQueries;
    if (qbase >= nQ) return;                     /* wave-uniform */
    const int cur = P.curSlot;
    const int* gOff = gridOff + (size_t)cur * (DRFE_GRID_CELLS + 1);
    const uint4* cK = cellKp + (size_t)cur * maxKp;
    const uint4* cD = cellDesc + (size_t)cur * maxKp * 2;
    const bool have = qbase + g < nQ;
    const size_t qo = (size_t)P.queryBase + (have ? qbase + g : qbase);
    const MatchQuery q = queries[qo];
    int nMinCellX = 0, nMaxCellX = 0, nMinCellY = 0, nMaxCellY = 0;
    const bool inRange = have && q.valid && window_cells(q, cam, invW, invH, nMinCellX, nMaxCellX, nMinCellY, nMaxCellY);
    const int ncol = nMaxCellX - nMinCellX + 1;
    /* lane = column: run of records per window column, inclusive prefix inside the row of 16 lanes */
    int runB = 0, runN = 0;
    const bool narrow = inRange && ncol <= 16;
    if (narrow && l < ncol) {
        const int cb = (nMinCellX + l) * DRFE_GRID_ROWS;
        runB = gOff[cb + nMinCellY];
        runN = gOff[cb + nMaxCellY + 1] - runB;
    }
    int incl = runN;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);      /* row_shr:1, 0 shifted in */
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
    const int T = __shfl(incl, 15, 16);
    const bool small = narrow && T <= WQ_TAB;    /* uniform inside the group */
    /* column owners write the record position of every visit position of their run */
    {
        const int excl = incl - runN;
        const int n = small ? runN : 0;
        for (int k = 0; __any(k < n); k++)
            if (k < n) sTab[wv][g][excl + k] = runB + k;
    }
    const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
    const uint64_t q0 = (uint64_t)q.desc[0] | ((uint64_t)q.desc[1] << 32), q1 = (uint64_t)q.desc[2] | ((uint64_t)q.desc[3] << 32),
                   q2 = (uint64_t)q.desc[4] | ((uint64_t)q.desc[5] << 32), q3 = (uint64_t)q.desc[6] | ((uint64_t)q.desc[7] << 32);
    uint32_t* oIdx = candIdx + qo * DRFE_MATCH_MAX_CAND;
    uint32_t* oKey = candKey + qo * DRFE_MATCH_MAX_CAND;
    const int Tg = small ? T : 0;
    int nOut = 0;
    uint32_t myKey = 0xFFFFFFFFu, myIdx = 0;
    for (int s0 = 0; __any(s0 < Tg); s0 += 16) {
        const int sq = s0 + l;
        bool pass = false;
        uint32_t key = 0xFFFFFFFFu, id = 0;
        if (sq < Tg) pass = window_candidate(q, bCheckLevels, cK, cD, sTab[wv][g][sq], sq, q0, q1, q2, q3, key, id);
        const uint32_t m16 = (uint32_t)(__ballot(pass) >> (g * 16)) & 0xFFFFu;
        const int pos = nOut + __popc(m16 & ((1u << l) - 1u));
        if (pass) {                              /* Tg <= WQ_TAB < DRFE_MATCH_MAX_CAND: no overflow on this path */
            oIdx[pos] = id;
            oKey[pos] = key;
            if (key < myKey) { myKey = key; myIdx = id; }
        }
        nOut += __popc(m16);
    }
    /* group minimum of (key, lane): xor butterflies inside the row of 16 lanes */
    uint32_t mn = myKey;
    mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0xB1, 0xf, 0xf, false));     /* quad_perm [1,0,3,2] */
    mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0x4E, 0xf, 0xf, false));     /* quad_perm [2,3,0,1] */
    mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0x141, 0xf, 0xf, false));    /* row_half_mirror */
    mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0x140, 0xf, 0xf, false));    /* row_mirror */
    const uint32_t who16 = (uint32_t)(__ballot(myKey == mn) >> (g * 16)) & 0xFFFFu;
    const uint32_t bIdx = (uint32_t)__shfl((int)myIdx, __ffs((int)who16) - 1, 16);
    if (have && l == 0 && (small || !inRange)) {
        candCnt[qo] = nOut;                                     /* 0 for an invalid query or a window off the grid */
        candBest[qo] = small ? make_uint2(mn, bIdx) : make_uint2(0xFFFFFFFFu, 0u);
    }
    /* the wide or long windows of this quartet: the whole wavefront, one query at a time */
    unsigned long long wide = __ballot(inRange && !small && l == 0);
    while (wide) {
        const int gg = (__ffsll((long long)wide) - 1) >> 4;
        wide &= wide - 1;
        const size_t qw = (size_t)P.queryBase + qbase + gg;
        const MatchQuery qq = queries[qw];
        int x0, x1, y0, y1;
        window_cells(qq, cam, invW, invH, x0, x1, y0, y1);
        window_query_wave(qq, qw, lane, x0, x1, y0, y1, gOff, cK, cD, candIdx, candKey, candCnt, candBest, status);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* sequential claim replay: one wavefront per frame pair                                             */

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return drfe_wave_min_u32(v); }

/* SearchByProjection(CurrentFrame, LastFrame, th, bMono): best unclaimed candidate per map point in
 * index order, TH_HIGH gate, rotation histogram with the reference's 1/HISTO_LENGTH factor quirk
 * (SURVEY.md §9.6), ComputeThreeMaxima, un-assignment of the other bins.
 *
 * One workgroup per frame pair, all map points at once, as a fixed-point iteration instead of a
 * sequential replay.  State: pick[i] = the candidate map point i takes.  One sweep computes
 *   owner[k]  = lowest i with obs(i) > 0 whose pick is keypoint k (the claim that blocks later points),
 *   pick'[i]  = best (distance, visit position) candidate k with k not claimed on entry and owner[k] >= i.
 * pick[i] depends only on picks of points j < i, so the sequential loop's result is the unique fixed
 * point and sweep t has points 0..t right; real frames settle in a handful of sweeps because
 * displacement chains are short.  A point whose overall best candidate is free (the common case) never
 * touches its candidate list. */
#define RS_THREADS 512
#define RS_MAX_T 8                                /* map points per thread: nQ <= 4096.  512 threads: a workgroup of 1024 needs sixteen free wave slots on one CU
                                                     at once, which it waits for while other batches' kernels fill the device (1.6 ms per launch with three
                                                     batches in flight against 0.09 alone); 512 x 8 is faster alone too (match stage 0.229 -> 0.209 ms) and
                                                     gave +3 % on the step, 256 x 16 the same step but 0.239 ms alone */
__global__ __launch_bounds__(RS_THREADS) void k_resolve_last(const MatchPair* __restrict__ pairs,
                                                             const MatchQuery* __restrict__ queries,
                                                             const drfe_keypoint* __restrict__ kps,
                                                             const int* __restrict__ kpCount, int maxKp,
                                                             const uint32_t* __restrict__ candIdx,
                                                             const uint32_t* __restrict__ candKey,
                                                             const int* __restrict__ candCnt,
                                                             const uint2* __restrict__ candBest, int checkOri,
                                                             int* __restrict__ match /* [curSlot][maxKp] in/out */,
                                                             const uint8_t* __restrict__ initObs,
                                                             int* __restrict__ matchCount,
                                                             uint16_t* __restrict__ histScratch /* [pair][2*maxKp] */)
{
    extern __shared__ int rs_smem[];
    int* owner = rs_smem;                                             /* [maxKp] */
    unsigned char* claim = reinterpret_cast<unsigned char*>(rs_smem + maxKp); /* bit0 claimed, bit1 obs>0 (on entry) */
    __shared__ int hist[30];
    __shared__ int sNm, sEnt, sInd[3], sChanged[3];
    const MatchPair P = pairs[blockIdx.x];
    const int tid = threadIdx.x;
    const int nQ = P.mpSlot >= 0 ? kpCount[P.lastSlot] : P.nQueries;
    const int nCur = kpCount[P.curSlot];
    int* M = match + (size_t)P.curSlot * maxKp;
    const drfe_keypoint* Kc = kps + (size_t)P.curSlot * maxKp;
    const drfe_keypoint* Kl = kps + (size_t)P.lastSlot * maxKp;
    uint16_t* hs = histScratch + (size_t)blockIdx.x * 2 * maxKp;
    for (int i = tid; i < nCur; i += RS_THREADS) {
        unsigned char c = 0;
        if (M[i] >= 0) c = 1 | ((initObs ? initObs[i] : 1) ? 2 : 0);
        claim[i] = c;
    }
    if (tid < 30) hist[tid] = 0;
    if (tid == 0) { sNm = 0; sEnt = 0; sChanged[0] = 0; }
    const float factor = 1.0f / 30;
    int cnt[RS_MAX_T], obs[RS_MAX_T];
    uint32_t bKey[RS_MAX_T], bIdx[RS_MAX_T], pKey[RS_MAX_T], pIdx[RS_MAX_T];
#pragma unroll
    for (int t = 0; t < RS_MAX_T; t++) {
        const int i = tid + t * RS_THREADS;
        cnt[t] = 0; obs[t] = 0; bKey[t] = 0xFFFFFFFFu; bIdx[t] = 0; pKey[t] = 0xFFFFFFFFu; pIdx[t] = 0;
        if (i < nQ) {
            const size_t qo = (size_t)P.queryBase + i;
            cnt[t] = candCnt[qo];
            if (cnt[t] > 0) { const uint2 b = candBest[qo]; bKey[t] = b.x; bIdx[t] = b.y; obs[t] = queries[qo].obs; }
        }
    }
    for (int sweep = 0;; sweep++) {
        for (int k = tid; k < nCur; k += RS_THREADS) owner[k] = 0x7FFFFFFF;
        if (tid == 0) sChanged[(sweep + 1) % 3] = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RS_MAX_T; t++)
            if (obs[t] && pKey[t] != 0xFFFFFFFFu && (int)(pKey[t] >> 22) <= 100)       /* TH_HIGH */
                atomicMin(&owner[pIdx[t] & 0xFFFFFF], tid + t * RS_THREADS);
        __syncthreads();
        bool changed = false;
#pragma unroll
        for (int t = 0; t < RS_MAX_T; t++) {
            if (cnt[t] == 0) continue;
            const int i = tid + t * RS_THREADS;
            uint32_t key = 0xFFFFFFFFu, idx = 0;
            const int kb = (int)(bIdx[t] & 0xFFFFFF);
            if ((claim[kb] & 3) != 3 && owner[kb] >= i) { key = bKey[t]; idx = bIdx[t]; }
            else {
                const size_t qo = ((size_t)P.queryBase + i) * DRFE_MATCH_MAX_CAND;
                /* eight candidates per trip: four independent 16-byte loads in flight instead of a chain
                 * of dependent dword loads (the list rows are 1 KB aligned) */
                for (int j0 = 0; j0 < cnt[t]; j0 += 8) {
                    const uint4 ka = *reinterpret_cast<const uint4*>(candKey + qo + j0);
                    const uint4 kb2 = *reinterpret_cast<const uint4*>(candKey + qo + j0 + 4);
                    const uint4 ia = *reinterpret_cast<const uint4*>(candIdx + qo + j0);
                    const uint4 ib = *reinterpret_cast<const uint4*>(candIdx + qo + j0 + 4);
                    const uint32_t kv[8] = {ka.x, ka.y, ka.z, ka.w, kb2.x, kb2.y, kb2.z, kb2.w};
                    const uint32_t iv[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        if (j0 + u >= cnt[t] || kv[u] >= key) continue;
                        const int kk = (int)(iv[u] & 0xFFFFFF);
                        if ((claim[kk] & 3) != 3 && owner[kk] >= i) { key = kv[u]; idx = iv[u]; }
                    }
                }
            }
            if (key != pKey[t] || idx != pIdx[t]) { changed = true; pKey[t] = key; pIdx[t] = idx; }
        }
        if (changed) sChanged[sweep % 3] = 1;
        __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(sChanged[sweep % 3])) break;      /* an LDS flag as a wave-uniform scalar: a scalar branch around the sweep's barriers */
    }
    /* commit: M[k] = the LAST point that took k (an obs == 0 claim does not block, a later point overwrites
     * it, :1507-1510 / :1518); every success counts and enters the rotation histogram */
    for (int k = tid; k < nCur; k += RS_THREADS) owner[k] = -1;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RS_MAX_T; t++) {
        if (pKey[t] == 0xFFFFFFFFu || (int)(pKey[t] >> 22) > 100) continue;
        const int i = tid + t * RS_THREADS;
        const int i2 = (int)(pIdx[t] & 0xFFFFFF);
        atomicMax(&owner[i2], i);
        atomicAdd(&sNm, 1);
        if (checkOri) {
            float rot = Kl[i].angle - Kc[i2].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == 30) bin = 0;
            const int e = atomicAdd(&sEnt, 1);
            hs[2 * e] = (uint16_t)bin;
            hs[2 * e + 1] = (uint16_t)i2;
            atomicAdd(&hist[bin], 1);
        }
    }
    __syncthreads();
    for (int k = tid; k < nCur; k += RS_THREADS)
        if (owner[k] >= 0) M[k] = owner[k];
    __syncthreads();
    if (checkOri) {
        if (tid == 0) {   /* ComputeThreeMaxima, src/ORBmatcher.cc:1666-1707 */
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < 30; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
            sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3;
        }
        __syncthreads();
        const int nEnt = sEnt;
        for (int e = tid; e < nEnt; e += RS_THREADS) {
            const int bin = hs[2 * e];
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) { M[hs[2 * e + 1]] = -1; atomicSub(&sNm, 1); }
        }
        __syncthreads();
    }
    if (tid == 0) matchCount[P.curSlot] = sNm;
}

/* SearchByProjection(Frame&, vector<MapPoint*>&, th): best and second best with their octaves, ratio
 * test only when both are on the same level (src/ORBmatcher.cc:99-125). */
__global__ __launch_bounds__(WAVE) void k_resolve_map(const MatchPair* __restrict__ pairs,
                                                      const MatchQuery* __restrict__ queries,
                                                      const int* __restrict__ kpCount, int maxKp,
                                                      const uint32_t* __restrict__ candIdx,
                                                      const uint32_t* __restrict__ candKey,
                                                      const int* __restrict__ candCnt, float nnratio,
                                                      int* __restrict__ match, const uint8_t* __restrict__ initObs,
                                                      int* __restrict__ matchCount)
{
    extern __shared__ unsigned char claim[];
    const MatchPair P = pairs[blockIdx.x];
    const int lane = threadIdx.x;
    const int nQ = P.nQueries;
    const int nCur = kpCount[P.curSlot];
    int* M = match + (size_t)P.curSlot * maxKp;
    for (int i = lane; i < nCur; i += WAVE) {
        unsigned char c = 0;
        if (M[i] >= 0) c = 1 | ((initObs ? initObs[i] : 1) ? 2 : 0);
        claim[i] = c;
    }
    __syncthreads();
    int nmatches = 0;
    for (int i = 0; i < nQ; i++) {
        const size_t qo = (size_t)P.queryBase + i;
        const int cnt = candCnt[qo];
        if (cnt == 0) continue;
        /* the reference's scan keeps (best, second) with strict '<' in visit order: best = min by
         * (dist, visit); second = min by (dist, visit) of the rest */
        uint32_t k1 = 0xFFFFFFFFu, i1 = 0, k2 = 0xFFFFFFFFu, i2 = 0;
        for (int c0 = 0; c0 < cnt; c0 += WAVE) {
            uint32_t key = 0xFFFFFFFFu, idx = 0;
            if (c0 + lane < cnt) {
                idx = candIdx[qo * DRFE_MATCH_MAX_CAND + c0 + lane];
                key = candKey[qo * DRFE_MATCH_MAX_CAND + c0 + lane];
                if ((claim[idx & 0xFFFFFF] & 3) == 3) key = 0xFFFFFFFFu;
            }
            for (int pass = 0; pass < 2; pass++) {
                const uint32_t mn = wave_min_u32(key);
                if (mn == 0xFFFFFFFFu) break;
                const unsigned long long who = __ballot(key == mn);
                const int src = __ffsll((long long)who) - 1;
                const uint32_t ix = (uint32_t)__shfl((int)idx, src);
                if (mn < k1) { k2 = k1; i2 = i1; k1 = mn; i1 = ix; }
                else if (mn < k2) { k2 = mn; i2 = ix; }
                if (lane == src) key = 0xFFFFFFFFu;
            }
        }
        if (k1 == 0xFFFFFFFFu) continue;
        const int bestDist = (int)(k1 >> 22);
        const int bestDist2 = (k2 == 0xFFFFFFFFu) ? 256 : (int)(k2 >> 22);
        const int bestLevel = (int)(i1 >> 24);
        const int bestLevel2 = (k2 == 0xFFFFFFFFu) ? -1 : (int)(i2 >> 24);
        if (bestDist <= 100) {
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            const int bi = (int)(i1 & 0xFFFFFF);
            if (lane == 0) { M[bi] = i; claim[bi] = 1 | (queries[qo].obs ? 2 : 0); }
            nmatches++;
            __syncthreads();
        }
    }
    if (lane == 0) matchCount[P.curSlot] = nmatches;
}

/* ------------------------------------------------------------------------------------------------ */
/* brute-force Hamming k-NN (k <= 2): 64 queries per workgroup, train tiles of 64 descriptors in LDS  */

__global__ __launch_bounds__(64) void k_bf_knn(const uint8_t* __restrict__ Q, int nq, const uint8_t* __restrict__ T,
                                               int nt, int k, int* __restrict__ outIdx, int* __restrict__ outDist)
{
    __shared__ uint64_t tile[64 * 4];
    const int lane = threadIdx.x;
    const int qi = blockIdx.x * 64 + lane;
    uint64_t q[4] = {0, 0, 0, 0};
    if (qi < nq) {
        const uint64_t* p = reinterpret_cast<const uint64_t*>(Q + (size_t)qi * 32);
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = p[3];
    }
    int d1 = 1 << 30, i1 = -1, d2 = 1 << 30, i2 = -1;
    for (int t0 = 0; t0 < nt; t0 += 64) {
        __syncthreads();
        if (t0 + lane < nt) {
            const uint64_t* p = reinterpret_cast<const uint64_t*>(T + (size_t)(t0 + lane) * 32);
            tile[lane * 4 + 0] = p[0]; tile[lane * 4 + 1] = p[1]; tile[lane * 4 + 2] = p[2]; tile[lane * 4 + 3] = p[3];
        }
        __syncthreads();
        const int m = min(64, nt - t0);
        for (int j = 0; j < m; j++) {   /* ascending train index: strict '<' keeps the first minimum */
            const int d = __popcll(q[0] ^ tile[j * 4]) + __popcll(q[1] ^ tile[j * 4 + 1]) +
                          __popcll(q[2] ^ tile[j * 4 + 2]) + __popcll(q[3] ^ tile[j * 4 + 3]);
            if (d < d1) { d2 = d1; i2 = i1; d1 = d; i1 = t0 + j; }
            else if (d < d2) { d2 = d; i2 = t0 + j; }
        }
    }
    if (qi < nq) {
        outIdx[(size_t)qi * k] = i1;
        outDist[(size_t)qi * k] = i1 >= 0 ? d1 : -1;
        if (k > 1) {
            outIdx[(size_t)qi * k + 1] = i2;
            outDist[(size_t)qi * k + 1] = i2 >= 0 ? d2 : -1;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* launchers                                                                                         */


/* the pixel ComputeStereoFromRGBD reads the depth of keypoint i at (u | v << 16; 0xFFFFFFFF = outside the image): what a host
 * that keeps the depth images needs to gather one raw value per keypoint instead of shipping whole depth frames */
__global__ __launch_bounds__(256) void k_kp_pixels(const drfe_keypoint* __restrict__ kps, const int* __restrict__ kpCount, int maxKp,
                                                   int w, int h, uint32_t* __restrict__ uv)
{
    const int slot = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= maxKp) return;
    uint32_t o = 0xFFFFFFFFu;
    if (i < kpCount[slot]) {
        const drfe_keypoint kp = kps[(size_t)slot * maxKp + i];
        const int v = (int)kp.y, u = (int)kp.x;
        if (u >= 0 && u < w && v >= 0 && v < h) o = (uint32_t)u | ((uint32_t)v << 16);
    }
    uv[(size_t)slot * maxKp + i] = o;
}

hipError_t drfe_launch_kp_pixels(drfe_ctx* c, int nframes, uint32_t* d_uv, hipStream_t s)
{
    hipLaunchKernelGGL(k_kp_pixels, dim3((c->maxKp + 255) / 256, nframes), dim3(256), 0, s, c->d_kps, c->d_kpCount, c->maxKp,
                       c->geom.imgW, c->geom.imgH, d_uv);
    return hipGetLastError();
}

hipError_t drfe_launch_glue(drfe_ctx* c, const uint16_t* d_depth, size_t frameStride, size_t rowStride,
                            const drfe_camera& cam, int nframes, hipStream_t s)
{
    const float invW = (float)DRFE_GRID_COLS / (float)(cam.max_x - cam.min_x);
    const float invH = (float)DRFE_GRID_ROWS / (float)(cam.max_y - cam.min_y);
    prof_begin(c, DRFE_STAGE_GLUE, s);
    if (c->dist.enabled)
        hipLaunchKernelGGL(k_undistort, dim3((c->maxKp + 255) / 256, nframes), dim3(256), 0, s, c->d_kps, c->d_kpCount,
                           c->maxKp, c->dist, c->d_kpsUn);
    hipLaunchKernelGGL(k_stereo, dim3((c->maxKp + 255) / 256, nframes), dim3(256), 0, s, c->d_kps, drfe_kps_un(c), c->d_kpCount,
                       c->maxKp, d_depth, frameStride, rowStride, c->geom.imgW, c->geom.imgH, cam, c->d_uRight,
                       c->d_depth);
    hipLaunchKernelGGL(k_grid, dim3(nframes), dim3(256), 0, s, drfe_kps_un(c), c->d_kpCount, c->maxKp, cam, invW, invH,
                       c->d_uRight, c->d_desc, c->d_gridOff, c->d_gridIdx, c->d_cellKp, c->d_cellDesc);
    prof_end(c, DRFE_STAGE_GLUE, s);
    return hipGetLastError();
}

/* AssignFeaturesToGrid alone, for slots whose keypoints, mvuRight and descriptors came from the host (drfe_frame_load) */
hipError_t drfe_launch_grid(drfe_ctx* c, const drfe_camera& cam, int nframes, hipStream_t s)
{
    const float invW = (float)DRFE_GRID_COLS / (float)(cam.max_x - cam.min_x);
    const float invH = (float)DRFE_GRID_ROWS / (float)(cam.max_y - cam.min_y);
    hipLaunchKernelGGL(k_grid, dim3(nframes), dim3(256), 0, s, drfe_kps_un(c), c->d_kpCount, c->maxKp, cam, invW, invH,
                       c->d_uRight, c->d_desc, c->d_gridOff, c->d_gridIdx, c->d_cellKp, c->d_cellDesc);
    return hipGetLastError();
}

/* p[0..n) = v as a kernel: memset nodes of a captured graph did not execute on this ROCm (see drfe_launch_orb), and the
 * per-frame flow replays the matcher from one */
__global__ __launch_bounds__(256) void k_fill_i32(int* __restrict__ p, int n, int v)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
hipError_t drfe_launch_fill_i32(int* d_p, int n, int v, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, s, d_p, n, v);
    return hipGetLastError();
}

/* p[0..n) = v and *q = w in one launch (the per-frame flow clears a slot's matches and their count) */
__global__ __launch_bounds__(256) void k_fill_i32_and_word(int* __restrict__ p, int n, int v, int* __restrict__ q, int w)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
    if (i == 0) *q = w;
}
hipError_t drfe_launch_fill_i32_and_word(int* d_p, int n, int v, int* d_q, int w, hipStream_t s)
{
    hipLaunchKernelGGL(k_fill_i32_and_word, dim3((n > 0 ? n + 255 : 256) / 256), dim3(256), 0, s, d_p, n, v, d_q, w);
    return hipGetLastError();
}

/* The results of one frame slot gathered into one staging buffer, so that a captured per-frame graph ends in ONE download
 * instead of one copy node per array (a copy node costs ~5 us of graph time; the arrays are a few KB each).  Segments are
 * dword ranges laid end to end in dst; a null source leaves its range untouched. */
__global__ __launch_bounds__(256) void k_pack_segments(DrfePackArgs A, uint32_t* __restrict__ dst)
{
    uint32_t total = 0;
    for (int k = 0; k < A.n; k++) total += A.dwords[k];
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        uint32_t base = 0;
        int k = 0;
        while (i - base >= A.dwords[k]) { base += A.dwords[k]; k++; }
        if (A.src[k]) dst[i] = A.src[k][i - base];
    }
}
hipError_t drfe_launch_pack_segments(const DrfePackArgs& A, uint32_t* d_dst, hipStream_t s)
{
    uint32_t total = 0;
    for (int k = 0; k < A.n; k++) total += A.dwords[k];
    if (!total) return hipSuccess;
    const unsigned blocks = (total + 1023) / 1024;            /* four dwords per thread */
    hipLaunchKernelGGL(k_pack_segments, dim3(blocks), dim3(256), 0, s, A, d_dst);
    return hipGetLastError();
}

hipError_t drfe_launch_window_match(drfe_ctx* c, const MatchBuffers& mb, const drfe_camera& cam, int npairs,
                                    int maxQueries, int mode, float th, float nnratio, int checkOri,
                                    const uint8_t* d_initObs, hipStream_t s, int statusWord)
{
    const float invW = (float)DRFE_GRID_COLS / (float)(cam.max_x - cam.min_x);
    const float invH = (float)DRFE_GRID_ROWS / (float)(cam.max_y - cam.min_y);
    if (mode == 0 && maxQueries > RS_THREADS * RS_MAX_T) return hipErrorInvalidValue;
    /* the matcher's overflow flag is its own word and is cleared by the search that owns it: d_status[1] for the host-buffer
     * searches (an overflowing window fails THAT call only, the next call on the same extracted batch starts clean),
     * d_status[2] for drfe_match_consecutive_batch, whose check is deferred to the download / pipeline sync - no other
     * search in between can erase it */
    (void)drfe_launch_fill_i32(c->d_status + statusWord, 1, 0, s);
    if (mode == 0)
        hipLaunchKernelGGL(k_queries_last, dim3((maxQueries + 255) / 256, npairs), dim3(256), 0, s, mb.d_pairs,
                           drfe_kps_un(c), c->d_kpCount, c->maxKp, mb.d_mps, cam, mb.d_scale, th, mb.d_queries);
    hipLaunchKernelGGL(k_window_candidates, dim3((maxQueries + WQ_PER_BLOCK - 1) / WQ_PER_BLOCK, npairs), dim3(WQ_THREADS), 0, s, mb.d_pairs,
                       mb.d_queries, c->d_kpCount, c->maxKp, c->d_gridOff, c->d_cellKp, c->d_cellDesc, cam, invW, invH,
                       mb.d_candIdx, mb.d_candKey, mb.d_candCnt, mb.d_candBest, c->d_status + statusWord,
                       drfe_div_magic((uint32_t)((maxQueries + WQ_PER_BLOCK - 1) / WQ_PER_BLOCK)));
    const size_t lds = (size_t)c->maxKp;
    if (mode == 0)
        hipLaunchKernelGGL(k_resolve_last, dim3(npairs), dim3(RS_THREADS), (size_t)c->maxKp * 5 + 16, s, mb.d_pairs,
                           mb.d_queries, drfe_kps_un(c), c->d_kpCount, c->maxKp, mb.d_candIdx, mb.d_candKey, mb.d_candCnt,
                           mb.d_candBest, checkOri, c->d_match, d_initObs, c->d_matchCount, mb.d_hist);
    else
        hipLaunchKernelGGL(k_resolve_map, dim3(npairs), dim3(WAVE), lds, s, mb.d_pairs, mb.d_queries, c->d_kpCount,
                           c->maxKp, mb.d_candIdx, mb.d_candKey, mb.d_candCnt, nnratio, c->d_match, d_initObs,
                           c->d_matchCount);
    return hipGetLastError();
}

/* the window gather alone (candidate lists in GetFeaturesInArea order with their distances): for searches whose acceptance
 * rule is replayed on the host */
hipError_t drfe_launch_window_candidates(drfe_ctx* c, const MatchBuffers& mb, const drfe_camera& cam, int npairs, int maxQueries,
                                         hipStream_t s)
{
    const float invW = (float)DRFE_GRID_COLS / (float)(cam.max_x - cam.min_x);
    const float invH = (float)DRFE_GRID_ROWS / (float)(cam.max_y - cam.min_y);
    (void)hipMemsetAsync(c->d_status + 1, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_window_candidates, dim3((maxQueries + WQ_PER_BLOCK - 1) / WQ_PER_BLOCK, npairs), dim3(WQ_THREADS), 0, s, mb.d_pairs,
                       mb.d_queries, c->d_kpCount, c->maxKp, c->d_gridOff, c->d_cellKp, c->d_cellDesc, cam, invW, invH,
                       mb.d_candIdx, mb.d_candKey, mb.d_candCnt, mb.d_candBest, c->d_status + 1,
                       drfe_div_magic((uint32_t)((maxQueries + WQ_PER_BLOCK - 1) / WQ_PER_BLOCK)));
    return hipGetLastError();
}

hipError_t drfe_launch_mappoints_last(drfe_ctx* c, const MatchBuffers& mb, const drfe_camera& cam, const float* d_Twc,
                                      int nframes, hipStream_t s)
{
    hipLaunchKernelGGL(k_mappoints_last, dim3((c->maxKp + 255) / 256, nframes), dim3(256), 0, s, drfe_kps_un(c), c->d_desc,
                       c->d_kpCount, c->maxKp, c->d_depth, cam, d_Twc, mb.d_mps);
    return hipGetLastError();
}

hipError_t drfe_launch_bf_knn(const uint8_t* dQ, int nq, const uint8_t* dT, int nt, int k, int* dIdx, int* dDist,
                              hipStream_t s)
{
    hipLaunchKernelGGL(k_bf_knn, dim3((nq + 63) / 64), dim3(64), 0, s, dQ, nq, dT, nt, k, dIdx, dDist);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------ */
/* LSDmatcher::SearchByProjection                                               */

/* projection part of the (Frame, Frame) overload, src/LSDmatcher.cpp:39-85 */
__global__ __launch_bounds__(64) void k_line_queries_last(const drfe_map_line* __restrict__ lines, int n,
                                                          const float* __restrict__ TcwCur, drfe_camera cam, int forward,
                                                          int backward, const float* __restrict__ scale, float th,
                                                          LineQuery* __restrict__ out)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const drfe_map_line ml = lines[i];
    LineQuery q;
    q.valid = 0; q.obs = ml.obs_positive ? 1 : 0; q.minLevel = 0; q.maxLevel = 0;
    q.x1 = q.y1 = q.x2 = q.y2 = q.r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) q.desc[k] = reinterpret_cast<const uint32_t*>(ml.desc)[k];
    if (ml.valid) {
        const float SP[3] = {(float)ml.world[0], (float)ml.world[1], (float)ml.world[2]};
        const float EP[3] = {(float)ml.world[3], (float)ml.world[4], (float)ml.world[5]};
        float T[16];
#pragma unroll
        for (int k = 0; k < 16; k++) T[k] = TcwCur[k];
        float SPc[3], EPc[3];
        mat3_mul_add(T, SP, SPc);
        mat3_mul_add(T, EP, EPc);
        bool ok = !(SPc[2] < 0.0f || EPc[2] < 0.0f);
        const float invz1 = 1.0f / SPc[2];
        const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
        if (u1 < cam.min_x || u1 > cam.max_x || v1 < cam.min_y || v1 > cam.max_y) ok = false;
        const float invz2 = 1.0f / EPc[2];
        const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
        if (u2 < cam.min_x || u2 > cam.max_x || v2 < cam.min_y || v2 > cam.max_y) ok = false;
        if (ok) {
            const int oct = ml.octave;
            q.valid = 1;
            q.x1 = u1; q.y1 = v1; q.x2 = u2; q.y2 = v2;
            q.r = th * scale[oct];
            if (forward) { q.minLevel = oct; q.maxLevel = -1; }
            else if (backward) { q.minLevel = 0; q.maxLevel = oct; }
            else { q.minLevel = oct - 1; q.maxLevel = oct + 1; }
        }
    }
    out[i] = q;
}

/* One wavefront replays the map-line loop: lanes over the current key lines (Frame::GetLinesInArea test,
 * claim check, LBD Hamming distance), wave reductions for best = min (distance, index) and second = the
 * same over the rest — what the reference's strict-< scan in index order leaves in bestDist/bestLevel and
 * bestDist2/bestLevel2 (src/LSDmatcher.cpp:95-136). */
__global__ __launch_bounds__(64) void k_line_search(const LineQuery* __restrict__ queries, int n,
                                                    const LineCur* __restrict__ cur, const uint8_t* __restrict__ desc,
                                                    int nCur, float nnratio, const uint8_t* __restrict__ claimIn /* bit0 held, bit1 obs */,
                                                    int* __restrict__ curMl, int* __restrict__ nmatchesOut)
{
    extern __shared__ unsigned char claim[];               /* [nCur] */
    const int lane = threadIdx.x;
    for (int k = lane; k < nCur; k += WAVE) claim[k] = claimIn[k];
    __syncthreads();
    int nmatches = 0;
    for (int i = 0; i < n; i++) {
        const LineQuery q = queries[i];
        if (!q.valid) continue;
        const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel > 0);
        const float r2 = q.r * q.r;
        const double rs = (double)q.r * 0.01;
        const float sl0 = (q.y1 - q.y2) / (q.x1 - q.x2);
        const double mxq = 0.5 * (double)(q.x1 + q.x2), myq = 0.5 * (double)(q.y1 + q.y2);
        const uint64_t q0 = (uint64_t)q.desc[0] | ((uint64_t)q.desc[1] << 32), q1 = (uint64_t)q.desc[2] | ((uint64_t)q.desc[3] << 32),
                       q2 = (uint64_t)q.desc[4] | ((uint64_t)q.desc[5] << 32), q3 = (uint64_t)q.desc[6] | ((uint64_t)q.desc[7] << 32);
        /* per-lane best and second over its stride of lines; key = distance << 16 | index */
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        for (int idx = lane; idx < nCur; idx += WAVE) {
            const LineCur kl = cur[idx];
            const double mx = mxq - (double)kl.ptX, my = myq - (double)kl.ptY;
            const float distance = (float)(mx * mx + my * my);
            if (distance > r2) continue;
            const float slope = sl0 - kl.angle;
            if ((double)slope > rs) continue;
            if (bCheckLevels) {
                if (kl.octave < q.minLevel) continue;
                if (q.maxLevel >= 0 && kl.octave > q.maxLevel) continue;
            }
            if ((claim[idx] & 3) == 3) continue;
            const uint64_t* d = reinterpret_cast<const uint64_t*>(desc + (size_t)idx * 32);
            const int dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
            const uint32_t key = ((uint32_t)dist << 16) | (uint32_t)idx;
            if (key < k1) { k2 = k1; k1 = key; }
            else if (key < k2) k2 = key;
        }
        const uint32_t b1 = wave_min_u32(k1);
        if (b1 == 0xFFFFFFFFu) continue;                   /* vIndices empty or every candidate claimed */
        const uint32_t b2 = wave_min_u32(k1 == b1 ? k2 : k1);
        const int bestDist = (int)(b1 >> 16), bestIdx = (int)(b1 & 0xFFFF);
        int bestDist2 = 256, bestLevel2 = -1;
        if (b2 != 0xFFFFFFFFu) { bestDist2 = (int)(b2 >> 16); bestLevel2 = cur[b2 & 0xFFFF].octave; }
        const int bestLevel = cur[bestIdx].octave;
        if (bestDist <= 100) {                             /* TH_HIGH */
            if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            if (lane == 0) {
                curMl[bestIdx] = i;
                claim[bestIdx] = (uint8_t)(1 | (q.obs ? 2 : 0));
            }
            __syncthreads();                               /* one wave: orders the LDS claim write */
            nmatches++;
        }
    }
    if (lane == 0) *nmatchesOut = nmatches;
}

hipError_t drfe_launch_line_projection(const drfe_map_line* d_lines, int n, const float* d_TcwCur, const drfe_camera& cam,
                                       int forward, int backward, const float* d_scale, float th, LineQuery* d_q,
                                       hipStream_t s)
{
    hipLaunchKernelGGL(k_line_queries_last, dim3((n + 63) / 64), dim3(64), 0, s, d_lines, n, d_TcwCur, cam, forward,
                       backward, d_scale, th, d_q);
    return hipGetLastError();
}

hipError_t drfe_launch_line_search(const LineQuery* d_q, int n, const LineCur* d_cur, const uint8_t* d_desc, int nCur,
                                   float nnratio, uint8_t* d_claim, int* d_curMl, int* d_nmatches, hipStream_t s)
{
    hipLaunchKernelGGL(k_line_search, dim3(1), dim3(64), (size_t)((nCur + 15) & ~15), s, d_q, n, d_cur, d_desc, nCur, nnratio, d_claim, d_curMl,
                       d_nmatches);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------ */
/* Frame::isInFrustum (src/Frame.cc:602-727)                                                          */

/* cv::norm / Mat::dot of 3x1 CV_32F: double accumulation */
__device__ __forceinline__ float norm3_f(const float v[3]) { return (float)sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }
__device__ __forceinline__ double dot3_d(const float a[3], const float b[3]) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; }

__global__ __launch_bounds__(256) void k_frustum_points(const drfe_frustum_point* __restrict__ pts, int n, FrustumPose P,
                                                        drfe_camera cam, drfe_tracked_point* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const drfe_frustum_point p = pts[i];
    drfe_tracked_point o = out[i];
    o.track_in_view = 0; o.level = 0; o.proj_x = o.proj_y = o.proj_xr = o.view_cos = 0.f;
    float Pc[3];
    mat3_mul_add(P.T, p.world, Pc);
    bool ok = !(Pc[2] < 0.0f);
    const float invz = 1.0f / Pc[2];
    const float u = cam.fx * Pc[0] * invz + cam.cx, v = cam.fy * Pc[1] * invz + cam.cy;
    if (u < cam.min_x || u > cam.max_x || v < cam.min_y || v > cam.max_y) ok = false;
    const float maxDistance = 1.2f * p.max_distance, minDistance = 0.8f * p.min_distance;
    const float PO[3] = {p.world[0] - P.Ow[0], p.world[1] - P.Ow[1], p.world[2] - P.Ow[2]};
    const float dist = norm3_f(PO);
    if (dist < minDistance || dist > maxDistance) ok = false;
    const float viewCos = (float)(dot3_d(PO, p.normal) / (double)dist);
    if (viewCos < P.limit) ok = false;
    if (ok) {
        const float ratio = p.max_distance / dist;
        int nScale = (int)ceilf(drfe_logf(ratio) / P.logScale);
        if (nScale < 0) nScale = 0;
        else if (nScale >= P.nLevels) nScale = P.nLevels - 1;
        o.track_in_view = 1; o.proj_x = u; o.proj_xr = u - P.bf * invz; o.proj_y = v; o.level = nScale; o.view_cos = viewCos;
    }
    out[i] = o;
}

__global__ __launch_bounds__(256) void k_frustum_lines(const drfe_frustum_line* __restrict__ lines, int n, FrustumPose P,
                                                       drfe_camera cam, drfe_tracked_line* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const drfe_frustum_line l = lines[i];
    drfe_tracked_line o = out[i];
    o.in_view = 0; o.level = 0; o.x1 = o.y1 = o.x2 = o.y2 = o.view_cos = 0.f;
    const float SP[3] = {(float)l.world[0], (float)l.world[1], (float)l.world[2]};
    const float EP[3] = {(float)l.world[3], (float)l.world[4], (float)l.world[5]};
    float SPc[3], EPc[3];
    mat3_mul_add(P.T, SP, SPc);
    mat3_mul_add(P.T, EP, EPc);
    bool ok = !(SPc[2] < 0.0f || EPc[2] < 0.0f);
    const float invz1 = 1.0f / SPc[2];
    const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
    if (u1 < cam.min_x || u1 > cam.max_x || v1 < cam.min_y || v1 > cam.max_y) ok = false;
    const float invz2 = 1.0f / EPc[2];
    const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
    if (u2 < cam.min_x || u2 > cam.max_x || v2 < cam.min_y || v2 > cam.max_y) ok = false;
    const float maxDistance = 1.2f * l.max_distance, minDistance = 0.8f * l.min_distance;
    float OM[3];
#pragma unroll
    for (int k = 0; k < 3; k++) OM[k] = (SP[k] + EP[k]) * 0.5f - P.Ow[k];
    const float dist = norm3_f(OM);
    if (dist < minDistance || dist > maxDistance) ok = false;
    const float pn[3] = {(float)l.normal[0], (float)l.normal[1], (float)l.normal[2]};
    const float viewCos = (float)(dot3_d(OM, pn) / (double)dist);
    if (viewCos < P.limit) ok = false;
    if (ok) {
        const float ratio = l.max_distance / dist;
        o.in_view = 1; o.x1 = u1; o.y1 = v1; o.x2 = u2; o.y2 = v2; o.view_cos = viewCos;
        o.level = (int)ceilf(drfe_logf(ratio) / P.logScale);       /* MapLine::PredictScale does not clamp */
    }
    out[i] = o;
}

hipError_t drfe_launch_frustum_points(const drfe_frustum_point* d_pts, int n, const FrustumPose& P, const drfe_camera& cam,
                                      drfe_tracked_point* d_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_frustum_points, dim3((n + 255) / 256), dim3(256), 0, s, d_pts, n, P, cam, d_out);
    return hipGetLastError();
}
hipError_t drfe_launch_frustum_lines(const drfe_frustum_line* d_lines, int n, const FrustumPose& P, const drfe_camera& cam,
                                     drfe_tracked_line* d_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_frustum_lines, dim3((n + 255) / 256), dim3(256), 0, s, d_lines, n, P, cam, d_out);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------ */
/* LSDmatcher::Fuse(KeyFrame*, vector<MapLine*>, th): the search (src/LSDmatcher.cpp:901-991)          */

/* One wavefront per map line.  The projection of both end points, the image-bound tests, the distance band, the
 * 60-degree cone and MapLine::PredictScale are wave-uniform (every lane evaluates them); the lanes then stride over the
 * keyframe's key lines with KeyFrame::GetLinesInArea's midpoint / slope test and the octave window, and the wave
 * minimum of distance << 16 | index is the reference's first strict minimum in index order. */
__global__ __launch_bounds__(256) void k_line_fuse_search(const drfe_frustum_line* __restrict__ lines, const uint8_t* __restrict__ descs,
                                                          const uint8_t* __restrict__ skip, int n, FrustumPose P, drfe_camera cam,
                                                          const float* __restrict__ scale, float th,
                                                          const LineCur* __restrict__ kf, const uint8_t* __restrict__ kfDesc, int nKF,
                                                          int* __restrict__ bestIdx, int* __restrict__ bestDist, int sim3,
                                                          LineSim3 C, int* __restrict__ distRow)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (256 / WAVE) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (i >= n) return;                          /* wave-uniform */
    int outIdx = -1, outDist = 0x7FFFFFFF;
    bool ok = !(skip && skip[i]);
    const drfe_frustum_line l = lines[i];
    const float SP[3] = {(float)l.world[0], (float)l.world[1], (float)l.world[2]};
    const float EP[3] = {(float)l.world[3], (float)l.world[4], (float)l.world[5]};
    float SPc[3], EPc[3];
    mat3_mul_add(P.T, SP, SPc);
    mat3_mul_add(P.T, EP, EPc);
    if (sim3) {                                  /* LSDmatcher::SearchBySim3: SPc2 = sR * (R1w * SP + t1w) + t, :566-574 */
        float a[3];
#pragma unroll
        for (int r = 0; r < 3; r++) a[r] = (C.sR[r * 3] * SPc[0] + C.sR[r * 3 + 1] * SPc[1] + C.sR[r * 3 + 2] * SPc[2]) + C.t[r];
#pragma unroll
        for (int r = 0; r < 3; r++) SPc[r] = a[r];
#pragma unroll
        for (int r = 0; r < 3; r++) a[r] = (C.sR[r * 3] * EPc[0] + C.sR[r * 3 + 1] * EPc[1] + C.sR[r * 3 + 2] * EPc[2]) + C.t[r];
#pragma unroll
        for (int r = 0; r < 3; r++) EPc[r] = a[r];
    }
    if (SPc[2] < 0.0f || EPc[2] < 0.0f) ok = false;
    const float invz1 = 1.0f / SPc[2];
    const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
    const float invz2 = 1.0f / EPc[2];
    const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
    const float maxDistance = 1.2f * l.max_distance, minDistance = 0.8f * l.min_distance;
    float dist;
    if (sim3) {                                  /* KeyFrame::IsInImage, distance of the midpoint in the target camera, no cone */
        if (!(u1 >= cam.min_x && u1 < cam.max_x && v1 >= cam.min_y && v1 < cam.max_y)) ok = false;
        if (!(u2 >= cam.min_x && u2 < cam.max_x && v2 >= cam.min_y && v2 < cam.max_y)) ok = false;
        const float mid[3] = {(SPc[0] + EPc[0]) * 0.5f, (SPc[1] + EPc[1]) * 0.5f, (SPc[2] + EPc[2]) * 0.5f};
        dist = norm3_f(mid);
        if (dist < minDistance || dist > maxDistance) ok = false;
    } else {
        if (u1 < cam.min_x || u1 > cam.max_x || v1 < cam.min_y || v1 > cam.max_y) ok = false;
        if (u2 < cam.min_x || u2 > cam.max_x || v2 < cam.min_y || v2 > cam.max_y) ok = false;
        float OM[3];
#pragma unroll
        for (int k = 0; k < 3; k++) OM[k] = (SP[k] + EP[k]) * 0.5f - P.Ow[k];
        dist = norm3_f(OM);
        if (dist < minDistance || dist > maxDistance) ok = false;
        const float pn[3] = {(float)l.normal[0], (float)l.normal[1], (float)l.normal[2]};
        if (dot3_d(OM, pn) < 0.5 * (double)dist) ok = false;
    }
    if (distRow)                                 /* per-key-line distances for the host's first-come replay: -1 = no candidate */
        for (int idx = lane; idx < nKF; idx += WAVE) distRow[(size_t)i * nKF + idx] = -1;
    if (ok) {
        const float ratio = l.max_distance / dist;
        const int level = (int)ceilf(drfe_logf(ratio) / P.logScale);
        if (level < 0 || level >= P.nLevels) {
            outIdx = -2;
        } else {
            const float r = th * scale[level];
            const float r2 = r * r;
            const double rs = (double)r * 0.01;
            const float sl0 = (v1 - v2) / (u1 - u2);
            const double mxq = 0.5 * (double)(u1 + u2), myq = 0.5 * (double)(v1 + v2);
            const uint64_t* q = reinterpret_cast<const uint64_t*>(descs + (size_t)i * 32);
            const uint64_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            uint32_t best = 0xFFFFFFFFu;
            for (int idx = lane; idx < nKF; idx += WAVE) {
                const LineCur kl = kf[idx];
                const double mx = mxq - (double)kl.ptX, my = myq - (double)kl.ptY;
                const float distance = (float)(mx * mx + my * my);
                if (distance > r2) continue;
                const float slope = sl0 - kl.angle;
                if ((double)slope > rs) continue;
                if (kl.octave < level - 1 || kl.octave > level) continue;
                const uint64_t* d = reinterpret_cast<const uint64_t*>(kfDesc + (size_t)idx * 32);
                const int hd = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
                const uint32_t key = ((uint32_t)hd << 16) | (uint32_t)idx;
                if (key < best) best = key;
                if (distRow) distRow[(size_t)i * nKF + idx] = hd;
            }
            const uint32_t mn = wave_min_u32(best);
            if (mn != 0xFFFFFFFFu) { outIdx = (int)(mn & 0xFFFF); outDist = (int)(mn >> 16); }
        }
    }
    if (lane == 0) { bestIdx[i] = outIdx; bestDist[i] = outDist; }
}

hipError_t drfe_launch_line_fuse_search(const drfe_frustum_line* d_lines, const uint8_t* d_descs, const uint8_t* d_skip, int n,
                                        const FrustumPose& P, const drfe_camera& cam, const float* d_scale, float th,
                                        const LineCur* d_kf, const uint8_t* d_kfDesc, int nKF, int* d_bestIdx, int* d_bestDist,
                                        hipStream_t s, const LineSim3* sim3, int* d_distRow)
{
    LineSim3 C;
    memset(&C, 0, sizeof(C));
    if (sim3) C = *sim3;
    hipLaunchKernelGGL(k_line_fuse_search, dim3((n + 3) / 4), dim3(256), 0, s, d_lines, d_descs, d_skip, n, P, cam, d_scale, th,
                       d_kf, d_kfDesc, nKF, d_bestIdx, d_bestDist, sim3 ? 1 : 0, C, d_distRow);
    return hipGetLastError();
}

/* ------------------------------------------------------------------------------------------------ */
/* ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th): the search (src/ORBmatcher.cc:846-953)          */

/* One wavefront per map point.  Projection, KeyFrame::IsInImage, distance band, 60-degree cone and PredictScale are
 * wave-uniform; the window is gathered exactly like k_window_candidates (one contiguous run of cell-sorted records per
 * grid column, lanes on candidates) with KeyFrame::GetFeaturesInArea's rules (no level filter), then the octave
 * window, the chi-square reprojection gate and a wave minimum of distance << 22 | visit position (first minimum). */
__global__ __launch_bounds__(256) void k_fuse_search(const drfe_frustum_point* __restrict__ pts, const uint8_t* __restrict__ descs,
                                                     const uint8_t* __restrict__ skip, int n, FuseParams P, drfe_camera cam,
                                                     float invW, float invH, const int* __restrict__ gridOff,
                                                     const uint4* __restrict__ cellKp, const uint4* __restrict__ cellDesc,
                                                     int* __restrict__ bestIdx, int* __restrict__ bestDist,
                                                     const uint8_t* __restrict__ taken, int2* __restrict__ list,
                                                     int* __restrict__ listCount)
{
    __shared__ uint32_t sKey[256 / WAVE][WAVE];
    __shared__ uint32_t sIdx[256 / WAVE][WAVE];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i = blockIdx.x * (256 / WAVE) + w;
    if (i >= n) return;                          /* wave-uniform */
    int outIdx = -1, outDist = 256;
    int nListed = 0;                             /* candidates within listTh, wave-uniform */
    if (list) sKey[w][lane] = 0xFFFFFFFFu;
    bool ok = !(skip && skip[i]);
    const drfe_frustum_point p = pts[i];
    float Pc[3];
    mat3_mul_add(P.T, p.world, Pc);
    if (P.sim3 == 2) {                                      /* p3Dc2 = sR21 * p3Dc1 + t21 */
        const float q[3] = {Pc[0], Pc[1], Pc[2]};
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const float d = P.sR2[r * 3] * q[0] + P.sR2[r * 3 + 1] * q[1] + P.sR2[r * 3 + 2] * q[2];
            Pc[r] = d + P.t2[r];
        }
    }
    const bool reloc = P.sim3 == 4;
    if (!reloc && Pc[2] < 0.0f) ok = false;                 /* the relocalisation search has no depth test */
    const float invz = (P.sim3 == 1 || P.sim3 == 2 || reloc) ? (float)(1.0 / (double)Pc[2]) : 1 / Pc[2];
    const float x = Pc[0] * invz, y = Pc[1] * invz;
    /* `fx*xc*invzc+cx` at :1568 associates the other way round */
    const float u = reloc ? cam.fx * Pc[0] * invz + cam.cx : cam.fx * x + cam.cx;
    const float v = reloc ? cam.fy * Pc[1] * invz + cam.cy : cam.fy * y + cam.cy;
    if (reloc) {
        if (u < cam.min_x || u > cam.max_x || v < cam.min_y || v > cam.max_y) ok = false;
        if (!(Pc[2] != 0.0f)) ok = false;                   /* z = 0 (or NaN): the reference projects to infinity */
    } else if (!(u >= cam.min_x && u < cam.max_x && v >= cam.min_y && v < cam.max_y)) ok = false;
    const float ur = u - P.bf * invz;
    const float maxDistance = 1.2f * p.max_distance, minDistance = 0.8f * p.min_distance;
    const float PO[3] = {p.world[0] - P.Ow[0], p.world[1] - P.Ow[1], p.world[2] - P.Ow[2]};
    const float dist3D = P.sim3 == 2 ? norm3_f(Pc) : norm3_f(PO);
    if (dist3D < minDistance || dist3D > maxDistance) ok = false;
    if (P.sim3 != 2 && !reloc && dot3_d(PO, p.normal) < 0.5 * (double)dist3D) ok = false;
    if (ok) {
        const float ratio = p.max_distance / dist3D;
        int level = (int)ceilf(drfe_logf(ratio) / P.logScale);
        if (level < 0) level = 0;
        else if (level >= P.nLevels) level = P.nLevels - 1;
        const float r = P.th * P.scale[level];
        const int nMinCellX = max(0, (int)floorf((u - cam.min_x - r) * invW));
        const int nMaxCellX = min(DRFE_GRID_COLS - 1, (int)ceilf((u - cam.min_x + r) * invW));
        const int nMinCellY = max(0, (int)floorf((v - cam.min_y - r) * invH));
        const int nMaxCellY = min(DRFE_GRID_ROWS - 1, (int)ceilf((v - cam.min_y + r) * invH));
        if (!(nMinCellX >= DRFE_GRID_COLS || nMaxCellX < 0 || nMinCellY >= DRFE_GRID_ROWS || nMaxCellY < 0)) {
            const uint64_t* q = reinterpret_cast<const uint64_t*>(descs + (size_t)i * 32);
            const uint64_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const int ncol = nMaxCellX - nMinCellX + 1;
            int runB = 0, runN = 0;
            if (lane < ncol) {
                const int cb = (nMinCellX + lane) * DRFE_GRID_ROWS;
                runB = gridOff[cb + nMinCellY];
                runN = gridOff[cb + nMaxCellY + 1] - runB;
            }
            const int incl = drfe_wave_incl_scan(runN, lane);
            const int T = __builtin_amdgcn_readlane(incl, 63);
            uint32_t myKey = 0xFFFFFFFFu, myIdx = 0;
            for (int s0 = 0; s0 < T; s0 += WAVE) {
                const int sq = s0 + lane;
                uint32_t lKey = 0xFFFFFFFFu, lIdx = 0;
                int base = __builtin_amdgcn_readfirstlane(runB);
                for (int j = 0; j + 1 < ncol; j++) {
                    const int inclJ = __builtin_amdgcn_readlane(incl, j), nextB = __builtin_amdgcn_readlane(runB, j + 1);
                    if (inclJ <= sq) base = nextB - inclJ;
                }
                if (sq < T) {
                    const int pp = base + sq;
                    const uint4 k4 = cellKp[pp];
                    const float kx = __uint_as_float(k4.x), ky = __uint_as_float(k4.y), kr = __uint_as_float(k4.z);
                    const int oct = (int)(k4.w >> 24);
                    const float dx = kx - u, dy = ky - v;
                    bool c = fabsf(dx) < r && fabsf(dy) < r;                /* KeyFrame::GetFeaturesInArea */
                    if (oct < level - 1 || oct > level + (reloc ? 1 : 0)) c = false;  /* reloc: Frame::GetFeaturesInArea(level-1, level+1) */
                    const float ex = u - kx, ey = v - ky;
                    if (P.sim3) {
                    } else if (kr >= 0) {
                        const float er = ur - kr;
                        const float e2 = ex * ex + ey * ey + er * er;
                        if ((double)(e2 * P.invSigma2[oct & 15]) > 7.8) c = false;
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        if ((double)(e2 * P.invSigma2[oct & 15]) > 5.99) c = false;
                    }
                    if (c && taken && taken[k4.w & 0xFFFFFF]) c = false;    /* vpMatched[idx] */
                    if (c) {
                        const uint4 da = cellDesc[2 * pp], db = cellDesc[2 * pp + 1];
                        const int dist = __popcll(q0 ^ ((uint64_t)da.x | ((uint64_t)da.y << 32))) + __popcll(q1 ^ ((uint64_t)da.z | ((uint64_t)da.w << 32))) +
                                         __popcll(q2 ^ ((uint64_t)db.x | ((uint64_t)db.y << 32))) + __popcll(q3 ^ ((uint64_t)db.z | ((uint64_t)db.w << 32)));
                        const uint32_t key = ((uint32_t)dist << 22) | (uint32_t)min(sq, (1 << 22) - 1);
                        if (key < myKey) { myKey = key; myIdx = k4.w & 0xFFFFFF; }
                        if (dist <= P.listTh) { lKey = key; lIdx = k4.w & 0xFFFFFF; }
                    }
                }
                if (list) {
                    /* append this step's candidates within listTh: slot = running count + rank in the ballot, folded
                     * onto the 64 slots with a minimum (a fold only happens past 64 such candidates; the overall
                     * minimum survives it and listCount tells the host the list is incomplete) */
                    const unsigned long long bal = __ballot(lKey != 0xFFFFFFFFu);
                    if (lKey != 0xFFFFFFFFu) {
                        const int slot = (nListed + __popcll(bal & ((1ull << lane) - 1))) & 63;
                        if (lKey < sKey[w][slot]) { sKey[w][slot] = lKey; sIdx[w][slot] = lIdx; }
                    }
                    nListed += __popcll(bal);
                }
            }
            const uint32_t mn = wave_min_u32(myKey);
            if (mn != 0xFFFFFFFFu) {
                const unsigned long long who = __ballot(myKey == mn);
                outIdx = __shfl((int)myIdx, __ffsll((long long)who) - 1);
                outDist = (int)(mn >> 22);
            }
        }
    }
    if (lane == 0) { bestIdx[i] = outIdx; bestDist[i] = outDist; }
    if (list) {
        /* the FUSE_LIST_K smallest keys of the 64 slots, in order (same-wave LDS traffic: no barrier needed) */
        uint32_t k = sKey[w][lane];
        const uint32_t id = sIdx[w][lane];
        for (int r = 0; r < FUSE_LIST_K; r++) {
            const uint32_t mn = wave_min_u32(k);
            int2 e = make_int2(-1, 256);
            if (mn != 0xFFFFFFFFu) {
                const int src = __ffsll((long long)__ballot(k == mn)) - 1;
                e = make_int2(__shfl((int)id, src), (int)(mn >> 22));
                if (lane == src) k = 0xFFFFFFFFu;
            }
            if (lane == 0) list[(size_t)i * FUSE_LIST_K + r] = e;
        }
        if (lane == 0) listCount[i] = nListed;
    }
}

hipError_t drfe_launch_fuse_search(drfe_ctx* c, int slot, const drfe_frustum_point* d_pts, const uint8_t* d_descs,
                                   const uint8_t* d_skip, int n, const FuseParams& P, const drfe_camera& cam, int* d_bestIdx,
                                   int* d_bestDist, hipStream_t s, const uint8_t* d_taken, int2* d_list, int* d_listCount)
{
    const float invW = (float)DRFE_GRID_COLS / (float)(cam.max_x - cam.min_x);
    const float invH = (float)DRFE_GRID_ROWS / (float)(cam.max_y - cam.min_y);
    hipLaunchKernelGGL(k_fuse_search, dim3((n + 3) / 4), dim3(256), 0, s, d_pts, d_descs, d_skip, n, P, cam, invW, invH,
                       c->d_gridOff + (size_t)slot * (DRFE_GRID_CELLS + 1), c->d_cellKp + (size_t)slot * c->maxKp,
                       c->d_cellDesc + (size_t)slot * c->maxKp * 2, d_bestIdx, d_bestDist, d_taken, d_list, d_listCount);
    return hipGetLastError();
}
