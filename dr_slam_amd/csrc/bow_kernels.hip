/* bow_kernels.hip — bag-of-words kernels.
 *
 *   k_bow_transform     TemplatedVocabulary::transform(feature, word, weight, nid, levelsup)
 *                       reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1216-1259, FORB::distance
 *   k_bow_match_groups  inner loops of ORBmatcher::SearchByBoW        src/ORBmatcher.cc:190-262
 *   k_bow_rot_filter    rotation-consistency part of SearchByBoW      src/ORBmatcher.cc:270-290
 *
 * transform: 32 lanes per descriptor — lane c scores child c of the current node (k <= 20), a
 * half-wave min over (distance << 8 | c) reproduces "first child with the strictly smallest distance".
 * SearchByBoW: the claim `if(vpMapPointMatches[realIdxF]) continue;` only couples features of the same
 * vocabulary node, so every common node is an independent sequential problem: one wavefront per node,
 * lanes over the frame features of the node, a short loop over its keyframe features.
 */
#include "drfe_internal.h"
#include "bow_internal.h"

#define WAVE 64

__global__ __launch_bounds__(256) void k_bow_transform(const VocDev voc, const uint8_t* __restrict__ desc,
                                                       const int* __restrict__ kpCount, int maxKp, int levelsup,
                                                       int* __restrict__ word, double* __restrict__ weight,
                                                       int* __restrict__ nid)
{
    const int slot = blockIdx.y;
    const int f = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int c = threadIdx.x & 31;
    if (f >= kpCount[slot]) return;                 /* uniform per 32-lane group */
    const uint64_t* q = reinterpret_cast<const uint64_t*>(desc + ((size_t)slot * maxKp + f) * 32);
    const uint64_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const int nid_level = voc.L - levelsup;
    int node = 0, level = 0, nidOut = 0;
    while (true) {
        const int cb = voc.childBegin[node], nc = voc.childBegin[node + 1] - cb;
        if (nc == 0) break;                         /* leaf */
        ++level;
        uint32_t key = 0xFFFFFFFFu;
        if (c < nc) {
            const int child = voc.children[cb + c];
            const uint64_t* d = reinterpret_cast<const uint64_t*>(voc.desc + (size_t)child * 32);
            const int dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
            key = ((uint32_t)dist << 8) | (uint32_t)c;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o, 32));
        node = voc.children[cb + (int)(key & 0xFF)];
        if (level == nid_level) nidOut = node;
    }
    if (c == 0) {
        const size_t o = (size_t)slot * maxKp + f;
        word[o] = voc.wordId[node];
        weight[o] = voc.weight[node];
        nid[o] = nidOut;
    }
}

/* one wavefront per common vocabulary node */
__global__ __launch_bounds__(WAVE) void k_bow_match_groups(const BowGroup* __restrict__ groups,
                                                           const int* __restrict__ kfIdx, const int* __restrict__ fIdx,
                                                           const uint8_t* __restrict__ descKF,
                                                           const uint8_t* __restrict__ descF,
                                                           const drfe_keypoint* __restrict__ kpKF,
                                                           const drfe_keypoint* __restrict__ kpF,
                                                           const int* __restrict__ kfMP, const int* __restrict__ fMP /* NULL: frame */,
                                                           int thLow /* 50: <=, 49: the keyframe overload's < 50 */,
                                                           float nnratio, int checkOri,
                                                           int* __restrict__ match /* per F keypoint, -1 init */,
                                                           int* __restrict__ counters /* [0]=nmatches [1]=entries */,
                                                           int* __restrict__ hist /* 30 */,
                                                           uint16_t* __restrict__ entries /* (bin, idx) pairs */)
{
    const BowGroup g = groups[blockIdx.x];
    const int lane = threadIdx.x;
    const int nF = g.fEnd - g.fBegin;
    for (int a = g.kfBegin; a < g.kfEnd; a++) {
        const int iKF = kfIdx[a];
        if (kfMP[iKF] < 0) continue;                       /* !pMP || pMP->isBad() */
        const uint64_t* q = reinterpret_cast<const uint64_t*>(descKF + (size_t)iKF * 32);
        const uint64_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;     /* two smallest (dist << 16 | position) keys */
        for (int b0 = 0; b0 < nF; b0 += WAVE) {
            uint32_t key = 0xFFFFFFFFu;
            const int b = b0 + lane;
            if (b < nF) {
                const int iF = fIdx[g.fBegin + b];
                /* claims are written by lane 0 of this very wavefront; agent-scope accesses keep the
                 * vector L1 out of the picture */
                if (__hip_atomic_load(&match[iF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0 && (!fMP || fMP[iF] >= 0)) {
                    const uint64_t* d = reinterpret_cast<const uint64_t*>(descF + (size_t)iF * 32);
                    const int dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
                    key = ((uint32_t)dist << 16) | (uint32_t)b;
                }
            }
            for (int pass = 0; pass < 2; pass++) {
                uint32_t mn = key;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
                if (mn == 0xFFFFFFFFu) break;
                if (mn < k1) { k2 = k1; k1 = mn; } else if (mn < k2) k2 = mn;
                if (key == mn) key = 0xFFFFFFFFu;         /* keys are unique (position) */
            }
        }
        if (k1 == 0xFFFFFFFFu) continue;
        const int bestDist1 = (int)(k1 >> 16);
        const int bestDist2 = (k2 == 0xFFFFFFFFu) ? 256 : (int)(k2 >> 16);
        if (bestDist1 <= thLow && (float)bestDist1 < nnratio * (float)bestDist2) {   /* TH_LOW, ratio */
            const int iF = fIdx[g.fBegin + (int)(k1 & 0xFFFF)];
            if (lane == 0) {
                __hip_atomic_store(&match[iF], iKF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(&counters[0], 1);
                if (checkOri) {
                    float rot = kpKF[iKF].angle - kpF[iF].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * (1.0f / 30));
                    if (bin == 30) bin = 0;
                    const int e = atomicAdd(&counters[1], 1);
                    entries[2 * e] = (uint16_t)bin;
                    entries[2 * e + 1] = (uint16_t)iF;
                    atomicAdd(&hist[bin], 1);
                }
            }
            __threadfence_block();
            __syncthreads();                               /* later KF features of this node see the claim */
        }
    }
}

__global__ __launch_bounds__(256) void k_bow_rot_filter(int* __restrict__ match, int* __restrict__ counters,
                                                        const int* __restrict__ hist,
                                                        const uint16_t* __restrict__ entries)
{
    __shared__ int sInd[3];
    if (threadIdx.x == 0) {   /* ComputeThreeMaxima, src/ORBmatcher.cc:1666-1707 */
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < 30; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) ind3 = -1;
        sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3;
    }
    __syncthreads();
    const int n = counters[1];
    int removed = 0;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int bin = entries[2 * e];
        if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) { match[entries[2 * e + 1]] = -1; removed++; }
    }
    if (removed) atomicSub(&counters[0], removed);
}

hipError_t drfe_launch_bow_transform(drfe_ctx* c, const VocDev& voc, int levelsup, int nframes, int* d_word,
                                     double* d_weight, int* d_nid, hipStream_t s)
{
    hipLaunchKernelGGL(k_bow_transform, dim3((c->maxKp + 7) / 8, nframes), dim3(256), 0, s, voc, c->d_desc, c->d_kpCount,
                       c->maxKp, levelsup, d_word, d_weight, d_nid);
    return hipGetLastError();
}

/* ORBmatcher::SearchForTriangulation inner loops (src/ORBmatcher.cc:695-793): one wavefront per common vocabulary
 * node.  For a keypoint of KF1 the reference keeps, among the KF2 keypoints of the node that pass the static tests
 * (no map point, stereo rule, distance <= TH_LOW, not closer than 10 px (scaled) to the epipole when both are
 * monocular, CheckDistEpipolarLine), the LAST one with the smallest distance (`dist > bestDist -> continue` lets
 * equal distances replace): wave minimum of dist << 16 | (0xFFFF - position).  This reference never marks KF2
 * keypoints as taken, so KF1 keypoints are independent. */
__global__ __launch_bounds__(WAVE) void k_bow_triangulation_groups(const BowGroup* __restrict__ groups,
                                                                   const int* __restrict__ idx1s, const int* __restrict__ idx2s,
                                                                   const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ desc2,
                                                                   const drfe_keypoint* __restrict__ kp1, const drfe_keypoint* __restrict__ kp2,
                                                                   const float* __restrict__ ur1, const float* __restrict__ ur2,
                                                                   const int* __restrict__ mp1, const int* __restrict__ mp2, TriParams P,
                                                                   int* __restrict__ match12, int* __restrict__ counters,
                                                                   int* __restrict__ hist, uint16_t* __restrict__ entries)
{
    const BowGroup g = groups[blockIdx.x];
    const int lane = threadIdx.x;
    const int n2 = g.fEnd - g.fBegin;
    for (int a = g.kfBegin; a < g.kfEnd; a++) {
        const int i1 = idx1s[a];
        if (mp1[i1] >= 0) continue;
        const bool bStereo1 = ur1[i1] >= 0;
        if (P.onlyStereo && !bStereo1) continue;
        const drfe_keypoint k1 = kp1[i1];
        const uint64_t* q = reinterpret_cast<const uint64_t*>(desc1 + (size_t)i1 * 32);
        const uint64_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        /* epipolar line l = x1' F12 (CheckDistEpipolarLine, :143-146) */
        const float la = k1.x * P.F[0] + k1.y * P.F[3] + P.F[6];
        const float lb = k1.x * P.F[1] + k1.y * P.F[4] + P.F[7];
        const float lc = k1.x * P.F[2] + k1.y * P.F[5] + P.F[8];
        const float den = la * la + lb * lb;
        uint32_t best = 0xFFFFFFFFu;
        for (int b0 = 0; b0 < n2; b0 += WAVE) {
            const int b = b0 + lane;
            uint32_t key = 0xFFFFFFFFu;
            if (b < n2) {
                const int i2 = idx2s[g.fBegin + b];
                bool ok = mp2[i2] < 0;
                const bool bStereo2 = ur2[i2] >= 0;
                if (P.onlyStereo && !bStereo2) ok = false;
                if (ok) {
                    const uint64_t* d = reinterpret_cast<const uint64_t*>(desc2 + (size_t)i2 * 32);
                    const int dist = __popcll(q0 ^ d[0]) + __popcll(q1 ^ d[1]) + __popcll(q2 ^ d[2]) + __popcll(q3 ^ d[3]);
                    if (dist > 50) ok = false;                                     /* TH_LOW */
                    const drfe_keypoint k2 = kp2[i2];
                    if (ok && !bStereo1 && !bStereo2) {
                        const float distex = P.ex - k2.x, distey = P.ey - k2.y;
                        if (distex * distex + distey * distey < 100 * P.scale[k2.octave]) ok = false;
                    }
                    if (ok) {
                        const float num = la * k2.x + lb * k2.y + lc;
                        if (den == 0) ok = false;
                        else {
                            const float dsqr = num * num / den;
                            if (!((double)dsqr < 3.84 * (double)P.sigma2[k2.octave])) ok = false;
                        }
                    }
                    if (ok) key = ((uint32_t)dist << 16) | (uint32_t)(0xFFFF - b);
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o));
            best = min(best, key);
        }
        if (best == 0xFFFFFFFFu) continue;
        if (lane == 0) {
            const int i2 = idx2s[g.fBegin + (0xFFFF - (int)(best & 0xFFFF))];
            match12[i1] = i2;
            atomicAdd(&counters[0], 1);
            if (P.checkOri) {
                float rot = k1.angle - kp2[i2].angle;
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * (1.0f / 30));
                if (bin == 30) bin = 0;
                const int e = atomicAdd(&counters[1], 1);
                entries[2 * e] = (uint16_t)bin;
                entries[2 * e + 1] = (uint16_t)i1;
                atomicAdd(&hist[bin], 1);
            }
        }
    }
}

hipError_t drfe_launch_bow_triangulation(drfe_ctx* c, int slot1, int slot2, const BowGroup* d_groups, int ngroups,
                                         const int* d_idx1, const int* d_idx2, const int* d_mp1, const int* d_mp2,
                                         const TriParams& P, int* d_match12, int* d_counters, int* d_hist,
                                         uint16_t* d_entries, hipStream_t s)
{
    const size_t o1 = (size_t)slot1 * c->maxKp, o2 = (size_t)slot2 * c->maxKp;
    if (ngroups > 0)
        hipLaunchKernelGGL(k_bow_triangulation_groups, dim3(ngroups), dim3(WAVE), 0, s, d_groups, d_idx1, d_idx2,
                           c->d_desc + o1 * 32, c->d_desc + o2 * 32, drfe_kps_un(c) + o1, drfe_kps_un(c) + o2, c->d_uRight + o1,
                           c->d_uRight + o2, d_mp1, d_mp2, P, d_match12, d_counters, d_hist, d_entries);
    if (P.checkOri)
        hipLaunchKernelGGL(k_bow_rot_filter, dim3(1), dim3(256), 0, s, d_match12, d_counters, d_hist, d_entries);
    return hipGetLastError();
}

hipError_t drfe_launch_bow_match(drfe_ctx* c, int kfSlot, int fSlot, const BowGroup* d_groups, int ngroups,
                                 const int* d_kfIdx, const int* d_fIdx, const int* d_kfMP, const int* d_fMP, int thLow,
                                 float nnratio, int checkOri, int* d_match, int* d_counters, int* d_hist,
                                 uint16_t* d_entries, hipStream_t s)
{
    if (ngroups > 0)
        hipLaunchKernelGGL(k_bow_match_groups, dim3(ngroups), dim3(WAVE), 0, s, d_groups, d_kfIdx, d_fIdx,
                           c->d_desc + (size_t)kfSlot * c->maxKp * 32, c->d_desc + (size_t)fSlot * c->maxKp * 32,
                           c->d_kps + (size_t)kfSlot * c->maxKp, c->d_kps + (size_t)fSlot * c->maxKp, d_kfMP, d_fMP, thLow,
                           nnratio, checkOri, d_match, d_counters, d_hist, d_entries);
    if (checkOri)
        hipLaunchKernelGGL(k_bow_rot_filter, dim3(1), dim3(256), 0, s, d_match, d_counters, d_hist, d_entries);
    return hipGetLastError();
}
