/* lsd_order_kernels.hip — cv::LineSegmentDetectorImpl's pseudo-ordering on the device: std::sort(ordered_points, compare_norm)
 * of OpenCV 3.4 lsd.cpp (behind reference src/LSDextractor.cpp:12-43).
 *
 * compare_norm looks at the gradient bin only, 196 000 pixels share 1024 bins, and std::sort is not stable: the order of equal
 * bins - which is the seed order of region growing - is whatever libstdc++'s introsort leaves.  Only the same algorithm gives
 * the same permutation, so this file executes introsort's element moves (bits/stl_algo.h: median-of-three Hoare partitions
 * down to ranges of 16 under a depth limit of 2 lg n, then one insertion sort) the way introsort_restated.h restates them for
 * the host, with the parallelism the restatement exposes:
 *   - the two sub-ranges a partition leaves are independent: ranges above 8192 records are partitioned by the whole workgroup
 *     one after the other, everything below is dealt to the sixteen wavefronts, each of which finishes its range depth-first;
 *   - a Hoare partition swaps the k-th record from the left that does not go before the pivot ("left stopper") with the k-th
 *     from the right the pivot does not go before, while the former lies left of the latter, and cuts at min(L[K], R[K-1]):
 *     stopper positions are compacted in rank order by ballot / popcount prefix sums, K is the length of the prefix of pairs
 *     still in order, and the K swaps are independent;
 *   - the final insertion sort moves a record left past records it goes before only - it is the stable sort of what the
 *     partitions left: two stable counting passes over five bits of the bin.
 * A range that exhausts the depth limit (heap sort in libstdc++; never reached on image data) flags the frame and the host
 * orders it.  One workgroup of 1024 threads per frame; keys and results stay in HBM, the device region growing reads them there. */
#include "drfe_internal.h"
#include "lines_internal.h"

#define ORD_T 1024
#define ORD_WAVES (ORD_T / 64)
#define ORD_BIG 8192             /* ranges above this many records: one at a time by the whole workgroup */
#define ORD_QCAP 1024            /* pending ranges a frame can hold (LDS; 128 KB of it are the counting passes') */
#define ORD_STACK 48             /* depth-first stack of a wavefront (>= the depth limit 2 lg n of any array that fits) */

namespace {

struct Seg { uint32_t first, last; int depth; };

__device__ __forceinline__ uint32_t bin_of(uint32_t k) { return k >> 22; }
/* compare_norm: a goes before b iff its bin is larger */
__device__ __forceinline__ bool before(uint32_t a, uint32_t b) { return bin_of(a) > bin_of(b); }

/* std::__move_median_to_first(result, x, y, z) by one thread */
__device__ __forceinline__ void median_to_first(uint32_t* a, uint32_t result, uint32_t x, uint32_t y, uint32_t z)
{
    const uint32_t ax = a[x], ay = a[y], az = a[z];
    uint32_t pick;
    if (before(ax, ay)) {
        if (before(ay, az)) pick = y;
        else if (before(ax, az)) pick = z;
        else pick = x;
    } else if (before(ax, az)) pick = x;
    else if (before(ay, az)) pick = z;
    else pick = y;
    const uint32_t t = a[result];
    a[result] = a[pick];
    a[pick] = t;
}

/* std::__unguarded_partition(a + first + 1, a + last, a + first) by a group of NT threads (64: one wavefront; ORD_T: the
 * workgroup).  tid = thread index inside the group.  posL / posR: scratch of the range's length at [first, last).  wcnt: LDS,
 * ORD_WAVES + 2 ints (workgroup variant).  Returns the cut to every thread. */
template <int NT>
__device__ __forceinline__ uint32_t hoare_cut(uint32_t* a, uint32_t first, uint32_t last, uint32_t* posL, uint32_t* posR, int tid, int* wcnt)
{
    const uint32_t lo = first + 1, hi = last;
    const uint32_t pk = bin_of(a[first]);
    const int lane = tid & 63, wv = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto group_sync = [&]() { if (NT > 64) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); };
    /* exclusive rank of a flagged thread inside the group's block + the block's total */
    auto block_rank = [&](bool f, uint32_t& total) -> uint32_t {
        const unsigned long long m = __ballot(f);
        uint32_t r = (uint32_t)__popcll(m & lt);
        if (NT > 64) {
            __syncthreads();
            if (lane == 0) wcnt[wv] = __popcll(m);
            __syncthreads();
            uint32_t before_ = 0, all = 0;
            for (int k = 0; k < NT / 64; k++) { const uint32_t c = (uint32_t)wcnt[k]; if (k < wv) before_ += c; all += c; }
            r += before_;
            total = all;
        } else total = (uint32_t)__popcll(m);
        return r;
    };
    uint32_t cntL = 0, cntR = 0;
    for (uint32_t base = lo; base < hi; base += NT) {                      /* left stoppers, ascending */
        const uint32_t p = base + tid;
        const bool f = p < hi && !(bin_of(a[p]) > pk);
        uint32_t tot;
        const uint32_t r = block_rank(f, tot);
        if (f) posL[first + cntL + r] = p;
        cntL += tot;
    }
    for (uint32_t off = 0; lo + off < hi; off += NT) {                     /* right stoppers, descending */
        const uint32_t back = off + tid;
        const bool in = back < hi - lo;
        const uint32_t p = hi - 1 - (in ? back : 0);
        const bool f = in && !(pk > bin_of(a[p]));
        uint32_t tot;
        const uint32_t r = block_rank(f, tot);
        if (f) posR[first + cntR + r] = p;
        cntR += tot;
    }
    group_sync();
    /* K = pairs still in order: a prefix of the rank order */
    const uint32_t m = cntL < cntR ? cntL : cntR;
    uint32_t K = 0;
    for (uint32_t base = 0; base < m; base += NT) {
        const uint32_t k = base + tid;
        const bool f = k < m && posL[first + k] < posR[first + k];
        uint32_t tot;
        (void)block_rank(f, tot);
        K += tot;
        if (tot < (uint32_t)NT && base + NT < m) break;                   /* the prefix ended inside this block */
    }
    for (uint32_t k = tid; k < K; k += NT) {
        const uint32_t pl = posL[first + k], pr = posR[first + k];
        const uint32_t x = a[pl], y = a[pr];
        a[pl] = y; a[pr] = x;
    }
    const uint32_t l = K < cntL ? posL[first + K] : hi, r = K > 0 ? posR[first + K - 1] : hi;
    group_sync();
    return l < r ? l : r;
}

} // namespace

extern "C" __global__ __launch_bounds__(ORD_T) void k_lsd_order(uint32_t* __restrict__ keysBase, size_t keyStride, int n,
                                                                uint32_t* __restrict__ posLBase, uint32_t* __restrict__ posRBase,
                                                                size_t posStride, int* __restrict__ statusBase, int statusStride,
                                                                int depthLimit)
{
    extern __shared__ uint32_t dyn[];                  /* counting passes: 32 x ORD_T counters */
    __shared__ Seg queue[ORD_QCAP];
    __shared__ Seg stack[ORD_WAVES][ORD_STACK];
    __shared__ int qHead, qTail, qOverflow, heapNeeded, wcnt[ORD_WAVES + 2];
    __shared__ uint32_t cutShared;
    uint32_t* a = keysBase + keyStride * blockIdx.x;
    uint32_t* posL = posLBase + posStride * blockIdx.x;
    uint32_t* posR = posRBase + posStride * blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { qHead = 0; qTail = 0; qOverflow = 0; heapNeeded = 0; }
    __syncthreads();
    if (n > 16 && tid == 0) { queue[0].first = 0; queue[0].last = (uint32_t)n; queue[0].depth = depthLimit; qTail = 1; }
    __syncthreads();

    /* ---- ranges above ORD_BIG: the workgroup partitions them one after the other ---- */
    for (;;) {
        /* find a big range in the queue (thread 0), move it out by swapping with the head */
        if (tid == 0) {
            int found = -1;
            for (int k = qHead; k < qTail; k++) if (queue[k].last - queue[k].first > ORD_BIG) { found = k; break; }
            if (found >= 0) { const Seg s = queue[found]; queue[found] = queue[qHead]; queue[qHead] = s; qHead++; cutShared = 1; }
            else cutShared = 0;
        }
        __syncthreads();
        if (!cutShared) break;
        const Seg s = queue[qHead - 1];
        __syncthreads();
        if (s.depth == 0) { if (tid == 0) heapNeeded = 1; continue; }
        if (tid == 0) median_to_first(a, s.first, s.first + 1, s.first + (s.last - s.first) / 2, s.last - 1);
        __syncthreads();
        const uint32_t cut = hoare_cut<ORD_T>(a, s.first, s.last, posL, posR, tid, wcnt);
        if (tid == 0) {
            if (s.last - cut > 16) { if (qTail < ORD_QCAP) { queue[qTail].first = cut; queue[qTail].last = s.last; queue[qTail].depth = s.depth - 1; qTail++; } else qOverflow = 1; }
            if (cut - s.first > 16) { if (qTail < ORD_QCAP) { queue[qTail].first = s.first; queue[qTail].last = cut; queue[qTail].depth = s.depth - 1; qTail++; } else qOverflow = 1; }
        }
        __syncthreads();
    }

    /* ---- everything else: a wavefront takes a range and finishes it depth-first ---- */
    for (;;) {
        int q = 0;
        if (lane == 0) q = atomicAdd(&qHead, 1);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= qTail) break;                                  /* qTail is final: only the stage above appends */
        int sp = 0;
        Seg s = queue[q];
        for (;;) {
            /* std::__introsort_loop on s */
            while (s.last - s.first > 16) {
                if (s.depth == 0) { if (lane == 0) heapNeeded = 1; break; }
                s.depth--;
                if (lane == 0) median_to_first(a, s.first, s.first + 1, s.first + (s.last - s.first) / 2, s.last - 1);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                const uint32_t cut = hoare_cut<64>(a, s.first, s.last, posL, posR, lane, nullptr);
                if (s.last - cut > 16) {
                    if (sp < ORD_STACK) {
                        if (lane == 0) { stack[wv][sp].first = cut; stack[wv][sp].last = s.last; stack[wv][sp].depth = s.depth; }
                        sp++;
                    } else if (lane == 0) qOverflow = 1;
                }
                s.last = cut;
            }
            if (sp == 0) break;
            sp--;
            s = stack[wv][sp];
        }
    }
    __syncthreads();

    /* ---- std::__final_insertion_sort = the stable sort by descending bin: two stable counting passes (5 bits each) ---- */
    uint32_t* src = a;
    uint32_t* dst = posL;
    const uint32_t E = ((uint32_t)n + ORD_T - 1) / ORD_T;
    const uint32_t c0 = min((uint32_t)n, (uint32_t)tid * E), c1 = min((uint32_t)n, c0 + E);
    for (int pass = 0; pass < 2; pass++) {
        const int sh = pass * 5;
        for (int b = 0; b < 32; b++) dyn[b * ORD_T + tid] = 0;
        for (uint32_t p = c0; p < c1; p++) dyn[(((1023u - bin_of(src[p])) >> sh) & 31u) * ORD_T + tid]++;
        __syncthreads();
        /* exclusive scan of the 32 x ORD_T counters in (digit, thread) order: thread i owns entries [32 i, 32 i + 32) */
        uint32_t loc = 0;
        for (int k = 0; k < 32; k++) loc += dyn[tid * 32 + k];
        uint32_t inc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (lane == 63) wcnt[wv] = (int)inc;
        __syncthreads();
        uint32_t wbase = 0;
        for (int k = 0; k < wv; k++) wbase += (uint32_t)wcnt[k];
        uint32_t run = wbase + inc - loc;
        for (int k = 0; k < 32; k++) { const uint32_t c = dyn[tid * 32 + k]; dyn[tid * 32 + k] = run; run += c; }
        __syncthreads();
        for (uint32_t p = c0; p < c1; p++) {
            const uint32_t v = src[p];
            const uint32_t slot = (((1023u - bin_of(v)) >> sh) & 31u) * ORD_T + tid;
            dst[dyn[slot]++] = v;
        }
        __syncthreads();
        uint32_t* t = src; src = dst; dst = t;
    }
    if (tid == 0) statusBase[(size_t)statusStride * blockIdx.x] = (heapNeeded ? 1 : 0) | (qOverflow ? 2 : 0);
}

hipError_t drfe_launch_lsd_order(uint32_t* d_keys, size_t keyStride, int n, uint32_t* d_posL, uint32_t* d_posR, size_t posStride,
                                 int* d_status, int statusStride, int nframes, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    int lg = 0;
    for (size_t v = (size_t)n; v > 1; v >>= 1) lg++;
    const size_t lds = 32 * ORD_T * sizeof(uint32_t);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lsd_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k_lsd_order, dim3(nframes), dim3(ORD_T), lds, s, d_keys, keyStride, n, d_posL, d_posR, posStride, d_status,
                       statusStride, 2 * lg);
    return hipGetLastError();
}
