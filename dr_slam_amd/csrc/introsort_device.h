/* introsort_device.h — libstdc++'s std::sort (introsort) as a device routine: the element moves of bits/stl_algo.h executed by
 * one workgroup on an array in HBM, so that the PERMUTATION OF EQUAL KEYS is std::sort's.  Two callers need
 * exactly that (the same two introsort_restated.h serves on the host):
 *   LSD's pseudo-ordering     std::sort(ordered_points, compare_norm) of cv::LineSegmentDetectorImpl (lsd_order_kernels.hip)
 *   pcl::VoxelGrid            std::sort(index_vector) on the leaf index (voxel_kernels.hip)
 * Traits: `Rec` (record type), `key(Rec)` -> uint32 with "a goes before b" == key(a) < key(b), `KEY_BITS`.
 *   - median-of-three Hoare partitions down to ranges of 16 under a depth limit of 2 lg n: the two sub-ranges a partition leaves are
 *     independent, so ranges above ORD_BIG records are partitioned by the whole workgroup one after the other and everything
 *     below is dealt to the sixteen wavefronts, each of which finishes its range depth-first;
 *   - a Hoare partition swaps the k-th record from the left that does not go before the pivot ("left stopper") with the k-th
 *     from the right the pivot does not go before, while the former lies left of the latter, and cuts at min(L[K], R[K-1]):
 *     stopper positions are compacted in rank order by ballot / popcount prefix sums, K is the length of the prefix of pairs
 *     still in order, and the K swaps are independent;
 *   - the final insertion sort moves a record left past records it goes before only - it is the stable sort of what the
 *     partitions left: stable counting passes over five key bits each.
 *   - what one wavefront writes and reads back (a range it owns, in HBM or LDS) needs no wait: a wavefront's memory operations
 *     are performed in order, so the fences between its steps are wavefront-scope (compiler ordering only); data changes hands
 *     between wavefronts at the workgroup barriers only.
 *   - a range that exhausts the depth limit is heap-sorted the way libstdc++ does it (std::__partial_sort(first, last, last) =
 *     std::__make_heap + std::__sort_heap, every __adjust_heap / __push_heap move in its order) by ONE lane: the moves of a heap
 *     sort are one dependent chain.  On the data this serves such ranges are a few dozen records (25 on the synthetic planes
 *     that reach the branch: depth 2 lg n is only exhausted far down the recursion); ranges above ORD_HEAP_MAX records - never
 *     seen - set bit 0 of the returned status instead and the caller sorts on the host, so that one lane never holds a CU for
 *     seconds. */
#ifndef DRFE_INTROSORT_DEVICE_H
#define DRFE_INTROSORT_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

/* NTH = threads of the workgroup (a multiple of 64, template parameter): 1024 for the 196 000 keys of a frame's pixel ordering,
 * 256 for a plane's voxel records - a workgroup's LDS is 128 bytes per thread for the counting passes plus the range queue, and
 * a 1024-thread sort fills a CU's LDS alone (no other kernel's wavefronts beside it) */
#ifndef ORD_BIG
#define ORD_BIG 8192             /* ranges above this many records: one at a time by the whole workgroup */
#endif
#ifndef ORD_BIG_SHIFT
#define ORD_BIG_SHIFT 4
#endif
#ifndef ORD_BIG_FLOOR
#define ORD_BIG_FLOOR 1024
#endif
#define ORD_QCAP 256             /* pending ranges an array can hold (LDS, 12 bytes each).  A workgroup partition leaves at most two and the
                                  * wavefront phase adds none: ~2 n / ORD_BIG entries for keys that split evenly (48 for a frame's
                                  * 196 000 pixels); with 1024 entries a sort held 47.5 KB of LDS - three per CU - with 256 it is
                                  * 38.3 KB, four per CU.  An overflow flags the array for the host. */
#define ORD_STACK 48             /* depth-first stack of a wavefront (>= the depth limit 2 lg n of any array that fits) */
#ifndef ORD_DYN_WORDS
#define ORD_DYN_WORDS 16         /* LDS words per thread: a wavefront's share (4 KB) holds the ranges it partitions in LDS.  Measured per
                                  * frame / per plane cloud (k_lsd_order / k_voxel_grid, ms) at 32 | 24 | 16 | 12 | 8 words: 14.1 / 1.99 |
                                  * 14.5 / 2.09 | 15.4 / 2.27 | 16.0 / 2.38 | 18.3 / 2.63 - and a 256-thread sort holds 38.4 | 30.7 | 21.8 |
                                  * 17.9 | 14.1 KB of its CU (4 | 5 | 7 | 9 | 11 per CU): LDS x time is what the mix is short of */
#endif
#define ORD_DYN_LDS_BYTES(NTH) (ORD_DYN_WORDS * (NTH) * 4)
#define ORD_HEAP_MAX 1024        /* longest range one lane heap-sorts (about 20 000 dependent moves) */

/* phase marks of a profiling build: the includer defines ISD_TP(k) (k = 0: workgroup partitions done, 1: wavefront phase done,
 * 2: counting passes done) */
#ifndef ISD_TP
#define ISD_TP(k)
#endif
/* wavefront-phase split of a profiling build (wavefront 0 only): ISD_WT0() marks, ISD_WT(k) adds the time since the mark to slot k
 * (0 partitions in HBM, 1 copy into LDS, 2 wavefront partitions in LDS, 3 one range per lane, 4 copy back) and marks again */
#ifndef ISD_WT
#define ISD_WT0()
#define ISD_WT(k)
#endif

namespace isd {

struct Seg { uint32_t first, last; int depth; };

/* LDS state of one sort (declare one per kernel: __shared__ isd::Shared<NTH> sh;) */
template <int NTH>
struct Shared {
    Seg queue[ORD_QCAP];
    Seg stack[NTH / 64][ORD_STACK];
    int qHead, qTail, bigTop, qOverflow, heapNeeded, wcnt[8 * (NTH / 64) + 2];
    Seg cur;                        /* the range the workgroup partitions now */
    uint32_t cutShared;
    unsigned long long tp;          /* profiling builds: thread 0's last phase mark */
};

/* std::__move_median_to_first(result, x, y, z) by one thread */
template <class T, class RecPtr>
__device__ __forceinline__ void median_to_first(RecPtr a, uint32_t result, uint32_t x, uint32_t y, uint32_t z)
{
    const uint32_t kx = T::key(a[x]), ky = T::key(a[y]), kz = T::key(a[z]);
    uint32_t pick;
    if (kx < ky) {
        if (ky < kz) pick = y;
        else if (kx < kz) pick = z;
        else pick = x;
    } else if (kx < kz) pick = x;
    else if (ky < kz) pick = z;
    else pick = y;
    const typename T::Rec t = a[result];
    a[result] = a[pick];
    a[pick] = t;
}

/* std::__adjust_heap(a + first, hole, len, value, before) of bits/stl_heap.h followed by its std::__push_heap, by one thread */
template <class T, class RecPtr>
__device__ __forceinline__ void adjust_heap(RecPtr a, uint32_t first, int hole, int len, typename T::Rec value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (T::key(a[first + child]) < T::key(a[first + child - 1])) child--;
        a[first + hole] = a[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[first + hole] = a[first + child - 1];
        hole = child - 1;
    }
    const uint32_t kv = T::key(value);
    int parent = (hole - 1) / 2;
    while (hole > top && T::key(a[first + parent]) < kv) {
        a[first + hole] = a[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[first + hole] = value;
}

/* std::__partial_sort(a + first, a + last, a + last, before): __heap_select with middle == last is __make_heap alone, then
 * __sort_heap.  One thread. */
template <class T, class RecPtr>
__device__ __forceinline__ void heap_sort_range(RecPtr a, uint32_t first, uint32_t last)
{
    const int len = (int)(last - first);
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; parent--) {
        const typename T::Rec v = a[first + parent];
        adjust_heap<T>(a, first, parent, len, v);
        if (parent == 0) break;
    }
    for (int end = len - 1; end >= 1; end--) {                 /* __pop_heap(first, end, end) */
        const typename T::Rec v = a[first + end];
        a[first + end] = a[first];
        adjust_heap<T>(a, first, 0, end, v);
    }
}

/* std::__introsort_loop on a[first, last), at most ORD_TINY records, by ONE LANE (every lane of the wavefront its own range, side
 * by side): median-of-three, the textbook __unguarded_partition, the right part parked while the left one is finished.  With at
 * most ORD_TINY records (ORD_TINY + 16) / 17 parts wait at most (every parked part has 17 records or more and the part being split
 * at least one more).  Returns false if that bound failed (never: the caller flags the array). */
#ifndef ORD_TINY
#define ORD_TINY 64
#endif
#define ORD_TINY_STACK ((ORD_TINY + 16) / 17)
template <class T, class RecPtr>
__device__ __forceinline__ bool lane_introsort(RecPtr a, uint32_t first, uint32_t last, int depth)
{
    uint32_t e[ORD_TINY_STACK];                /* parked ranges: first | last << 11 | depth << 22; e[0] = the newest (static indices only) */
#pragma unroll
    for (int k = 0; k < ORD_TINY_STACK; k++) e[k] = 0;
    int sp = 0;
    for (;;) {
        while (last - first > 16) {
            if (depth == 0) { heap_sort_range<T>(a, first, last); break; }
            depth--;
            median_to_first<T>(a, first, first + 1, first + (last - first) / 2, last - 1);
            const uint32_t pk = T::key(a[first]);
            uint32_t i = first + 1, j = last;
            for (;;) {
                while (T::key(a[i]) < pk) i++;
                j--;
                while (pk < T::key(a[j])) j--;
                if (!(i < j)) break;
                const typename T::Rec x = a[i], y = a[j];
                a[i] = y; a[j] = x;
                i++;
            }
            if (last - i > 16) {
                if (sp >= ORD_TINY_STACK) return false;
#pragma unroll
                for (int k = ORD_TINY_STACK - 1; k > 0; k--) e[k] = e[k - 1];
                e[0] = i | last << 11 | (uint32_t)depth << 22; sp++;
            }
            last = i;
        }
        if (sp == 0) return true;
        first = e[0] & 0x7FFu; last = (e[0] >> 11) & 0x7FFu; depth = (int)(e[0] >> 22);
#pragma unroll
        for (int k = 0; k + 1 < ORD_TINY_STACK; k++) e[k] = e[k + 1];
        sp--;
    }
}

/* std::__unguarded_partition(a + first + 1, a + last, a + first) by a group of NT threads (64: one wavefront; NTH: the
 * workgroup).  tid = thread index inside the group.  posL / posR: scratch of the range's length at [first, last).  wcnt: LDS,
 * NT / 64 + 2 ints (workgroup variant).  Returns the cut to every thread. */
template <int NT, class T, class RecPtr, class PosPtr>
__device__ __forceinline__ uint32_t hoare_cut(RecPtr a, uint32_t first, uint32_t last, PosPtr posL, PosPtr posR, int tid, int* wcnt)
{
    const uint32_t lo = first + 1, hi = last;
    const uint32_t pk = T::key(a[first]);
    const int lane = tid & 63, wv = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto group_sync = [&]() { if (NT > 64) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); };
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
    /* Round 6: ONE scan finds both kinds of stoppers - a record's key decides "does not go before the pivot" (left stopper) and "the
     * pivot does not go before it" (right stopper) alike - so every record is read once per partition instead of twice and a round's
     * chain (load -> ballot -> barrier -> LDS -> barrier -> store) carries both lists.  The right stoppers land in ASCENDING position
     * order; the k-th from the right - what the pairing below asks for - is entry cntR - 1 - k. */
    uint32_t cntL = 0, cntR = 0;
    if (NT == 64) {
        /* one wavefront: four blocks of 64 per round, their loads in flight together (the scan is bound by memory latency) */
        for (uint32_t base = lo; base < hi; base += 256) {
            bool fl[4], fr[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t p = base + 64 * u + lane;
                const uint32_t k = T::key(a[p < hi ? p : lo]);
                fl[u] = p < hi && !(k < pk);
                fr[u] = p < hi && !(pk < k);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned long long ml = __ballot(fl[u]), mr = __ballot(fr[u]);
                if (fl[u]) posL[first + cntL + (uint32_t)__popcll(ml & lt)] = base + 64 * u + lane;
                if (fr[u]) posR[first + cntR + (uint32_t)__popcll(mr & lt)] = base + 64 * u + lane;
                cntL += (uint32_t)__popcll(ml);
                cntR += (uint32_t)__popcll(mr);
            }
        }
    } else {
        /* the workgroup: four blocks of NT per round - their loads in flight together and ONE pair of barriers for the four
         * (a round is a chain load -> ballot -> barrier -> LDS -> barrier -> store, ~1 us whatever it carries) */
        constexpr int NWV = NT / 64;
        for (uint32_t base = lo; base < hi; base += 4 * NT) {
            bool fl[4], fr[4];
            unsigned long long ml[4], mr[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t p = base + NT * u + tid;
                const uint32_t k = T::key(a[p < hi ? p : lo]);
                fl[u] = p < hi && !(k < pk);
                fr[u] = p < hi && !(pk < k);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { ml[u] = __ballot(fl[u]); mr[u] = __ballot(fr[u]); }
            __syncthreads();
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 4; u++) { wcnt[u * NWV + wv] = __popcll(ml[u]); wcnt[(4 + u) * NWV + wv] = __popcll(mr[u]); }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t bl = 0, al = 0, br = 0, ar = 0;
                for (int k = 0; k < NWV; k++) {
                    const uint32_t c = (uint32_t)wcnt[u * NWV + k], d = (uint32_t)wcnt[(4 + u) * NWV + k];
                    if (k < wv) { bl += c; br += d; }
                    al += c; ar += d;
                }
                if (fl[u]) posL[first + cntL + bl + (uint32_t)__popcll(ml[u] & lt)] = base + NT * u + tid;
                if (fr[u]) posR[first + cntR + br + (uint32_t)__popcll(mr[u] & lt)] = base + NT * u + tid;
                cntL += al;
                cntR += ar;
            }
        }
    }
    group_sync();
    /* K = pairs still in order: a prefix of the rank order */
    const uint32_t m = cntL < cntR ? cntL : cntR;
    /* "the k-th left stopper lies left of the k-th right stopper" holds for a prefix of k (one list ascends, the other descends):
     * its length by an NT-ary search, log_NT(m) rounds of one probe per thread */
    uint32_t klo = 0, khi = m;
    while (klo < khi) {
        const uint32_t span = khi - klo, step = (span + NT - 1) / NT, np = (span + step - 1) / step;
        const uint32_t k = klo + (uint32_t)tid * step;
        const bool f = k < khi && posL[first + k] < posR[first + cntR - 1 - k];
        uint32_t c;
        (void)block_rank(f, c);
        const uint32_t nlo = c > 0 ? klo + (c - 1) * step + 1 : klo, nhi = c < np ? klo + c * step : khi;
        klo = nlo; khi = nhi;
    }
    const uint32_t K = klo;
    for (uint32_t k0 = tid; k0 < K; k0 += 4 * NT) {
        uint32_t pl[4], pr[4];
        typename T::Rec x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t k = k0 + NT * u; const bool in = k < K; pl[u] = in ? (uint32_t)posL[first + k] : first; pr[u] = in ? (uint32_t)posR[first + cntR - 1 - k] : first; }
#pragma unroll
        for (int u = 0; u < 4; u++) { x[u] = a[pl[u]]; y[u] = a[pr[u]]; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (k0 + NT * u < K) { a[pl[u]] = y[u]; a[pr[u]] = x[u]; }
    }
    const uint32_t l = K < cntL ? (uint32_t)posL[first + K] : hi, r = K > 0 ? (uint32_t)posR[first + cntR - K] : hi;
    group_sync();
    return l < r ? l : r;
}

/* a[0..n) into std::sort's order (the NTH threads of the workgroup call this together).  posL / posR: n uint32 each; tmp: n records
 * (ping-pong buffer of the counting passes); dyn: ORD_DYN_LDS_BYTES(NTH) of LDS; keyBits: significant bits of the keys present.
 * Returns (to every thread) 0, or bit 0 = a range above ORD_HEAP_MAX records needs heap sort, bit 1 = internal queue overflow or a loop bound exceeded (bits 2-4: which):
 * result unusable. */
template <int NTH, class T>
__device__ __forceinline__ int sort(typename T::Rec* a, int n, uint32_t* posL, uint32_t* posR, typename T::Rec* tmp, uint32_t* dyn, Shared<NTH>& sh,
                                    int depthLimit, int keyBits)
{
    typedef typename T::Rec Rec;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __syncthreads();
    /* one array, two lists: ranges of at most ORD_BIG records (the wavefronts' work) fill it from the bottom [qHead, qTail), ranges
     * above ORD_BIG (the workgroup's) are a stack growing down from the top [bigTop, ORD_QCAP) - no search for the next large
     * range, and a consumed entry's slot is free again */
    if (tid == 0) { sh.qHead = 0; sh.qTail = 0; sh.bigTop = ORD_QCAP; sh.qOverflow = 0; sh.heapNeeded = 0; }
    __syncthreads();
    /* what the workgroup partitions together: ranges above ORD_BIG records - and, in an array of fewer than 16 ORD_BIG records,
     * above a sixteenth of it (not below 1024), so that a plane cloud of a few thousand points still leaves every wavefront a
     * few ranges instead of the whole array to the one that takes it */
    const uint32_t bigAbove = min((uint32_t)ORD_BIG, max((uint32_t)ORD_BIG_FLOOR, (uint32_t)n >> ORD_BIG_SHIFT));
    auto push_range = [&](uint32_t f, uint32_t l, int d) {                /* thread 0 */
        if (l - f <= 16) return;
        if (sh.qTail >= sh.bigTop) { sh.qOverflow |= 1; return; }
        const int at = l - f > bigAbove ? --sh.bigTop : sh.qTail++;
        sh.queue[at].first = f; sh.queue[at].last = l; sh.queue[at].depth = d;
    };
    if (tid == 0) push_range(0, (uint32_t)n, depthLimit);
    __syncthreads();

    /* ---- ranges above ORD_BIG: the workgroup partitions them one after the other ---- */
    uint32_t wgIter = 0;
    for (;;) {
        if (tid == 0) {
            if (sh.bigTop < ORD_QCAP) { sh.cur = sh.queue[sh.bigTop]; sh.bigTop++; sh.cutShared = 1; }
            else sh.cutShared = 0;
        }
        __syncthreads();
        /* the loop's conditions come out of LDS: made wave-uniform scalars, so that the branches around the barriers below are scalar
         * branches and not EXEC-masked regions a wavefront could skip */
        if (!__builtin_amdgcn_readfirstlane((int)sh.cutShared)) break;
        /* every data-dependent loop of this routine carries a bound far above what a correct run needs: a workgroup that ran past it
         * would otherwise hold its CU for ever (bit 2 / 3 / 4 of the status: which loop) */
        if (++wgIter > 4u * ORD_QCAP + 64u) { if (tid == 0) sh.qOverflow |= 4; break; }
        Seg s = sh.cur;
        s.first = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.first); s.last = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.last);
        s.depth = __builtin_amdgcn_readfirstlane(s.depth);
        __syncthreads();
        /* every thread runs the same barriers on both arms and meets the others at the one below: no `continue` around it.  (With a
         * `continue` here a workgroup stopped for good the first time a range above ORD_BIG ran out of depth - a 10 270-record
         * range of a 12 905-point plane cloud, found by a parity soak; ranges that run out of depth in the wavefront phase, the
         * common case, were never affected.) */
        if (s.depth == 0) { if (tid == 0) sh.heapNeeded = 1; }            /* above ORD_BIG > ORD_HEAP_MAX records: the host's */
        else {
            if (tid == 0) median_to_first<T>(a, s.first, s.first + 1, s.first + (s.last - s.first) / 2, s.last - 1);
            __syncthreads();
            const uint32_t cut = hoare_cut<NTH, T>(a, s.first, s.last, posL, posR, tid, sh.wcnt);
            if (tid == 0) { push_range(cut, s.last, s.depth - 1); push_range(s.first, cut, s.depth - 1); }
        }
        __syncthreads();
    }

    ISD_TP(0);
    /* ---- everything else: a wavefront takes a range and finishes it depth-first.  A range of at most LCAP records moves into the
     * wavefront's share of the dynamic LDS (records + 16-bit stopper positions) and is partitioned there down to the ranges of 16:
     * the partitions of small ranges are chains of dependent accesses, a few per range, and most ranges are small ---- */
    constexpr uint32_t LBYTES = ORD_DYN_LDS_BYTES(NTH) / (NTH / 64);
    constexpr uint32_t LCAP = (LBYTES - 256) / (sizeof(Rec) + 4);
    static_assert(LCAP < 2048, "lane_introsort packs positions into 11 bits");
    Rec* lrec = (Rec*)((uint8_t*)dyn + (size_t)wv * LBYTES);
    uint16_t* lposL = (uint16_t*)(lrec + LCAP);
    uint16_t* lposR = lposL + LCAP;
    /* ranges of at most ORD_TINY records met inside the LDS block: not partitioned by the wavefront (a partition is a chain of a
     * few dependent LDS accesses whatever its size, and three partitions in four are of such ranges) but listed here, and when the
     * block's larger ranges are done every lane finishes one of them on its own (lane_introsort) */
    uint32_t* tiny = (uint32_t*)((uint8_t*)dyn + (size_t)wv * LBYTES + LBYTES - 256);
    uint32_t waveIter = 0;
    unsigned long long wtMark = 0; (void)wtMark;         /* profiling builds */
    const uint32_t waveMax = 8u * (uint32_t)n + 4096u;           /* partitions one wavefront can legitimately run: far fewer */
    bool runaway = false;
    for (; !runaway;) {
        int q = 0;
        if (lane == 0) q = atomicAdd(&sh.qHead, 1);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= sh.qTail) break;                               /* qTail is final: only the stage above appends */
        int sp = 0;
        Seg s = sh.queue[q];
        for (; !runaway;) {
            /* std::__introsort_loop on s */
            while (s.last - s.first > 16) {
                if (++waveIter > waveMax) { if (lane == 0) sh.qOverflow |= 8; runaway = true; break; }
                const uint32_t m = s.last - s.first;
                if (s.depth == 0) {
                    if (m > ORD_HEAP_MAX) { if (lane == 0) sh.heapNeeded = 1; }
                    else {
                        if (lane == 0) heap_sort_range<T>(a, s.first, s.last);
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    }
                    break;
                }
                if (m <= LCAP) {
                    ISD_WT0();
                    for (uint32_t i = lane; i < m; i += 64) lrec[i] = a[s.first + i];
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    ISD_WT(1);
                    const int base = sp;
                    int ntiny = 0;
                    Seg t; t.first = 0; t.last = m; t.depth = s.depth;
                    for (; !runaway;) {
                        while (t.last - t.first > 16) {
                            if (++waveIter > waveMax) { if (lane == 0) sh.qOverflow |= 16; runaway = true; break; }
                            if (t.last - t.first <= ORD_TINY && ntiny < 64) {
                                if (lane == 0) tiny[ntiny] = t.first | t.last << 11 | (uint32_t)t.depth << 22;
                                ntiny++;
                                break;
                            }
                            if (t.depth == 0) {
                                if (lane == 0) heap_sort_range<T>(lrec, t.first, t.last);
                                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                                break;
                            }
                            t.depth--;
                            if (lane == 0) median_to_first<T>(lrec, t.first, t.first + 1, t.first + (t.last - t.first) / 2, t.last - 1);
                            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                            const uint32_t cut = hoare_cut<64, T>(lrec, t.first, t.last, lposL, lposR, lane, nullptr);
                            if (t.last - cut > 16) {
                                if (sp < ORD_STACK) {
                                    if (lane == 0) { sh.stack[wv][sp].first = cut; sh.stack[wv][sp].last = t.last; sh.stack[wv][sp].depth = t.depth; }
                                    sp++;
                                } else if (lane == 0) sh.qOverflow |= 1;
                            }
                            t.last = cut;
                        }
                        if (sp == base) break;
                        sp--;
                        t = sh.stack[wv][sp];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    ISD_WT(2);
                    if (lane < ntiny) {
                        const uint32_t e = tiny[lane];
                        if (!lane_introsort<T>(lrec, e & 0x7FFu, (e >> 11) & 0x7FFu, (int)(e >> 22))) sh.qOverflow |= 32;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    ISD_WT(3);
                    for (uint32_t i = lane; i < m; i += 64) a[s.first + i] = lrec[i];
                    ISD_WT(4);
                    break;
                }
                s.depth--;
                ISD_WT0();
                if (lane == 0) median_to_first<T>(a, s.first, s.first + 1, s.first + (s.last - s.first) / 2, s.last - 1);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                const uint32_t cut = hoare_cut<64, T>(a, s.first, s.last, posL, posR, lane, nullptr);
                ISD_WT(0);
                if (s.last - cut > 16) {
                    if (sp < ORD_STACK) {
                        if (lane == 0) { sh.stack[wv][sp].first = cut; sh.stack[wv][sp].last = s.last; sh.stack[wv][sp].depth = s.depth; }
                        sp++;
                    } else if (lane == 0) sh.qOverflow |= 1;
                }
                s.last = cut;
            }
            if (sp == 0) break;
            sp--;
            s = sh.stack[wv][sp];
        }
    }
    __syncthreads();
    ISD_TP(1);

    /* ---- std::__final_insertion_sort = the stable sort by key: stable counting passes, five bits each.  A wavefront owns a
     * contiguous share of the array and takes it in tiles of 64 consecutive records (one coalesced load): the lanes of a tile that
     * hold the same digit find each other with five ballots, the first of them adds their number to the digit's counter of this
     * wavefront (32 x NTH / 64 counters in LDS - 512 bytes at 256 threads, where a counter per thread and digit took 32 KB and every
     * thread walked its own stretch of the array, 64 cache lines per load), and in the second sweep a record's place is the
     * counter's value plus its rank among those lanes.  Order within a digit = wavefront, tile, lane: stable. ---- */
    Rec* src = a;
    Rec* dst = tmp;
    constexpr int NW = NTH / 64;
    const uint32_t tiles = ((uint32_t)n + 63u) >> 6, tilesPerWave = (tiles + NW - 1) / NW;
    const uint32_t t0 = min(tiles, (uint32_t)wv * tilesPerWave), t1 = min(tiles, t0 + tilesPerWave);
    uint32_t* cnt = dyn;                                    /* [digit][wavefront] */
    const unsigned long long ltm = (1ull << lane) - 1ull;
    /* the lanes of the tile whose digit equals this lane's (valid lanes only) */
    auto same_digit = [&](bool valid, uint32_t d) -> unsigned long long {
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 5; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        return peers;
    };
    const int passes = (keyBits + 4) / 5;
    for (int pass = 0; pass < passes; pass++) {
        const int shft = pass * 5;
        for (int i = tid; i < 32 * NW; i += NTH) cnt[i] = 0;
        __syncthreads();
        /* four tiles per trip, their loads in flight together: a sweep is a chain of load -> ballots -> LDS per tile, and the load is
         * most of it (round 6: one tile per trip waited a memory round trip 3 000 times per wavefront and sort) */
        for (uint32_t tb = t0; tb < t1; tb += 4) {
            uint32_t kk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t p = ((tb + u) << 6) + lane; kk[u] = (tb + u < t1 && p < (uint32_t)n) ? T::key(src[p]) : 0u; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (tb + u >= t1) break;                               /* wave-uniform */
                const uint32_t p = ((tb + u) << 6) + lane;
                const bool valid = p < (uint32_t)n;
                const uint32_t d = valid ? (kk[u] >> shft) & 31u : 0u;
                const unsigned long long peers = same_digit(valid, d);
                if (valid && !(peers & ltm)) atomicAdd(&cnt[d * NW + wv], (uint32_t)__popcll(peers));
            }
        }
        __syncthreads();
        /* exclusive scan of the counters in (digit, wavefront) order by wavefront 0: lane i owns NW / 2 consecutive entries */
        if (wv == 0) {
            constexpr int EPL = (32 * NW) / 64;
            uint32_t loc = 0;
#pragma unroll
            for (int k = 0; k < EPL; k++) loc += cnt[lane * EPL + k];
            const uint32_t inc = (uint32_t)drfe_wave_incl_scan((int)loc, lane);      /* wavefront 0 whole: wv is wave-uniform */
            uint32_t run = inc - loc;
#pragma unroll
            for (int k = 0; k < EPL; k++) { const uint32_t c = cnt[lane * EPL + k]; cnt[lane * EPL + k] = run; run += c; }
        }
        __syncthreads();
        for (uint32_t tb = t0; tb < t1; tb += 4) {
            Rec vv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t p = ((tb + u) << 6) + lane; vv[u] = (tb + u < t1 && p < (uint32_t)n) ? src[p] : src[0]; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (tb + u >= t1) break;                               /* wave-uniform */
                const uint32_t p = ((tb + u) << 6) + lane;
                const bool valid = p < (uint32_t)n;
                const Rec v = vv[u];
                const uint32_t d = valid ? (T::key(v) >> shft) & 31u : 0u;
                const unsigned long long peers = same_digit(valid, d);
                if (valid) {
                    const uint32_t base = cnt[d * NW + wv];
                    dst[base + (uint32_t)__popcll(peers & ltm)] = v;
                    if (!(peers & ltm)) cnt[d * NW + wv] = base + (uint32_t)__popcll(peers);
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   /* the next tile reads the counters this one advanced */
            }
        }
        __syncthreads();
        Rec* t = src; src = dst; dst = t;
    }
    if (src != a) {                                         /* odd number of passes: the result sits in tmp */
        for (uint32_t p = tid; p < (uint32_t)n; p += NTH) a[p] = src[p];
        __syncthreads();
    }
    ISD_TP(2);
    return (sh.heapNeeded ? 1 : 0) | (sh.qOverflow ? 2 : 0) | (sh.qOverflow & ~3);
}

} // namespace isd
#endif
