/* lsd_order_sort.h — LSD's pseudo-ordering, `std::sort(ordered_points, compare_norm)` of cv::LineSegmentDetectorImpl::ll_angle
 * (modules/imgproc/src/lsd.cpp of OpenCV 3.4), with the PERMUTATION libstdc++'s std::sort produces and a fraction of its
 * branch mispredictions.
 *
 * The comparator looks at the gradient bin only (descending), so a third of a million pixels share 1024 keys and the order of
 * equal bins - which decides the seed order of region growing, hence the segments - is whatever the sorting algorithm leaves.
 * The oracle calls std::sort; this file restates what std::sort does (libstdc++ bits/stl_algo.h: introsort = median-of-three
 * Hoare partitions down to ranges of 16, heap sort below a depth limit of 2 lg n, one final insertion sort) and executes the
 * same element moves differently:
 *   - a Hoare partition swaps the k-th element from the left that is not before the pivot with the k-th from the right that is
 *     not after it, for as long as the former lies left of the latter, and cuts at min(L[K], R[K-1]) (K swaps done).  The two
 *     stopper sets are found 64 elements at a time as bit masks (vector compares), the swaps read them off with ctz / clz: one
 *     unpredictable branch per 64 elements instead of one per element;
 *   - the final insertion sort moves an element left past strictly smaller bins only: it is THE stable sort of the array the
 *     partitions left, computed here as a counting sort over the 1024 bins;
 *   - the heap-sort fallback (never reached on image data, kept for exactness) is libstdc++'s own std::partial_sort.
 * tests/test_host_cpu.py compares the result with std::sort's on random, constant, sorted, organ-pipe and image-like key
 * arrays of many sizes (drfe_debug_lsd_order_sort); the GPU parity tests compare the segments with the oracle's. */
#ifndef DRFE_LSD_ORDER_SORT_H
#define DRFE_LSD_ORDER_SORT_H

#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <cstring>
#include <vector>
#include <immintrin.h>

namespace lsd_order {

enum { BIN_SHIFT = 22, NBINS = 1024 };           /* key = bin << 22 | y << 11 | x (lines_lsd.cpp) */
static inline uint32_t bin_of(uint32_t k) { return k >> BIN_SHIFT; }
/* compare_norm: a before b iff a's bin is larger */
struct Before { bool operator()(uint32_t a, uint32_t b) const { return bin_of(a) > bin_of(b); } };

/* std::__move_median_to_first(result, x, y, z) */
static inline void median_to_first(uint32_t* a, size_t result, size_t x, size_t y, size_t z)
{
    const Before before;
    if (before(a[x], a[y])) {
        if (before(a[y], a[z])) std::swap(a[result], a[y]);
        else if (before(a[x], a[z])) std::swap(a[result], a[z]);
        else std::swap(a[result], a[x]);
    } else if (before(a[x], a[z])) std::swap(a[result], a[x]);
    else if (before(a[y], a[z])) std::swap(a[result], a[z]);
    else std::swap(a[result], a[y]);
}

/* the final insertion sort = the stable sort by descending bin.  The array arrives in runs of <= 16 that are already in order
 * among themselves, so neighbours share bins and one counter per bin would serialise the scatter on its store-to-load chain:
 * four quarters of the array scatter side by side from their own start offsets. */
static inline void stable_by_bin(uint32_t* a, size_t n, std::vector<uint32_t>& tmp)
{
    enum { Q = 4 };
    static thread_local uint32_t cnt[Q][NBINS];
    std::memset(cnt, 0, sizeof(cnt));
    const size_t q = (n + Q - 1) / Q;
    size_t lo[Q], hi[Q];
    for (int s = 0; s < Q; s++) { lo[s] = std::min(n, (size_t)s * q); hi[s] = std::min(n, lo[s] + q); }
    for (size_t k = 0; k < q; k++)
        for (int s = 0; s < Q; s++)
            if (lo[s] + k < hi[s]) cnt[s][bin_of(a[lo[s] + k])]++;
    uint32_t run = 0;
    for (int b = NBINS - 1; b >= 0; b--)
        for (int s = 0; s < Q; s++) { const uint32_t c = cnt[s][b]; cnt[s][b] = run; run += c; }
    tmp.resize(n);
    uint32_t* out = tmp.data();
    for (size_t k = 0; k < q; k++)
        for (int s = 0; s < Q; s++)
            if (lo[s] + k < hi[s]) { const uint32_t v = a[lo[s] + k]; out[cnt[s][bin_of(v)]++] = v; }
    std::memcpy(a, out, n * sizeof(uint32_t));
}

#define LSD_ORDER_NS scalar_impl
#define LSD_ORDER_TARGET
#define LSD_ORDER_AVX2 0
#include "lsd_order_sort_impl.inc"
#undef LSD_ORDER_NS
#undef LSD_ORDER_TARGET
#undef LSD_ORDER_AVX2
#define LSD_ORDER_NS avx2_impl
#define LSD_ORDER_TARGET __attribute__((target("avx2,bmi,bmi2,lzcnt")))
#define LSD_ORDER_AVX2 1
#include "lsd_order_sort_impl.inc"
#undef LSD_ORDER_NS
#undef LSD_ORDER_TARGET
#undef LSD_ORDER_AVX2

/* a[0..n) into std::sort(a, a + n, Before())'s order.  mode: -1 best for this CPU, 0 scalar masks, 1 AVX2 masks.
 * depthOverride >= 0 replaces the depth limit 2 lg n (tests reach the heap-sort branch with it). */
static inline void sort(uint32_t* a, size_t n, std::vector<uint32_t>& tmp, int mode = -1, int depthOverride = -1)
{
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    if (n < 2) return;
    if (mode < 0) mode = avx2 ? 1 : 0;
    if (mode == 1 && avx2) avx2_impl::partition_phase(a, n, depthOverride);
    else scalar_impl::partition_phase(a, n, depthOverride);
    stable_by_bin(a, n, tmp);
}

/* the plain transcription of std::__introsort_loop / std::__final_insertion_sort with a settable depth limit: the checker of
 * the heap-sort branch in tests (at the natural depth limit it is compared with std::sort itself) */
static inline void reference_sort(uint32_t* a, size_t n, int depthOverride)
{
    const Before before;
    struct R {
        static size_t partition(uint32_t* a, size_t first, size_t last, size_t pivot, const Before& before)
        {
            for (;;) {
                while (before(a[first], a[pivot])) ++first;
                --last;
                while (before(a[pivot], a[last])) --last;
                if (!(first < last)) return first;
                std::swap(a[first], a[last]);
                ++first;
            }
        }
        static void loop(uint32_t* a, size_t first, size_t last, int depth, const Before& before)
        {
            while (last - first > 16) {
                if (depth == 0) { std::partial_sort(a + first, a + last, a + last, before); return; }
                --depth;
                median_to_first(a, first, first + 1, first + (last - first) / 2, last - 1);
                const size_t cut = partition(a, first + 1, last, first, before);
                loop(a, cut, last, depth, before);
                last = cut;
            }
        }
    };
    if (n < 2) return;
    int lg = 0;
    for (size_t v = n; v > 1; v >>= 1) lg++;
    R::loop(a, 0, n, depthOverride >= 0 ? depthOverride : 2 * lg, before);
    for (size_t k = 1; k < n; k++) {                 /* insertion sort: left past strictly smaller bins */
        const uint32_t v = a[k];
        size_t p = k;
        while (p > 0 && before(v, a[p - 1])) { a[p] = a[p - 1]; p--; }
        a[p] = v;
    }
}

}  // namespace lsd_order
#endif
