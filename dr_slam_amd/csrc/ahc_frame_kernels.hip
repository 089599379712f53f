/* ahc_frame_kernels.hip — PEAC's agglomerative plane extraction for a whole frame on the device: ahc::PlaneFitter::run after
 * the init-block fits (reference include/peac/AHCPlaneFitter.hpp: initGraph :760-880, ahCluster :986-1192, refineDetails with
 * findBlockMembership :488-590 and floodFill :431-479, the re-merge and relabel :299-382; restated for the host in
 * planes_ahc.cpp, which stays the low-latency single-frame path and the checker of this one).
 *
 * The algorithm is a chain of order-defined steps - a priority queue of nodes by plane-fit MSE, merges that rewrite the
 * neighbour lists, a FIFO flood fill whose arrival order decides the labels - so, like the line detector's region growing
 * (lsd_grow_kernels.hip), ONE WAVEFRONT runs one frame's sequence exactly, wave-uniform, and uses its lanes where the
 * sequence leaves room:
 *   - the trial merges of a popped node (one plane fit per neighbour: nine sums added, a 3 x 3 symmetric eigen-solve in f64)
 *     are independent: one lane each, folded in neighbour order afterwards;
 *   - rewriting the neighbours' lists after a merge touches a different list per neighbour: one lane each;
 *   - block membership, seeds, relabelling, member lists are data-parallel passes with ballot / popcount prefix sums;
 *   - the flood fill evaluates the four neighbours of a queue entry in four lanes (depth -> point -> distance to the plane),
 *     sixteen entries per step, then applies them in the reference's order: visits of the same pixel form chains (found by
 *     sorting the 64 pixel | lane keys across the wavefront: chain_sort64) and are applied in rounds by depth in the chain.
 * The priority queue lives in LDS as a 64-ary heap (its size never exceeds the number of init blocks; keys as floats, exact ties
 * from the nodes; the order is total, so the pop sequence does not depend on the heap's shape); nodes, neighbour lists, the union-find,
 * membership / distance maps and the flood-fill queue are in HBM.  Throughput comes from frames in flight: a launch carries one
 * wavefront per frame.  Overflowing any fixed capacity (neighbour pool, queue, planes) or an uncertified cosine flags the
 * frame and the host redoes it. */
#include "drfe_internal.h"
#include "planes_internal.h"
#include "ahc_math.h"
#include "cr_sincos.h"

/* Two instantiations of each kernel (template parameters HEAP, LIST):
 *   640 x 480-class frames (<= 3200 init blocks): HEAP 3200, LIST 376.  The longest neighbour list seen on 256 frames of the four
 *     scene kinds is 125; a plane that fills a 64 x 48 grid of blocks has ~224 neighbours.  k_ahc_cluster then holds 22 992 bytes
 *     of LDS - seven frames per CU - and k_ahc_refine 17 008: nine;
 *   up to 12 800 init blocks (BASELINE config 5: 1280 x 960 = 128 x 96 blocks): HEAP 12800, LIST 1024 (a plane that fills the
 *     grid has ~448 neighbours): 86 KB for k_ahc_cluster_big - one frame per CU, the queue still entirely in LDS.
 * HEAP = priority-queue capacity (>= init blocks); LIST = a neighbour list staged in LDS (a longer one hands the frame to the host) */
#define AHCD_HEAP_SMALL 3200
#define AHCD_LIST_SMALL 376
#define AHCD_HEAP_BIG 12800
#define AHCD_LIST_BIG 1024
#define AHCD_PIXBITS 21           /* a flood-fill queue entry is pixel | plane << 21: pixels below 2^21 (1280 x 960 = 1 228 800), planes below 128 */
#define AHCD_PIXMASK ((1u << AHCD_PIXBITS) - 1u)
#define AHCD_MAXEX 128            /* extracted planes before the re-merge */
/* the words k_ahc_cluster leaves for k_ahc_refine (AhcDevFrame::handoff, AHC_HANDOFF_INTS of planes_internal.h): [0] extracted
 * nodes, [1] flood-fill seeds, [2] nodes, [3] neighbour pool fill, then the node ids, their kept-block flags, phase timers */
#define AHCD_HO_EX 4
#define AHCD_HO_VALID (4 + AHCD_MAXEX)
#define AHCD_HO_TP (4 + 2 * AHCD_MAXEX)
#define AHCD_HO_LABELS (AHCD_HO_TP + 8)   /* 1: k_ahc_refine left the frame's raw labels, plane map and plane count for the k_ahc_labels_* kernels */
#define GLOBAL_AS __attribute__((address_space(1)))

namespace {

__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double rl_d(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni_b(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
/* one wavefront per workgroup: what a lane stores and another lane loads later goes through the same in-order memory pipeline, so
 * the fences between the steps are wavefront-scope - the compiler keeps the order, the hardware has nothing to wait for */
__device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
/* ordering between the LDS accesses of this one wavefront: the LDS serves a wavefront's instructions in order, so only the compiler
 * must not move them - no wait for the global stores in flight (the workgroup fence waits for those: ~1 us each) */
__device__ __forceinline__ void wave_order() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }


/* the key of lane ^ j without a trip through the LDS crossbar: DPP inside a row of sixteen lanes (quad permutes, row rotations), gfx950's
 * v_permlane16_swap / v_permlane32_swap across rows and halves.  A 64-key bitonic sort on these runs in 0.23 us against 0.72 us on
 * __shfl_xor (ds_bpermute for j = 4, 16, 32): tools/ubench_permlane.hip, which also checks the forms against __shfl_xor on the device. */
__device__ __forceinline__ uint32_t xor_partner(uint32_t key, int j, int lane)
{
    if (j == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xF, 0xF, false);          /* quad_perm [1, 0, 3, 2] */
    if (j == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xF, 0xF, false);          /* quad_perm [2, 3, 0, 1] */
    if (j == 4) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x124, 0xF, 0xF, false);     /* row_ror:4: from lane - 4 */
        const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x12C, 0xF, 0xF, false);     /* row_ror:12: from lane + 4 */
        return (lane & 4) ? a : b;
    }
    if (j == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xF, 0xF, false);         /* row_ror:8 */
    if (j == 16) { const auto r = __builtin_amdgcn_permlane16_swap(key, key, false, false); return (lane & 16) ? r[0] : r[1]; }
    const auto r = __builtin_amdgcn_permlane32_swap(key, key, false, false);
    return (lane & 32) ? r[0] : r[1];
}

/* Chains of lanes that target the same pixel, in lane (= visiting) order, for the flood fill: every live lane gets the previous
 * lane with its pixel (-1: none), its depth in the chain and whether it is the last.  The 64 keys pixel << 6 | lane are sorted
 * across the wavefront by a bitonic network (21 compare-exchange stages; a dead lane's key sorts behind every live one and is
 * its own pixel), runs of equal pixels are read off lane masks, and one forward permute takes the answers home.  ~150
 * instructions whatever the step holds - the search over the step's entries it replaces cost 65 per ENTRY. */
__device__ __forceinline__ void chain_sort64(bool live, int pixel, int lane, int& prev, int& depth, bool& isLast)
{
    uint32_t key = (live ? (uint32_t)pixel : (0x3FFFFC0u | (uint32_t)lane)) << 6 | (uint32_t)lane;   /* pixels are below 2^21 */
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint32_t other = xor_partner(key, j, lane);
            const bool keepMin = ((lane & j) == 0) == ((lane & k) == 0);
            const uint32_t lo = key < other ? key : other, hi = key < other ? other : key;
            key = keepMin ? lo : hi;
        }
    }
    /* lane i holds the i-th smallest key */
    const uint32_t before = (uint32_t)__shfl_up((int)key, 1);
    const bool same = lane > 0 && (before >> 6) == (key >> 6);
    const unsigned long long S = __ballot(same), H = ~S;                         /* run heads */
    const unsigned long long le = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
    const int runStart = 63 - __builtin_clzll(H & le);                           /* bit 0 of H is always set */
    const bool last = lane == 63 || !((S >> (lane + 1)) & 1ull);
    const uint32_t packed = (uint32_t)(lane - runStart) | (same ? ((before & 63u) + 1u) << 6 : 0u) | (last ? 1u << 13 : 0u);
    /* home: the lane a key came from receives its answer (ds_permute: a forward permute, lane i writes to lane key & 63) */
    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)((key & 63u) << 2), (int)packed);
    depth = (int)(got & 63u);
    prev = (int)((got >> 6) & 127u) - 1;
    isLast = (got >> 13) & 1u;
}

/* The same for 128 visits, two per lane: visit id = slot * 64 + lane, visiting order = id order (round 6: the flood fill takes 32
 * queue entries per step).  Keys pixel << 7 | id sorted by a 128-element bitonic network over (slot, lane) - 28 compare-exchange
 * stages, the one with partner distance 64 inside the lane - then runs of equal pixels are read off two lane masks and every visit's
 * answer goes home through a 128-word LDS table (tmp). */
__device__ __forceinline__ void chain_sort128(bool live0, int pixel0, bool live1, int pixel1, int lane, uint32_t* tmp,
                                              int& prev0, int& depth0, bool& isLast0, int& prev1, int& depth1, bool& isLast1)
{
    /* a dead visit's key sorts behind every live one and is its own pixel (pixels are below 2^21, ids below 2^7) */
    uint32_t k0 = (live0 ? (uint32_t)pixel0 : (0x200000u | (uint32_t)lane)) << 7 | (uint32_t)lane;
    uint32_t k1 = (live1 ? (uint32_t)pixel1 : (0x200040u | (uint32_t)lane)) << 7 | (uint32_t)(64 + lane);
#pragma unroll
    for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j == 64) {                                   /* k == 128: the partner is the lane's other slot, ascending */
                const uint32_t lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
                k0 = lo; k1 = hi;
            } else {
                const uint32_t o0 = xor_partner(k0, j, lane), o1 = xor_partner(k1, j, lane);
                /* element index i = slot * 64 + lane: (i & j) is (lane & j); (i & k) is (lane & k) below 64, the slot at 64, zero at 128 */
                const bool low = (lane & j) == 0;
                const bool up0 = k == 128 ? true : (k == 64 ? true : (lane & k) == 0);
                const bool up1 = k == 128 ? true : (k == 64 ? false : (lane & k) == 0);
                const uint32_t lo0 = k0 < o0 ? k0 : o0, hi0 = k0 < o0 ? o0 : k0;
                const uint32_t lo1 = k1 < o1 ? k1 : o1, hi1 = k1 < o1 ? o1 : k1;
                k0 = (low == up0) ? lo0 : hi0;
                k1 = (low == up1) ? lo1 : hi1;
            }
        }
    }
    /* sorted position i = slot * 64 + lane holds the i-th smallest key */
    const uint32_t b0 = (uint32_t)__shfl_up((int)k0, 1);                     /* slot 0: the element before (lane 0: none) */
    uint32_t b1 = (uint32_t)__shfl_up((int)k1, 1);                           /* slot 1, lane 0: slot 0's last */
    const uint32_t k0last = (uint32_t)__builtin_amdgcn_readlane((int)k0, 63);
    if (lane == 0) b1 = k0last;
    const bool same0 = lane > 0 && (b0 >> 7) == (k0 >> 7);
    const bool same1 = (b1 >> 7) == (k1 >> 7);
    const unsigned long long S0 = __ballot(same0), S1 = __ballot(same1), H0 = ~S0, H1 = ~S1;     /* run heads; bit 0 of H0 is always set */
    const unsigned long long le = lane == 63 ? ~0ull : (2ull << lane) - 1ull;
    const int rs0 = 63 - __builtin_clzll(H0 & le);
    const int rs1 = (H1 & le) ? 64 + 63 - __builtin_clzll(H1 & le) : 63 - __builtin_clzll(H0);
    const bool last0 = lane == 63 ? !(S1 & 1ull) : !((S0 >> (lane + 1)) & 1ull);
    const bool last1 = lane == 63 || !((S1 >> (lane + 1)) & 1ull);
    const uint32_t p0 = (uint32_t)(lane - rs0) | (same0 ? ((b0 & 127u) + 1u) << 7 : 0u) | (last0 ? 1u << 15 : 0u);
    const uint32_t p1 = (uint32_t)(64 + lane - rs1) | (same1 ? ((b1 & 127u) + 1u) << 7 : 0u) | (last1 ? 1u << 15 : 0u);
    tmp[k0 & 127u] = p0;
    tmp[k1 & 127u] = p1;
    wave_order();
    const uint32_t g0 = tmp[lane], g1 = tmp[64 + lane];
    wave_order();
    depth0 = (int)(g0 & 127u); prev0 = (int)((g0 >> 7) & 255u) - 1; isLast0 = (g0 >> 15) & 1u;
    depth1 = (int)(g1 & 127u); prev1 = (int)((g1 >> 7) & 255u) - 1; isLast1 = (g1 >> 15) & 1u;
}

#ifdef AHC_PROFILE
/* ahCluster of frame 0 (100 MHz ticks of lane 0): 0 pops, 1 heap pop, 2 loads of the popped node + its list, 3 trial merges, 4 merge
 * (node, heap push, union-find, list surgery), 5 no merge (extract / disconnect), 6 merges */
__device__ unsigned long long g_ahcProf[8];
#define CP_T() wall_clock64()
#define CP_ADD(k, t0) do { if (blockIdx.x == 0 && c.lane == 0) g_ahcProf[k] += wall_clock64() - (t0); } while (0)
#define CP_CNT(k) do { if (blockIdx.x == 0 && c.lane == 0) g_ahcProf[k] += 1; } while (0)
#else
#define CP_T() 0ull
#define CP_ADD(k, t0) (void)(t0)
#define CP_CNT(k) (void)0
#endif

struct SinCosR { double s, c; int ok; };
__device__ __noinline__ SinCosR cr_cos_call(double x)
{
    SinCosR r;
    r.ok = drfe_cr_sincos(x, &r.s, &r.c);
    return r;
}

struct Ctx {
    /* frame arrays (global address space) */
    const GLOBAL_AS uint16_t* depth; int rowStride;
    GLOBAL_AS double* S; GLOBAL_AS double* fit; GLOBAL_AS int* N; GLOBAL_AS int* rid; GLOBAL_AS uint8_t* nouse;
    GLOBAL_AS int* nbOff; GLOBAL_AS int* nbLen; GLOBAL_AS int* pool;
    GLOBAL_AS int* dsParent; GLOBAL_AS int* dsSize; GLOBAL_AS int* G; GLOBAL_AS int* blkMap; GLOBAL_AS int* ridToPlid;
    GLOBAL_AS int16_t* mem; GLOBAL_AS float* dist; GLOBAL_AS uint32_t* rf;
    /* LDS */
    float* heapKey; uint16_t* heapId; uint16_t* lA; uint16_t* lB; uint16_t* lU; double* win;      /* node ids are below 2 NB + 256 = 6400 */
    int heapSize, nNodes, poolUsed, status, lane;
    int listCap;                                     /* entries of lA / lB (lU: twice that) */
    AhcDevParams P;
};

/* --- priority queue: smallest MSE first, ties by the smaller node id (QCmp of planes_ahc.cpp; a total order, so the pop
 * sequence does not depend on the heap's internals). */
/* The queue keeps the keys as FLOATS (half the LDS: the clustering kernel's footprint decides how many frames a CU holds):
 * rounding to float is monotone, so two different floats order the doubles; equal floats (the doubles agree to 24 bits: rare)
 * are settled by the exact MSEs, which every node keeps in its fit record. */
__device__ __forceinline__ double heap_exact(const Ctx& c, int id) { return c.fit[8 * (size_t)id + 6]; }
__device__ __forceinline__ bool heap_less(const Ctx& c, float ka, int ia, float kb, int ib)
{
    if (ka != kb) return ka < kb;
    const double da = heap_exact(c, ia), db = heap_exact(c, ib);
    return da < db || (da == db && ia < ib);
}
/* (key, id) not in the node's record yet, or just written: its exact key comes along */
__device__ __forceinline__ bool heap_less_new(const Ctx& c, double key, float kf, int id, float kb, int ib)
{
    if (kf != kb) return kf < kb;
    const double db = heap_exact(c, ib);
    return key < db || (key == db && id < ib);
}
/* The queue is a 64-ary heap in LDS (children of entry i: 64 i + 1 .. 64 i + 64): three levels hold 3200 entries (four: 12 800), a level of
 * sift-down is ONE read by the 64 lanes and a minimum across the wavefront, and sift-up looks at two parents at most.  (The binary
 * heap walked by one lane cost twelve levels of dependent LDS reads per pop: 2.1 us, a fifth of the clustering.) */
__device__ void heap_push(Ctx& c, double key, int id)                 /* one lane */
{
    const float kf = (float)key;
    int i = c.heapSize++;
    while (i > 0) {
        const int p = (i - 1) >> 6;
        const float kp = c.heapKey[p];
        const int ip = c.heapId[p];
        if (!heap_less_new(c, key, kf, id, kp, ip)) break;
        c.heapKey[i] = kp; c.heapId[i] = (uint16_t)ip;
        i = p;
    }
    c.heapKey[i] = kf; c.heapId[i] = (uint16_t)id;
}
/* smallest float of the wavefront, to every lane */
__device__ __forceinline__ float wave_min_f32(float v)
{
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));     /* quad_perm [1, 0, 3, 2] */
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));     /* quad_perm [2, 3, 0, 1] */
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));    /* row_half_mirror */
    v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)));    /* row_mirror */
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
                r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
/* (key, id) sinks from entry i of a heap of n entries.  The whole wavefront. */
__device__ void heap_sift_down(Ctx& c, int i, float key, int id, int n)
{
    const int lane = c.lane;
    for (;;) {
        const int base = 64 * i + 1;
        if (base >= n) break;
        const int ch = base + lane;
        const bool valid = ch < n;
        const float kc = valid ? c.heapKey[ch] : __int_as_float(0x7f800000);
        const int ic = valid ? (int)c.heapId[ch] : 0xFFFF;
        const float kmin = wave_min_f32(kc);
        unsigned long long ties = __ballot(valid && kc == kmin);
        int best = __builtin_ctzll(ties);
        ties &= ties - 1;
        while (ties) {                                     /* equal floats (rare): the exact keys and the ids decide */
            const int l = __builtin_ctzll(ties);
            ties &= ties - 1;
            if (heap_less(c, kmin, rl_i(ic, l), kmin, rl_i(ic, best))) best = l;
        }
        const int ib = rl_i(ic, best);
        if (!heap_less(c, kmin, ib, key, id)) break;
        if (lane == 0) { c.heapKey[i] = kmin; c.heapId[i] = (uint16_t)ib; }
        i = base + best;
    }
    if (lane == 0) { c.heapKey[i] = key; c.heapId[i] = (uint16_t)id; }
    wave_order();
}
__device__ int heap_pop(Ctx& c)                                        /* the whole wavefront; c.heapSize uniform */
{
    const int top = c.heapId[0];
    const int n = --c.heapSize;
    if (n > 0) {
        const float key = c.heapKey[n];
        const int id = c.heapId[n];
        heap_sift_down(c, 0, key, id, n);
    }
    return top;
}

/* A node's rid word: its representative block (low 16 bits: blocks are below 12 800) and the ROOT of that block's set (high 16 bits).
 * The sets of two living nodes are disjoint and a set changes only when its node merges - which ends the node - so the root a node
 * was created with stays its root for as long as it lives: a merge reads both roots from the words it has already loaded instead of
 * walking two chains of dependent loads (round 6: 2-4 memory round trips per merge). */
__device__ __forceinline__ int rid_block(int word) { return word & 0xFFFF; }
__device__ __forceinline__ int rid_root(int word) { return (int)((unsigned)word >> 16); }
__device__ __forceinline__ int rid_pack(int block, int root) { return block | (root << 16); }

/* union-find over the init blocks (DisjointSet.hpp): Find without path compression gives the same roots */
__device__ __forceinline__ int ds_find(const Ctx& c, int x)
{
    int p = c.dsParent[x];
    while (p != x) { x = p; p = c.dsParent[x]; }
    return x;
}

__device__ __forceinline__ double similarity(const Ctx& c, int a, int b)
{
    const GLOBAL_AS double* n = c.fit + 8 * (size_t)a + 3;
    const GLOBAL_AS double* m = c.fit + 8 * (size_t)b + 3;
    return fabs(n[0] * m[0] + n[1] * m[1] + n[2] * m[2]);
}

/* insertSorted without duplicates into the list of node a (capacity guaranteed by the caller); one lane */
__device__ __forceinline__ void list_insert(Ctx& c, int a, int v)
{
    GLOBAL_AS int* L = c.pool + c.nbOff[a];
    int len = c.nbLen[a], pos = 0;
    while (pos < len && L[pos] < v) pos++;
    if (pos < len && L[pos] == v) return;
    for (int k = len; k > pos; k--) L[k] = L[k - 1];
    L[pos] = v;
    c.nbLen[a] = len + 1;
}
/* eraseSorted; one lane.  Lists of up to eight neighbours (nearly all) are fetched with eight independent loads instead of a
 * chain of dependent ones */
__device__ __forceinline__ void list_erase(Ctx& c, int a, int v)
{
    GLOBAL_AS int* L = c.pool + c.nbOff[a];
    const int len = c.nbLen[a];
    if (len <= 8) {
        int e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = k < len ? L[k] : 0x7fffffff;
        int w = 0;
        bool found = false;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < len) {
                if (e[k] == v) found = true;
                else { if (found) L[w] = e[k]; w++; }
            }
        if (found) c.nbLen[a] = len - 1;
        return;
    }
    /* longer lists, eight entries at a time: the loads of a group are independent (one round trip per group instead of one per
     * entry); an entry behind v moves one place to the left - to a place at or before its own, so no group reads what an earlier one wrote */
    bool found = false;
    for (int k0 = 0; k0 < len; k0 += 8) {
        int e[8];
#pragma unroll
        for (int j = 0; j < 8; j++) e[j] = k0 + j < len ? L[k0 + j] : 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (k0 + j < len) {
                if (e[j] == v) found = true;
                else if (found) L[k0 + j - 1] = e[j];
            }
    }
    if (found) c.nbLen[a] = len - 1;
}

/* what a merge of p and q into the new node id does to the list of a common neighbour a: p and q leave, id (the largest id so
 * far) is appended - disconnectAll(p), disconnectAll(q) and the connects of the new node in one pass over the list */
__device__ __forceinline__ void list_replace2(Ctx& c, int a, int p, int q, int id)
{
    GLOBAL_AS int* L = c.pool + c.nbOff[a];
    const int len = c.nbLen[a];
    int w = 0;
    if (len <= 8) {
        int e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = k < len ? L[k] : 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (k < len && e[k] != p && e[k] != q) { if (w != k) L[w] = e[k]; w++; }
    } else {
        /* eight entries at a time (independent loads: a round trip per group); an entry is stored at or before its own place */
        for (int k0 = 0; k0 < len; k0 += 8) {
            int e[8];
#pragma unroll
            for (int k = 0; k < 8; k++) e[k] = k0 + k < len ? L[k0 + k] : 0x7fffffff;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k0 + k < len && e[k] != p && e[k] != q) { if (w != k0 + k) L[w] = e[k]; w++; }
        }
    }
    L[w] = id;
    c.nbLen[a] = w + 1;
}

/* ParamSet::T_ang(P_INIT, z) with the millimetre defaults (planes_ahc.cpp tAngInit); cosine correctly rounded */
__device__ double t_ang_init(Ctx& c, double z)
{
    const double pi = 3.14159265358979323846;
    const double z_near = 500, z_far = 4000, a_near = 15.0 * pi / 180.0, a_far = 90.0 * pi / 180.0;
    double cz = z > z_near ? z : z_near;
    cz = cz < z_far ? cz : z_far;
    const double factor = (a_far - a_near) / (z_far - z_near);
    const SinCosR r = cr_cos_call(factor * cz + a_near - factor * z_near);
    if (!r.ok) c.status |= 1;
    return r.c;
}

/* disconnectAll(p) for the staged list lX of length len (the neighbours of p): every neighbour drops p; lane per neighbour */
__device__ __forceinline__ void disconnect_staged(Ctx& c, const uint16_t* lX, int len, int p)
{
    for (int k = c.lane; k < len; k += 64) list_erase(c, lX[k], p);
    fence();
}

/* ahCluster (planes_ahc.cpp cluster()): pops until the queue is empty; extracted nodes (N >= minSupport) appended to ex[] */
__device__ void cluster(Ctx& c, int* ex, int& nEx)
{
    const int lane = c.lane;
    while (uni_i(c.heapSize) > 0) {
        int p = 0;
        const unsigned long long cp0 = CP_T();
        CP_CNT(0);
        c.heapSize = uni_i(c.heapSize);
        p = uni_i((int)c.heapId[0]);
        /* everything that depends on p alone in ONE round trip - flags, list head, sums, normal - issued BEFORE the queue is repaired: the
         * sift-down touches LDS only, so the round trip runs beside it (round 6) */
        const int nouseP = c.nouse[p], lenP = c.nbLen[p], offP = c.nbOff[p];
        double Sp[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Sp[k] = c.S[9 * (size_t)p + k];
        const int Np = c.N[p], ridP = c.rid[p];
        const double npx = c.fit[8 * (size_t)p + 3], npy = c.fit[8 * (size_t)p + 4], npz = c.fit[8 * (size_t)p + 5];
        (void)heap_pop(c);
        CP_ADD(1, cp0);
        const unsigned long long cp1 = CP_T();
        if (uni_b(nouseP != 0)) continue;
        const int sizeP = c.dsSize[rid_root(ridP)];             /* in flight beside the list loads below */
        const int Lp = uni_i(lenP);
#ifdef AHC_PROFILE
#if defined(AHC_PROFILE_CHUNKS)
        if (blockIdx.x == 0 && c.lane == 0) g_ahcProf[7] += (unsigned long long)((Lp + 63) / 64) | ((unsigned long long)Lp << 32);      /* trial chunks | neighbours, summed over the pops */
#elif defined(AHC_PROFILE_HOT)
        if (blockIdx.x == 0 && c.lane == 0 && p == c.nNodes - 1 && p >= c.P.NB / 2) g_ahcProf[7] += 1;      /* pops of the node the last merge made */
#else
        if (c.lane == 0) atomicMax(&g_ahcProf[7], (unsigned long long)Lp);
#endif
#endif
        if (Lp > c.listCap) { c.status |= 2; return; }
        const GLOBAL_AS int* listP = c.pool + offP;
        for (int k = lane; k < Lp; k += 64) c.lA[k] = (uint16_t)listP[k];
#ifdef AHC_PROFILE
        if (__ballot(Lp > 0 && c.lA[0] == 0xFFFF && Sp[0] == 12345.678)) c.status |= 4;      /* wait for the loads here */
#endif
        CP_ADD(2, cp1);
        const unsigned long long cp2 = CP_T();
        /* trial merges, one lane per neighbour; the fold keeps the reference's order and tie rule */
        bool haveCand = false;
        double candMse = 0;
        int candN = 0, candNb = -1, candLen = 0, candRid = 0, candSize = 0;
        for (int base = 0; base < Lp; base += 64) {
            const int k = base + lane;
            bool ok = false;
            double S[9];
            AhcFit f;
            int Nn = 0, ridN = 0, nb = 0, nbLenK = 0, ridNbK = 0, sizeNbK = 0, nbHead[8];
            f.mse = 0;
            if (k < Lp) {
                nb = c.lA[k];
                /* what depends on the neighbour alone, fetched together (one round trip): its normal for the similarity gate, its
                 * sums and size for the trial, and where its own list lies */
                const double mx = c.fit[8 * (size_t)nb + 3], my = c.fit[8 * (size_t)nb + 4], mz = c.fit[8 * (size_t)nb + 5];
                nbLenK = c.nbLen[nb];
                const int nbOffK = c.nbOff[nb];
                const int Nb = c.N[nb], ridNb = c.rid[nb];
                ridNbK = ridNb;
#pragma unroll
                for (int q = 0; q < 9; q++) S[q] = c.S[9 * (size_t)nb + q];
                /* the neighbour's own list (its first eight entries: nearly always all of it): if this trial wins, the merge
                 * needs it and would wait two more round trips for it */
                const GLOBAL_AS int* listN = c.pool + nbOffK;
#pragma unroll
                for (int q = 0; q < 8; q++) nbHead[q] = q < nbLenK ? listN[q] : 0;
                sizeNbK = c.dsSize[rid_root(ridNb)];            /* its set's size, should this trial win: same round trip as the list head */
                if (!(fabs(npx * mx + npy * my + npz * mz) < c.P.cos60)) {
                    ok = true;
#pragma unroll
                    for (int q = 0; q < 9; q++) S[q] = Sp[q] + S[q];
                    Nn = Np + Nb;
                    ridN = rid_block(Np >= Nb ? ridP : ridNb);
                    ahc_plane_from_sums(S, Nn, &f);
                }
            }
            unsigned long long m = __ballot(ok);
            int winLane = -1;
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const double mse = rl_d(f.mse, l);
                if (!haveCand || candMse > mse || (candMse == mse && (double)candN < mse)) {
                    haveCand = true; candMse = mse; candN = rl_i(Nn, l); candNb = rl_i(nb, l); candLen = rl_i(nbLenK, l); candRid = rl_i(ridNbK, l); candSize = rl_i(sizeNbK, l); winLane = l;
                }
            }
            if (winLane >= 0 && lane == winLane) {          /* this chunk's winner parks its merged node in LDS */
#pragma unroll
                for (int q = 0; q < 9; q++) c.win[q] = S[q];
                c.win[9] = f.center[0]; c.win[10] = f.center[1]; c.win[11] = f.center[2];
                c.win[12] = f.normal[0]; c.win[13] = f.normal[1]; c.win[14] = f.normal[2];
                c.win[15] = f.mse; c.win[16] = f.curvature;
                ((int*)(c.win + 17))[0] = Nn; ((int*)(c.win + 17))[1] = ridN;
#pragma unroll
                for (int q = 0; q < 8; q++) c.lB[q] = (uint16_t)nbHead[q];
            }
        }
        CP_ADD(3, cp2);
        const unsigned long long cp3 = CP_T();
        bool merge = false;
        if (haveCand) {
            const double z = c.win[11];
            const double t = 1.6e-6 * z * z + 8.0;
            merge = candMse < t * t;
        }
        if (uni_b(merge)) {
            const int id = c.nNodes;
            const int Lc = candLen;
            if (id >= c.P.maxNodes || Lc > c.listCap) { c.status |= 2; return; }
            c.nNodes = id + 1;
            if (lane < 9) c.S[9 * (size_t)id + lane] = c.win[lane];
            if (lane < 8) c.fit[8 * (size_t)id + lane] = c.win[9 + lane];
            /* mergeNbsFrom: union by size of the two root blocks - both roots and their sizes came with the words loaded above */
            const int xr = rid_root(ridP), yr = rid_root(candRid), sxr = sizeP, syr = candSize;
            if (lane == 0) {
                const int newRoot = (xr != yr && sxr < syr) ? yr : xr;
                c.N[id] = ((int*)(c.win + 17))[0]; c.rid[id] = rid_pack(((int*)(c.win + 17))[1], newRoot); c.nouse[id] = 0;
                heap_push(c, c.win[15], id);
                if (xr != yr) {
                    if (sxr < syr) { c.dsParent[xr] = yr; c.dsSize[yr] = syr + sxr; }
                    else { c.dsParent[yr] = xr; c.dsSize[xr] = sxr + syr; }
                }
            }
            c.heapSize = uni_i(c.heapSize);
            if (Lc > 8) {                                       /* the winner parked the first eight: the rest of a long list */
                const GLOBAL_AS int* listC = c.pool + c.nbOff[candNb];
                for (int k = 8 + lane; k < Lc; k += 64) c.lB[k] = (uint16_t)listC[k];
            }
            fence();
            /* u = nbs(p) U nbs(cand) \ {p, cand}, sorted.  Short lists (the rule): lane i holds one element of A ++ B; it is kept
             * unless it is p, cand or an element of B that A holds too, and its place in u is the number of kept elements
             * smaller than it.  Long lists: the same count through binary searches (below). */
            int Lu = 0;
            if (Lp + Lc <= 64) {
                const int tot = Lp + Lc;
                const bool fromB = lane >= Lp;
                const int v = lane < tot ? (fromB ? c.lB[lane - Lp] : c.lA[lane]) : 0x7fffffff;
                bool keep = lane < tot && v != p && v != candNb;
                int smaller = 0;
                for (int l = 0; l < tot; l++) {
                    const int o = rl_i(v, l);
                    if (fromB && l < Lp && o == v) keep = false;          /* A holds it too */
                }
                const unsigned long long km = __ballot(keep);
                for (int l = 0; l < tot; l++) {
                    const int o = rl_i(v, l);
                    if (((km >> l) & 1ull) && o < v) smaller++;
                }
                if (keep) c.lU[smaller] = (uint16_t)v;
                Lu = __popcll(km);
            } else {
                /* long lists (a node in the middle of a big plane: up to a few hundred neighbours), all lanes: an element's place in u is
                 * the number of kept elements below it, counted without a merge loop - A = nbs(p) and B = nbs(cand) are sorted and free of
                 * duplicates, cand is in A and p in B (they are neighbours) and nowhere else, a value both lists hold is kept from A:
                 *   v = A[i], v != cand:   i - [cand < v] + |{b in B: b < v}| - |{common values < v}| - [p < v]
                 *   v = B[j], v != p, not in A:   |{a in A: a < v}| - [cand < v] + j - |{common values < v}| - [p < v]
                 * |{... < v}| in the other list is a binary search (LDS), the common values below v a running ballot count in list order.
                 * (One lane merging the two lists cost ~100 cycles per element: 6 us of a merge's 16 at 1280 x 960.) */
                auto lower_bound = [&](const uint16_t* L, int n, int v, bool& has) -> int {
                    int lo = 0, hi = n;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)L[mid] < v) lo = mid + 1; else hi = mid; }
                    has = lo < n && (int)L[lo] == v;
                    return lo;
                };
                bool pInB = false;
                (void)lower_bound(c.lB, Lc, p, pInB);              /* adjacency is symmetric, so it is; the count below does not assume it */
                const int pB = pInB ? 1 : 0;
                int common = 0;
                for (int base = 0; base < Lp; base += 64) {
                    const int i = base + lane;
                    const bool on = i < Lp;
                    const int v = on ? (int)c.lA[i] : 0x7fffffff;
                    bool inB = false;
                    const int lb = on ? lower_bound(c.lB, Lc, v, inB) : 0;
                    const unsigned long long cm = __ballot(on && inB);
                    if (on && v != candNb) c.lU[i - (candNb < v ? 1 : 0) + lb - (common + __popcll(cm & ((1ull << lane) - 1ull))) - (p < v ? pB : 0)] = (uint16_t)v;
                    common += __popcll(cm);
                }
                int commonB = 0;
                for (int base = 0; base < Lc; base += 64) {
                    const int j = base + lane;
                    const bool on = j < Lc;
                    const int v = on ? (int)c.lB[j] : 0x7fffffff;
                    bool inA = false;
                    const int lb = on ? lower_bound(c.lA, Lp, v, inA) : 0;
                    const unsigned long long cm = __ballot(on && inA);
                    if (on && !inA && v != p) c.lU[lb - (candNb < v ? 1 : 0) + j - (commonB + __popcll(cm & ((1ull << lane) - 1ull))) - (p < v ? pB : 0)] = (uint16_t)v;
                    commonB += __popcll(cm);
                }
                Lu = Lp + Lc - common - 1 - pB;
            }
            Lu = uni_i(Lu);
            fence();
            if (c.poolUsed + Lu > c.P.poolCap) { c.status |= 2; return; }
            /* disconnectAll(p), disconnectAll(cand) and the new node's connects: every member of u (a neighbour of p, of cand
             * or of both) drops them and gets the new node - the largest id so far: appended - in one pass over its list; the
             * lists of p and cand themselves die with the nodes.  The new node takes u. */
            const int off = c.poolUsed;
            c.poolUsed += Lu;
            for (int k = lane; k < Lu; k += 64) {
                const int nb = c.lU[k];
                c.pool[off + k] = nb;
                list_replace2(c, nb, p, candNb, id);
            }
            if (lane == 0) { c.nbOff[id] = off; c.nbLen[id] = Lu; c.nbLen[p] = 0; c.nbLen[candNb] = 0; c.nouse[p] = 1; c.nouse[candNb] = 1; }
            fence();
            CP_ADD(4, cp3); CP_CNT(6);
        } else {
            if (Np >= AHC_MIN_SUPPORT) {
                if (nEx >= AHCD_MAXEX) { c.status |= 2; return; }
                if (lane == 0) ex[nEx] = p;
                nEx++;
            }
            disconnect_staged(c, c.lA, Lp, p);
            if (lane == 0) c.nbLen[p] = 0;
            fence();
            CP_ADD(5, cp3);
        }
    }
    /* std::stable_sort by N, larger first: a handful of planes, insertion sort */
    if (lane == 0)
        for (int i = 1; i < nEx; i++) {
            const int v = ex[i], nv = c.N[v];
            int j = i;
            while (j > 0 && c.N[ex[j - 1]] < nv) { ex[j] = ex[j - 1]; j--; }
            ex[j] = v;
        }
    fence();
}

} // namespace

template <int HEAP, int LIST>
__device__ __forceinline__ void ahc_cluster_frame(const AhcDevFrame* __restrict__ frames, const AhcDevParams& P)
{
    __shared__ float heapKey[HEAP];
    __shared__ uint16_t heapId[HEAP];
    __shared__ uint16_t lA[LIST], lB[LIST], lU[2 * LIST];
    __shared__ double win[18];
    __shared__ int ex[AHCD_MAXEX];
    __shared__ uint8_t isValid[AHCD_MAXEX];
    const AhcDevFrame F = frames[blockIdx.x];
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    Ctx c;
    c.depth = (const GLOBAL_AS uint16_t*)F.depth; c.rowStride = (int)F.rowStride;
    c.S = (GLOBAL_AS double*)F.nodeS; c.fit = (GLOBAL_AS double*)F.nodeFit; c.N = (GLOBAL_AS int*)F.nodeN; c.rid = (GLOBAL_AS int*)F.nodeRid;
    c.nouse = (GLOBAL_AS uint8_t*)F.nodeNouse; c.nbOff = (GLOBAL_AS int*)F.nbOff; c.nbLen = (GLOBAL_AS int*)F.nbLen; c.pool = (GLOBAL_AS int*)F.nbPool;
    c.dsParent = (GLOBAL_AS int*)F.dsParent; c.dsSize = (GLOBAL_AS int*)F.dsSize; c.G = (GLOBAL_AS int*)F.G; c.blkMap = (GLOBAL_AS int*)F.blkMap;
    c.ridToPlid = (GLOBAL_AS int*)F.ridToPlid; c.mem = (GLOBAL_AS int16_t*)F.membership; c.dist = (GLOBAL_AS float*)F.distMap; c.rf = (GLOBAL_AS uint32_t*)F.rf;
    c.heapKey = heapKey; c.heapId = heapId; c.lA = lA; c.lB = lB; c.lU = lU; c.win = win;
    c.heapSize = 0; c.nNodes = 0; c.poolUsed = 0; c.status = 0; c.lane = lane; c.listCap = LIST; c.P = P;
    const GLOBAL_AS AhcBlockRec* blocks = (const GLOBAL_AS AhcBlockRec*)F.blocks;
    GLOBAL_AS int* out = (GLOBAL_AS int*)F.out;
    const int w = P.w, h = P.h, Nw = P.Nw, Nh = P.Nh, NB = P.NB, npx = w * h;
    /* no cloud for k_voxel_grid unless the frame runs to its end */
    if (F.jobs) for (int i = lane; i < 2 * P.planeCap; i += 64) ((GLOBAL_AS int*)F.jobs)[i] = 0;
    if (NB > HEAP || npx > (1 << AHCD_PIXBITS)) { if (lane == 0) { out[0] = 0; out[1] = 4; } return; }

#ifdef AHC_PROFILE
    unsigned long long tp[8]; int tpi = 0;
#define TP() tp[tpi++] = wall_clock64()
#else
#define TP() (void)0
#endif
    TP();
    /* ---- initGraph: nodes of the valid blocks, in block order ---- */
    for (int base = 0; base < NB; base += 64) {
        const int b = base + lane;
        const bool v = b < NB && blocks[b].valid != 0;
        const unsigned long long m = __ballot(v);
        if (b < NB) {
            c.dsParent[b] = b; c.dsSize[b] = 1;
            if (v) {
                const int id = c.nNodes + __popcll(m & lt);
                c.G[b] = id;
                for (int k = 0; k < 9; k++) c.S[9 * (size_t)id + k] = blocks[b].sums[k];
                for (int k = 0; k < 3; k++) { c.fit[8 * (size_t)id + k] = blocks[b].center[k]; c.fit[8 * (size_t)id + 3 + k] = blocks[b].normal[k]; }
                c.fit[8 * (size_t)id + 6] = blocks[b].mse; c.fit[8 * (size_t)id + 7] = blocks[b].curvature;
                c.N[id] = blocks[b].N; c.rid[id] = rid_pack(b, b); c.nouse[id] = 0;
                c.nbOff[id] = 4 * id; c.nbLen[id] = 0;
                heapKey[id] = (float)blocks[b].mse; heapId[id] = (uint16_t)id;
            } else c.G[b] = -1;
        }
        c.nNodes += __popcll(m);
    }
    c.poolUsed = 4 * c.nNodes;
    c.heapSize = c.nNodes;
    fence();
    for (int sidx = (c.heapSize - 2) / 64; sidx >= 0 && c.heapSize > 1; sidx--) {          /* heapify: the parents, last first */
        const float key = heapKey[sidx];
        const int id = heapId[sidx];
        heap_sift_down(c, sidx, key, id, c.heapSize);
    }
    /* edges: the row pass (a lane per block row), then the column pass (a lane per block column), each with the reference's
     * skip pattern; a pass only touches the lists of its own row / column */
    for (int i = lane; i < Nh; i += 64)
        for (int j = 1; j < Nw; j += 2) {
            const int cidx = i * Nw + j;
            if (c.G[cidx - 1] < 0) { --j; continue; }
            if (c.G[cidx] < 0) continue;
            if (j < Nw - 1 && c.G[cidx + 1] < 0) { ++j; continue; }
            const double th = t_ang_init(c, c.fit[8 * (size_t)c.G[cidx] + 2]);
            if ((j < Nw - 1 && similarity(c, c.G[cidx - 1], c.G[cidx + 1]) >= th) || (j == Nw - 1 && similarity(c, c.G[cidx], c.G[cidx - 1]) >= th)) {
                list_insert(c, c.G[cidx], c.G[cidx - 1]); list_insert(c, c.G[cidx - 1], c.G[cidx]);
                if (j < Nw - 1) { list_insert(c, c.G[cidx], c.G[cidx + 1]); list_insert(c, c.G[cidx + 1], c.G[cidx]); }
            } else --j;
        }
    fence();
    for (int j = lane; j < Nw; j += 64)
        for (int i = 1; i < Nh; i += 2) {
            const int cidx = i * Nw + j;
            if (c.G[cidx - Nw] < 0) { --i; continue; }
            if (c.G[cidx] < 0) continue;
            if (i < Nh - 1 && c.G[cidx + Nw] < 0) { ++i; continue; }
            const double th = t_ang_init(c, c.fit[8 * (size_t)c.G[cidx] + 2]);
            if ((i < Nh - 1 && similarity(c, c.G[cidx - Nw], c.G[cidx + Nw]) >= th) || (i == Nh - 1 && similarity(c, c.G[cidx], c.G[cidx - Nw]) >= th)) {
                list_insert(c, c.G[cidx], c.G[cidx - Nw]); list_insert(c, c.G[cidx - Nw], c.G[cidx]);
                if (i < Nh - 1) { list_insert(c, c.G[cidx], c.G[cidx + Nw]); list_insert(c, c.G[cidx + Nw], c.G[cidx]); }
            } else --i;
        }
    fence();
    c.status = (int)(__ballot(c.status != 0) != 0);          /* a lane's uncertified cosine */

    TP();
    /* ---- ahCluster ---- */
    int nEx = 0;
    cluster(c, ex, nEx);
    nEx = uni_i(nEx);
    if (uni_b(c.status != 0)) { if (lane == 0) { out[0] = 0; out[1] = c.status; } return; }

    TP();
    /* ---- refineDetails: findBlockMembership ---- */
    for (int k = lane; k < NB; k += 64) c.ridToPlid[k] = -1;
    for (int k = lane; k < npx; k += 64) { c.mem[k] = -1; c.dist[k] = 3.402823466e+38f; }
    for (int k = lane; k < nEx; k += 64) {
        isValid[k] = 0;
    }
    fence();
    if (lane == 0)
        for (int plid = 0; plid < nEx; plid++) {             /* std::map::insert: the first plane of a root keeps it */
            const int r = rid_block(c.rid[ex[plid]]);
            if (c.ridToPlid[r] < 0) c.ridToPlid[r] = plid;
        }
    fence();
    for (int b = lane; b < NB; b += 64) {
        const int i = b / Nw, j = b - i * Nw;
        const int setid = ds_find(c, b);
        int bm = -1;
        if (c.dsSize[setid] * (AHC_WIN * AHC_WIN) >= AHC_MIN_SUPPORT) {
            bool same = true;
            if (j > 0 && ds_find(c, b - 1) != setid) same = false;
            if (same && j < Nw - 1 && ds_find(c, b + 1) != setid) same = false;
            if (same && i > 0 && ds_find(c, b - Nw) != setid) same = false;
            if (same && i < Nh - 1 && ds_find(c, b + Nw) != setid) same = false;
            if (same) { const int v = c.ridToPlid[setid]; bm = v < 0 ? 0 : v; }     /* std::map::operator[]: a missing root reads 0 */
        }
        c.blkMap[b] = bm;
        if (bm >= 0) {
            isValid[bm] = 1;
            for (int y = i * AHC_WIN; y < (i + 1) * AHC_WIN; y++)
                for (int x = j * AHC_WIN; x < (j + 1) * AHC_WIN; x++) c.mem[(size_t)y * w + x] = (int16_t)bm;
        }
    }
    fence();
    /* seeds of the flood fill, in block raster order: the border pixels between a block and its upper / left neighbour */
    int nRf = 0;
    for (int base = 0; base < NB; base += 64) {
        const int b = base + lane;
        int cnt = 0, bm = -1, up = -1, left = -1, i = 0, j = 0;
        if (b < NB) {
            i = b / Nw; j = b - i * Nw;
            bm = c.blkMap[b];
            up = i > 0 ? c.blkMap[b - Nw] : -2;
            left = j > 0 ? c.blkMap[b - 1] : -2;
            if (bm < 0) cnt = (up >= 0 ? AHC_WIN - 1 : 0) + (left >= 0 ? AHC_WIN - 1 : 0);
            else cnt = ((i > 0 && up != bm) ? AHC_WIN - 1 : 0) + ((j > 0 && left != bm) ? AHC_WIN - 1 : 0);
        }
        const int incl = drfe_wave_incl_scan(cnt, lane);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (nRf + tot > P.rfCap) { c.status |= 2; break; }
        int at = nRf + incl - cnt;
        if (cnt) {
            if (bm < 0) {
                if (up >= 0) { const int spix = (i * AHC_WIN - 1) * w + j * AHC_WIN; for (int k = 1; k < AHC_WIN; ++k) c.rf[at++] = (uint32_t)(spix + k) | (uint32_t)up << AHCD_PIXBITS; }
                if (left >= 0) { const int spix = (i * AHC_WIN) * w + j * AHC_WIN - 1; for (int k = 0; k < AHC_WIN - 1; ++k) c.rf[at++] = (uint32_t)(spix + k * w) | (uint32_t)left << AHCD_PIXBITS; }
            } else {
                if (i > 0 && up != bm) { const int spix = (i * AHC_WIN) * w + j * AHC_WIN; for (int k = 0; k < AHC_WIN - 1; ++k) c.rf[at++] = (uint32_t)(spix + k) | (uint32_t)bm << AHCD_PIXBITS; }
                if (j > 0 && left != bm) { const int spix = (i * AHC_WIN) * w + j * AHC_WIN; for (int k = 1; k < AHC_WIN; ++k) c.rf[at++] = (uint32_t)(spix + k * w) | (uint32_t)bm << AHCD_PIXBITS; }
            }
        }
        nRf += tot;
    }
    nRf = uni_i(nRf);
    /* the extracted nodes start the flood fill without neighbours (disconnectAll); connects between them get fresh lists */
    if (c.poolUsed + nEx * nEx > P.poolCap) c.status |= 2;
    if (uni_b(__ballot(c.status != 0) != 0)) { if (lane == 0) { out[0] = 0; out[1] = 2; } return; }
    for (int k = lane; k < nEx; k += 64) { c.nbOff[ex[k]] = c.poolUsed + k * nEx; c.nbLen[ex[k]] = 0; }
    c.poolUsed += nEx * nEx;
    fence();

    /* ---- hand over to k_ahc_refine: the extracted nodes, which of them kept blocks, the seeds' count, the graph's fill ---- */
    GLOBAL_AS int* ho = (GLOBAL_AS int*)F.handoff;
    for (int k = lane; k < nEx; k += 64) { ho[AHCD_HO_EX + k] = ex[k]; ho[AHCD_HO_VALID + k] = isValid[k]; }
    if (lane == 0) { ho[0] = nEx; ho[1] = nRf; ho[2] = c.nNodes; ho[3] = c.poolUsed; ho[AHCD_HO_LABELS] = 0; out[0] = 0; out[1] = -1; }
#ifdef AHC_PROFILE
    TP();
    if (lane == 0) for (int k = 0; k + 1 < tpi; k++) ho[AHCD_HO_TP + k] = (int)(tp[k + 1] - tp[k]);
#endif
}

extern "C" __global__ __launch_bounds__(64) void k_ahc_cluster(const AhcDevFrame* __restrict__ frames, AhcDevParams P)
{
    ahc_cluster_frame<AHCD_HEAP_SMALL, AHCD_LIST_SMALL>(frames, P);
}
extern "C" __global__ __launch_bounds__(64) void k_ahc_cluster_big(const AhcDevFrame* __restrict__ frames, AhcDevParams P)
{
    ahc_cluster_frame<AHCD_HEAP_BIG, AHCD_LIST_BIG>(frames, P);
}

/* the second half of a frame: flood fill from the seeds, the re-merge of the grown planes, labels, member lists, plane clouds.
 * Its own kernel because its LDS need is half of the clustering's (no 3200-entry queue): the CU holds five of these wavefronts,
 * or three of k_ahc_cluster, where the single kernel's 57 KB allowed two - and whatever LDS these long-running wavefronts hold
 * is what the line path's growth (27 KB per frame) cannot use. */
template <int HEAP, int LIST>
__device__ __forceinline__ void ahc_refine_frame(const AhcDevFrame* __restrict__ frames, const AhcDevParams& P)
{
    __shared__ float heapKey[AHCD_MAXEX];                     /* the re-merge's queue: at most the extracted planes */
    __shared__ uint16_t heapId[AHCD_MAXEX];
    __shared__ uint16_t lA[LIST], lB[LIST], lU[2 * LIST];
    __shared__ double win[18];
    __shared__ int ex[AHCD_MAXEX], ex2[AHCD_MAXEX], plidmap[AHCD_MAXEX];
    __shared__ uint8_t isValid[AHCD_MAXEX];
    __shared__ double plN[AHCD_MAXEX][3], plC[AHCD_MAXEX][3], plMse[AHCD_MAXEX];
    __shared__ int8_t blkLds[HEAP];                           /* flood fill: 1 = the block is kept whole (its pixels are final) */
    __shared__ uint32_t ffTmp[128];                           /* flood fill: chain_sort128's way home */
    const AhcDevFrame F = frames[blockIdx.x];
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    Ctx c;
    c.depth = (const GLOBAL_AS uint16_t*)F.depth; c.rowStride = (int)F.rowStride;
    c.S = (GLOBAL_AS double*)F.nodeS; c.fit = (GLOBAL_AS double*)F.nodeFit; c.N = (GLOBAL_AS int*)F.nodeN; c.rid = (GLOBAL_AS int*)F.nodeRid;
    c.nouse = (GLOBAL_AS uint8_t*)F.nodeNouse; c.nbOff = (GLOBAL_AS int*)F.nbOff; c.nbLen = (GLOBAL_AS int*)F.nbLen; c.pool = (GLOBAL_AS int*)F.nbPool;
    c.dsParent = (GLOBAL_AS int*)F.dsParent; c.dsSize = (GLOBAL_AS int*)F.dsSize; c.G = (GLOBAL_AS int*)F.G; c.blkMap = (GLOBAL_AS int*)F.blkMap;
    c.ridToPlid = (GLOBAL_AS int*)F.ridToPlid; c.mem = (GLOBAL_AS int16_t*)F.membership; c.dist = (GLOBAL_AS float*)F.distMap; c.rf = (GLOBAL_AS uint32_t*)F.rf;
    c.heapKey = heapKey; c.heapId = heapId; c.lA = lA; c.lB = lB; c.lU = lU; c.win = win;
    c.heapSize = 0; c.status = 0; c.lane = lane; c.listCap = LIST; c.P = P;
    GLOBAL_AS int* out = (GLOBAL_AS int*)F.out;
    const GLOBAL_AS int* ho = (const GLOBAL_AS int*)F.handoff;
    const int w = P.w, h = P.h, Nw = P.Nw, Nh = P.Nh, NB = P.NB;
    if (out[1] != -1) return;                                 /* k_ahc_cluster gave the frame back to the host */
    const int nEx = ho[0], nRf = ho[1];
    c.nNodes = ho[2]; c.poolUsed = ho[3];
    for (int k = lane; k < nEx; k += 64) {
        const int nd = ho[AHCD_HO_EX + k];
        ex[k] = nd; isValid[k] = (uint8_t)ho[AHCD_HO_VALID + k];
        for (int q = 0; q < 3; q++) { plC[k][q] = c.fit[8 * (size_t)nd + q]; plN[k][q] = c.fit[8 * (size_t)nd + 3 + q]; }
        plMse[k] = c.fit[8 * (size_t)nd + 6];
    }
    fence();
#ifdef AHC_PROFILE
    unsigned long long tp[8]; int tpi = 0;
#endif
    TP();
    int rfTotal = 0;
    /* ---- floodFill: FIFO over (pixel, plane).  Sixteen queue entries per step, their four neighbours each in the 64 lanes.
     * What a lane needs that never changes (block map, depth -> point -> distance to the entry's plane, the inlier test) and
     * the fetch of what does change (the neighbour's label and its best distance so far) happen for all 64 at once.  The
     * reference applies the 64 (entry, neighbour) visits one after the other; the only thing one visit passes to a later one
     * is the state of a PIXEL both look at - queue appends keep lane order under a ballot prefix, and connecting two planes
     * is a set insertion.  So every lane finds the previous lane of the step that targets its pixel, the visits are applied in
     * rounds by depth in those chains (94 % of the steps have one; usually depth 1, more when a pixel sits in the queue several
     * times), a lane taking its input state from its predecessor's output, and the last lane of a chain stores the pixel.  One
     * memory fence per step. ---- */
    {
#ifdef AHC_PROFILE
        unsigned ffSteps = 0, ffDup = 0, ffDepth = 0, ffHave = 0;
#endif
        int head = 0, tail = nRf;
        const double fx = P.fx, fy = P.fy, cx = P.cx, cy = P.cy, factor = P.factor;
        for (int k = lane; k < NB; k += 64) blkLds[k] = (int8_t)(c.blkMap[k] >= 0 ? 1 : 0);
        fence();
        const int eLane = lane >> 2, nbLane = lane & 3;          /* entry of a half-step, neighbour (left, right, up, down) */
        const uint32_t magicW = 0xFFFFFFFFu / (uint32_t)w + 1u;   /* ceil(2^32 / w): n / w == umulhi(n, magicW) while n * (magicW * w - 2^32) < 2^32, i.e. for every pixel index below 2^21 at w <= 2048 */
        /* Round 6: THIRTY-TWO queue entries per step - two visits per lane (slot 0: entries 0..15 of the step, slot 1: entries 16..31; visit
         * id = slot * 64 + lane = visiting order).  What a step pays once whatever it holds - the round trip of the labels / distances /
         * depths, the fence, the queue bookkeeping - is now paid per 128 visits; the chains of same-pixel visits come from a 128-key sort. */
        uint32_t eNext0 = 0, eNext1 = 0;
        int nextFrom0 = -1, nextFrom1 = -1;                      /* queue positions eNext0 / eNext1 were prefetched for (this lane) */
        struct Visit { bool have, in, push, dirty, distDirty, meets; int cIdx, plid, trail, other; float cdist, old; };
#ifdef AHC_PROFILE_FF
        unsigned long long ffc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FFC(k) do { const unsigned long long t__ = __builtin_readcyclecounter(); ffc[k] += t__ - ffT; ffT = t__; } while (0)
#else
#define FFC(k) (void)0
#endif
        while (head < tail) {
#ifdef AHC_PROFILE_FF
            unsigned long long ffT = __builtin_readcyclecounter();
#endif
            const int cnt = min(32, tail - head);
            Visit V[2];
            uint32_t e[2];
            {
                const int q0 = head + eLane, q1 = head + 16 + eLane;
                e[0] = eLane < cnt ? (nextFrom0 == q0 ? eNext0 : c.rf[q0]) : 0u;
                e[1] = 16 + eLane < cnt ? (nextFrom1 == q1 ? eNext1 : c.rf[q1]) : 0u;
                /* the next step's entries, when the queue already holds them: their fetch overlaps this step */
                if (q0 + 32 < tail) { eNext0 = c.rf[q0 + 32]; nextFrom0 = q0 + 32; }
                if (q1 + 32 < tail) { eNext1 = c.rf[q1 + 32]; nextFrom1 = q1 + 32; }
            }
            int cyn[2], cxn[2];
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                Visit& v = V[sl];
                const bool mine = 16 * sl + eLane < cnt;
                const int sIdx = (int)(e[sl] & AHCD_PIXMASK);
                v.plid = (int)(e[sl] >> AHCD_PIXBITS);
                const int sy = (int)__umulhi((uint32_t)sIdx, magicW), sx = sIdx - sy * w;      /* sIdx / w: exact for sIdx < 2^21, w <= 2048 */
                cxn[sl] = sx; cyn[sl] = sy;
                bool have = false;
                if (nbLane == 0) { have = sx > 0; cxn[sl] = sx - 1; }
                else if (nbLane == 1) { have = sx < w - 1; cxn[sl] = sx + 1; }
                else if (nbLane == 2) { have = sy > 0; cyn[sl] = sy - 1; }
                else { have = sy < h - 1; cyn[sl] = sy + 1; }
                have = have && mine;
                v.cIdx = cyn[sl] * w + cxn[sl];
                if (have) {
                    const int by = cyn[sl] / AHC_WIN, bx = cxn[sl] / AHC_WIN;
                    if (by < Nh && bx < Nw && blkLds[by * Nw + bx]) have = false;      /* pixels of kept blocks are final */
                }
                v.have = have; v.in = false; v.push = false; v.dirty = false; v.distDirty = false; v.meets = false;
                v.trail = 0; v.other = -1; v.cdist = -1.f; v.old = 0.f;
            }
            FFC(0);
            /* both slots' loads in one round trip */
            uint16_t dz[2] = {0, 0};
#pragma unroll
            for (int sl = 0; sl < 2; sl++)
                if (V[sl].have) {
                    V[sl].trail = c.mem[V[sl].cIdx];
                    V[sl].old = c.dist[V[sl].cIdx];
                    dz[sl] = c.depth[(size_t)cyn[sl] * c.rowStride + cxn[sl]];
                }
#ifdef AHC_PROFILE_FF
            if (__ballot(dz[0] == 65535 && dz[1] == 65535 && V[0].old == 12345.f && V[1].trail == 777) == ~0ull) c.status |= 8;     /* wait for the loads here */
#endif
            FFC(1);
#pragma unroll
            for (int sl = 0; sl < 2; sl++)
                if (V[sl].have) {
                    Visit& v = V[sl];
                    double z = (double)dz[sl] * factor;
                    if (z > 5.0) z = 0.0;
                    if (z != 0.0) {
                        const double px = ((double)cxn[sl] - cx) * z / fx, py = ((double)cyn[sl] - cy) * z / fy;
                        const int pl = v.plid;
                        const double sd = plN[pl][0] * (px - plC[pl][0]) + plN[pl][1] * (py - plC[pl][1]) + plN[pl][2] * (z - plC[pl][2]);
                        v.cdist = (float)fabs(sd);
                        const double cd = (double)v.cdist;
                        v.in = cd * cd < 9 * plMse[pl] + 1e-5;
                    }
                }
            FFC(2);
            /* chains of visits that target the same pixel, in visiting order: the depth in the chain is the number of visitors before */
            const unsigned long long live0 = __ballot(V[0].have), live1 = __ballot(V[1].have);
            int prev[2] = {-1, -1}, depth[2] = {0, 0};
            bool isLast[2] = {true, true};
            if (live1) chain_sort128(V[0].have, V[0].cIdx, V[1].have, V[1].cIdx, lane, ffTmp, prev[0], depth[0], isLast[0], prev[1], depth[1], isLast[1]);
            else if (live0) chain_sort64(V[0].have, V[0].cIdx, lane, prev[0], depth[0], isLast[0]);
            FFC(3);
            int maxDepth = 0;
            while (__ballot(depth[0] > maxDepth || depth[1] > maxDepth)) maxDepth++;
#ifdef AHC_PROFILE
            ffSteps++; if (maxDepth > 0) ffDup++; ffDepth += maxDepth; ffHave += __popcll(live0) + __popcll(live1);
#endif
            for (int r = 0; r <= maxDepth; r++) {
                if (r > 0) {
                    /* a visit of depth r continues from its predecessor's output: label | flags in one word, the distance */
                    const int w0 = (V[0].trail & 0xFFFF) | (V[0].dirty ? 0x10000 : 0) | (V[0].distDirty ? 0x20000 : 0);
                    const int w1 = (V[1].trail & 0xFFFF) | (V[1].dirty ? 0x10000 : 0) | (V[1].distDirty ? 0x20000 : 0);
                    const float o0 = V[0].old, o1 = V[1].old;
#pragma unroll
                    for (int sl = 0; sl < 2; sl++) {
                        const int src = prev[sl] < 0 ? lane : prev[sl];
                        const int sa = __shfl(w0, src & 63), sb = __shfl(w1, src & 63);
                        const float oa = __shfl(o0, src & 63), ob = __shfl(o1, src & 63);
                        if (V[sl].have && depth[sl] == r) {
                            const int sIn = (src & 64) ? sb : sa;
                            V[sl].trail = (int)(int16_t)(sIn & 0xFFFF); V[sl].dirty = (sIn & 0x10000) != 0; V[sl].distDirty = (sIn & 0x20000) != 0;
                            V[sl].old = (src & 64) ? ob : oa;
                        }
                    }
                }
#pragma unroll
                for (int sl = 0; sl < 2; sl++) {
                    Visit& v = V[sl];
                    if (v.have && depth[sl] == r) {
                        const bool active = !(v.trail <= -6) && !(v.trail >= 0 && v.trail == v.plid);
                        if (active) {
                            if (v.in && v.trail >= 0) { v.meets = true; v.other = v.trail; }
                            if (v.in && v.cdist < v.old) { v.trail = v.plid; v.old = v.cdist; v.push = true; v.dirty = true; v.distDirty = true; }
                            else if (v.trail < 0) { v.trail = v.trail - 1; v.dirty = true; }
                        }
                    }
                }
            }
            FFC(4);
            /* planes that meet and are similar enough are connected for the re-merge (rare: one lane at a time; a set insertion) */
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                unsigned long long cm = __ballot(V[sl].meets);
                while (cm) {
                    const int l = __builtin_ctzll(cm);
                    cm &= cm - 1;
                    const int o = ex[rl_i(V[sl].other, l)], me = ex[rl_i(V[sl].plid, l)];
                    if (lane == 0 && similarity(c, me, o) >= P.cos30) { list_insert(c, o, me); list_insert(c, me, o); }
                    fence();
                }
            }
#pragma unroll
            for (int sl = 0; sl < 2; sl++)
                if (V[sl].have && isLast[sl] && V[sl].dirty) {
                    c.mem[V[sl].cIdx] = (int16_t)V[sl].trail;
                    if (V[sl].distDirty) c.dist[V[sl].cIdx] = V[sl].old;
                }
            const unsigned long long pm0 = __ballot(V[0].push), pm1 = __ballot(V[1].push);
            if (pm0 | pm1) {
                const int np0 = __popcll(pm0), np = np0 + __popcll(pm1);
                if (tail + np > P.rfCap) { c.status |= 2; break; }
                if (V[0].push) c.rf[tail + __popcll(pm0 & lt)] = (uint32_t)V[0].cIdx | (uint32_t)V[0].plid << AHCD_PIXBITS;
                if (V[1].push) c.rf[tail + np0 + __popcll(pm1 & lt)] = (uint32_t)V[1].cIdx | (uint32_t)V[1].plid << AHCD_PIXBITS;
                tail += np;
            }
            head += cnt;
            fence();
            FFC(5);
        }
#ifdef AHC_PROFILE_FF
        if (blockIdx.x == 0 && lane == 0) printf("flood fill, frame 0, shader cycles: entries + geometry %llu, loads %llu, f64 distances %llu, chain sort %llu, rounds %llu, connects + stores + appends + fence %llu\n", ffc[0], ffc[1], ffc[2], ffc[3], ffc[4], ffc[5]);
#endif
        rfTotal = tail;
#ifdef AHC_PROFILE
        if (lane == 0) { ((GLOBAL_AS int*)F.handoff)[AHCD_HO_TP + 4] = (int)ffSteps; ((GLOBAL_AS int*)F.handoff)[AHCD_HO_TP + 5] = (int)ffDup; ((GLOBAL_AS int*)F.handoff)[AHCD_HO_TP + 6] = (int)ffDepth; ((GLOBAL_AS int*)F.handoff)[AHCD_HO_TP + 7] = (int)ffHave; }
#endif
    }
    if (uni_b(c.status != 0)) { if (lane == 0) { out[0] = 0; out[1] = c.status; } return; }
    TP();

    /* ---- re-merge the grown planes (the valid ones, by MSE) and relabel ---- */
    c.heapSize = 0;
    if (lane == 0) {
        for (int i = 0; i < nEx; i++)
            if (isValid[i]) heap_push(c, c.fit[8 * (size_t)ex[i] + 6], ex[i]);
    }
    c.heapSize = uni_i(c.heapSize);
    int nFinal = 0;
    cluster(c, ex2, nFinal);
    nFinal = uni_i(nFinal);
    if (uni_b(c.status != 0) || nFinal > P.planeCap || nFinal > 254) { if (lane == 0) { out[0] = nFinal; out[1] = c.status | 2; } return; }
    for (int i = lane; i < nEx; i += 64) {
        int pm = -1;
        if (isValid[i]) {
            const int r = ds_find(c, rid_block(c.rid[ex[i]]));
            for (int j = 0; j < nFinal; j++)
                if (r == rid_block(c.rid[ex2[j]])) { pm = j; break; }
        }
        plidmap[i] = pm;
    }
    GLOBAL_AS drfe_plane* planes = (GLOBAL_AS drfe_plane*)F.planes;
    for (int i = lane; i < nFinal; i += 64) {
        const int nd = ex2[i];
        for (int q = 0; q < 3; q++) { planes[i].normal[q] = c.fit[8 * (size_t)nd + 3 + q]; planes[i].center[q] = c.fit[8 * (size_t)nd + q]; }
        planes[i].mse = c.fit[8 * (size_t)nd + 6]; planes[i].curvature = c.fit[8 * (size_t)nd + 7];
        planes[i].n_points = c.N[nd]; planes[i].rid = rid_block(c.rid[nd]);
    }
    fence();
    TP();
    /* final plane per pixel, the label image, the member lists (raster order per plane) and each plane's cloud are pure functions of
     * what is in memory now - the raw labels, this map and the depth image: the k_ahc_labels_* kernels compute them with 128
     * wavefronts per frame instead of this one (round 5; they were 5 of this kernel's 22 ms) */
    {
        GLOBAL_AS int* how = (GLOBAL_AS int*)F.handoff;
        for (int i = lane; i < nEx; i += 64) how[AHCD_HO_EX + i] = plidmap[i];
        fence();
        if (lane == 0) how[AHCD_HO_LABELS] = 1;
    }
    TP();
    if (lane == 0) { out[0] = nFinal; out[1] = 0; out[2] = rfTotal; out[3] = c.nNodes; }
#ifdef AHC_PROFILE
    /* phase times (100 MHz ticks) into the head of the flood-fill queue, which nobody reads any more */
    if (lane == 0) {
        for (int k = 0; k < 3; k++) c.rf[k] = (uint32_t)ho[AHCD_HO_TP + k];
        for (int k = 0; k + 1 < tpi; k++) c.rf[3 + k] = (uint32_t)(tp[k + 1] - tp[k]);
    }
#endif
}

#ifdef AHC_REFINE_WAVES_PER_EU         /* experiment builds: cap the allocation (3 -> 168 VGPRs, 19 spilled) */
#define AHC_REFINE_OCC __attribute__((amdgpu_waves_per_eu(AHC_REFINE_WAVES_PER_EU, AHC_REFINE_WAVES_PER_EU)))
#else
#define AHC_REFINE_OCC
#endif
extern "C" __global__ __launch_bounds__(64) AHC_REFINE_OCC void k_ahc_refine(const AhcDevFrame* __restrict__ frames, AhcDevParams P)
{
    ahc_refine_frame<AHCD_HEAP_SMALL, AHCD_LIST_SMALL>(frames, P);
}
extern "C" __global__ __launch_bounds__(64) void k_ahc_refine_big(const AhcDevFrame* __restrict__ frames, AhcDevParams P)
{
    ahc_refine_frame<AHCD_HEAP_BIG, AHCD_LIST_BIG>(frames, P);
}


/* ---- labels, member lists, plane clouds: three small kernels behind k_ahc_refine -------------------------------------------------
 * AHCL_NBLK wavefronts per frame, each on a contiguous range of pixels (raster order):
 *   k_ahc_labels_count    final label of every pixel (raw label -> plane through the re-merge's map), the label image, and per range
 *                         the number of members / of kept cloud points of every plane;
 *   k_ahc_labels_prefix   one wavefront per frame: the ranges' counts into offsets (plane by plane, range by range), memberOff, jobs;
 *   k_ahc_labels_scatter  member indices and cloud points of a range written from its offsets on: raster order per plane, exactly
 *                         the single wavefront's lists (PlaneFitter's membership lists and Frame::ComputePlanes' gather,
 *                         src/Frame.cc:985-1000).
 * Scratch: the flood-fill queue, which nobody reads any more (behind the words the profile build keeps there). */
#define AHCL_NBLK 128
#define AHCL_STRIDE 128                  /* counts per range: one per final plane (<= planeCap < 128) */
#define AHCL_SCRATCH 64                  /* first word of the scratch inside rf */
__device__ __forceinline__ void ahcl_range(int npx, int b, int& k0, int& k1)
{
    const int per = (((npx + AHCL_NBLK - 1) / AHCL_NBLK) + 63) & ~63;
    k0 = min(npx, b * per); k1 = min(npx, k0 + per);
}

extern "C" __global__ __launch_bounds__(64) void k_ahc_labels_count(const AhcDevFrame* __restrict__ frames, AhcDevParams P)
{
    __shared__ int plidmap[AHCD_MAXEX];
    __shared__ int cnt[AHCL_STRIDE], kcnt[AHCL_STRIDE];
    const AhcDevFrame F = frames[blockIdx.y];
    const int lane = threadIdx.x, b = blockIdx.x;
    const GLOBAL_AS int* out = (const GLOBAL_AS int*)F.out;
    const GLOBAL_AS int* ho = (const GLOBAL_AS int*)F.handoff;
    if (out[1] != 0 || ho[AHCD_HO_LABELS] != 1) return;
    const int nFinal = out[0], nEx = ho[0];
    const int w = P.w, npx = P.w * P.h;
    for (int i = lane; i < nEx; i += 64) plidmap[i] = ho[AHCD_HO_EX + i];
    for (int i = lane; i < AHCL_STRIDE; i += 64) { cnt[i] = 0; kcnt[i] = 0; }
    fence();
    GLOBAL_AS int16_t* mem = (GLOBAL_AS int16_t*)F.membership;
    GLOBAL_AS uint8_t* seg = (GLOBAL_AS uint8_t*)F.seg;
    const GLOBAL_AS uint16_t* depth = (const GLOBAL_AS uint16_t*)F.depth;
    const int rowStride = (int)F.rowStride;
    const double gfactor = P.factor;
    const float maxPointDist = P.maxPointDist;
    auto depth_at = [&](int k) -> uint16_t { const int row = k / w, col = k - row * w; return depth[(size_t)row * rowStride + col]; };
    int k0, k1;
    ahcl_range(npx, b, k0, k1);
    int rawNext = k0 + lane < k1 ? (int)mem[k0 + lane] : -1;
    uint16_t depNext = k0 + lane < k1 ? depth_at(k0 + lane) : (uint16_t)0;
    for (int base = k0; base < k1; base += 64) {
        const int k = base + lane;
        const int raw = rawNext;
        const uint16_t dep = depNext;
        if (k + 64 < k1) { rawNext = mem[k + 64]; depNext = depth_at(k + 64); }
        int pl = -1;
        bool keep = false;
        if (k < k1) {
            pl = raw >= 0 ? plidmap[raw] : -1;
            mem[k] = (int16_t)pl;
            seg[k] = (uint8_t)(pl + 1);
            if (pl >= 0) {
                double z = (double)dep * gfactor;
                if (z > 5.0) z = 0.0;
                keep = !((float)z > maxPointDist);
            }
        }
        /* per-plane counts: the lanes of a plane are counted once per chunk */
        const unsigned long long kept = __ballot(keep);
        unsigned long long todo = __ballot(pl >= 0);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            const int v = rl_i(pl, l);
            const unsigned long long same = __ballot(pl == v);
            if (lane == l) { cnt[v] += __popcll(same); kcnt[v] += __popcll(same & kept); }
            todo &= ~same;
        }
    }
    fence();
    GLOBAL_AS int* sc = (GLOBAL_AS int*)F.rf + AHCL_SCRATCH;
    for (int v = lane; v < nFinal; v += 64) {
        sc[(size_t)b * AHCL_STRIDE + v] = cnt[v];
        sc[(size_t)(AHCL_NBLK + b) * AHCL_STRIDE + v] = kcnt[v];
    }
}

extern "C" __global__ __launch_bounds__(64) void k_ahc_labels_prefix(const AhcDevFrame* __restrict__ frames, AhcDevParams P)
{
    __shared__ int tot[AHCL_STRIDE + 1], ktot[AHCL_STRIDE + 1];
    const AhcDevFrame F = frames[blockIdx.x];
    const int lane = threadIdx.x;
    const GLOBAL_AS int* out = (const GLOBAL_AS int*)F.out;
    const GLOBAL_AS int* ho = (const GLOBAL_AS int*)F.handoff;
    if (out[1] != 0 || ho[AHCD_HO_LABELS] != 1) return;
    const int nFinal = out[0];
    GLOBAL_AS int* sc = (GLOBAL_AS int*)F.rf + AHCL_SCRATCH;
    /* range by range, for the lane's plane(s): a range's count becomes the number of the plane's members in the ranges before it */
    for (int v = lane; v < nFinal; v += 64) {
        int run = 0, krun = 0;
        for (int b = 0; b < AHCL_NBLK; b++) {
            const int x = sc[(size_t)b * AHCL_STRIDE + v], kx = sc[(size_t)(AHCL_NBLK + b) * AHCL_STRIDE + v];
            sc[(size_t)b * AHCL_STRIDE + v] = run; sc[(size_t)(AHCL_NBLK + b) * AHCL_STRIDE + v] = krun;
            run += x; krun += kx;
        }
        tot[v + 1] = run; ktot[v + 1] = krun;
    }
    if (lane == 0) { tot[0] = 0; ktot[0] = 0; }
    fence();
    GLOBAL_AS int* memberOff = (GLOBAL_AS int*)F.memberOff;
    GLOBAL_AS int* jobs = (GLOBAL_AS int*)F.jobs;
    if (lane == 0) {
        for (int i = 0; i < nFinal; i++) { tot[i + 1] += tot[i]; ktot[i + 1] += ktot[i]; }
        for (int i = 0; i <= nFinal; i++) memberOff[i] = tot[i];
        if (jobs) for (int i = 0; i < nFinal; i++) { jobs[2 * i] = F.ptsBase + ktot[i]; jobs[2 * i + 1] = ktot[i + 1] - ktot[i]; }
    }
    fence();
    /* where each plane's list / cloud starts: behind the ranges' tables */
    for (int v = lane; v < nFinal; v += 64) {
        sc[(size_t)(2 * AHCL_NBLK) * AHCL_STRIDE + v] = tot[v];
        sc[(size_t)(2 * AHCL_NBLK + 1) * AHCL_STRIDE + v] = ktot[v];
    }
}

extern "C" __global__ __launch_bounds__(64) void k_ahc_labels_scatter(const AhcDevFrame* __restrict__ frames, AhcDevParams P)
{
    __shared__ int counts[AHCL_STRIDE], kcounts[AHCL_STRIDE];
    const AhcDevFrame F = frames[blockIdx.y];
    const int lane = threadIdx.x, b = blockIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const GLOBAL_AS int* out = (const GLOBAL_AS int*)F.out;
    const GLOBAL_AS int* ho = (const GLOBAL_AS int*)F.handoff;
    if (out[1] != 0 || ho[AHCD_HO_LABELS] != 1) return;
    const int nFinal = out[0];
    const int w = P.w, npx = P.w * P.h;
    const GLOBAL_AS int* sc = (const GLOBAL_AS int*)F.rf + AHCL_SCRATCH;
    for (int v = lane; v < nFinal; v += 64) {
        counts[v] = sc[(size_t)(2 * AHCL_NBLK) * AHCL_STRIDE + v] + sc[(size_t)b * AHCL_STRIDE + v];
        kcounts[v] = sc[(size_t)(2 * AHCL_NBLK + 1) * AHCL_STRIDE + v] + sc[(size_t)(AHCL_NBLK + b) * AHCL_STRIDE + v];
    }
    fence();
    const GLOBAL_AS int16_t* mem = (const GLOBAL_AS int16_t*)F.membership;
    const GLOBAL_AS uint16_t* depth = (const GLOBAL_AS uint16_t*)F.depth;
    const int rowStride = (int)F.rowStride;
    GLOBAL_AS int* memberIdx = (GLOBAL_AS int*)F.memberIdx;
    GLOBAL_AS float* pts = (GLOBAL_AS float*)F.pts;
    const double gfx = P.fx, gfy = P.fy, gcx = P.cx, gcy = P.cy, gfactor = P.factor;
    const float maxPointDist = P.maxPointDist;
    auto depth_at = [&](int k) -> uint16_t { const int row = k / w, col = k - row * w; return depth[(size_t)row * rowStride + col]; };
    int k0, k1;
    ahcl_range(npx, b, k0, k1);
    int rawNext = k0 + lane < k1 ? (int)mem[k0 + lane] : -1;
    uint16_t depNext = k0 + lane < k1 ? depth_at(k0 + lane) : (uint16_t)0;
    for (int base = k0; base < k1; base += 64) {
        const int k = base + lane;
        const int pl = k < k1 ? rawNext : -1;
        const uint16_t dep = depNext;
        if (k + 64 < k1) { rawNext = mem[k + 64]; depNext = depth_at(k + 64); }
        bool keep = false;
        float X = 0.f, Y = 0.f, Z = 0.f;
        if (pl >= 0 && pts) {
            /* PlaneDetection::readDepthImage (src/PlaneExtractor.cpp:39-52): doubles, K floats promoted; the cloud holds floats */
            const int row = k / w, col = k - row * w;
            const double z = (double)dep * gfactor;
            if (!(z > 5.0)) {
                X = (float)(((double)col - gcx) * z / gfx);
                Y = (float)(((double)row - gcy) * z / gfy);
                Z = (float)z;
            }
            keep = !(Z > maxPointDist);
        }
        const unsigned long long kept = __ballot(keep);
        unsigned long long todo = __ballot(pl >= 0);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            const int v = rl_i(pl, l);
            const unsigned long long same = __ballot(pl == v);
            if (pl == v) {
                memberIdx[counts[v] + __popcll(same & lt)] = k;
                if (keep) {
                    const size_t q = 3 * (size_t)(kcounts[v] + __popcll(same & kept & lt));
                    pts[q] = X; pts[q + 1] = Y; pts[q + 2] = Z;
                }
            }
            wave_order();
            if (lane == l) { counts[v] += __popcll(same); kcounts[v] += __popcll(same & kept); }
            todo &= ~same;
        }
        wave_order();
    }
}

int drfe_ahc_device_fits(int w, int h)
{
    /* w <= 2048: the flood fill divides pixel indices by w with a 32-bit reciprocal (exact for indices below 2^21 up to that width) */
    return w >= AHC_WIN && w <= 2048 && h >= AHC_WIN && (w / AHC_WIN) * (h / AHC_WIN) <= AHCD_HEAP_BIG && (size_t)w * h <= ((size_t)1 << AHCD_PIXBITS);
}

hipError_t drfe_launch_ahc_frames(const AhcDevFrame* d_frames, int nframes, const AhcDevParams& P, hipStream_t s, hipEvent_t* ev3)
{
    if (nframes <= 0) return hipSuccess;
    static_assert(AHC_HANDOFF_INTS >= AHCD_HO_LABELS + 1, "handoff words");
    if (P.NB > AHCD_HEAP_BIG || (size_t)P.w * P.h > ((size_t)1 << AHCD_PIXBITS) || P.w > 2048) return hipErrorInvalidValue;
    if (P.NB <= AHCD_HEAP_SMALL) {
        hipLaunchKernelGGL(k_ahc_cluster, dim3(nframes), dim3(64), 0, s, d_frames, P);
        if (ev3) (void)hipEventRecord(ev3[0], s);
        hipLaunchKernelGGL(k_ahc_refine, dim3(nframes), dim3(64), 0, s, d_frames, P);
    } else {
        hipLaunchKernelGGL(k_ahc_cluster_big, dim3(nframes), dim3(64), 0, s, d_frames, P);
        if (ev3) (void)hipEventRecord(ev3[0], s);
        hipLaunchKernelGGL(k_ahc_refine_big, dim3(nframes), dim3(64), 0, s, d_frames, P);
    }
    if (ev3) (void)hipEventRecord(ev3[1], s);
    if (P.planeCap >= AHCL_STRIDE || (size_t)P.rfCap < (size_t)AHCL_SCRATCH + (size_t)(2 * AHCL_NBLK + 2) * AHCL_STRIDE) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_ahc_labels_count, dim3(AHCL_NBLK, nframes), dim3(64), 0, s, d_frames, P);
    hipLaunchKernelGGL(k_ahc_labels_prefix, dim3(nframes), dim3(64), 0, s, d_frames, P);
    hipLaunchKernelGGL(k_ahc_labels_scatter, dim3(AHCL_NBLK, nframes), dim3(64), 0, s, d_frames, P);
    if (ev3) (void)hipEventRecord(ev3[2], s);
    return hipGetLastError();
}

#ifdef AHC_PROFILE
extern "C" int drfe_debug_ahc_cluster_profile(unsigned long long* out8)
{
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ahcProf), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ahcProf), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
