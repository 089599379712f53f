/* orb_kernels.hip — gfx950 kernels of the ORB extraction path.
 *
 * Every kernel is batch-wide: blockIdx.y (or .z) is the frame slot, so one launch covers all frames
 * of a batch and the grid is >> 256 workgroups even though a single 640x480 frame is only ~7 MB of
 * traffic.  All arithmetic that decides an output bit is integer, or float32 without FMA contraction
 * (-ffp-contract=off) through include/drfe_math.h.
 *
 *   k_pyr_level0   copyMakeBorder(REFLECT_101) of the input           (reference src/ORBextractor.cc:1127)
 *   k_pyr_resize   cv::resize(INTER_LINEAR) cascade + border          (:1120-1123)
 *   k_fast_cells   per-cell cv::FAST(20) -> NMS -> fallback cv::FAST(7) (:789-829)
 *   k_quadtree     DistributeOctTree                                   (:539-763)
 *   k_blur         GaussianBlur 7x7 sigma 2 on the interior            (:1085-1086)
 *   k_orient_desc  IC_Angle + computeOrbDescriptor + keypoint finishing (:77-147, :837-847, :1095-1101)
 */
#include "drfe_internal.h"
#include "../../include/drfe_math.h"

#define WAVE 64

__device__ __forceinline__ int reflect101(int p, int n)
{
    /* single reflection: callers guarantee |overshoot| < n */
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p;
}

/* ------------------------------------------------------------------------------------------------ */
/* pyramid                                                                                          */

/* copyMakeBorder(REFLECT_101) of the input into level 0.  Streaming kernels on this part want 16 bytes per lane
 * (tools/ubench_copy.hip: 16 B/lane copies run at 6-7 TB/s, 4 B/lane at 4-5).
 *   k_pyr_level0_wide   thread = 16 bordered pixels of one row whose sources are plain interior bytes: bordered column
 *                       x16 maps to source column x16 - 19 = 13 (mod 16), so two aligned 16-byte loads and a 13-byte
 *                       shift (v_alignbyte) per row, no divergence; needs a 16-byte aligned source
 *   k_pyr_level0_edge   thread = 4 bordered pixels, byte by byte with reflection: the left and right column bands the
 *                       wide kernel leaves out (or everything, for unaligned sources) */
#define PYR0_ROWS 2
__global__ __launch_bounds__(256) void k_pyr_level0_wide(const DevLevel L, int pyrSlotBytes, const uint8_t* __restrict__ gray,
                                                         size_t frameStride, size_t rowStride, int x16First, int x16Last,
                                                         uint8_t* __restrict__ pyr)
{
    const int slot = blockIdx.z;
    const int y0 = (blockIdx.y * 4 + threadIdx.y) * PYR0_ROWS;     /* bordered row */
    const int x16 = x16First + (blockIdx.x * 64 + threadIdx.x) * 16;
    const int bh = L.h + 2 * DRFE_EDGE;
    if (y0 >= bh || x16 > x16Last) return;
    const int sx = x16 - DRFE_EDGE;
    const uint8_t* frame = gray + (size_t)slot * frameStride;
    uint4 out[PYR0_ROWS];
#pragma unroll
    for (int r = 0; r < PYR0_ROWS; r++) {
        const int y = min(y0 + r, bh - 1);
        const uint8_t* src = frame + (size_t)reflect101(y - DRFE_EDGE, L.h) * rowStride;
        const uint4 A = *reinterpret_cast<const uint4*>(src + (sx - 13));
        const uint4 B = *reinterpret_cast<const uint4*>(src + (sx + 3));
        out[r] = make_uint4(__builtin_amdgcn_alignbyte(B.x, A.w, 1), __builtin_amdgcn_alignbyte(B.y, B.x, 1),
                            __builtin_amdgcn_alignbyte(B.z, B.y, 1), __builtin_amdgcn_alignbyte(B.w, B.z, 1));
    }
    uint8_t* dst = pyr + (size_t)slot * pyrSlotBytes + L.pyrOff + x16;
#pragma unroll
    for (int r = 0; r < PYR0_ROWS; r++)
        if (y0 + r < bh) *reinterpret_cast<uint4*>(dst + (size_t)(y0 + r) * L.pyrPitch) = out[r];
}

/* columns [0, leftEnd) and [rightBegin, pitch) of every bordered row; thread = one dword, items numbered densely
 * (row-major over the nLeft + nRight dwords of a row) so that no lane idles */
__global__ __launch_bounds__(256) void k_pyr_level0_edge(const DevLevel L, int pyrSlotBytes, const uint8_t* __restrict__ gray,
                                                         size_t frameStride, size_t rowStride, int leftEnd, int rightBegin,
                                                         uint8_t* __restrict__ pyr)
{
    const int slot = blockIdx.y;
    const int nLeft = leftEnd / 4, nPer = nLeft + (L.pyrPitch - rightBegin) / 4;
    const int bw = L.w + 2 * DRFE_EDGE, bh = L.h + 2 * DRFE_EDGE;
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= nPer * bh) return;
    const int y = item / nPer, t = item - y * nPer;
    const int x4 = t < nLeft ? t * 4 : rightBegin + (t - nLeft) * 4;
    const uint8_t* src = gray + (size_t)slot * frameStride + (size_t)reflect101(y - DRFE_EDGE, L.h) * rowStride;
    uint32_t o = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = x4 + k;
        uint32_t v = 0;
        if (x < bw) v = src[reflect101(x - DRFE_EDGE, L.w)];
        o |= v << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(pyr + (size_t)slot * pyrSlotBytes + L.pyrOff + (size_t)y * L.pyrPitch + x4) = o;
}

/* One thread = 4 horizontally adjacent pixels of the BORDERED level; block = 64 x 4 threads, so a wave
 * is 256 contiguous pixels of one row.  The tap tables are indexed by bordered coordinates (reflection
 * folded in on the host), so a thread needs one 8-byte row tap, two 16-byte column-tap loads and, per
 * source row, three aligned dwords that cover the <= 9 source pixels its four outputs read; the bytes
 * are picked out of registers.  Level descriptors travel as kernel arguments (no dependent load). */
__device__ __forceinline__ uint32_t pick_byte(uint32_t d0, uint32_t d1, uint32_t d2, int o)
{
    const uint32_t w = o < 4 ? d0 : (o < 8 ? d1 : d2);
    return (w >> (8 * (o & 3))) & 0xFFu;
}

#define PYR_ROWS 4                                /* output rows per thread */
__global__ __launch_bounds__(256) void k_pyr_resize(const DevLevel L, const DevLevel P, int pyrSlotBytes,
                                                    const ResizeTap* __restrict__ taps, uint8_t* __restrict__ pyr)
{
    const int slot = blockIdx.z;
    const int y0 = (blockIdx.y * 4 + threadIdx.y) * PYR_ROWS;
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int bh = L.h + 2 * DRFE_EDGE;
    if (y0 >= bh || x4 >= L.pyrPitch) return;
    uint8_t* base = pyr + (size_t)slot * pyrSlotBytes;
    /* the row-tap table is padded to a multiple of PYR_ROWS entries, 16-byte aligned */
    const uint4 tya = *reinterpret_cast<const uint4*>(taps + L.ytabOff + y0);
    const uint4 tyb = *reinterpret_cast<const uint4*>(taps + L.ytabOff + y0 + 2);
    const uint4 ta = *reinterpret_cast<const uint4*>(taps + L.xtabOff + x4);       /* taps of x4, x4+1 */
    const uint4 tb = *reinterpret_cast<const uint4*>(taps + L.xtabOff + x4 + 2);   /* x4+2, x4+3 */
    const uint32_t syp[PYR_ROWS] = {tya.x, tya.z, tyb.x, tyb.z}, wyp[PYR_ROWS] = {tya.y, tya.w, tyb.y, tyb.w};
    const uint32_t sp[4] = {ta.x, ta.z, tb.x, tb.z};      /* s0 | s1 << 16 */
    const uint32_t wp[4] = {ta.y, ta.w, tb.y, tb.w};      /* w0 | w1 << 16 (int16) */
    int lo = 0xFFFF;
#pragma unroll
    for (int k = 0; k < 4; k++) lo = min(lo, (int)(sp[k] & 0xFFFF));
    const int ws = (lo + DRFE_EDGE) & ~3;                 /* bordered source column of the window, dword aligned */
    const uint8_t* srcw = base + P.pyrOff + (size_t)DRFE_EDGE * P.pyrPitch + ws;
    uint32_t a[PYR_ROWS][3], c[PYR_ROWS][3];
#pragma unroll
    for (int r = 0; r < PYR_ROWS; r++) {                  /* 6 * PYR_ROWS independent dword loads in flight */
        const uint32_t* R0 = reinterpret_cast<const uint32_t*>(srcw + (size_t)(syp[r] & 0xFFFF) * P.pyrPitch);
        const uint32_t* R1 = reinterpret_cast<const uint32_t*>(srcw + (size_t)(syp[r] >> 16) * P.pyrPitch);
        a[r][0] = R0[0]; a[r][1] = R0[1]; a[r][2] = R0[2];
        c[r][0] = R1[0]; c[r][1] = R1[1]; c[r][2] = R1[2];
    }
    int o0[4], o1[4], w0[4], w1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        o0[k] = (int)(sp[k] & 0xFFFF) + DRFE_EDGE - ws; o1[k] = (int)(sp[k] >> 16) + DRFE_EDGE - ws;
        w0[k] = (int)(short)(wp[k] & 0xFFFF); w1[k] = (int)(short)(wp[k] >> 16);
    }
#pragma unroll
    for (int r = 0; r < PYR_ROWS; r++) {
        if (y0 + r >= bh) break;
        const int b0 = (int)(short)(wyp[r] & 0xFFFF), b1 = (int)(short)(wyp[r] >> 16);
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int h0 = (int)pick_byte(a[r][0], a[r][1], a[r][2], o0[k]) * w0[k] + (int)pick_byte(a[r][0], a[r][1], a[r][2], o1[k]) * w1[k];
            const int h1 = (int)pick_byte(c[r][0], c[r][1], c[r][2], o0[k]) * w0[k] + (int)pick_byte(c[r][0], c[r][1], c[r][2], o1[k]) * w1[k];
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = min(255, max(0, v));
            out |= (uint32_t)v << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(base + L.pyrOff + (size_t)(y0 + r) * L.pyrPitch + x4) = out;
    }
}

/* The same resize with the source staged in LDS: a 64 x 4 block (256 output columns x 16 rows) first copies the
 * source window its taps touch — host-computed per block, <= 24 rows x 88 dwords — into LDS with coalesced dword loads
 * (6-7 per thread instead of 24), and every tap becomes one ds_read_u8 instead of a byte picked out of registers.
 * Arithmetic and results are identical to k_pyr_resize; measured speed is the same (~2.3 TB/s: neither version is
 * VALU- or issue-bound), so this is the default only because it is the simpler inner loop.  (Tried: 16 pixels per
 * thread with 16-byte loads and stores — slower, the strided LDS byte reads conflict.) */
#define RES_PITCH (DRFE_RESIZE_LDS_WD * 4 + 4)      /* bytes per LDS row: odd dword count, no bank aliasing between rows */
__global__ __launch_bounds__(256) void k_pyr_resize_lds(const DevLevel L, const DevLevel P, int pyrSlotBytes,
                                                        const ResizeTap* __restrict__ taps, uint8_t* __restrict__ pyr,
                                                        uint32_t magicXY, uint32_t magicX)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[DRFE_RESIZE_LDS_ROWS * RES_PITCH];
    int bx, by, bz;
    drfe_xcd_swizzle_3d(magicXY, magicX, bx, by, bz);   /* a frame's tiles on one XCD: neighbouring source windows overlap */
    bx = __builtin_amdgcn_readfirstlane(bx); by = __builtin_amdgcn_readfirstlane(by);    /* block-uniform: scalar bases */
    const int slot = __builtin_amdgcn_readfirstlane(bz);
    const int tid = threadIdx.y * 64 + threadIdx.x;
    uint8_t* base = pyr + (size_t)slot * pyrSlotBytes;
    const ResizeTap wx = taps[L.xwinOff + bx], wy = taps[L.ywinOff + by];
    /* this thread's taps first: their latency overlaps the tile fill instead of following the barrier */
    /* a wavefront is one row of the 64 x 4 block: its four output rows, their source rows and vertical weights are
     * wave-uniform, and readfirstlane tells the compiler so (scalar row bases, scalar branches below) */
    const int y0 = (by * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y)) * PYR_ROWS;
    const int x4 = (bx * 64 + threadIdx.x) * 4;
    const int bh = L.h + 2 * DRFE_EDGE;
    const bool active = y0 < bh && x4 < L.pyrPitch;
    const int yT = y0 < bh ? y0 : 0, xT = active ? x4 : 0;
    const uint4 tya = *reinterpret_cast<const uint4*>(taps + L.ytabOff + yT);
    const uint4 tyb = *reinterpret_cast<const uint4*>(taps + L.ytabOff + yT + 2);
    const uint4 ta = *reinterpret_cast<const uint4*>(taps + L.xtabOff + xT);
    const uint4 tb = *reinterpret_cast<const uint4*>(taps + L.xtabOff + xT + 2);
    /* The 19 border rows above and below the level are mirror images of interior rows (copyMakeBorder REFLECT_101 of the resized
     * interior, :1122-1123): the thread that computes interior row p in [1, 19] or [h - 20, h - 2] stores it a second time at
     * its mirror row, and nobody computes a border row - 38 of a level's h + 38 rows, a fifth of the small levels.  A block
     * whose sixteen rows are all border rows has nothing to do (block-uniform: before the tile fill and the barrier). */
    {
        const int r0 = by * (4 * PYR_ROWS);
        if (r0 + 4 * PYR_ROWS <= DRFE_EDGE || r0 >= DRFE_EDGE + L.h) return;
    }
    const int ws = ((int)wx.s0 + DRFE_EDGE) & ~3;                       /* bordered source column of tile byte 0 */
    const int wd = (((int)wx.s1 + DRFE_EDGE - ws) >> 2) + 1;           /* dwords per tile row */
    const int nr = (int)wy.s1 - (int)wy.s0 + 1;
    const uint8_t* srcw = base + P.pyrOff + (size_t)((int)wy.s0 + DRFE_EDGE) * P.pyrPitch + ws;
    {   /* (row, dword) of element tid + 256 k, carried incrementally: one division per thread, none in the loop */
        const int q256 = 256 / wd, r256 = 256 - q256 * wd;
        int r = tid / wd, cdw = tid - r * wd;
        while (r < nr) {
            *reinterpret_cast<uint32_t*>(&tile[r * RES_PITCH + cdw * 4]) =
                *reinterpret_cast<const uint32_t*>(srcw + (size_t)r * P.pyrPitch + cdw * 4);
            r += q256; cdw += r256;
            if (cdw >= wd) { cdw -= wd; r++; }
        }
    }
    __syncthreads();
    if (y0 >= bh) return;                        /* wave-uniform; columns past the pitch compute on column 0 and store nothing */
    if (y0 + PYR_ROWS <= DRFE_EDGE || y0 >= DRFE_EDGE + L.h) return;    /* four border rows: nothing to compute */
    uint8_t* const dstLevel = base + L.pyrOff;   /* block-uniform: the stores address with one 32-bit offset (a level is < 2^24 bytes) */
    const uint32_t syp[PYR_ROWS] = {tya.x, tya.z, tyb.x, tyb.z}, wyp[PYR_ROWS] = {tya.y, tya.w, tyb.y, tyb.w};
    const uint32_t sp[4] = {ta.x, ta.z, tb.x, tb.z}, wp[4] = {ta.y, ta.w, tb.y, tb.w};
    /* The second tap of a column is the pixel right of the first (s1 == s0 + 1), except at the level's last column where the
     * table repeats s0 with weight 0: reading s0 + 1 there multiplies whatever the tile holds by zero, so ONE address per
     * column serves both taps (the second is the +1 immediate of the LDS load; a tile row has a spare dword behind it):
     * four address adds per source row instead of eight, 22 VGPRs instead of 30 */
    int o0[4], w0[4], w1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        o0[k] = (int)(sp[k] & 0xFFFF) + DRFE_EDGE - ws;
        w0[k] = (int)(short)(wp[k] & 0xFFFF); w1[k] = (int)(short)(wp[k] >> 16);
    }
    /* horizontal pass of one source row for this thread's four columns, already shifted (the vertical pass only ever
     * uses h >> 4) */
    auto hrow = [&](int srow, uint32_t (&H)[4]) {
        const uint8_t* R = &tile[srow * RES_PITCH];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            /* two byte loads on purpose: merged into one 16-bit load (what the compiler does with plain accesses) they are
             * misaligned for half of the columns, and the LDS serves those at a third of the speed (measured: pyramid stage
             * 0.50 -> 1.43 ms).  The volatile LDS-address-space pointer keeps them apart. */
            typedef const volatile __attribute__((address_space(3))) uint8_t* lds_u8p;
            lds_u8p q = (lds_u8p)(R + o0[k]);
            H[k] = (uint32_t)(((int)q[0] * w0[k] + (int)q[1] * w1[k]) >> 4);
        }
    };
    /* Consecutive output rows mostly share a source row (scale 1.2: rows (s, s+1), (s+1, s+2), ...): the lower row's
     * horizontal pass is kept for the next output row, 5 instead of 8 row passes per thread.  (b * h) >> 16 is the high
     * half of h * (b << 16): one v_mul_hi_u32 instead of a multiply and a shift (weights and sums are non-negative). */
    int prevB = -1;
    uint32_t Hp[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < PYR_ROWS; r++) {
        if (y0 + r >= bh) break;
        const int pr = y0 + r - DRFE_EDGE;                               /* interior row; wave-uniform */
        if (pr < 0 || pr >= L.h) continue;                               /* border row: written with its mirror image below */
        const uint32_t sy = (uint32_t)__builtin_amdgcn_readfirstlane((int)syp[r]), wy2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)wyp[r]);
        const int a = (int)(sy & 0xFFFF) - (int)wy.s0, b = (int)(sy >> 16) - (int)wy.s0;
        uint32_t Ha[4], Hb[4];
        if (a == prevB) {
#pragma unroll
            for (int k = 0; k < 4; k++) Ha[k] = Hp[k];
        } else hrow(a, Ha);
        if (b == a) {
#pragma unroll
            for (int k = 0; k < 4; k++) Hb[k] = Ha[k];
        } else hrow(b, Hb);
        const uint32_t b0s = wy2 << 16, b1s = wy2 & 0xFFFF0000u;
        /* no saturation needed: H <= (255 * 2049) >> 4 = 32655 and b0 + b1 <= 2049 (two independently rounded 11-bit
         * weights), so the two floored products sum to at most 1020 and (1020 + 2) >> 2 = 255.  The four 10-bit sums are
         * shifted two at a time (v_pk_lshrrev_b16 on 16-bit halves) and their low bytes gathered by one v_perm_b32 */
        uint32_t sm[4];
#pragma unroll
        for (int k = 0; k < 4; k++) sm[k] = __umulhi(Ha[k], b0s) + __umulhi(Hb[k], b1s) + 2u;
        typedef unsigned short u16x2r __attribute__((ext_vector_type(2)));
        const u16x2r two = {2, 2};
        const uint32_t q01 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2r, sm[0] | (sm[1] << 16)) >> two);
        const uint32_t q23 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2r, sm[2] | (sm[3] << 16)) >> two);
        const uint32_t out = __builtin_amdgcn_perm(q23, q01, 0x06040200u);
        if (active) {
            *reinterpret_cast<uint32_t*>(dstLevel + (uint32_t)(__umul24((uint32_t)(y0 + r), (uint32_t)L.pyrPitch) + (uint32_t)x4)) = out;
            if (pr >= 1 && pr <= DRFE_EDGE)                              /* mirrors into the top border: bordered row 19 - p */
                *reinterpret_cast<uint32_t*>(dstLevel + (uint32_t)(__umul24((uint32_t)(DRFE_EDGE - pr), (uint32_t)L.pyrPitch) + (uint32_t)x4)) = out;
            if (pr >= L.h - 1 - DRFE_EDGE && pr <= L.h - 2)              /* ... into the bottom border: 19 + 2 (h - 1) - p */
                *reinterpret_cast<uint32_t*>(dstLevel + (uint32_t)(__umul24((uint32_t)(DRFE_EDGE + 2 * (L.h - 1) - pr), (uint32_t)L.pyrPitch) + (uint32_t)x4)) = out;
        }
        prevB = b;
#pragma unroll
        for (int k = 0; k < 4; k++) Hp[k] = Hb[k];
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* FAST-9/16 per cell                                                                               */

/* Threshold-free corner strength: cornerScore<16>(p, 0) = max(0, S_dark, S_bright) - 1 where
 * S = max over the 16 nine-pixel arcs of the min |difference|.  "Corner at t" is
 * exactly "strength >= t", so one strength map serves both FAST thresholds. Returned clamped to >= 0
 * (values below minThFAST never take part in a decision).
 *
 * Two vertically adjacent pixels A=(x,y), B=(x,y+1) ride in the two 16-bit halves of every register
 * (v_pk_sub/min/max_i16).  The eight 8-pixel windows that start at odd ring positions are shared by
 * two 9-arcs each (arc j-1..j+7 and arc j..j+8), and max(min(w,a), min(w,b)) == min(w, max(a,b)):
 *   dark:   S = max_j min(w8lo[j], max(d[j-1], d[j+8]))      bright: -min_j max(w8hi[j], min(d[j-1], d[j+8]))
 * = 16 sub + 2*(8+8+8) + 16 + 16 + 14 packed ops for two pixels. */
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk2(uint32_t lo, uint32_t hi) { return __builtin_bit_cast(s16x2, lo | (hi << 16)); }
__device__ __forceinline__ s16x2 pmin(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 umax2(u16x2 a, u16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u16x2 umin2(u16x2 a, u16x2 b) { return __builtin_elementwise_min(a, b); }

/* The arithmetic runs on the packed HALF-FLOAT pipe: an integer n in [-1023, 1023] written into a 16-bit
 * lane as sign|magnitude IS the f16 subnormal n * 2^-24, subnormal add/sub/min/max are exact (f16
 * denormals are never flushed in this code object: .amdhsa_float_denorm_mode_16_64 3), and gfx950 has
 * three-input packed min/max (v_pk_minimum3_f16 / v_pk_maximum3_f16) which the integer pipe lacks.
 * The trees run on the ring pixels themselves, not on the 16 differences: min over an arc of (v - p) is v - max over
 * the arc of p, so the centre is subtracted twice at the end instead of 16 times at the start:
 * 2*(8+8+8+8+4) + 5 = 77 packed ops for two pixels instead of 120 (integer pipe) or 92 (differences first). */
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h16x2 hpk(uint32_t lo, uint32_t hi) { return __builtin_bit_cast(h16x2, lo | (hi << 16)); }
__device__ __forceinline__ h16x2 hmin(h16x2 a, h16x2 b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ h16x2 hmax(h16x2 a, h16x2 b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ h16x2 hmin3(h16x2 a, h16x2 b, h16x2 c) { return hmin(hmin(a, b), c); }
__device__ __forceinline__ h16x2 hmax3(h16x2 a, h16x2 b, h16x2 c) { return hmax(hmax(a, b), c); }

/* returns strength(A) | strength(B) << 16: two ds_read_u8 and one v_lshl_or per ring position (DESIGN.md section 4 lists
 * the alternatives that were measured: a row-interleaved 16-bit tile, d16 / d16_hi LDS loads) */
__device__ __forceinline__ h16x2 hpair(const uint8_t* c, int i, int P) { return hpk(c[i], c[i + P]); }
__device__ __forceinline__ uint32_t fast_strength2(const uint8_t* c, const int P)
{
    /* c points at the top-left corner of A's 7x7 patch (B = the pixel below A), so every ring offset is a
     * non-negative immediate of the LDS load */
    const int C = 3 * P + 3;
    const h16x2 v = hpair(c, C, P);
    h16x2 d[16];                            /* the ring pixels p[k] (as exact f16 subnormals) */
#define RING(k, o) d[k] = hpair(c, C + (o), P)
    RING(0, 3 * P);       RING(1, 3 * P + 1);   RING(2, 2 * P + 2);    RING(3, P + 3);
    RING(4, 3);           RING(5, -P + 3);      RING(6, -2 * P + 2);   RING(7, -3 * P + 1);
    RING(8, -3 * P);      RING(9, -3 * P - 1);  RING(10, -2 * P - 2);  RING(11, -P - 3);
    RING(12, -3);         RING(13, P - 3);      RING(14, 2 * P - 2);   RING(15, 3 * P - 1);
#undef RING
    h16x2 lo2[8], hi2[8], lo4[8], hi4[8], t[8], u[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {           /* j = 2m+1: pairs (1,2) (3,4) ... (15,0) */
        lo2[m] = hmin(d[2 * m + 1], d[(2 * m + 2) & 15]);
        hi2[m] = hmax(d[2 * m + 1], d[(2 * m + 2) & 15]);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) { lo4[m] = hmin(lo2[m], lo2[(m + 1) & 7]); hi4[m] = hmax(hi2[m], hi2[(m + 1) & 7]); }
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const h16x2 e0 = d[2 * m], e1 = d[(2 * m + 9) & 15];   /* d[j-1], d[j+8] */
        t[m] = hmin3(lo4[m], lo4[(m + 2) & 7], hmax(e0, e1));  /* min(p[j..j+7], max(p[j-1], p[j+8])) */
        u[m] = hmax3(hi4[m], hi4[(m + 2) & 7], hmin(e0, e1));
    }
    /* a = the brightest arc floor, b = the darkest arc ceiling: S_bright = a - v, S_dark = v - b */
    const h16x2 a = hmax(hmax3(hmax3(t[0], t[1], t[2]), t[3], t[4]), hmax3(t[5], t[6], t[7]));
    const h16x2 b = hmin(hmin3(hmin3(u[0], u[1], u[2]), u[3], u[4]), hmin3(u[5], u[6], u[7]));
    const h16x2 one = __builtin_bit_cast(h16x2, 0x00010001u), zero = __builtin_bit_cast(h16x2, 0u);
    const h16x2 r = hmax(hmax(a - v, v - b) - one, zero);
    return __builtin_bit_cast(uint32_t, r);
}

#define FAST_TILE_PITCH 68                       /* pixels per window-tile row: 16 data dwords + 1 pad dword so that lanes two
                                                    rows apart hit different banks */
#define FAST_MAX_EVAL (DRFE_FAST_MAX_WIN - 6)    /* 54 */
#define FAST_SC_PITCH 64

/* LDS bytes of k_fast_cells for windows up to maxWh rows: tile rows (one spare row in the byte layout: the unstored B pixel
 * of an odd-height area reads it), then (maxWh - 6 + 2) score rows */
#define FAST_TILE_ROW_BYTES FAST_TILE_PITCH
static inline int fast_sc_off(int maxWh) { return ((maxWh + 1) * FAST_TILE_ROW_BYTES + 15) & ~15; }
static inline int fast_lds_bytes(int maxWh) { return fast_sc_off(maxWh) + (maxWh - 6 + 2) * FAST_SC_PITCH; }

/* One wavefront per FAST cell (one cv::FAST call of the reference).  Window rows arrive as aligned
 * dwords; each lane scores ~16 pixels; the strict-3x3-maximum flags stay in two 64-bit lane masks
 * (>= iniThFAST / >= minThFAST); `__any` decides the per-cell fallback; a wave prefix sum gives every
 * lane its slots behind ONE global atomic per cell. */
__global__ __launch_bounds__(64) void k_fast_cells(const FastCell* __restrict__ cells, int nlevels, int pyrSlotBytes,
                                                   int candSlotElems, int iniTh, int minTh,
                                                   const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand0,
                                                   uint32_t* __restrict__ cand1, int* __restrict__ candCount,
                                                   int* __restrict__ status, int scOff, uint32_t gxMagic)
{
    int bx, by;
    drfe_xcd_swizzle_2d(gxMagic, bx, by);        /* all cells of a frame on one XCD: window rows share lines */
    /* dynamic LDS sized for the tallest cell window of this geometry (36 rows for 30-px cells, not the 60-row worst
     * case): window tile, then the score tile at byte scOff (fast_lds_bytes below) */
    extern __shared__ __attribute__((aligned(16))) unsigned char fastLds[];
    uint32_t* tile = reinterpret_cast<uint32_t*>(fastLds);
    uint8_t* sc = fastLds + scOff;

    const FastCell fc = cells[bx];                /* everything below depends on this one record only */
    const int slot = by;
    const int lane = threadIdx.x;
    const int ww = fc.ww, wh = fc.wh;
    const int ew = ww - 6, eh = wh - 6;           /* evaluated area */
    const int off = fc.off;
    const uint8_t* src = pyr + (size_t)slot * pyrSlotBytes + fc.srcOff;
    {
        /* window rows as 16-byte chunks (dword-aligned addresses): lane = (row, chunk), 16 rows per trip, so a 36-row
         * window fills in 3 trips instead of 9 with single dwords; a chunk is loaded only if the window needs its first
         * byte, which keeps the over-read inside the level's bordered row */
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        const int nch = (ww + off + 15) >> 4;     /* chunks per window row (<= 4) */
        const int c = lane & 3;
        if (c < nch)
            for (int r = lane >> 2; r < wh; r += 16) {
                const u32x4_a4 g = *reinterpret_cast<const u32x4_a4*>(src + (size_t)r * fc.pitch + c * 16);
                uint32_t* t = &tile[r * (FAST_TILE_PITCH / 4) + c * 4];
                t[0] = g.x; t[1] = g.y; t[2] = g.z; t[3] = g.w;
            }
    }
    /* score tile: pixel (x, y) of the evaluated area lives at byte (y+1)*64 + (x+4); everything else
     * (apron rows 0 / eh+1, bytes left of 4 and right of ew+3) stays 0 = "neighbour outside the cell" */
    for (int i = lane; i < (eh + 2) * (FAST_SC_PITCH / 16); i += 64) reinterpret_cast<uint4*>(sc)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tile) + off;   /* tb[y*PITCH + x] = window pixel (x, y) */
    {   /* score pass: item = two vertically adjacent pixels (rows 2r, 2r+1) of one column; (rp, x) of item
           lane + 64k is carried incrementally (no integer division in the loop) */
        const int nrp = (eh + 1) >> 1;
        const int q64 = 64 / ew, r64 = 64 - q64 * ew;
        int rp = lane / ew, x = lane - rp * ew;
        while (rp < nrp) {
            const int yA = 2 * rp;                                    /* odd eh: the last B is scored but not stored */
            const uint32_t s2 = fast_strength2(&tb[__mul24(yA, FAST_TILE_PITCH) + x], FAST_TILE_PITCH);
            sc[(yA + 1) * FAST_SC_PITCH + (x + 4)] = (uint8_t)(s2 & 0xFF);
            if (2 * rp + 1 < eh) sc[(yA + 2) * FAST_SC_PITCH + (x + 4)] = (uint8_t)(s2 >> 16);
            rp += q64; x += r64;
            if (x >= ew) { x -= ew; rp++; }
        }
    }
    __syncthreads();
    /* strict 3x3 maximum, four horizontally adjacent pixels per item, branch-free: the nine score dwords
     * around the quad are split into even / odd byte lanes (u16x2) so that v_pk_max_u16 reduces the eight
     * neighbours of two pixels at a time.  Bit 4k+j of a lane's mask = pixel j of its k-th quad. */
    const int nq = (ew + 3) >> 2;                 /* quads per row (<= 14) */
                                                  /* nq * eh <= 756 items -> <= 12 quads per lane */
    const uint32_t* scw = reinterpret_cast<const uint32_t*>(sc);
    unsigned long long m20 = 0, m7 = 0;
    {
        const int q64 = 64 / nq, r64 = 64 - q64 * nq;
        int y = lane / nq, q = lane - y * nq;
        const u16x2 thMin = {(unsigned short)(minTh - 1), (unsigned short)(minTh - 1)};
        const u16x2 thIni = {(unsigned short)(iniTh - 1), (unsigned short)(iniTh - 1)};
        const u16x2 one = {1, 1};
        for (int k = 0; y < eh; k += 4) {
            const uint32_t* w = scw + y * (FAST_SC_PITCH / 4) + q;          /* dword left of the quad, row above */
            u16x2 E[3], O[3], LE[3], RO[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const uint32_t a = w[r * 16], b = w[r * 16 + 1], c = w[r * 16 + 2];
                E[r] = __builtin_bit_cast(u16x2, b & 0x00FF00FFu);                          /* px 0, 2 */
                O[r] = __builtin_bit_cast(u16x2, (b >> 8) & 0x00FF00FFu);                   /* px 1, 3 */
                LE[r] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0C050C03u)); /* px -1, 1 */
                RO[r] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(c, b, 0x0C040C02u)); /* px 2, 4 */
            }
            const u16x2 H = umax2(umax2(E[0], O[0]), umax2(E[2], O[2]));
            const u16x2 mE = umax2(umax2(umax2(H, LE[0]), umax2(LE[2], LE[1])), O[1]);
            const u16x2 mO = umax2(umax2(umax2(H, RO[0]), umax2(RO[2], RO[1])), E[1]);
            const u16x2 fE = __builtin_elementwise_sub_sat(E[1], mE), fO = __builtin_elementwise_sub_sat(O[1], mO);
            const uint32_t e7 = __builtin_bit_cast(uint32_t, umin2(umin2(fE, __builtin_elementwise_sub_sat(E[1], thMin)), one));
            const uint32_t o7 = __builtin_bit_cast(uint32_t, umin2(umin2(fO, __builtin_elementwise_sub_sat(O[1], thMin)), one));
            const uint32_t e20 = __builtin_bit_cast(uint32_t, umin2(umin2(fE, __builtin_elementwise_sub_sat(E[1], thIni)), one));
            const uint32_t o20 = __builtin_bit_cast(uint32_t, umin2(umin2(fO, __builtin_elementwise_sub_sat(O[1], thIni)), one));
            const uint32_t t7 = e7 | (o7 << 1), t20 = e20 | (o20 << 1);    /* bits 0,1 = px 0,1; bits 16,17 = px 2,3 */
            m7 |= (unsigned long long)((t7 | (t7 >> 14)) & 0xFu) << k;
            m20 |= (unsigned long long)((t20 | (t20 >> 14)) & 0xFu) << k;
            y += q64; q += r64;
            if (q >= nq) { q -= nq; y++; }
        }
    }
    unsigned long long emit = __any(m20 != 0) ? m20 : m7;         /* fallback decided per cell after NMS@ini */
    const int cnt = __popcll(emit);
    const int incl = drfe_wave_incl_scan(cnt, lane);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (total == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&candCount[DRFE_CC_IDX(slot, fc.level)], total);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base + total > (int)fc.candCap) { if (lane == 0) atomicOr(status, 1); return; }
    size_t pos = (size_t)slot * candSlotElems + fc.candOff + base + (incl - cnt);
    const uint32_t magic = 0xFFFFFFFFu / (uint32_t)nq + 1u;     /* i / nq == umulhi(i, magic) for i < 2^16, nq > 1 */
    while (emit) {
        const int k = __builtin_ctzll(emit);
        emit &= emit - 1;
        const int i = lane + 64 * (k >> 2);
        const int y = nq == 1 ? i : (int)__umulhi((uint32_t)i, magic), x = 4 * (i - y * nq) + (k & 3);
        const uint32_t s = sc[(y + 1) * FAST_SC_PITCH + (x + 4)];
        /* keypoint coordinates as the reference leaves them in vToDistributeKeys (:822-823):
         * cv::FAST coordinate inside the window + (j*wCell, i*hCell) */
        const uint32_t kx = (uint32_t)(x + 3 + fc.offX), ky = (uint32_t)(y + 3 + fc.offY);
        cand0[pos] = kx | (ky << 12) | (s << 24);
        cand1[pos] = (fc.cellIdx << 12) | ((uint32_t)y << 6) | (uint32_t)x;   /* emission order */
        pos++;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* FAST per cell, column-pair layout (the kernel the batch path runs when the geometry fits)          */

/* k_fast_cells_cols: same result as k_fast_cells, a third fewer VALU instructions.
 *  - the window tile holds one pixel per 16-bit lane, so an aligned ds_read_b32 is the exact f16 pair (x, x+1) of a ring
 *    position with an even pixel offset (10 of 17) and two dwords + one v_alignbit of the others: 7 packing instructions
 *    per pixel pair instead of 17, dwords instead of bytes from LDS;
 *  - lane = (column pair cp, row block rb): a lane scores rpl consecutive rows of its two columns and keeps the scores
 *    in registers; the strict 3x3 maximum takes vertical neighbours from the lane's own registers (block seams: one
 *    ds_bpermute each way), horizontal ones through wave_shr:1 / wave_shl:1 DPP moves of the column maxima - there is
 *    no score tile in LDS and no byte unpacking;
 *  - two horizontally adjacent pixels cannot both be strict maxima, so a lane emits at most one candidate per row.
 * Emission order inside a cell differs from k_fast_cells; the order key in cand1 is what the quadtree ties on. */
#define FASTC_P16 52                              /* pixels per tile row: 26 dwords, so row blocks 8 rows apart start 16 banks apart */
#define FASTC_PB (FASTC_P16 * 2)
/* tile, then the survivor list of the screened paths: FASTC_LIST_CAP 16-bit items (the tile offset of a surviving pair row's patch, later the pair's two scores).
 * 320 of the wavefront's 512 pixel-pair rows: on textured frames 24-50 % of a cell's pair rows pass the screen at iniThFAST
 * (profiles/r05_fast_screen_survivors.txt); a cell with more survivors takes the plain path.  640 bytes = one LDS granule more per
 * workgroup than the 128-entry list of rounds 3-4 (25 instead of 32 workgroups per CU); measured: the plain path does not notice */
#define FASTC_LIST_CAP 320
static inline int fastc_list_off(int rows) { return (rows * FASTC_PB + 16 + 15) & ~15; }
/* entries of the list: at least FASTC_LIST_CAP, and whatever else the LDS allocation granule (1280 bytes) leaves behind them - up to `want`
 * (every pair row of the wavefront: 512 for the 8-row instantiation, 768 for the 12-row one) - so that a cell never falls off the list
 * when a longer one costs nothing */
static inline int fastc_list_cap(int rows, int want)
{
    const int off = fastc_list_off(rows), total = (off + FASTC_LIST_CAP * 2 + 1279) / 1280 * 1280;
    return std::max(FASTC_LIST_CAP, std::min(want, (total - off) / 2));
}
static inline int fastc_lds_bytes(int rows, int want) { return fastc_list_off(rows) + fastc_list_cap(rows, want) * 2; }
#define FASTC_DPP_SHR 0x138                        /* wave_shr:1 */
#define FASTC_DPP_SHL 0x130                        /* wave_shl:1 */

/* ring position (DX, DY) of the pixel pair whose 7x7 patch starts at c (a 4-byte-aligned tile address: the fill drops the odd
 * part of the window's byte alignment).  Even pixel offsets are one aligned ds_read_b32; odd ones take the two aligned dwords
 * around them and one v_alignbit - misaligned LDS dwords are legal on gfx950 but run ~20x slower (measured: 3.4 ms against
 * 0.8 ms for this kernel). */
template <int DX, int DY>
__device__ __forceinline__ h16x2 fastc_ld(const uint8_t* c)
{
    constexpr int k = (DY + 3) * FASTC_P16 + (DX + 3);
    if constexpr ((k & 1) != 0) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(c + (k - 1) * 2);
        return __builtin_bit_cast(h16x2, __builtin_amdgcn_alignbit(p[1], p[0], 16));
    } else {
        return __builtin_bit_cast(h16x2, *reinterpret_cast<const uint32_t*>(c + k * 2));
    }
}

/* c -> the 16-bit tile at the top-left corner of the 7x7 patch of pixel A; returns strength(A) | strength(A+1) << 16 */
__device__ __forceinline__ uint32_t fastc_strength2(const uint8_t* c)
{
#define PX(dx, dy) fastc_ld<dx, dy>(c)
    const h16x2 v = PX(0, 0);
    h16x2 d[16];
    d[0] = PX(0, 3);    d[1] = PX(1, 3);    d[2] = PX(2, 2);    d[3] = PX(3, 1);
    d[4] = PX(3, 0);    d[5] = PX(3, -1);   d[6] = PX(2, -2);   d[7] = PX(1, -3);
    d[8] = PX(0, -3);   d[9] = PX(-1, -3);  d[10] = PX(-2, -2); d[11] = PX(-3, -1);
    d[12] = PX(-3, 0);  d[13] = PX(-3, 1);  d[14] = PX(-2, 2);  d[15] = PX(-1, 3);
#undef PX
    h16x2 lo2[8], hi2[8], lo4[8], hi4[8], t[8], u[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        lo2[m] = hmin(d[2 * m + 1], d[(2 * m + 2) & 15]);
        hi2[m] = hmax(d[2 * m + 1], d[(2 * m + 2) & 15]);
    }
#pragma unroll
    for (int m = 0; m < 8; m++) { lo4[m] = hmin(lo2[m], lo2[(m + 1) & 7]); hi4[m] = hmax(hi2[m], hi2[(m + 1) & 7]); }
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const h16x2 e0 = d[2 * m], e1 = d[(2 * m + 9) & 15];
        t[m] = hmin3(lo4[m], lo4[(m + 2) & 7], hmax(e0, e1));
        u[m] = hmax3(hi4[m], hi4[(m + 2) & 7], hmin(e0, e1));
    }
    const h16x2 a = hmax(hmax3(hmax3(t[0], t[1], t[2]), t[3], t[4]), hmax3(t[5], t[6], t[7]));
    const h16x2 b = hmin(hmin3(hmin3(u[0], u[1], u[2]), u[3], u[4]), hmin3(u[5], u[6], u[7]));
    const h16x2 one = __builtin_bit_cast(h16x2, 0x00010001u);
    const h16x2 r = hmax3(a - v, v - b, one) - one;              /* max(S - 1, 0) */
    return __builtin_bit_cast(uint32_t, r);
}

/* The compass screen: a 9-arc of the 16-pixel ring always holds two ADJACENT compass pixels (ring positions 0, 4, 8, 12), so a
 * pixel whose strength reaches th + 1 has two adjacent compass pixels all brighter than v + th or all darker than v - th.
 * Returns a non-zero half where the half's pixel passes (strength may reach th + 1), zero where it cannot.  d0 / d4 / d8 / d12 =
 * ring pixels (0, 3), (3, 0), (0, -3), (-3, 0); th as the f16 bit pattern of the integer threshold. */
__device__ __forceinline__ uint32_t fastc_screen(h16x2 v, h16x2 d0, h16x2 d4, h16x2 d8, h16x2 d12, h16x2 th)
{
    const h16x2 hi = v + th, lo = v - th;
    /* brightest adjacent pair's darker pixel: the four adjacent pairs (0,4) (4,8) (8,12) (12,0) are exactly the pairs of one pixel of
     * {0, 8} with one of {4, 12}, and max over x in X, y in Y of min(x, y) = min(max X, max Y): three operations instead of six */
    const h16x2 M = hmin(hmax(d0, d8), hmax(d4, d12));
    const h16x2 m = hmax(hmin(d0, d8), hmin(d4, d12));
    const h16x2 zero = __builtin_bit_cast(h16x2, 0u);
    return __builtin_bit_cast(uint32_t, hmax3(M - hi, lo - m, zero));
}

__device__ __forceinline__ uint32_t u2max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, umax2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t u2min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, umin2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t u2subs(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

template <int RMAX>
__global__ __launch_bounds__(64) void k_fast_cells_cols(const FastCell* __restrict__ cells, int nlevels, int pyrSlotBytes,
                                                        int candSlotElems, int iniTh, int minTh,
                                                        const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand0,
                                                        uint32_t* __restrict__ cand1, int* __restrict__ candCount,
                                                        int* __restrict__ status, uint32_t gxMagic, int cellFirst, int screen, int listOff, int listCap)
{
    int bx, by;
    drfe_xcd_swizzle_2d(gxMagic, bx, by);
    extern __shared__ __attribute__((aligned(16))) unsigned char fastLds[];
    const FastCell fc = cells[cellFirst + bx];
    const int slot = by;
    const int lane = threadIdx.x;
    const int ww = fc.ww, wh = fc.wh;
    const int ew = ww - 6, eh = wh - 6;
    const int off = fc.off;
    const uint8_t* src = pyr + (size_t)__builtin_amdgcn_readfirstlane(slot) * pyrSlotBytes + fc.srcOff;    /* block-uniform: a scalar base */
    {   /* window rows as 16-byte chunks, widened to one pixel per 16-bit lane.  Tile index k of a row holds byte k + sh of
           the aligned source row, sh = off & 1: pixel (wx, wy) of the window sits at index (off & 2) + wx, so every pixel
           pair a lane reads starts at an even index.  A chunk is loaded only if the window needs its first byte (and the
           dword behind it only if the window reaches it), which keeps the over-read inside the level's bordered row */
        typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
        const int need = ww + off;                /* bytes of the aligned row the window covers */
        const int nch = (need + 15) >> 4;         /* chunks per window row (<= 3) */
        const int nitems = wh * nch;
        const int sh = off & 1;
        const uint32_t selE = sh ? 0x0C020C01u : 0x0C010C00u, selO = sh ? 0x0C040C03u : 0x0C030C02u;
        for (int i = lane; i < nitems; i += 64) {
            const int r = nch == 1 ? i : nch == 2 ? (i >> 1) : (int)(((uint32_t)i * 21846u) >> 16);
            const int c = i - r * nch;
            const uint8_t* p = src + (uint32_t)(__umul24((uint32_t)r, fc.pitch) + (uint32_t)c * 16u);      /* 32-bit offset: a window is a few KB */
            const u32x4_a4 g = *reinterpret_cast<const u32x4_a4*>(p);
            uint32_t g4 = 0;
            if (sh && c * 16 + 16 < need) g4 = *reinterpret_cast<const uint32_t*>(p + 16);
            uint2* t = reinterpret_cast<uint2*>(fastLds + r * FASTC_PB + c * 32);
            t[0] = make_uint2(__builtin_amdgcn_perm(0u, g.x, selE), __builtin_amdgcn_perm(g.y, g.x, selO));
            t[1] = make_uint2(__builtin_amdgcn_perm(0u, g.y, selE), __builtin_amdgcn_perm(g.z, g.y, selO));
            t[2] = make_uint2(__builtin_amdgcn_perm(0u, g.z, selE), __builtin_amdgcn_perm(g.w, g.z, selO));
            t[3] = make_uint2(__builtin_amdgcn_perm(0u, g.w, selE), __builtin_amdgcn_perm(g4, g.w, selO));
        }
    }
    __syncthreads();
    const int ncp = fc.ncp, R = fc.rpl, nrb = fc.nrb;
    const int rb = (int)(((uint32_t)lane * fc.ncpMagic) >> 16), cp = lane - rb * ncp;
    const int x = 2 * cp, y0 = rb * R;
    const bool laneOn = rb < nrb;
    /* pixels outside the evaluated area (odd width, rows past the last block's end, idle lanes) score 0 = "neighbour
       outside the cell"; what the tile holds there is never looked at */
    const uint32_t mcol = laneOn ? ((x < ew ? 0xFFFFu : 0u) | (x + 1 < ew ? 0xFFFF0000u : 0u)) : 0u;
    const uint8_t* base = fastLds + (laneOn ? y0 * FASTC_PB + ((off & 2) + x) * 2 : 0);
    /* rows past the area's last one exist only at the end of the last row block: rows i >= iInv of the lanes with
       rb == nrb - 1 (iInv is wave-uniform) */
    const int iInv = eh - (nrb - 1) * R;
    const uint32_t mlast = rb == nrb - 1 ? 0u : mcol;
    uint32_t s[RMAX];
    /* Which way this cell is scored.  The strength tree costs ~100 instructions per pixel-pair row whatever the pixels hold; on a
     * low-texture cell a twentieth of the pair rows can reach minThFAST at all, and on a textured one a third can reach iniThFAST.
     * The compass screen (fastc_screen, ~25 instructions) of every lane's FIRST row - one row of every row block: a sample spread
     * over the cell - decides per wavefront:
     *   first (screen >= 2; round 5): an eighth to two thirds of the sampled pair rows pass the screen AT iniTh -> screen
     *     every row at iniTh and run the tree on the compacted survivors only (if they fit the list).  A pair row that fails scores 0
     *     instead of its true strength (< iniTh): no comparison at iniTh can see that - a pixel kept there has s > max(N, iniTh - 1) -
     *     so if the cell HAS a corner at iniTh (the reference's first cv::FAST call returns something, ORBextractor.cc:809-812)
     *     its candidates are exact and the wavefront is done; if it has none the strengths below iniTh are needed and the cell
     *     starts again below;
     *   then: fewer than a quarter of the sampled pair rows pass the screen at minTh -> the same at minTh (exact for both
     *     thresholds: a value below minTh never reaches an output); otherwise the plain path, which costs a textured cell the
     *     samples and nothing else.
     * The stages are lambdas inlined at their two uses, straight-line code on both ways (a loop over the two attempts kept the
     * score arrays of one attempt alive across the other: 81 VGPRs instead of 56). */
    const uint32_t th7 = (uint32_t)(minTh - 1) * 0x00010001u, th20 = (uint32_t)(iniTh - 1) * 0x00010001u;
    const uint32_t selA = cp == 0 ? 0x05040C0Cu : 0x05040302u;          /* [own col x   | left lane's col x-1] */
    const uint32_t selB = cp == ncp - 1 ? 0x0C0C0302u : 0x05040302u;    /* [right lane's col x+2 | own col x+1] */
    uint32_t dm[RMAX], nb[RMAX];
    /* the first row of every lane through the screen at thS: how many of the lanes that hold a pixel pair pass */
    auto sample = [&](h16x2 thS, int& nS, int& nAll) -> uint32_t {
        const uint32_t p0 = fastc_screen(fastc_ld<0, 0>(base), fastc_ld<0, 3>(base), fastc_ld<3, 0>(base), fastc_ld<0, -3>(base),
                                         fastc_ld<-3, 0>(base), thS) & (0 < iInv ? mcol : mlast);
        nS = __popcll(__ballot(p0 != 0)); nAll = __popcll(__ballot(mcol != 0));
        return p0;
    };
    /* screen every row at thS, the tree on the compacted survivors, their scores back into s[]; false: they do not fit the list */
    /* p0: the first row's screen (the sample's) */
    auto screened_fill = [&](h16x2 thS, uint32_t p0) -> bool {
        unsigned long long pm[RMAX];                 /* wave-uniform: which lanes' pair row i passed the screen */
        uint32_t total = 0;
        uint16_t* lst = reinterpret_cast<uint16_t*>(fastLds + listOff);
        const uint32_t baseOff = (uint32_t)(base - fastLds);
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
            uint32_t pb = p0;
            if (i > 0) pb = 0;
            if (i > 0 && i < R)
                pb = fastc_screen(fastc_ld<0, 0>(base + i * FASTC_PB), fastc_ld<0, 3>(base + i * FASTC_PB), fastc_ld<3, 0>(base + i * FASTC_PB),
                                  fastc_ld<0, -3>(base + i * FASTC_PB), fastc_ld<-3, 0>(base + i * FASTC_PB), thS) & (i < iInv ? mcol : mlast);
            pm[i] = __ballot(pb != 0);
            /* the survivor's place in the list: the rows before it, then the lanes before it in its row */
            const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm[i], 0u));
            if (pb != 0 && pos < (uint32_t)listCap) lst[pos] = (uint16_t)(baseOff + (uint32_t)(i * FASTC_PB));    /* where the pair's patch starts in the tile */
            total += (uint32_t)__popcll(pm[i]);
        }
        if (total > (uint32_t)listCap) return false;
        __syncthreads();
        /* the tree on the survivors, 64 at a time; a survivor's list slot then takes its two 8-bit scores, which the owner
         * reads back below (no register of the owner is live across the passes: the register budget stays the plain path's) */
        const uint32_t npass = (total + 63u) >> 6;
        for (uint32_t k = 0; k < npass; k++) {
            const uint32_t p = 64u * k + (uint32_t)lane;
            if (p < total) {
                const uint32_t v2 = fastc_strength2(fastLds + lst[p]);
                lst[p] = (uint16_t)((v2 & 0xFFu) | ((v2 >> 8) & 0xFF00u));
            }
        }
        __syncthreads();
        uint32_t start = 0;
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
            s[i] = 0;
            if ((pm[i] >> lane) & 1ull) {
                const uint32_t v2 = lst[start + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm[i], 0u))];
                s[i] = ((v2 & 0xFFu) | ((v2 & 0xFF00u) << 8)) & (i < iInv ? mcol : mlast);
            }
            start += (uint32_t)__popcll(pm[i]);
        }
        return true;
    };
    /* strict 3x3 maximum of s[] into nb[], dm[] = how far each score is above max(N, iniTh - 1); true: the cell has a corner at iniTh.
     * Block seams: the row above this lane's first row is the last row of lane - ncp, the row below its last one the first row of
     * lane + ncp.  C3 = column maximum incl. the pixel; the left / right column maxima come from the neighbouring lanes (wave shifts;
     * the selectors blank the cell's outer columns); N = the eight neighbours' maximum.  A pixel is kept at threshold t iff
     * s > max(N, t - 1).  Rows i >= R hold zero scores: they run through the same instructions (no control flow around register
     * arrays) and come out as "no maximum". */
    auto nms_ini = [&]() -> bool {
        uint32_t last = 0;
#pragma unroll
        for (int i = 0; i < RMAX; i++)
            if (i == R - 1) last = s[i];
        uint32_t up = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - ncp) << 2, (int)last);
        uint32_t down = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + ncp) << 2, (int)s[0]);
        if (rb == 0) up = 0;
        if (rb + 1 >= nrb) down = 0;
        uint32_t acc20 = 0;
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
            const uint32_t sup = i == 0 ? up : s[i > 0 ? i - 1 : 0];
            const uint32_t sdn = (i == R - 1 || i == RMAX - 1) ? down : s[i + 1 < RMAX ? i + 1 : i];
            const uint32_t V = u2max(sup, sdn), C3 = u2max(V, s[i]);
            const uint32_t L = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C3, FASTC_DPP_SHR, 0xF, 0xF, true);
            const uint32_t Rr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)C3, FASTC_DPP_SHL, 0xF, 0xF, true);
            const uint32_t A = __builtin_amdgcn_perm(C3, L, selA), B = __builtin_amdgcn_perm(Rr, C3, selB);
            nb[i] = __builtin_bit_cast(uint32_t, hmax3(__builtin_bit_cast(h16x2, V), __builtin_bit_cast(h16x2, A),
                                                       __builtin_bit_cast(h16x2, B)));     /* small integers = exact f16 subnormals */
            dm[i] = u2subs(s[i], u2max(nb[i], th20));
            acc20 |= dm[i];
        }
        return __any(acc20 != 0) != 0;
    };
    auto emit = [&]() {
        /* where a candidate goes inside the cell's run: row by row, the lanes of a row in lane order - one ballot per row, the
         * row's start is a scalar popcount sum and a lane's place two v_mbcnt (the order inside a run is free: the quadtree ties on
         * the order key of cand1, not on the array position) */
        unsigned long long rowMask[RMAX];
        int total = 0;
#pragma unroll
        for (int i = 0; i < RMAX; i++) { rowMask[i] = __ballot(dm[i] != 0); total += __popcll(rowMask[i]); }
        if (total == 0) return;
        int cbase = 0;
        if (lane == 0) cbase = atomicAdd(&candCount[DRFE_CC_IDX(slot, fc.level)], total);
        cbase = __builtin_amdgcn_readfirstlane(cbase);
        if (cbase + total > (int)fc.candCap) { if (lane == 0) atomicOr(status, 1); return; }
        /* the cell's output run starts at a wave-uniform element (scalar base); a lane adds its 32-bit offset */
        const size_t run = (size_t)__builtin_amdgcn_readfirstlane(slot) * (size_t)candSlotElems + fc.candOff + (size_t)cbase;
        uint32_t* const out0 = cand0 + run;
        uint32_t* const out1 = cand1 + run;
        uint32_t rowStart = 0;                            /* scalar: candidates of the rows before this one */
        /* the constant parts of the two records: keypoint coordinates as the reference leaves them in vToDistributeKeys
         * (:822-823), and the emission-order key */
        const uint32_t k0base = ((uint32_t)x + 3 + fc.offX) | (((uint32_t)y0 + 3 + fc.offY) << 12);
        const uint32_t k1base = (fc.cellIdx << 12) | ((uint32_t)y0 << 6) | (uint32_t)x;
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
            if (dm[i] != 0) {
                const uint32_t hi = dm[i] >> 16 ? 1u : 0u;
                const uint32_t sc = hi ? s[i] >> 16 : s[i] & 0xFFFFu;
                /* x + hi and y0 + i never carry out of their fields (x + 1 < 64, y0 + i < 64; kx, ky < 4096) */
                /* byte offset in 32-bit arithmetic: the store address stays scalar base + VGPR offset */
                const uint32_t pos = (rowStart + __builtin_amdgcn_mbcnt_hi((uint32_t)(rowMask[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)rowMask[i], 0u))) * 4u;
                *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out0) + pos) = (k0base + hi + ((uint32_t)i << 12)) | (sc << 24);
                *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out1) + pos) = k1base + hi + ((uint32_t)i << 6);
            }
            rowStart += (uint32_t)__popcll(rowMask[i]);
        }
    };
    if (screen >= 2) {
        int nS, nAll;
        const h16x2 thIni = __builtin_bit_cast(h16x2, (uint32_t)iniTh * 0x00010001u);
        const uint32_t p0 = sample(thIni, nS, nAll);
        /* a sample with (nearly) no survivor says "probably no corner at iniTh here" - a low-texture cell, which would pay for both
         * attempts: straight to the exact ways.  Between an eighth and two thirds of the sampled pair rows: worth trying */
        if (8 * nS >= nAll && 3 * nS < 2 * nAll && screened_fill(thIni, p0)) {
            if (nms_ini()) { emit(); return; }
            __syncthreads();           /* the list is written again below */
        }
    }
    bool plain = true;
    if (screen) {
        int nS, nAll;
        const h16x2 thMin = __builtin_bit_cast(h16x2, (uint32_t)minTh * 0x00010001u);
        const uint32_t p0 = sample(thMin, nS, nAll);
        if (4 * nS < nAll) plain = !screened_fill(thMin, p0);          /* fewer than a quarter of the lanes that hold a pixel pair */
    }
    if (plain) {
#pragma unroll
        for (int i = 0; i < RMAX; i++) {
            s[i] = 0;
            if (i < R) {
                const uint32_t r = fastc_strength2(base + i * FASTC_PB);
                s[i] = r & (i < iInv ? mcol : mlast);
            }
        }
    }
    if (!nms_ini()) {                                                  /* fallback decided per cell after NMS@ini */
#pragma unroll
        for (int i = 0; i < RMAX; i++) dm[i] = u2subs(s[i], u2max(nb[i], th7));
    }
    emit();
}

/* ------------------------------------------------------------------------------------------------ */
/* quadtree                                                                                          */

/* k_quadtree<QT_THREADS, QT_KPT, QT_MAXN>: workgroup size, candidate keys a thread keeps in registers,
 * node-list capacity.  The host picks <512,16,*> for the large levels and <256,8,256> for the small ones
 * (less LDS and fewer registers -> more workgroups per CU). */

template <int QT_THREADS, int QT_MAXN> struct QtShared {
    short x0[2][QT_MAXN], x1[2][QT_MAXN], y0[2][QT_MAXN], y1[2][QT_MAXN]; /* UL.x, UR.x, UL.y, BR.y */
    int cnt[2][QT_MAXN];
    unsigned char isNew[2][QT_MAXN];      /* created in the previous round (phase-2 candidates) */
    int ccnt[QT_MAXN * 4];                /* child key counts of this round; reused as child positions */
    int newPos[QT_MAXN];                  /* list position of a surviving node in the next list (steps E-F);
                                             before that (steps A-C) the node's largest-first sort key */
    uint32_t mid[QT_MAXN];                /* candidate of this round: 1<<31 | split y << 12 | split x */
    unsigned char proc[QT_MAXN];          /* node is divided this round */
    unsigned short order[QT_MAXN];        /* processing order of divided nodes */
    int wtot[3][QT_THREADS / 64];         /* wave totals of the three block scans of a round */
    int len, phase, finish, err, take, total, nExp;
};

/* Exclusive block scan over 2*QT_THREADS values, thread t holding elements 2t (a) and 2t+1 (b);
 * returns the grand total.  `wtot` must not be reused before the next barrier after the call. */
template <int QT_THREADS> __device__ __forceinline__ int qt_scan2(int a, int b, int* wtot, int& exA, int& exB)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = a + b;
    const int incl = drfe_wave_incl_scan(s, lane);
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < QT_THREADS / 64; k++) {
        const int t = wtot[k];
        if (k < w) before += t;
        total += t;
    }
    exA = before + incl - s;
    exB = exA + a;
    return total;
}

/* arr[tgt] += 1 for every lane with tgt >= 0, one LDS atomic per distinct counter in the wave.
 * Must be reached by all lanes of the wave. */
__device__ __forceinline__ void qt_wave_add(int* arr, int tgt)
{
    const int lane = threadIdx.x & 63;
    unsigned long long active = __ballot(tgt >= 0);
    while (active) {
        const int leader = __ffsll((long long)active) - 1;
        const int t = __builtin_amdgcn_readlane(tgt, leader);
        const unsigned long long m = __ballot(tgt == t);
        if (lane == leader) atomicAdd(&arr[t], __popcll(m));
        active &= ~m;
    }
}

/* DivideNode key association (:512-526): child counter 4*i+q the key falls in, or -1 if node i is not
 * divided this round */
template <class SH> __device__ __forceinline__ int qt_child_slot(const SH& S, int cur, int i, uint32_t key)
{
    const uint32_t m = S.mid[i];
    if (!(m >> 31)) return -1;
    const int x = (int)(key & 0xFFF), y = (int)((key >> 12) & 0xFFF);
    const int mx = (int)(m & 0xFFF), my = (int)((m >> 12) & 0xFFF);
    (void)cur;
    return 4 * i + ((x < mx) ? ((y < my) ? 0 : 2) : ((y < my) ? 1 : 3));
}

/* One workgroup runs DistributeOctTree for one (slot, level).  The reference's std::list is kept as
 * an array in list order, rebuilt every round:
 *   new list = children of the LAST divided node (n4,n3,n2,n1), ..., children of the FIRST divided
 *              node, followed by the undivided nodes in their old order
 * which is what push_front + erase produce (src/ORBextractor.cc:612-660).  Sweep rounds (:597-666)
 * divide every multi-key node in list order; once a sweep could overshoot N (:673) the rounds divide
 * the previous round's children largest-first (sort at :684, ties by creation order — the canonical
 * rule of SURVEY.md §9.1) and stop as soon as the list holds N nodes (:730).  Keys never move: each
 * carries the list position of its node. */
template <int QT_THREADS, int QT_KPT, int QT_MAXN>
__global__ __launch_bounds__(QT_THREADS, QT_THREADS == 256 ? 8 : 6) void k_quadtree(const DevGeom* __restrict__ G, int levelBase,
                                                         const uint32_t* __restrict__ cand0,
                                                         const uint32_t* __restrict__ cand1,
                                                         uint16_t* __restrict__ node,
                                                         const int* __restrict__ candCount,
                                                         uint32_t* __restrict__ sel, int* __restrict__ selCount,
                                                         int* __restrict__ status)
{
    __shared__ QtShared<QT_THREADS, QT_MAXN> S;
    const int level = levelBase + blockIdx.x, slot = blockIdx.y, tid = threadIdx.x;
    const DevLevel& L = G->lv[level];
    const int n = min(candCount[DRFE_CC_IDX(slot, level)], L.candCap);
    const size_t coff = (size_t)slot * G->candSlotElems + L.candOff;
    const uint32_t* k0 = cand0 + coff;
    const uint32_t* k1 = cand1 + coff;
    uint16_t* nd = node + coff;
    const int N = L.quota;
    const int maxNodes = L.kpCap;
    if (n == 0) { if (tid == 0) selCount[slot * G->nlevels + level] = 0; return; }

    /* root nodes, :543-566 */
    const int nIni = L.nIni;
    for (int i = tid; i < nIni; i += QT_THREADS) {
        S.x0[0][i] = (short)(int)(L.hX * (float)i);
        S.x1[0][i] = (short)(int)(L.hX * (float)(i + 1));
        S.y0[0][i] = 0;
        S.y1[0][i] = (short)(L.maxBY - L.minBY);
        S.cnt[0][i] = 0;
        S.isNew[0][i] = 0;
    }
    if (tid == 0) { S.err = 0; S.finish = 0; S.phase = 1; S.nExp = 0; S.take = 0x7FFFFFFF; }
    __syncthreads();
    /* keys (x | y << 12 | score << 24) and their node positions live in registers: QT_KPT per thread;
     * only a level with more than QT_KPT * QT_THREADS candidates spills the rest to the global arrays */
    uint32_t rk[QT_KPT], rn2[QT_KPT / 2];     /* node positions: two 16-bit fields per register */
#define RN_GET(t) ((rn2[(t) >> 1] >> (((t) & 1) * 16)) & 0xFFFFu)
#define RN_SET(t, v) (rn2[(t) >> 1] = (rn2[(t) >> 1] & ~(0xFFFFu << (((t) & 1) * 16))) | ((uint32_t)(v) << (((t) & 1) * 16)))
#pragma unroll
    for (int t = 0; t < QT_KPT; t++) { const int k = tid + t * QT_THREADS; rk[t] = k < n ? k0[k] : 0u; rn2[t >> 1] = 0; }
#define QT_FOR_KEYS(...)                                                                               \
    _Pragma("unroll") for (int t = 0; t < QT_KPT; t++) {                                               \
        if (tid + t * QT_THREADS < n) { const uint32_t KEY = rk[t]; uint32_t NODE = RN_GET(t); (void)KEY; __VA_ARGS__; RN_SET(t, NODE); } \
    }                                                                                                  \
    for (int k = tid + QT_KPT * QT_THREADS; k < n; k += QT_THREADS) {                                  \
        const uint32_t KEY = k0[k]; uint32_t NODE = nd[k]; (void)KEY; __VA_ARGS__; nd[k] = (uint16_t)NODE;               \
    }
#pragma unroll
    for (int t = 0; t < QT_KPT; t++) {
        if (t * QT_THREADS >= n) break;
        int r = -1;
        if (tid + t * QT_THREADS < n) { r = min((int)((float)(int)(rk[t] & 0xFFF) / L.hX), nIni - 1); RN_SET(t, r); }
        qt_wave_add(S.cnt[0], r);
    }
    for (int kb = QT_KPT * QT_THREADS; kb < n; kb += QT_THREADS) {
        int r = -1;
        if (kb + tid < n) { r = min((int)((float)(int)(k0[kb + tid] & 0xFFF) / L.hX), nIni - 1); nd[kb + tid] = (uint16_t)r; }
        qt_wave_add(S.cnt[0], r);
    }
    __syncthreads();
    /* erase empty roots, :577-590 */
    if (tid == 0) {
        int m = 0;
        for (int i = 0; i < nIni; i++) {
            S.newPos[i] = m;
            if (S.cnt[0][i] > 0) {
                S.x0[0][m] = S.x0[0][i]; S.x1[0][m] = S.x1[0][i]; S.y0[0][m] = S.y0[0][i]; S.y1[0][m] = S.y1[0][i];
                S.cnt[0][m] = S.cnt[0][i];
                m++;
            }
        }
        S.len = m;
    }
    __syncthreads();
    if (S.len != nIni)
        QT_FOR_KEYS({ NODE = (uint32_t)S.newPos[NODE]; })
    __syncthreads();

    int cur = 0;
    while (true) {
        const int len = S.len;
        const int phase = S.phase;
        /* A. candidates of this round */
        for (int i = tid; i < len; i += QT_THREADS) {
            const int ci = S.cnt[cur][i];
            const bool c = ci > 1 && (phase == 1 || S.isNew[cur][i]);
            S.proc[i] = c ? 1 : 0;
            const int mx = S.x0[cur][i] + ((S.x1[cur][i] - S.x0[cur][i] + 1) >> 1);
            const int my = S.y0[cur][i] + ((S.y1[cur][i] - S.y0[cur][i] + 1) >> 1);
            S.mid[i] = c ? (0x80000000u | (uint32_t)mx | ((uint32_t)my << 12)) : 0u;
            /* largest-first order of the second phase (:684): bigger size first, equal sizes -> later
             * created first == smaller list position (children sit reversed at the list front) */
            S.newPos[i] = c ? (int)(((uint32_t)ci << 10) | (uint32_t)(QT_MAXN - 1 - i)) : 0;
            S.ccnt[4 * i] = 0; S.ccnt[4 * i + 1] = 0; S.ccnt[4 * i + 2] = 0; S.ccnt[4 * i + 3] = 0;
        }
        __syncthreads();
        /* B. DivideNode key association (:512-526) for every candidate node */
        {
            /* while the list is short thousands of keys fall on a handful of counters: same-address LDS
             * atomics serialise lane by lane, so a wave first merges its lanes per counter */
            const bool agg = len <= 4;
#pragma unroll
            for (int t = 0; t < QT_KPT; t++) {
                if (t * QT_THREADS >= n) break;                       /* block-uniform */
                int tgt = -1;
                if (tid + t * QT_THREADS < n) tgt = qt_child_slot(S, cur, (int)RN_GET(t), rk[t]);
                if (agg) qt_wave_add(S.ccnt, tgt); else if (tgt >= 0) atomicAdd(&S.ccnt[tgt], 1);
            }
            for (int kb = QT_KPT * QT_THREADS; kb < n; kb += QT_THREADS) {
                int tgt = -1;
                if (kb + tid < n) tgt = qt_child_slot(S, cur, (int)nd[kb + tid], k0[kb + tid]);
                if (agg) qt_wave_add(S.ccnt, tgt); else if (tgt >= 0) atomicAdd(&S.ccnt[tgt], 1);
            }
        }
        __syncthreads();
        /* C. which candidates are divided, in which order.  Thread t owns list positions (sweep rounds) or
         * ranks (largest-first rounds) 2t, 2t+1.  Outputs: for each of its two elements whether that node
         * is divided (dv), which node it is (i), how many children earlier-processed nodes create (c);
         * for its two list positions whether the node survives undivided (s) and how many survivors
         * precede it (ps). */
        const int e0 = 2 * tid, e1 = 2 * tid + 1;
        int i0 = e0, i1 = e1, c0, c1, ps0, ps1, total, survivors;
        bool dv0, dv1, s0, s1;
#define QT_NCH(i) ((S.ccnt[4 * (i)] > 0) + (S.ccnt[4 * (i) + 1] > 0) + (S.ccnt[4 * (i) + 2] > 0) + (S.ccnt[4 * (i) + 3] > 0))
        if (phase == 1) {
            /* sweep (:597-666): every candidate is divided, in list order -> one scan of the packed pair
             * (children | survivors << 16) over the list */
            dv0 = e0 < len && S.proc[e0]; dv1 = e1 < len && S.proc[e1];
            s0 = e0 < len && !dv0; s1 = e1 < len && !dv1;
            const int v0 = (dv0 ? QT_NCH(e0) : 0) | ((s0 ? 1 : 0) << 16), v1 = (dv1 ? QT_NCH(e1) : 0) | ((s1 ? 1 : 0) << 16);
            int x0, x1;
            const int tot = qt_scan2<QT_THREADS>(v0, v1, S.wtot[0], x0, x1);
            c0 = x0 & 0xFFFF; c1 = x1 & 0xFFFF; ps0 = x0 >> 16; ps1 = x1 >> 16;
            total = tot & 0xFFFF; survivors = tot >> 16;
        } else {
            int m;
            {
                const int p0 = e0 < len ? S.proc[e0] : 0, p1 = e1 < len ? S.proc[e1] : 0;
                int x0, x1;
                m = qt_scan2<QT_THREADS>(p0, p1, S.wtot[0], x0, x1);
            }
            /* rank sort on the keys step A left in newPos[] (0 for nodes that are not candidates) */
            for (int i = tid; i < len; i += QT_THREADS) {
                const uint32_t ki = (uint32_t)S.newPos[i];
                if (ki == 0) continue;
                int rank = 0;
                for (int j = 0; j < len; j++) rank += ((uint32_t)S.newPos[j] > ki) ? 1 : 0;
                S.order[rank] = (unsigned short)i;
            }
            __syncthreads();
            /* children per candidate in processing order; the division stops with the node that brings the
             * list to N (:730) */
            int nch0 = 0, nch1 = 0;
            i0 = 0; i1 = 0;
            if (e0 < m) { i0 = S.order[e0]; nch0 = QT_NCH(i0); }
            if (e1 < m) { i1 = S.order[e1]; nch1 = QT_NCH(i1); }
            qt_scan2<QT_THREADS>(nch0, nch1, S.wtot[1], c0, c1);
            if (e0 < m && len + (c0 + nch0) - (e0 + 1) >= N) atomicMin(&S.take, e0 + 1);
            if (e1 < m && len + (c1 + nch1) - (e1 + 1) >= N) atomicMin(&S.take, e1 + 1);
            __syncthreads();
            const int take = min(S.take, m);
            if (e0 >= take && e0 < m) S.proc[i0] = 0;
            if (e1 >= take && e1 < m) S.proc[i1] = 0;
            if (e0 == take - 1) S.total = c0 + nch0;
            if (e1 == take - 1) S.total = c1 + nch1;
            __syncthreads();
            total = take > 0 ? S.total : 0;
            dv0 = e0 < take; dv1 = e1 < take;
            s0 = e0 < len && !S.proc[e0]; s1 = e1 < len && !S.proc[e1];
            survivors = qt_scan2<QT_THREADS>(s0 ? 1 : 0, s1 ? 1 : 0, S.wtot[2], ps0, ps1);
        }
#undef QT_NCH
        /* D/E. next list: children of the first processed node END the children block, survivors follow in
         * old order */
        const int nxt = cur ^ 1;
        const int newLen = total + survivors;
        if (newLen > maxNodes || newLen > QT_MAXN) {
            if (tid == 0) { S.err = 1; S.finish = 1; }
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = h ? i1 : i0;
                if (!(h ? dv1 : dv0)) continue;
                const int X0 = S.x0[cur][i], X1 = S.x1[cur][i], Y0 = S.y0[cur][i], Y1 = S.y1[cur][i];
                const int mx = X0 + ((X1 - X0 + 1) >> 1), my = Y0 + ((Y1 - Y0 + 1) >> 1);
                int pos = total - (h ? c1 : c0);
                int nExp = 0;
                for (int q = 0; q < 4; q++) {                  /* n1..n4 pushed front in this order */
                    const int c = S.ccnt[4 * i + q];
                    if (c == 0) { S.ccnt[4 * i + q] = -1; continue; }
                    pos--;
                    S.x0[nxt][pos] = (short)((q & 1) ? mx : X0);
                    S.x1[nxt][pos] = (short)((q & 1) ? X1 : mx);
                    S.y0[nxt][pos] = (short)((q & 2) ? my : Y0);
                    S.y1[nxt][pos] = (short)((q & 2) ? Y1 : my);
                    S.cnt[nxt][pos] = c;
                    S.isNew[nxt][pos] = 1;
                    if (c > 1) nExp++;
                    S.ccnt[4 * i + q] = pos;
                }
                if (nExp) atomicAdd(&S.nExp, nExp);
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int i = h ? e1 : e0;
                if (!(h ? s1 : s0)) continue;
                const int p2 = total + (h ? ps1 : ps0);
                S.x0[nxt][p2] = S.x0[cur][i]; S.x1[nxt][p2] = S.x1[cur][i];
                S.y0[nxt][p2] = S.y0[cur][i]; S.y1[nxt][p2] = S.y1[cur][i];
                S.cnt[nxt][p2] = S.cnt[cur][i];
                S.isNew[nxt][p2] = 0;
                S.newPos[i] = p2;
            }
        }
        __syncthreads();
        if (tid == 0 && !S.err) {
            /* termination, :662-666 / :733-734, and the switch to largest-first rounds, :668 */
            if (newLen >= N || newLen == len) S.finish = 1;
            else if (phase == 1 && newLen + S.nExp * 3 > N) S.phase = 2;
            S.len = newLen;
            S.nExp = 0; S.take = 0x7FFFFFFF;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(S.err)) break;          /* LDS flags as wave-uniform scalars: scalar branches around the barriers */
        /* F. keys follow their node */
        QT_FOR_KEYS({
            const int i = (int)NODE;
            /* a candidate the largest-first round did not reach keeps proc == 0 but still has mid set */
            const int slot4 = S.proc[i] ? qt_child_slot(S, cur, i, KEY) : -1;
            NODE = (uint32_t)(slot4 >= 0 ? S.ccnt[slot4] : S.newPos[i]);
        })
        cur = nxt;
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(S.finish)) break;
    }
    if (S.err) {
        if (tid == 0) { atomicOr(status, 2); selCount[slot * G->nlevels + level] = 0; }
        return;
    }
    /* retain the best key per node, first maximum in emission order wins (:744-760) */
    const int len = S.len;
    unsigned long long* best = reinterpret_cast<unsigned long long*>(S.ccnt);
    for (int i = tid; i < len; i += QT_THREADS) best[i] = 0ull;
    __syncthreads();
    uint32_t ord[QT_KPT];                                   /* ~emission order: larger = earlier */
#pragma unroll
    for (int t = 0; t < QT_KPT; t++) { const int k = tid + t * QT_THREADS; ord[t] = k < n ? ~k1[k] : 0u; }
#pragma unroll
    for (int t = 0; t < QT_KPT; t++)
        if (tid + t * QT_THREADS < n) atomicMax(&best[RN_GET(t)], ((unsigned long long)(rk[t] >> 24) << 32) | ord[t]);
    for (int k = tid + QT_KPT * QT_THREADS; k < n; k += QT_THREADS)
        atomicMax(&best[nd[k]], ((unsigned long long)(k0[k] >> 24) << 32) | (unsigned long long)(~k1[k]));
    __syncthreads();
    uint32_t* out = sel + (size_t)slot * G->kpSlotElems + L.kpOff;
#pragma unroll
    for (int t = 0; t < QT_KPT; t++)
        if (tid + t * QT_THREADS < n && best[RN_GET(t)] == (((unsigned long long)(rk[t] >> 24) << 32) | ord[t])) out[RN_GET(t)] = rk[t];
    for (int k = tid + QT_KPT * QT_THREADS; k < n; k += QT_THREADS) {
        const unsigned long long p = ((unsigned long long)(k0[k] >> 24) << 32) | (unsigned long long)(~k1[k]);
        if (best[nd[k]] == p) out[nd[k]] = k0[k];
    }
#undef QT_FOR_KEYS
#undef RN_GET
#undef RN_SET
    if (tid == 0) selCount[slot * G->nlevels + level] = len;
}

/* ------------------------------------------------------------------------------------------------ */
/* Gaussian blur 7x7, sigma 2, 8.8 fixed point                                     */

#define BLUR_ROWS (DRFE_BLUR_TH + 6)
#define BLUR_COLG (DRFE_BLUR_TW / 4)             /* threads across a tile row (4 px each) */
#define BLUR_ROWL (256 / BLUR_COLG)               /* tile rows a pass of the block covers */
/* Tile = DRFE_BLUR_TW x DRFE_BLUR_TH output pixels, thread = 4 horizontally adjacent pixels.  The source is the
 * BORDERED pyramid level: its 19-px frame already holds the REFLECT_101 image of the interior, which is
 * exactly what GaussianBlur(BORDER_REFLECT_101) of the cloned interior ROI reads (3 px needed), so no
 * reflection logic runs here.  Horizontal pass straight from three aligned dword loads into 8.8 sums
 * in LDS (v_dot4_u32_u8); vertical pass from LDS (v_mad_u32_u16). */
__global__ __launch_bounds__(256) void k_blur(const DevGeom* __restrict__ G, const BlurTile* __restrict__ tiles,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, uint32_t gxMagic)
{
    /* XCD-aware numbering (whole frames per XCD): neighbouring tiles share the 128-byte lines their 136-byte rows straddle and
     * the halo rows; dealt round-robin over eight private L2s every such line was fetched twice (2 x FETCH_SIZE = 2.4 x the
     * algorithmic bytes in round 1).  DRFE_BLUR_NOSWIZZLE=1 (compile time) restores the plain numbering: it measured 1.4 %
     * faster on this VALU-bound kernel, at twice the fabric traffic. */
#if defined(DRFE_BLUR_NOSWIZZLE)
    const int bx = blockIdx.x, by = blockIdx.y; (void)gxMagic;
#else
    int bx, by;
    drfe_xcd_swizzle_2d(gxMagic, bx, by);
#endif
    /* horizontal sums of rows 2p and 2p+1 interleaved per pixel: hb2[p][x] = H[2p][x] | H[2p+1][x] << 16, so that the
     * vertical pass multiplies two rows per instruction (v_dot2_u32_u16) */
    __shared__ __attribute__((aligned(16))) uint32_t hb2[(BLUR_ROWS / 2) * DRFE_BLUR_TW];
    static_assert(BLUR_ROWS % 2 == 0 && DRFE_BLUR_TH % 2 == 0, "row pairs");
    const BlurTile t = tiles[bx];
    const int slot = by, tid = threadIdx.x;
    const DevLevel& L = G->lv[t.level];
    /* block-uniform bases stay in SGPRs (the swizzled bx / by come out of vector arithmetic: readfirstlane says they are
     * uniform); a thread addresses with ONE 32-bit offset per source row and the three dwords of a row are immediate
     * offsets 0 / 4 / 8 of the same load (a level is < 2^24 bytes) */
    const size_t imgOff = (size_t)__builtin_amdgcn_readfirstlane(slot) * (size_t)G->pyrSlotBytes + (size_t)L.pyrOff;
    const uint8_t* img = pyr + imgOff;
    const int x0 = __builtin_amdgcn_readfirstlane(t.tx * DRFE_BLUR_TW), y0 = __builtin_amdgcn_readfirstlane(t.ty * DRFE_BLUR_TH);
    const int cg = tid & (BLUR_COLG - 1), rr = tid / BLUR_COLG;
    /* interior x maps to bordered column x+19; the window of pixels x..x+3 starts at x+16 (4-aligned).  No column clamp: a
     * block tile may overhang the row, its threads then read the bytes that follow (the next row; at the very end of the arena
     * the 256 spare bytes drfe_create allocates) and store nothing */
    const uint32_t col = (uint32_t)(x0 + cg * 4 + 16);
    const uint32_t pitch = (uint32_t)L.pyrPitch;
    const int lastRow = L.h + 2 * DRFE_EDGE - 1;
    /* byte k of (w0,w1,w2) is interior column x-3+k and pixel x+j needs bytes j..j+6 under the taps (18,34,49,55,49,34,18).  The
     * TAPS are shifted to where the bytes lie instead of the bytes to the taps: every product is a v_dot4_u32_u8 of an aligned
     * dword with a constant - 2 + 2 + 3 + 3 of them for the four pixels, no v_alignbyte (8 + 6 instructions before);
     * 257 * 255 = 65535 still fits 16 bits */
#define TAP4(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))
    auto hsum4 = [&](uint32_t off, uint32_t (&h)[4]) {
        const uint32_t* row = reinterpret_cast<const uint32_t*>(img + off);
        const uint32_t w0 = row[0], w1 = row[1], w2 = row[2];
        h[0] = __builtin_amdgcn_udot4(w0, TAP4(18, 34, 49, 55), __builtin_amdgcn_udot4(w1, TAP4(49, 34, 18, 0), 0u, false), false);
        h[1] = __builtin_amdgcn_udot4(w0, TAP4(0, 18, 34, 49), __builtin_amdgcn_udot4(w1, TAP4(55, 49, 34, 18), 0u, false), false);
        h[2] = __builtin_amdgcn_udot4(w0, TAP4(0, 0, 18, 34), __builtin_amdgcn_udot4(w1, TAP4(49, 55, 49, 34),
                                      __builtin_amdgcn_udot4(w2, TAP4(18, 0, 0, 0), 0u, false), false), false);
        h[3] = __builtin_amdgcn_udot4(w0, TAP4(0, 0, 0, 18), __builtin_amdgcn_udot4(w1, TAP4(34, 49, 55, 49),
                                      __builtin_amdgcn_udot4(w2, TAP4(34, 18, 0, 0), 0u, false), false), false);
    };
#undef TAP4
    /* low halves of two registers as one dword (the sums fit 16 bits): one v_perm_b32 */
    auto pack16 = [](uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x05040100u); };
    /* source rows y0-3 .. min(y0 + TH, h) + 2 are all the stored output rows read: a tile that overhangs the level's last row
     * (a third of the tile area on the small levels) stops there - whole wavefronts drop out, a wave is two tile rows */
    const int pEnd = __builtin_amdgcn_readfirstlane(min(BLUR_ROWS / 2, (min(DRFE_BLUR_TH, L.h - y0) + 7) >> 1));
    for (int p = rr; p < pEnd; p += BLUR_ROWL) {
        uint32_t ha[4], hc[4];
        const uint32_t ra = (uint32_t)min(y0 + 2 * p - 3 + DRFE_EDGE, lastRow), rc = (uint32_t)min(y0 + 2 * p - 2 + DRFE_EDGE, lastRow);
        hsum4(__umul24(ra, pitch) + col, ha);
        hsum4(__umul24(rc, pitch) + col, hc);
        *reinterpret_cast<uint4*>(&hb2[p * DRFE_BLUR_TW + cg * 4]) =
            make_uint4(pack16(ha[0], hc[0]), pack16(ha[1], hc[1]), pack16(ha[2], hc[2]), pack16(ha[3], hc[3]));
    }
    __syncthreads();
    /* vertical pass, two output rows (2q, 2q+1) per thread from the four row pairs q..q+3:
     *   row 2q   = (18,34).P[q] + (49,55).P[q+1] + (49,34).P[q+2] + 18 * lo(P[q+3])
     *   row 2q+1 = 18 * hi(P[q]) + (34,49).P[q+1] + (55,49).P[q+2] + (34,18).P[q+3] */
    typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
    const u16x2v t1834 = {18, 34}, t4955 = {49, 55}, t4934 = {49, 34}, t3449 = {34, 49}, t5549 = {55, 49}, t3418 = {34, 18};
    const int blurPitch = L.blurPitch, levelH = L.h;
    if (x0 + cg * 4 >= blurPitch * DRFE_BTILE_W) return;      /* the block tile may overhang the last layout tile */
    uint8_t* base = blur + (size_t)__builtin_amdgcn_readfirstlane(slot) * (size_t)G->blurSlotBytes + (size_t)L.blurOff;
    for (int q = rr; q < DRFE_BLUR_TH / 2; q += BLUR_ROWL) {
        const int y = y0 + 2 * q;
        if (y >= levelH) continue;
        uint4 P[4];
#pragma unroll
        for (int k = 0; k < 4; k++) P[k] = *reinterpret_cast<const uint4*>(&hb2[(q + k) * DRFE_BLUR_TW + cg * 4]);
        /* the single-row term opens the sum together with the rounding constant (v_mad_u32_u16 picks the 16-bit half by
         * op_sel); (sum >> 16) saturated to a byte, two pixels at a time, is one v_ashr_pk_u8_i32 */
        uint32_t a[4], c2v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t p0 = (&P[0].x)[j], p1 = (&P[1].x)[j], p2 = (&P[2].x)[j], p3 = (&P[3].x)[j];
            asm("v_mad_u32_u16 %0, %1, 18, %2" : "=v"(a[j]) : "v"(p3), "s"(32768u));
            a[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, p0), t1834, a[j], false);
            a[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, p1), t4955, a[j], false);
            a[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, p2), t4934, a[j], false);
            asm("v_mad_u32_u16 %0, %1, 18, %2 op_sel:[1,0,0,0]" : "=v"(c2v[j]) : "v"(p0), "s"(32768u));
            c2v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, p1), t3449, c2v[j], false);
            c2v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, p2), t5549, c2v[j], false);
            c2v[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2v, p3), t3418, c2v[j], false);
        }
        const uint32_t outA = pack16((uint32_t)__builtin_amdgcn_ashr_pk_u8_i32((int)a[0], (int)a[1], 16),
                                     (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32((int)a[2], (int)a[3], 16));
        const uint32_t outB = pack16((uint32_t)__builtin_amdgcn_ashr_pk_u8_i32((int)c2v[0], (int)c2v[1], 16),
                                     (uint32_t)__builtin_amdgcn_ashr_pk_u8_i32((int)c2v[2], (int)c2v[3], 16));
        *reinterpret_cast<uint32_t*>(base + drfe_blur_offset(x0 + cg * 4, y, blurPitch)) = outA;
        if (y + 1 < levelH) *reinterpret_cast<uint32_t*>(base + drfe_blur_offset(x0 + cg * 4, y + 1, blurPitch)) = outB;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* orientation + rBRIEF + keypoint finishing: one wavefront per keypoint                             */

#define DESC_DISC_PER_LANE 12                     /* ceil(749 / 64) */
#ifndef DESC_KPW
#define DESC_KPW 4                                /* keypoints a wavefront carries at once */
#endif
#define DESC_REACH 18                             /* round(13 * sqrt(2)): how far the rotated pattern reaches; keypoints
                                                     sit >= 19 px inside their level (16-px region + FAST's 3-px margin) */
#define DESC_RAW_ROWS 31                          /* disc patch: rows y-15..y+15, 9 aligned dwords cover x-15..x+15 */
#define DESC_RAW_DW 9
#define DESC_BLUR_ROWS (2 * DESC_REACH + 1)       /* pattern patch: rows y-18..y+18, 10 aligned dwords cover x-18..x+18 */
#define DESC_BLUR_DW 10
#define DESC_RAW_N (DESC_RAW_ROWS * DESC_RAW_DW)      /* 279 dwords = 5 loads per lane */
#define DESC_BLUR_N (DESC_BLUR_ROWS * DESC_BLUR_DW)   /* 370 dwords = 6 loads per lane */
#define DESC_RAW_LD 5
#define DESC_BLUR_LD 6
#define DESC_LDS_DW 372
/* A byte gather with 64 scattered addresses costs the texture path ~30 cycles per instruction, and a keypoint needs 20
 * of them (12 for the moments, 8 for the pattern).  So each keypoint's two patches are fetched as aligned dwords along
 * rows instead (11 coalesced loads), parked in a wave-private LDS region, and the 20 gathers become ds_read_u8.  A
 * wavefront carries DESC_KPW keypoints through the phases together: every global load of the wave is issued up front,
 * and the lane-constant tables (disc offsets, pattern) are loaded once per wavefront.  The raw patch and the blurred
 * patch share the LDS region (LDS operations of one wave execute in order). */
#ifndef DESC_THREADS
#define DESC_THREADS 64               /* the wavefronts of this kernel never synchronise with each other: one per workgroup */
#endif
#define DESC_WAVES (DESC_THREADS / WAVE)
__global__ __launch_bounds__(DESC_THREADS) void k_orient_desc(const DevGeom* __restrict__ G, const uint8_t* __restrict__ pyr,
                                                     const uint8_t* __restrict__ blur,
                                                     const uint32_t* __restrict__ sel,
                                                     const int* __restrict__ selCount,
                                                     const int8_t* __restrict__ pattern,
                                                     const int16_t* __restrict__ disc, int discCount,
                                                     drfe_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                     int* __restrict__ kpCount, int maxKp, uint32_t gxMagic)
{
    __shared__ uint32_t sPatch[DESC_WAVES][DESC_KPW][DESC_LDS_DW];
    int bx, by;
    drfe_xcd_swizzle_2d(gxMagic, bx, by);        /* a frame's keypoints on one XCD: overlapping patches share its L2 */
    const int slot = by;
    const int lane = threadIdx.x & (WAVE - 1);
    /* first output index of this wave; readfirstlane makes it (and every level / pointer derived from it) wave-uniform
     * for the compiler: scalar ALU and SGPR base addresses instead of 64-bit vector address arithmetic per gather */
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g0 = (bx * DESC_WAVES + wv) * DESC_KPW;
    const int nl = G->nlevels;
    /* level of every output index: prefix sums of the per-level counts (level-major concatenation, :1103), kept
     * across lanes (lane l = level l) so that the lookup per keypoint is a compare, a ballot and a readlane instead of a
     * 16-deep scalar select chain (the scalar unit was the busiest one in this kernel) */
    const int cntL = lane < nl ? selCount[slot * nl + lane] : 0;
    int incl = cntL;                              /* inclusive prefix inside the first row of 16 lanes: four DPP adds */
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);      /* row_shr:1, 0 shifted in */
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);      /* row_shr:2 */
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);      /* row_shr:4 */
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);      /* row_shr:8 */
    static_assert(DRFE_MAX_LEVELS == 16, "the level prefix assumes one DPP row");
    const int excl = incl - cntL;
    const int total = __builtin_amdgcn_readlane(incl, DRFE_MAX_LEVELS - 1);
    if (g0 == 0 && lane == 0) kpCount[slot] = min(total, maxKp);
    if (g0 >= total || g0 >= maxKp) return;                   /* wave-uniform */
    int level[DESC_KPW], xi[DESC_KPW], yi[DESC_KPW], resp[DESC_KPW], pitchP[DESC_KPW], pitchB[DESC_KPW];
    int shP[DESC_KPW], shB[DESC_KPW];             /* byte position of the patch's first column inside its first dword */
    int bx0[DESC_KPW], by0[DESC_KPW];             /* blurred patch: first (dword-aligned) column and first row */
    bool live[DESC_KPW];
    const uint8_t* cP[DESC_KPW];
    const uint8_t* cB[DESC_KPW];
    uint32_t key[DESC_KPW];
#pragma unroll
    for (int j = 0; j < DESC_KPW; j++) {
        const int g = g0 + j;
        live[j] = g < total && g < maxKp;
        /* levels whose run ends at or before g; empty levels share their end with the previous one and are skipped */
        const int lv = live[j] ? __popcll(__ballot(lane < DRFE_MAX_LEVELS && g >= incl)) : 0;
        const int first = __builtin_amdgcn_readlane(excl, lv);
        level[j] = lv;
        key[j] = live[j] ? sel[(size_t)slot * G->kpSlotElems + G->lv[lv].kpOff + (g - first)] : 0u;
    }
#pragma unroll
    for (int j = 0; j < DESC_KPW; j++) {
        const DevLevel& L = G->lv[level[j]];
        xi[j] = (int)(key[j] & 0xFFF) + L.minBX; yi[j] = (int)((key[j] >> 12) & 0xFFF) + L.minBY;
        resp[j] = (int)(key[j] >> 24);
        pitchP[j] = L.pyrPitch; pitchB[j] = L.blurPitch;
        if (!live[j]) { xi[j] = L.minBX + 16; yi[j] = L.minBY + 16; }      /* a harmless in-bounds position */
        /* top-left corners of the patches, moved left to a dword boundary (slot and level offsets are multiples of
         * 256, pitches of 64, the arenas come from hipMalloc): the disc reaches 15 px, the rotated pattern 18 px */
        const size_t oP = (size_t)slot * G->pyrSlotBytes + L.pyrOff + (size_t)(yi[j] + DRFE_EDGE - 15) * L.pyrPitch + (xi[j] + DRFE_EDGE - 15);
        /* blurred level: tiled (drfe_blur_offset); cB = the level, (bx0, by0) = the patch corner moved left to a dword */
        shP[j] = (int)(oP & 3); shB[j] = (xi[j] - DESC_REACH) & 3;
        cP[j] = pyr + (oP & ~(size_t)3);
        cB[j] = blur + (size_t)slot * G->blurSlotBytes + L.blurOff;
        bx0[j] = (xi[j] - DESC_REACH) & ~3; by0[j] = yi[j] - DESC_REACH;
    }
    /* every global load of the wave: lane t of load k fetches dword (t / 9, t % 9) of the raw patch and (t / 10, t % 10)
     * of the blurred one (t = lane + 64 k) */
    uint32_t rw[DESC_KPW][DESC_RAW_LD], bw[DESC_KPW][DESC_BLUR_LD];
    {
        int rRow[DESC_RAW_LD], rCol[DESC_RAW_LD], bRow[DESC_BLUR_LD], bCol[DESC_BLUR_LD];
#pragma unroll
        for (int k = 0; k < DESC_RAW_LD; k++) {
            const int t = min(lane + k * WAVE, DESC_RAW_N - 1);               /* the tail repeats the last dword */
            rRow[k] = (int)(__umul24(t, 7282) >> 16); rCol[k] = (t - __mul24(rRow[k], DESC_RAW_DW)) * 4;
        }
#pragma unroll
        for (int k = 0; k < DESC_BLUR_LD; k++) {
            const int t = min(lane + k * WAVE, DESC_BLUR_N - 1);
            bRow[k] = (int)(__umul24(t, 6554) >> 16); bCol[k] = (t - __mul24(bRow[k], DESC_BLUR_DW)) * 4;
        }
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++) {
#pragma unroll
            for (int k = 0; k < DESC_RAW_LD; k++)
                rw[j][k] = *reinterpret_cast<const uint32_t*>(cP[j] + (uint32_t)(__mul24(rRow[k], pitchP[j]) + rCol[k]));   /* 24-bit multiply: full rate */
#pragma unroll
            for (int k = 0; k < DESC_BLUR_LD; k++)
                bw[j][k] = *reinterpret_cast<const uint32_t*>(cB[j] + drfe_blur_offset(bx0[j] + bCol[k], by0[j] + bRow[k], pitchB[j]));
        }
    }
    /* IC_Angle on the unblurred level: integer moments over the radius-15 disc (749 px = 12 offsets per lane) */
    const uint32_t* disc32 = reinterpret_cast<const uint32_t*>(disc);         /* u | v << 16, int16 each */
    uint32_t uv[DESC_DISC_PER_LANE];
#pragma unroll
    for (int k = 0; k < DESC_DISC_PER_LANE; k++) {
        const int t = lane + k * WAVE;
        uv[k] = t < discCount ? disc32[t] : 0u;               /* padding entries read the centre pixel, weight 0 */
    }
    int m10[DESC_KPW], m01[DESC_KPW];
    {
        int I[DESC_KPW][DESC_DISC_PER_LANE];
        int du[DESC_DISC_PER_LANE], dv[DESC_DISC_PER_LANE];
#pragma unroll
        for (int k = 0; k < DESC_DISC_PER_LANE; k++) { du[k] = (int)(short)(uv[k] & 0xFFFF); dv[k] = (int)(short)(uv[k] >> 16); }
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++)
#pragma unroll
            for (int k = 0; k < DESC_RAW_LD; k++) {
                const int t = lane + k * WAVE;
                if (t < DESC_RAW_N) sPatch[wv][j][t] = rw[j][k];
            }
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++) {
            const uint8_t* pb = reinterpret_cast<const uint8_t*>(sPatch[wv][j]) + shP[j];
#pragma unroll
            for (int k = 0; k < DESC_DISC_PER_LANE; k++) I[j][k] = pb[__mul24(dv[k] + 15, DESC_RAW_DW * 4) + (du[k] + 15)];
        }
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++) {
            m10[j] = 0; m01[j] = 0;
#pragma unroll
            for (int k = 0; k < DESC_DISC_PER_LANE; k++) {
                m10[j] += du[k] * I[j][k];
                m01[j] += dv[k] * I[j][k];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DESC_KPW; j++) { m10[j] = drfe_wave_sum_i32(m10[j]); m01[j] = drfe_wave_sum_i32(m01[j]); }      /* integer moments: any order */
    /* steered BRIEF on the blurred level: lane owns bit `lane` of each of the four 64-bit words */
    uint32_t pat[4];
#pragma unroll
    for (int r = 0; r < 4; r++) pat[r] = reinterpret_cast<const uint32_t*>(pattern)[r * 64 + lane];
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle[DESC_KPW], cosA[DESC_KPW], sinA[DESC_KPW];
    {   /* fastAtan2 and the float64 sin/cos once per wave: lane j evaluates keypoint j, the results are broadcast */
        int my01 = 0, my10 = 0;
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++)
            if (lane == j) { my01 = m01[j]; my10 = m10[j]; }
        const float ang = drfe_fast_atan2((float)my01, (float)my10);
        float sn, cs;
        drfe_sincos(ang * factorPI, &sn, &cs);
#pragma unroll
        for (int j = 0; j < DESC_KPW; j++) {
            angle[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ang), j));
            cosA[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cs), j));
            sinA[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sn), j));
        }
    }
    /* the blurred patches take over the LDS regions (the moment reads above are done: same wave, in order) */
#pragma unroll
    for (int j = 0; j < DESC_KPW; j++)
#pragma unroll
        for (int k = 0; k < DESC_BLUR_LD; k++) {
            const int t = lane + k * WAVE;
            if (t < DESC_BLUR_N) sPatch[wv][j][t] = bw[j][k];
        }
    /* The rotated pattern on the packed-f32 pipe: points 0 and 1 of a pair ride in one 64-bit register pair, so every
     * product and sum is one v_pk_*_f32 for both (each product and each sum still rounds on its own: no fused
     * multiply-add).  cvRound is the float add of 1.5 * 2^23: the sum's ulp is 1 there, round-to-nearest-even, and the
     * integer sits in the mantissa (0x4B400000 + n); v_mad_u32_u24 takes the low 24 bits of the row word (0x400000 + ry)
     * times the LDS row pitch plus the whole column word, and the constant excess is subtracted once. */
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 PX[4], PY[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint32_t q4 = pat[r];
        PX[r] = f32x2{(float)(int8_t)(q4 & 0xFF), (float)(int8_t)((q4 >> 16) & 0xFF)};
        PY[r] = f32x2{(float)(int8_t)((q4 >> 8) & 0xFF), (float)(int8_t)(q4 >> 24)};
    }
    const f32x2 magic = {12582912.0f, 12582912.0f};
    const uint32_t excess = 0x400000u * (DESC_BLUR_DW * 4) + 0x4B400000u;       /* what the two biased words add */
    int t0[DESC_KPW][4], t1[DESC_KPW][4];
#pragma unroll
    for (int j = 0; j < DESC_KPW; j++) {
        const f32x2 a = {cosA[j], cosA[j]}, b = {sinA[j], sinA[j]};
        /* the patch centre: (DESC_REACH, DESC_REACH) of the staged rows, plus the dword-alignment shift */
        const uint8_t* pc = reinterpret_cast<const uint8_t*>(sPatch[wv][j]) + shB[j] + DESC_REACH * (DESC_BLUR_DW * 4) + DESC_REACH;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const f32x2 ry = (PX[r] * b + PY[r] * a) + magic;      /* cvRound(x*b + y*a) */
            const f32x2 rx = (PX[r] * a - PY[r] * b) + magic;      /* cvRound(x*a - y*b) */
            /* elements copied to scalars first: __builtin_bit_cast applied to `v.y` directly reads element 0 (clang) */
            const float ry0 = ry.x, ry1 = ry.y, rx0 = rx.x, rx1 = rx.y;
            const uint32_t i0 = __umul24(__float_as_uint(ry0), DESC_BLUR_DW * 4) + __float_as_uint(rx0) - excess;
            const uint32_t i1 = __umul24(__float_as_uint(ry1), DESC_BLUR_DW * 4) + __float_as_uint(rx1) - excess;
            t0[j][r] = pc[(int)i0];
            t1[j][r] = pc[(int)i1];
        }
    }
#pragma unroll
    for (int j = 0; j < DESC_KPW; j++) {
        unsigned long long w4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) w4[r] = __ballot(t0[j][r] < t1[j][r]);
        if (lane == 0 && live[j]) {
            const int g = g0 + j;
            uint8_t* drow = desc + ((size_t)slot * maxKp + g) * 32;
            reinterpret_cast<ulonglong2*>(drow)[0] = make_ulonglong2(w4[0], w4[1]);
            reinterpret_cast<ulonglong2*>(drow)[1] = make_ulonglong2(w4[2], w4[3]);
            const DevLevel& L = G->lv[level[j]];
            drfe_keypoint kp;
            kp.x = (float)xi[j]; kp.y = (float)yi[j];
            if (level[j] != 0) { kp.x *= L.scale; kp.y *= L.scale; }
            kp.size = L.kpSize;
            kp.angle = angle[j];
            kp.response = (float)resp[j];
            kp.octave = level[j];
            kp.class_id = -1;
            kps[(size_t)slot * maxKp + g] = kp;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* host launcher                                                                                     */


__global__ void k_clear_counts(int* __restrict__ candCount, int n, int* __restrict__ status)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) candCount[(size_t)i * DRFE_CC_LINE] = 0;
    if (i == 0) status[0] = 0;
}

hipError_t drfe_launch_orb(drfe_ctx* c, const uint8_t* d_gray, size_t frameStride, size_t rowStride, int nframes,
                           hipStream_t s)
{
    const DevGeom& g = c->geom;
    const int nl = g.nlevels;
    hipError_t e;
    /* a kernel, not two hipMemsetAsync: memset nodes of a captured graph (the single-frame entry replays one) did not run
     * on this ROCm, and one launch is cheaper than two anyway */
    hipLaunchKernelGGL(k_clear_counts, dim3((nframes * DRFE_CC_SLOT / DRFE_CC_LINE + 255) / 256), dim3(256), 0, s, c->d_candCount, nframes * DRFE_CC_SLOT / DRFE_CC_LINE, c->d_status);
    e = hipGetLastError();
    if (e != hipSuccess) return e;

    prof_begin(c, DRFE_STAGE_PYRAMID, s);
    {
        const DevLevel& L = g.lv[0];
        const int bh = L.h + 2 * DRFE_EDGE;
        const bool aligned16 = (((uintptr_t)d_gray | frameStride | rowStride) & 15) == 0;
        /* wide kernel: 16-px groups whose two aligned source words lie inside the row: x16 - 19 >= 13, x16 <= w */
        int x16First = 32, x16Last = (L.w / 16) * 16, leftEnd = 0, rightBegin = 0;
        if (aligned16 && x16Last >= x16First) {
            const int nWide = (x16Last - x16First) / 16 + 1;
            hipLaunchKernelGGL(k_pyr_level0_wide, dim3((nWide + 63) / 64, (bh + 4 * PYR0_ROWS - 1) / (4 * PYR0_ROWS), nframes),
                               dim3(64, 4), 0, s, L, g.pyrSlotBytes, d_gray, frameStride, rowStride, x16First, x16Last, c->d_pyr);
            leftEnd = x16First; rightBegin = x16Last + 16;
        }
        const int nEdge = leftEnd / 4 + (L.pyrPitch - rightBegin) / 4;      /* leftEnd == rightBegin == 0: every dword */
        hipLaunchKernelGGL(k_pyr_level0_edge, dim3((nEdge * bh + 255) / 256, nframes), dim3(256), 0, s, L,
                           g.pyrSlotBytes, d_gray, frameStride, rowStride, leftEnd, rightBegin, c->d_pyr);
    }
    for (int l = 1; l < nl; l++) {
        const DevLevel& L = g.lv[l];
        dim3 grid((L.pyrPitch / 4 + 63) / 64, (L.h + 2 * DRFE_EDGE + 4 * PYR_ROWS - 1) / (4 * PYR_ROWS), nframes);
        if (L.resizeLds)
            hipLaunchKernelGGL(k_pyr_resize_lds, grid, dim3(64, 4), 0, s, L, g.lv[l - 1], g.pyrSlotBytes, c->d_taps, c->d_pyr,
                               drfe_div_magic(grid.x * grid.y), drfe_div_magic(grid.x));
        else
            hipLaunchKernelGGL(k_pyr_resize, grid, dim3(64, 4), 0, s, L, g.lv[l - 1], g.pyrSlotBytes, c->d_taps, c->d_pyr);
    }
    prof_end(c, DRFE_STAGE_PYRAMID, s);

    prof_begin(c, DRFE_STAGE_FAST, s);
    if (g.fastCols && !c->fastGeneric) {
        const int nSmall = g.fastColsSmall, nBig = g.totalCells - nSmall;
        if (nSmall > 0)
            hipLaunchKernelGGL((k_fast_cells_cols<8>), dim3(nSmall, nframes), dim3(64), (size_t)fastc_lds_bytes(g.fastColsRows, 8 * 64), s,
                               c->d_cells, g.nlevels, g.pyrSlotBytes, g.candSlotElems, g.iniTh, g.minTh, c->d_pyr, c->d_cand0,
                               c->d_cand1, c->d_candCount, c->d_status, drfe_div_magic((uint32_t)nSmall), 0, c->fastScreen, fastc_list_off(g.fastColsRows), fastc_list_cap(g.fastColsRows, 8 * 64));
        if (nBig > 0) {
            prof_end(c, DRFE_STAGE_FAST, s);
            prof_begin(c, DRFE_STAGE_FAST_B, s);
        }
        if (nBig > 0)
            hipLaunchKernelGGL((k_fast_cells_cols<DRFE_FASTC_MAX_RPL>), dim3(nBig, nframes), dim3(64),
                               (size_t)fastc_lds_bytes(g.fastColsRows, DRFE_FASTC_MAX_RPL * 64), s, c->d_cells, g.nlevels, g.pyrSlotBytes, g.candSlotElems,
                               g.iniTh, g.minTh, c->d_pyr, c->d_cand0, c->d_cand1, c->d_candCount, c->d_status,
                               drfe_div_magic((uint32_t)nBig), nSmall, c->fastScreen, fastc_list_off(g.fastColsRows), fastc_list_cap(g.fastColsRows, DRFE_FASTC_MAX_RPL * 64));
        if (nBig > 0) prof_end(c, DRFE_STAGE_FAST_B, s);
        else prof_end(c, DRFE_STAGE_FAST, s);
    } else {
        hipLaunchKernelGGL(k_fast_cells, dim3(g.totalCells, nframes), dim3(64), (size_t)fast_lds_bytes(g.fastMaxWh), s, c->d_cells,
                           g.nlevels, g.pyrSlotBytes, g.candSlotElems, g.iniTh, g.minTh, c->d_pyr,
                           c->d_cand0, c->d_cand1, c->d_candCount, c->d_status, fast_sc_off(g.fastMaxWh),
                           drfe_div_magic((uint32_t)g.totalCells));
        prof_end(c, DRFE_STAGE_FAST, s);
    }

    prof_begin(c, DRFE_STAGE_QUADTREE, s);
    {
        /* levels are ordered large -> small: [0, nBig) take the 512-thread variant, the rest the 256-thread one */
        int nBig = 0, capMax = 0;
        for (int l = 0; l < nl; l++) {
            if (g.lv[l].w * g.lv[l].h > 160000 || g.lv[l].kpCap > 256) nBig = l + 1;
            capMax = std::max(capMax, g.lv[l].kpCap);
        }
        /* a batch whose workgroups are all resident at once (2 x 512 threads per CU) finishes soonest as ONE
         * launch: the kernel is latency-bound and a second launch would only queue behind the first */
        if (nl * nframes <= 2 * 256) nBig = nl;
        if (nBig > 0) {
            if (capMax > 256)
                hipLaunchKernelGGL((k_quadtree<512, 16, DRFE_QT_MAX_NODES>), dim3(nBig, nframes), dim3(512), 0, s, c->d_geom, 0,
                                   c->d_cand0, c->d_cand1, c->d_node, c->d_candCount, c->d_sel, c->d_selCount, c->d_status);
            else
                hipLaunchKernelGGL((k_quadtree<512, 16, 256>), dim3(nBig, nframes), dim3(512), 0, s, c->d_geom, 0, c->d_cand0,
                                   c->d_cand1, c->d_node, c->d_candCount, c->d_sel, c->d_selCount, c->d_status);
        }
        if (nBig < nl)
            hipLaunchKernelGGL((k_quadtree<256, 8, 256>), dim3(nl - nBig, nframes), dim3(256), 0, s, c->d_geom, nBig, c->d_cand0,
                               c->d_cand1, c->d_node, c->d_candCount, c->d_sel, c->d_selCount, c->d_status);
    }
    prof_end(c, DRFE_STAGE_QUADTREE, s);

    prof_begin(c, DRFE_STAGE_BLUR, s);
    hipLaunchKernelGGL(k_blur, dim3(g.totalTiles, nframes), dim3(256), 0, s, c->d_geom, c->d_tiles, c->d_pyr,
                       c->d_blur, drfe_div_magic((uint32_t)g.totalTiles));
    prof_end(c, DRFE_STAGE_BLUR, s);

    prof_begin(c, DRFE_STAGE_DESC, s);
    hipLaunchKernelGGL(k_orient_desc, dim3((c->maxKp + DESC_WAVES * DESC_KPW - 1) / (DESC_WAVES * DESC_KPW), nframes), dim3(DESC_THREADS), 0, s, c->d_geom, c->d_pyr,
                       c->d_blur, c->d_sel, c->d_selCount, c->d_pattern, c->d_disc, c->discCount, c->d_kps,
                       c->d_desc, c->d_kpCount, c->maxKp, drfe_div_magic((uint32_t)((c->maxKp + DESC_WAVES * DESC_KPW - 1) / (DESC_WAVES * DESC_KPW))));
    prof_end(c, DRFE_STAGE_DESC, s);
    return hipGetLastError();
}

size_t drfe_quadtree_lds_bytes() { return sizeof(QtShared<512, DRFE_QT_MAX_NODES>); }
