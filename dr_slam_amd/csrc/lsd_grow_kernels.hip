/* lsd_grow_kernels.hip — the sequential core of cv::LineSegmentDetectorImpl on the device: seed loop, region_grow,
 * region2rect / get_theta, refine and reduce_region_radius (OpenCV 3.4 imgproc/src/lsd.cpp, the detector behind reference
 * src/LSDextractor.cpp:12-43; restated for the host in lines_lsd.cpp and for the oracle in oracle/lsd_oracle.cpp).
 *
 * The loop is order-defined: seeds are visited in the pseudo-ordering, a region claims pixels through a `used` map the
 * later seeds read, its running angle is a float sum in joining order, the rectangle moments are double sums in member
 * order.  One frame is therefore ONE WAVEFRONT that executes the reference's sequence exactly - wave-uniform control flow,
 * every order-defined value held identically by all lanes - and uses its 64 lanes where the sequence leaves room:
 *   - the seed scan reads 64 entries of the ordering at a time, two chunks ahead of the growth; entries whose pixel is already
 *     claimed are skipped by a ballot, and a seed none of whose free neighbours is aligned with its own angle is a one-pixel
 *     region whatever happens before its turn (the free set only shrinks), so a run of such seeds ahead of the first growing
 *     one is retired in one step;
 *   - a region grows inside the 7 x 7 window of its seed first (nineteen in twenty never leave it): one lane per window pixel,
 *     the window's free / aligned state as two scalar masks, nothing written until the region leaves the window or ends; beyond
 *     the window up to seven queued members per step, lane 9 m + k holding neighbour k of the step's m-th member, so that lane
 *     order is the visiting order and one pass over the lanes applies the step;
 *   - the alignment test |fastAtan2(sums) - angle(pixel)| <= tolerance is decided from the dot and cross product of the sums
 *     with the pixel's (cos, sin) wherever it is not within 0.02 degrees of the tolerance (fastAtan2's polynomial is within
 *     0.0096 degrees of the arctangent: align_class); only inside that band is the reference's arithmetic evaluated;
 *   - products, coordinate differences, rectangle extents and the alignment statistics are lane-parallel; the additions of
 *     the order-defined sums run one member after the other, three sums at once on three lanes walking LDS (ordered_sums3);
 *   - reduce_region_radius' swap-with-last removal is evaluated in closed form (the k-th removed position below the new
 *     size receives the k-th kept member from the back).
 * The `used` map is a bitmap in LDS (24 KB at 512 x 384); member lists live in HBM with the newest 128 entries mirrored in
 * LDS for the growth frontier - 26 624 bytes per frame, six frames per CU.  Throughput comes from frames in flight: a frame is
 * a dependency chain of ~10^5 steps, a launch carries one wavefront per frame and a CU holds as many as its LDS allows.
 *
 * cos / sin of region2rect and of the seed direction come from cr_sincos.h (correctly rounded, the same routine on the host
 * path); a frame whose rounding cannot be certified, or whose rectangle list overflows, is flagged and redone on the host. */
#include "drfe_internal.h"
#include "lines_internal.h"
#include "../../include/drfe_math.h"
#include "cr_sincos.h"
#include <type_traits>
#include <cmath>
#include <cstdlib>

#ifndef LSD_RING
#define LSD_RING 128          /* newest members mirrored in LDS.  With 128 a 512 x 384 frame needs 26 624 bytes of LDS and SIX frames
                               * fit a CU (the allocation granule is 1280 bytes: 512 entries = 28 160 bytes = five per CU, and so
                               * is 256); one call of 3072 frames: 6 195 -> 7 062 frames/s */
#endif
#ifdef LSD_PROFILE
#define PROF_T() wall_clock64()
#define PROF_ADD(k, t0) w.prof[k] += wall_clock64() - (t0)
#define PROF_CNT(k, v) w.prof[k] += (v)
#else
#define PROF_T() 0ull
#define PROF_ADD(k, t0) (void)(t0)
#define PROF_CNT(k, v) (void)0
#endif

namespace {

__device__ __forceinline__ int rl_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t rl_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float rl_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double rl_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
/* a wave-uniform condition as a scalar: keeps the control flow on the scalar unit and EXEC full */
__device__ __forceinline__ bool uni(bool c) { return __builtin_amdgcn_readfirstlane((int)c) != 0; }
__device__ __forceinline__ double uni_d(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uni_u64(unsigned long long v)
{
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
/* between a store and a load of the same address by different lanes of THIS wavefront (the only one of its workgroup): a
 * wavefront's memory operations are performed in order, so the compiler must keep the order and the hardware has nothing to wait for */
__device__ __forceinline__ void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

/* the frame's arrays as GLOBAL-address-space pointers: pointers read from the frame table are generic to the compiler, and a
 * generic (flat) store counts against the LDS counter as well - every member appended would stall the next bitmap read for a
 * full HBM round trip */
#define GLOBAL_AS __attribute__((address_space(1)))
struct FramePtrs {
    const GLOBAL_AS double* ang; const GLOBAL_AS float* cs; const GLOBAL_AS float* cs0; const GLOBAL_AS double* mod; const GLOBAL_AS uint32_t* order;
    GLOBAL_AS uint32_t* reg; GLOBAL_AS uint32_t* tmp; GLOBAL_AS double* rects; GLOBAL_AS int* out;
    int nOrder; uint32_t minSeedBin;
};

/* Overlay of one seed's processing in the multi-wave kernel (k_lsd_grow_mw, below): the pixels the seed being processed has
 * claimed so far - and the ones it claimed and released again - kept apart from the shared bitmap until the seed's turn to commit
 * comes.  An open-addressing table in LDS: entry = pixel index | OV_PRESENT while claimed (a released pixel keeps its entry: the
 * table is also the set J of every pixel the seed ever joined, which is what the commit validates). */
#define OV_SLOTS 128
#define OV_EMPTY 0xFFFFFFFFu
#define OV_PRESENT 0x40000000u
#define OV_PIX 0x3FFFFFFFu
#define OV_MAXLOAD 96                   /* entries beyond which a speculation gives up (99 regions in 100 join fewer pixels; the rest is processed at its turn with
                                         * the frame's overlay bitmap in HBM) */
#define LSD_STATUS_SPEC_OVERFLOW 0x100  /* internal to k_lsd_grow_mw: the overlay (or a wave's share of the member list) ran out */

struct Wave {
    FramePtrs F;
    uint32_t* bm;             /* LDS: bit set = pixel cannot join (claimed, or no level-line angle) */
    uint32_t* ring;           /* LDS: reg[j] for the newest LSD_RING members at ring[j & (LSD_RING - 1)] */
    double* col;              /* LDS: 64 x 3 doubles, the addends of region2rect's order-defined sums (ordered_sums3) */
    int W, H, lane;
    int status;
    /* multi-wave kernel only */
    uint32_t* ov;             /* LDS: this wave's overlay table (OV_SLOTS entries) */
    int ovl;                  /* 1: claims / releases go to the overlay table, `used` = bitmap | table; 2: to the frame's overlay BITMAP in HBM
                               * (regions too large for a table, processed at their turn); 0: straight to the bitmap (single-wave kernel) */
    GLOBAL_AS uint32_t* gbm;  /* the overlay bitmap in HBM (all zero between uses) */
    int ovCount;              /* entries in the overlay (wave-uniform) */
    int regCap;               /* entries of F.reg / F.tmp this wave owns */
    unsigned long long smallMask;   /* out of grow(): the window cells of a region that stayed small (nothing claimed) */
    int small;
#ifdef LSD_PROFILE
    unsigned long long prof[16];
#endif

    __device__ __forceinline__ bool bit(uint32_t q) const { return (bm[q >> 5] >> (q & 31)) & 1u; }
    __device__ __forceinline__ void set_bit_uniform(uint32_t q)           /* q wave-uniform */
    {
        if (lane == 0) atomicOr(&bm[q >> 5], 1u << (q & 31));      /* ds_or_b32 without return: nothing to wait for */
    }
    __device__ __forceinline__ uint32_t ov_slot(uint32_t q) const { return (q * 2654435761u) >> 25; }
    __device__ __forceinline__ bool ov_has(uint32_t q) const
    {
        uint32_t s = ov_slot(q);
        for (int k = 0; k < OV_SLOTS; k++) {
            const uint32_t e = ov[s];
            if (e == OV_EMPTY) return false;
            if ((e & OV_PIX) == q) return (e & OV_PRESENT) != 0;
            s = (s + 1) & (OV_SLOTS - 1);
        }
        return false;
    }
    /* per lane; returns true when a new entry was made */
    __device__ __forceinline__ bool ov_add(uint32_t q)
    {
        uint32_t s = ov_slot(q);
        for (int k = 0; k < OV_SLOTS; k++) {
            const uint32_t old = atomicCAS(&ov[s], OV_EMPTY, q | OV_PRESENT);
            if (old == OV_EMPTY) return true;
            if ((old & OV_PIX) == q) { atomicOr(&ov[s], OV_PRESENT); return false; }
            s = (s + 1) & (OV_SLOTS - 1);
        }
        return true;                                  /* full: the count below trips the overflow */
    }
    __device__ __forceinline__ void ov_drop(uint32_t q)
    {
        uint32_t s = ov_slot(q);
        for (int k = 0; k < OV_SLOTS; k++) {
            const uint32_t e = ov[s];
            if (e == OV_EMPTY) return;
            if ((e & OV_PIX) == q) { atomicAnd(&ov[s], ~OV_PRESENT); return; }
            s = (s + 1) & (OV_SLOTS - 1);
        }
    }
    /* can pixel q not join?  MW: the multi-wave kernel's form */
    template <bool MW> __device__ __forceinline__ bool used(uint32_t q) const
    {
        if (MW) {
            const bool b = bit(q);
            if (ovl == 2) return b || ((__hip_atomic_load(&gbm[q >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (q & 31)) & 1u);
            return b || (ovl && ov_has(q));
        }
        return bit(q);
    }
    /* the lanes for which `mine` holds claim their pixel q (called by all lanes of the wavefront) */
    template <bool MW> __device__ __forceinline__ void claim(bool mine, uint32_t q)
    {
        if (MW && ovl == 2) {
            /* returning atomics: the claim has reached L2 when the wave goes on, and the probes (agent-scope loads) read it there */
            if (mine) (void)__hip_atomic_fetch_or(&gbm[q >> 5], 1u << (q & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MW && ovl) {
            bool fresh = false;
            if (mine) fresh = ov_add(q);
            ovCount += __popcll(__ballot(fresh));
            if (ovCount > OV_MAXLOAD) status |= LSD_STATUS_SPEC_OVERFLOW;
        } else if (mine) atomicOr(&bm[q >> 5], 1u << (q & 31));
    }
    template <bool MW> __device__ __forceinline__ void release(bool mine, uint32_t q)
    {
        if (MW && ovl == 2) { if (mine) (void)__hip_atomic_fetch_and(&gbm[q >> 5], ~(1u << (q & 31)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else if (MW && ovl) { if (mine) ov_drop(q); }
        else if (mine) atomicAnd(&bm[q >> 5], ~(1u << (q & 31)));
    }
};

/* the two libm replacements as real functions: one copy each in the code object (inlined at their seven call sites they
 * pushed the kernel past the 64 KB instruction cache) */
struct SinCos { double s, c; int ok; };
__device__ __noinline__ SinCos cr_sincos_call(double x)
{
    SinCos r;
    r.ok = drfe_cr_sincos(x, &r.s, &r.c);
    return r;
}

__device__ __forceinline__ bool aligned_with(double theta, double a, double prec)
{
    double n = theta - a;
    if (n < 0) n = -n;
    if (n > 3.0 * 3.14159265358979323846 / 2.0) { n -= 2.0 * 3.14159265358979323846; if (n < 0) n = -n; }
    return n <= prec;
}

/* The 7 x 7 neighbourhood of a seed, one pixel per lane (lane t < 49 holds pixel (sx - 3 + t % 7, sy - 3 + t / 7)): level-line
 * angle and (cos, sin), fetched once per seed - and for the next seeds of the scan ahead of their turn, the fields never
 * change.  A region that stays inside it (nineteen in twenty do) grows without touching memory again: the neighbourhood's
 * state is a 49-bit scalar mask. */
struct Window { double a; float2 c; float2 s0; };      /* s0 (centre lane only): the seed's own direction, k_lsd_keys' field */

__device__ __forceinline__ Window load_window(const Wave& w, int sx, int sy)
{
    Window win;
    win.a = 0.0; win.c = make_float2(0.f, 0.f); win.s0 = make_float2(0.f, 0.f);
    const int t = w.lane, wy = (t * 37) >> 8, wx = t - 7 * wy;
    const int px = sx - 3 + wx, py = sy - 3 + wy;
    if (t < 49 && px >= 0 && py >= 0 && px < w.W && py < w.H) {
        const size_t q = (size_t)py * w.W + px;
        win.a = w.F.ang[q]; win.c = make_float2(w.F.cs[2 * q], w.F.cs[2 * q + 1]);
        if (t == 24) win.s0 = make_float2(w.F.cs0[2 * q], w.F.cs0[2 * q + 1]);
    }
    return win;
}

/* The alignment test of region_grow, |fastAtan2(sumdy, sumdx) * DEG2RAD - angle(pixel)| <= prec (mod 2 pi), decided WITHOUT the
 * arctangent wherever the decision is not close: the pixel's direction (c.x, c.y) is float(cos), float(sin) of its level-line
 * angle (k_lsd_gradient's field, correctly rounded), so the angle between the running direction S = (sumdx, sumdy) and the
 * pixel is atan2(cross, dot) of the two; cv::fastAtan2's polynomial is within 0.0096 degrees of the true arctangent (every
 * float ratio in [0, 1] scanned; its three `90 - a` style reflections add 5e-5) and the float roundings of dot / cross are
 * 1e-5 degrees, so with tLo = tan(prec - 0.02 deg), tHi = tan(prec + 0.02 deg)
 *     |cross| <= tLo * dot            : aligned, whatever the last bits of fastAtan2 are
 *     |cross| >= tHi * dot or dot <= 0: not aligned
 * and only a pixel in the 0.04-degree band between the two (2e-4 of them) is decided by the reference's own arithmetic.  A
 * running direction that nearly cancelled, or a NaN (seed direction not certified), is "uncertain" as well. */
struct AlignTan { float tLo, tHi; };
/* masks over the wavefront: lanes whose pixel is certainly aligned / not decided by the shortcut.  Plain compares into lane
 * masks and scalar mask arithmetic: no lane-divergent control flow */
__device__ __forceinline__ void align_class(float Sx, float Sy, float2 c, const AlignTan& T, unsigned long long& in, unsigned long long& unc)
{
    const float dot = __builtin_fmaf(Sx, c.x, Sy * c.y), crs = __builtin_fmaf(Sx, c.y, -(Sy * c.x));
    const float ac = fabsf(crs);
    const unsigned long long ok = __ballot(fabsf(dot) + ac > 1e-3f);
    /* dot <= 0 needs no test of its own: tHi > 0, so tHi * dot <= 0 <= |cross| and the pixel is "out" already */
    const unsigned long long mIn = __ballot(ac <= T.tLo * dot), mOut = __ballot(ac >= T.tHi * dot);
    in = ok & mIn;
    unc = ~(in | (ok & mOut));
}

/* the float sums of region_grow (before the first join: the seed's own direction - its sums start there) */
struct RegDir { float sx, sy; };

/* region_grow from seed (sx, sy): members to F.reg[0..n), their pixels claimed in the bitmap.  Returns n; regAngle out.
 * win = load_window(sx, sy).  CHEAP: the alignment tests go through align_class first (the caller's prec is the one T was
 * made for); refine's second growth runs with its own tolerance and the reference's arithmetic throughout. */
template <bool CHEAP, bool MW = false>
__device__ __forceinline__ int grow(Wave& w, int sx, int sy, double prec, double& regAngleOut, const Window& win, const AlignTan T, int minKeep)
{
    const int W = w.W, H = w.H, lane = w.lane;
    const double kDeg2Rad = 3.14159265358979323846 / 180.0;
    const double seedAngle = rl_f64(win.a, 24);
    /* float(cos), float(sin) of the seed's angle: correctly rounded per pixel by k_lsd_keys (NaN: not certified) */
    RegDir D;
    D.sx = rl_f32(win.s0.x, 24); D.sy = rl_f32(win.s0.y, 24);
    int n = 1, i = 0;
    /* the region's angle: the seed's own until a second member joins, then fastAtan2 of the sums - evaluated when a test needs
     * it (angleAt = the size of the region the cached value belongs to) */
    double curAngle = seedAngle;
    int angleAt = 1;
    auto angle_now = [&]() -> double {
        if (angleAt != n) { curAngle = (double)drfe_fast_atan2(D.sy, D.sx) * kDeg2Rad; angleAt = n; }
        return curAngle;
    };
    auto join_dir = [&](float cx, float cy) { D.sx += cx; D.sy += cy; };
    const bool shortcut = CHEAP && T.tLo > 0.f;
    const unsigned long long tg0 = PROF_T();
    PROF_CNT(10, 1);
    {
        /* inside the window: state of its pixels as a scalar mask, members as window indices in the lanes of `member`.  Nothing
         * is written while the region stays here: the window's free mask is the only state the growth reads, and the members'
         * bitmap bits (and, for a region that goes on, their list entries) are stored by their lanes at the end, side by side */
        const int wy0 = (lane * 37) >> 8, wx0 = lane - 7 * wy0;
        const int px = sx - 3 + wx0, py = sy - 3 + wy0;
        const bool inw = lane < 49 && px >= 0 && py >= 0 && px < W && py < H;
        unsigned long long wfree = __ballot(inw && !w.template used<MW>((uint32_t)(py * W + px))) & ~(1ull << 24);
        const unsigned long long wfree0 = wfree;
        /* which window pixels are aligned with the running angle: every lane tests its own pixel, so a member's probes are
         * scalar mask arithmetic and only a JOIN (which moves the angle) costs vector work */
        auto aligned_mask = [&]() -> unsigned long long {
            if (shortcut) {
                unsigned long long mi, mu;
                align_class(D.sx, D.sy, win.c, T, mi, mu);
                if (!(mu & wfree)) return mi;
                PROF_CNT(14, 1);
            }
            return __ballot(lane < 49 && aligned_with(angle_now(), win.a, prec));
        };
        unsigned long long walign = aligned_mask();
        int member = 24;                               /* lane j: window index of member j */
        while (i < n) {
            const int t = rl_i32(member, i);
            if (!((0x1F3E7CF9F00ull >> t) & 1ull)) break;             /* not one of the inner 5 x 5: its neighbours leave the window */
            unsigned long long nb = 0x1C287ull << (t - 8);                /* the 3 x 3 ring around t, raster order */
            for (;;) {
                const unsigned long long hit = nb & wfree & walign;       /* free neighbours that join now */
                if (!hit) break;
                const int l = __builtin_ctzll(hit);
                nb &= ~((2ull << l) - 1ull);                              /* the probes behind it come after the join */
                wfree &= ~(1ull << l);
                member = lane == n ? l : member;
                n++;
                join_dir(rl_f32(win.c.x, l), rl_f32(win.c.y, l));
                walign = aligned_mask();
            }
            i++;
        }
        if (MW) {
            /* the multi-wave kernel: a region that ended inside the window below the size that matters claims NOTHING here - its
             * cells go out as a mask (the cells that were free and are not any more, and the seed) and the commit writes them */
            w.small = (i >= n && n < minKeep) ? 1 : 0;
            w.smallMask = (wfree0 & ~wfree) | (1ull << 24);
        }
        if (!MW || !w.small) {
            const int my = (member * 37) >> 8, mx = member - 7 * my;
            const int jx = sx - 3 + mx, jy = sy - 3 + my;
            const uint32_t xy = (uint32_t)(jy << 16 | jx), q = (uint32_t)(jy * W + jx);
            w.template claim<MW>(lane < n, q);
            /* the list is read by the growth beyond the window and by region2rect: nineteen regions in twenty need neither */
            if (lane < n && (i < n || n >= minKeep)) { w.F.reg[lane] = xy; w.ring[lane] = xy; }
        }
    }
    PROF_ADD(3, tg0);
    PROF_CNT(11, i);
    const unsigned long long tg1 = PROF_T();
    /* beyond the window: up to seven queued members per step, their neighbours' fields fetched together.  Lane 9 m + k holds
     * neighbour k (raster order) of the step's m-th member, so LANE ORDER IS THE VISITING ORDER of region_grow: one pass over
     * the lanes whose pixel is free, a join removes every lane that holds the joined pixel and moves the direction the later
     * lanes are tested against.  The joined lanes store their list entries and bitmap bits together when the step ends. */
    const int m = lane / 9, k = lane - 9 * m;
    const int dyk = k / 3 - 1, dxk = k - 3 * (k / 3) - 1;
    const unsigned long long lt = (1ull << lane) - 1ull;
    while (i < n) {
        if (MW && uni((w.status & LSD_STATUS_SPEC_OVERFLOW) != 0 || n + 64 > w.regCap)) { w.status |= LSD_STATUS_SPEC_OVERFLOW; break; }
#ifdef LSD_PROFILE
        const unsigned long long tq0 = PROF_T();
#endif
        const int cnt = min(7, n - i);
        const bool act = m < cnt && lane < 63 && k != 4;
        uint32_t mxy = 0;
        if (uni(n - i <= LSD_RING)) { if (act) mxy = w.ring[(i + m) & (LSD_RING - 1)]; }
        else { wg_fence(); if (act) mxy = w.F.reg[i + m]; }
        const int nx = (int)(mxy & 0xFFFFu) + dxk, ny = (int)(mxy >> 16) + dyk;
        const bool inb = act && nx >= 0 && ny >= 0 && nx < W && ny < H;
        const uint32_t q = inb ? (uint32_t)(ny * W + nx) : 0xFFFFFFFFu;
        const bool want = inb && !w.template used<MW>(q);
        double a = 0.0;
        float2 c = make_float2(0.f, 0.f);
        if (want) { c = make_float2(w.F.cs[2 * (size_t)q], w.F.cs[2 * (size_t)q + 1]); if (!CHEAP) a = w.F.ang[q]; }
        unsigned long long qin = 0, qunc = ~0ull;
        auto classes = [&]() {
            if (!shortcut) return;                         /* the shortcut is off for this tolerance: every test the long way */
            align_class(D.sx, D.sy, c, T, qin, qunc);
        };
#ifdef LSD_PROFILE
        if (CHEAP) { if (__ballot(c.x == 12345.678f)) w.status |= 4; PROF_ADD(15, tq0); }      /* wait for the fields here */
#endif
        classes();
        unsigned long long mask = __ballot(want), joined = 0;
        for (;;) {
            /* the next lane whose pixel joins, or may: the free ones before it are certainly not aligned with the direction as
             * it stands, and it only moves at a join - they have had their turn */
            const unsigned long long todo = mask & (qin | qunc);
            if (!todo) break;
            const int l = __builtin_ctzll(todo);
            mask &= ~((2ull << l) - 1ull);
            bool al = true;
            if ((qunc >> l) & 1ull) {
                if (CHEAP) PROF_CNT(14, 1);
                const double av = CHEAP ? uni_d(w.F.ang[rl_u32(q, l)]) : rl_f64(a, l);
                al = uni(aligned_with(angle_now(), av, prec));
            }
            if (al) {
                joined |= 1ull << l;
                mask &= ~__ballot(q == rl_u32(q, l));      /* the same pixel as a later member's neighbour: claimed now */
                n++;
                join_dir(rl_f32(c.x, l), rl_f32(c.y, l));
                classes();
            }
        }
        if ((joined >> lane) & 1ull) {
            const int at = n - __popcll(joined) + __popcll(joined & lt);
            const uint32_t xy = (uint32_t)(ny << 16 | nx);
            w.F.reg[at] = xy; w.ring[at & (LSD_RING - 1)] = xy;
        }
        w.template claim<MW>((joined >> lane) & 1ull, q);
        i += cnt;
        PROF_CNT(12, 1);
    }
    PROF_ADD(4, tg1);
    /* only regions that get a second member need the seed's direction */
    if (n > 1 && rl_f32(win.s0.x, 24) != rl_f32(win.s0.x, 24)) w.status |= DRFE_LSD_STATUS_UNCERTAIN;
    /* the region's angle is region2rect's input: a region below the size its caller keeps (eighteen in twenty) never gets there, and
     * its fastAtan2 - a division and a polynomial on the chain - is not computed */
    regAngleOut = n >= minKeep ? angle_now() : 0.0;
    return n;
}


/* max (MAX) or min of a double over the wavefront, to every lane */
template <bool MAX>
__device__ __forceinline__ double wave_ext_f64(double v)
{
    auto step = [&](auto ctrl) {
        constexpr int C = decltype(ctrl)::value;
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), C, 0xF, 0xF, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), C, 0xF, 0xF, false);
        const double o = __hiloint2double(hi, lo);
        v = MAX ? fmax(v, o) : fmin(v, o);
    };
    step(std::integral_constant<int, 0xB1>());      /* quad_perm [1, 0, 3, 2] */
    step(std::integral_constant<int, 0x4E>());      /* quad_perm [2, 3, 0, 1] */
    step(std::integral_constant<int, 0x141>());     /* row_half_mirror */
    step(std::integral_constant<int, 0x140>());     /* row_mirror */
    const double r0 = rl_f64(v, 0), r1 = rl_f64(v, 16), r2 = rl_f64(v, 32), r3 = rl_f64(v, 48);
    return MAX ? fmax(fmax(r0, r1), fmax(r2, r3)) : fmin(fmin(r0, r1), fmin(r2, r3));
}

struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy; };

__device__ __forceinline__ double sq(double v) { return v * v; }
__device__ __forceinline__ double dist2d(double x1, double y1, double x2, double y2) { return sqrt(sq(x2 - x1) + sq(y2 - y1)); }
__device__ __forceinline__ double diff_signed(double a, double b)
{
    double d = a - b;
    while (d <= -3.14159265358979323846) d += 2.0 * 3.14159265358979323846;
    while (d > 3.14159265358979323846) d -= 2.0 * 3.14159265358979323846;
    return d;
}

/* acc (lane 0 / 1 / 2: three running sums) += v0 / v1 / v2 of lanes 0 .. cnt - 1, in lane order.  Lanes >= cnt must hold +0.0
 * (adding it changes nothing: the sums never hold -0.0, they start at +0.0).  Lanes above 2 walk lane 2's column and are ignored. */
__device__ __forceinline__ double ordered_sums3(Wave& w, double acc, double v0, double v1, double v2, int cnt)
{
    double* mine = w.col + 3 * w.lane;
    mine[0] = v0; mine[1] = v1; mine[2] = v2;
    const double* walk = w.col + min(w.lane, 2);
    for (int t = 0; t < cnt; t += 8) {
        const double e0 = walk[3 * t], e1 = walk[3 * t + 3], e2 = walk[3 * t + 6], e3 = walk[3 * t + 9], e4 = walk[3 * t + 12], e5 = walk[3 * t + 15],
                     e6 = walk[3 * t + 18], e7 = walk[3 * t + 21];
        acc += e0; acc += e1; acc += e2; acc += e3; acc += e4; acc += e5; acc += e6; acc += e7;
    }
    return acc;
}

/* region2rect (with get_theta) over F.reg[0..n): sums in member order, extents by wave reduction.  fromRing: the region
 * was just grown and has at most LSD_RING members, so the LDS mirror holds all of them (not after reduce_region_radius, which
 * reorders the list in HBM only).  A region of at most 64 members is fetched once and stays in registers for the three passes. */
__device__ __forceinline__ void to_rect(Wave& w, int n, double regAngle, double prec, Rect& rec, bool fromRing)
{
    const int lane = w.lane, W = w.W;
    const double kDeg2Rad = 3.14159265358979323846 / 180.0;
    const unsigned long long tr0 = PROF_T();
    PROF_CNT(13, 1);
    const bool ring = fromRing && n <= LSD_RING, cached = n <= 64;
    if (!ring) wg_fence();
    int cmx = 0, cmy = 0;
    double cmg = 0;
    if (cached && lane < n) {
        const uint32_t xy = ring ? w.ring[lane] : w.F.reg[lane];
        cmx = (int)(xy & 0xFFFFu); cmy = (int)(xy >> 16);
        cmg = w.F.mod[(size_t)cmy * W + cmx];
    }
    /* the order-defined sums: x += px[j], y += py[j], sum += mg[j] for j = 0 .. n - 1.  Three chains that do not touch each
     * other: lanes 0, 1, 2 each walk one of them through LDS (one ds_read + one add per member for all three, against six
     * lane broadcasts + three adds on the scalar path) */
    double acc = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane, cnt = min(64, n - base);
        double px = 0, py = 0, mg = 0;
        if (j < n) {
            int mx = cmx, my = cmy;
            mg = cmg;
            if (!cached) {
                const uint32_t xy = ring ? w.ring[j & (LSD_RING - 1)] : w.F.reg[j];
                mx = (int)(xy & 0xFFFFu); my = (int)(xy >> 16);
                mg = w.F.mod[(size_t)my * W + mx];
            }
            px = (double)mx * mg; py = (double)my * mg;
        }
        acc = ordered_sums3(w, acc, px, py, mg, cnt);
    }
    double x = rl_f64(acc, 0), y = rl_f64(acc, 1);
    const double sum = rl_f64(acc, 2);
    x /= sum; y /= sum;
    acc = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane, cnt = min(64, n - base);
        double a = 0, b = 0, c = 0;
        if (j < n) {
            int mx = cmx, my = cmy;
            double mg = cmg;
            if (!cached) {
                const uint32_t xy = ring ? w.ring[j & (LSD_RING - 1)] : w.F.reg[j];
                mx = (int)(xy & 0xFFFFu); my = (int)(xy >> 16);
                mg = w.F.mod[(size_t)my * W + mx];
            }
            const double dx = (double)mx - x, dy = (double)my - y;
            a = dy * dy * mg; b = dx * dx * mg; c = dx * dy * mg;
        }
        acc = ordered_sums3(w, acc, a, b, -c, cnt);                 /* Ixy -= c: adding -c is the same operation */
    }
    const double Ixx = rl_f64(acc, 0), Iyy = rl_f64(acc, 1), Ixy = rl_f64(acc, 2);
    const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)drfe_fast_atan2((float)(lambda - Ixx), (float)Ixy)
                                           : (double)drfe_fast_atan2((float)Ixy, (float)(lambda - Iyy));
    theta *= kDeg2Rad;
    if (fabs(diff_signed(theta, regAngle)) > prec) theta += 3.14159265358979323846;
    theta = uni_d(theta);
    const SinCos sc = cr_sincos_call(theta);
    const double dx = sc.c, dy = sc.s;
    if (!sc.ok) w.status |= DRFE_LSD_STATUS_UNCERTAIN;
    double lmin = 0, lmax = 0, wmin = 0, wmax = 0;
    for (int j = lane; j < n; j += 64) {
        int mx = cmx, my = cmy;
        if (!cached) {
            const uint32_t xy = ring ? w.ring[j & (LSD_RING - 1)] : w.F.reg[j];
            mx = (int)(xy & 0xFFFFu); my = (int)(xy >> 16);
        }
        const double rx = (double)mx - x, ry = (double)my - y;
        const double l = rx * dx + ry * dy, ww = -rx * dy + ry * dx;
        lmax = fmax(lmax, l); lmin = fmin(lmin, l);
        wmax = fmax(wmax, ww); wmin = fmin(wmin, ww);
    }
    /* wave-wide extremes: inside a row of 16 lanes by four DPP exchanges (the neighbour, the other pair, the mirrored half, the mirrored
     * row - max / min do not mind meeting a value twice), the four rows by lane broadcasts; a shuffle butterfly goes through the LDS
     * crossbar twelve times per value */
    lmax = wave_ext_f64<true>(lmax); lmin = wave_ext_f64<false>(lmin);
    wmax = wave_ext_f64<true>(wmax); wmin = wave_ext_f64<false>(wmin);
    rec.x1 = x + lmin * dx; rec.y1 = y + lmin * dy; rec.x2 = x + lmax * dx; rec.y2 = y + lmax * dy;
    rec.width = wmax - wmin;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy;
    if (rec.width < 1.0) rec.width = 1.0;
    PROF_ADD(5, tr0);
}

__device__ __forceinline__ double density_of(const Rect& rec, int n)
{
    return (double)n / (dist2d(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
}

/* reduce_region_radius: members farther than the shrinking radius from the seed are released and removed by
 * swap-with-last; the surviving order (it feeds the next region2rect's sums) in closed form */
template <bool MW = false>
__device__ __forceinline__ bool shrink(Wave& w, int& n, double regAngle, double prec, Rect& rec, double density, double densityTh)
{
    const int lane = w.lane, W = w.W;
    wg_fence();
    const uint32_t xy0 = w.F.reg[0];
    const double xc = (double)(int)(xy0 & 0xFFFFu), yc = (double)(int)(xy0 >> 16);
    const double r1 = sq(rec.x1 - xc) + sq(rec.y1 - yc), r2 = sq(rec.x2 - xc) + sq(rec.y2 - yc);
    double radSq = r1 > r2 ? r1 : r2;
    while (uni(density < densityTh)) {
        radSq *= 0.75 * 0.75;
        /* members kept; the removed ones are released */
        int K = 0;
        for (int base = 0; base < n; base += 64) {
            const int j = base + lane;
            bool good = false;
            if (j < n) {
                const uint32_t xy = w.F.reg[j];
                const int mx = (int)(xy & 0xFFFFu), my = (int)(xy >> 16);
                good = !(sq((double)mx - xc) + sq((double)my - yc) > radSq);
                if (!good) { const uint32_t q = (uint32_t)(my * W + mx); w.template release<MW>(true, q); }
            }
            K += __popcll(__ballot(good));
        }
        if (K < n) {
            /* kept members at positions >= K, from the back */
            int cf = 0;
            for (int base = ((n - 1) >> 6) << 6; base >= 0 && base + 63 >= K; base -= 64) {
                const int j = base + lane;
                bool f = false;
                uint32_t xy = 0;
                if (j >= K && j < n) {
                    xy = w.F.reg[j];
                    f = !(sq((double)(int)(xy & 0xFFFFu) - xc) + sq((double)(int)(xy >> 16) - yc) > radSq);
                }
                const unsigned long long mk = __ballot(f);
                if (f) w.F.tmp[cf + __popcll(lane < 63 ? (mk >> (lane + 1)) : 0ull)] = xy;
                cf += __popcll(mk);
            }
            wg_fence();
            /* removed positions below K, ascending, take them in that order */
            int cb = 0;
            for (int base = 0; base < K; base += 64) {
                const int j = base + lane;
                bool b = false;
                if (j < K) {
                    const uint32_t xy = w.F.reg[j];
                    b = sq((double)(int)(xy & 0xFFFFu) - xc) + sq((double)(int)(xy >> 16) - yc) > radSq;
                }
                const unsigned long long mk = __ballot(b);
                if (b) w.F.reg[j] = w.F.tmp[cb + __popcll(mk & ((1ull << lane) - 1ull))];
                cb += __popcll(mk);
            }
            n = K;
        }
        if (n < 2) return false;
        to_rect(w, n, regAngle, prec, rec, false);
        density = density_of(rec, n);
    }
    return true;
}

template <bool MW = false>
__device__ __forceinline__ bool refine(Wave& w, int& n, double& regAngle, double prec, Rect& rec, double densityTh, const Window& win, const AlignTan T,
                                       int sx, int sy)
{
    const int lane = w.lane, W = w.W;
    double density = density_of(rec, n);
    if (uni(density >= densityTh)) return true;
    wg_fence();
    /* the seed (member 0) and its angle are the caller's and the window's; the members of a region of at most LSD_RING come from
     * the LDS mirror (the region was grown a moment ago): no trip to HBM before the members' angles can be asked for */
    const double xc = (double)sx, yc = (double)sy, angC = rl_f64(win.a, 24);
    const bool fromRing = n <= LSD_RING;
    /* sum / ssum over the members within the rectangle's width of the seed, in member order: ordered_sums3 again (a member outside
     * adds +0.0, which changes neither sum: they start at +0.0 and a sum that started there never becomes -0.0) */
    double acc = 0;
    int cntIn = 0;
    for (int base = 0; base < n; base += 64) {
        const int j = base + lane;
        bool in = false;
        double d = 0, dd = 0;
        if (j < n) {
            const uint32_t xy = fromRing ? w.ring[j] : w.F.reg[j];
            const int mx = (int)(xy & 0xFFFFu), my = (int)(xy >> 16);
            const uint32_t q = (uint32_t)(my * W + mx);
            w.template release<MW>(true, q);
            if (dist2d(xc, yc, (double)mx, (double)my) < rec.width) {
                in = true;
                d = diff_signed(w.F.ang[q], angC);
                dd = d * d;
            }
        }
        cntIn += __popcll(__ballot(in));
        acc = ordered_sums3(w, acc, d, dd, 0.0, min(64, n - base));
    }
    const double sum = rl_f64(acc, 0), ssum = rl_f64(acc, 1);
    const double mean = sum / (double)cntIn;
    const double tau = 2.0 * sqrt((ssum - 2.0 * mean * sum) / (double)cntIn + mean * mean);
    /* the second growth runs with the tolerance tau: the shortcut's thresholds for it (tan to a relative 1e-6 is ample inside a
     * band of 0.02 degrees; a tolerance near 0 or 90 degrees switches the shortcut off) */
    AlignTan T2; T2.tLo = 0.f; T2.tHi = 0.f;
    const double tauU = uni_d(tau), band = 0.02 * 3.14159265358979323846 / 180.0;
    if (T.tLo > 0.f && tauU - band > 1e-3 && tauU + band < 1.5) {           /* T.tLo <= 0: the launch runs without the shortcut (DRFE_LSD_EXACT_ALIGN) */
        T2.tLo = (float)(tan(tauU - band) * (1.0 - 1e-5));
        T2.tHi = (float)(tan(tauU + band) * (1.0 + 1e-5));
    }
    n = grow<true, MW>(w, sx, sy, tauU, regAngle, win, T2, 0);
    if (MW && uni((w.status & LSD_STATUS_SPEC_OVERFLOW) != 0)) return false;
    if (n < 2) return false;
    to_rect(w, n, regAngle, prec, rec, true);
    density = density_of(rec, n);
    if (uni(density < densityTh)) return shrink<MW>(w, n, regAngle, prec, rec, density, densityTh);
    return true;
}

} // namespace

#ifdef LSD_GROW_WAVES_PER_EU           /* experiment builds: cap the allocation (3 -> 168 VGPRs, 14 spilled) */
#define LSD_GROW_OCC __attribute__((amdgpu_waves_per_eu(LSD_GROW_WAVES_PER_EU, LSD_GROW_WAVES_PER_EU)))
#else
#define LSD_GROW_OCC
#endif
extern "C" __global__ __launch_bounds__(64) LSD_GROW_OCC void k_lsd_grow(const LsdGrowFrame* __restrict__ frames, int W, int H, double prec, double p,
                                                            int minReg, double densityTh, int rectCap, float tLo, float tHi)
{
    extern __shared__ uint32_t lds[];
    Wave w;
    AlignTan T; T.tLo = tLo; T.tHi = tHi;
    {
        const LsdGrowFrame f = frames[blockIdx.x];
        w.F.ang = (const GLOBAL_AS double*)f.ang; w.F.cs = (const GLOBAL_AS float*)f.cs; w.F.cs0 = (const GLOBAL_AS float*)f.cs0; w.F.mod = (const GLOBAL_AS double*)f.mod;
        w.F.order = (const GLOBAL_AS uint32_t*)f.order; w.F.reg = (GLOBAL_AS uint32_t*)f.reg; w.F.tmp = (GLOBAL_AS uint32_t*)f.tmp;
        w.F.rects = (GLOBAL_AS double*)f.rects; w.F.out = (GLOBAL_AS int*)f.out; w.F.nOrder = f.nOrder;
        w.F.minSeedBin = f.meta ? 1024u - (uint32_t)(f.meta[1] & 0xFFFFFFFFull) : f.minSeedBin;
    }
#ifdef LSD_SETPRIO
    __builtin_amdgcn_s_setprio(LSD_SETPRIO);
#endif
    w.W = W; w.H = H; w.lane = threadIdx.x; w.status = 0;
    w.ov = nullptr; w.ovl = 0; w.gbm = nullptr; w.ovCount = 0; w.regCap = 0x7fffffff; w.small = 0; w.smallMask = 0;
#ifdef LSD_PROFILE
    for (int k = 0; k < 16; k++) w.prof[k] = 0;
#endif
    const unsigned long long tAll = PROF_T();
    const int lane = w.lane, npx = W * H, nWords = (npx + 31) >> 5;
    w.bm = lds;
    w.ring = lds + ((nWords + 1) & ~1);
    w.col = (double*)(w.ring + LSD_RING);
    /* bitmap: pixels without a level-line angle never join (NOTDEF, incl. the last row and column).  k_lsd_notdef left the words
     * (a wide kernel over all frames: this wavefront would spend 1.4 ms of its 73 reading 1.5 MB of angles for them) */
    const GLOBAL_AS uint32_t* notdef = (const GLOBAL_AS uint32_t*)frames[blockIdx.x].notdef;
    if (notdef) { for (int k = lane; k < nWords; k += 64) w.bm[k] = notdef[k]; }
    else
    for (int base = 0; base < npx; base += 256) {
        bool nd[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int q = base + 64 * u + lane; nd[u] = q >= npx || w.F.ang[q] == -1024.0; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned long long b = __ballot(nd[u]);
            const int wd = (base >> 5) + 2 * u;
            if (lane == 0) { if (wd < nWords) w.bm[wd] = (uint32_t)b; if (wd + 1 < nWords) w.bm[wd + 1] = (uint32_t)(b >> 32); }
        }
    }
    PROF_ADD(0, tAll);
    int nRects = 0;
    const int nOrder = w.F.nOrder;
    const uint32_t minSeedBin = w.F.minSeedBin;
    /* The seed scan runs two chunks (of 64 seeds of the ordering) ahead of the growth: a chunk's keys are fetched two iterations
     * before its turn, its candidates' own and neighbouring level-line angles one iteration before - with the bitmap of THAT
     * moment, a superset of what is free at the chunk's turn: a seed no free neighbour of which is aligned with it then has none
     * later either (the free set only shrinks), one that loses its last aligned neighbour in between grows into a one-pixel
     * region - the same bit set -, and every seed is looked up again at its turn.  So neither fetch is waited for. */
    struct Scan { bool valid, cand; int sx, sy; uint32_t q, nfMask; double sa, na[8]; };
    auto scan_keys = [&](int base) -> uint32_t { const int pos = base + lane; return pos < nOrder ? w.F.order[pos] : 0u; };
    auto scan_fields = [&](int base, uint32_t key, Scan& sc) {
        sc.valid = base + lane < nOrder && (key >> 22) >= minSeedBin;
        sc.sx = (int)(key & 0x7FFu); sc.sy = (int)((key >> 11) & 0x7FFu);
        sc.q = (uint32_t)(sc.sy * W + sc.sx);
        sc.cand = sc.valid && !w.bit(sc.q);
        sc.nfMask = 0; sc.sa = 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++) sc.na[j] = 0.0;
        if (sc.cand) {
            sc.sa = w.F.ang[sc.q];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int jj = j < 4 ? j : j + 1, dx = jj % 3 - 1, dy = jj / 3 - 1;
                const int nx = sc.sx + dx, ny = sc.sy + dy;
                const bool inb = nx >= 0 && ny >= 0 && nx < W && ny < H;
                const uint32_t nq = inb ? (uint32_t)(ny * W + nx) : 0u;
                const bool nf = inb && !w.bit(nq);
                if (nf) { sc.nfMask |= 1u << j; sc.na[j] = w.F.ang[nq]; }
            }
        }
    };
    Scan cur, nxt;
    uint32_t keyNext;
    {
        const uint32_t k0 = scan_keys(0);
        keyNext = scan_keys(64);
        scan_fields(0, k0, cur);
    }
    /* does any free neighbour join on the seed's own angle?  (the scan's loads of that chunk must have arrived) */
    auto scan_nontrivial = [&](const Scan& sc) -> bool {
        bool nt = false;
        if (sc.cand) {
#pragma unroll
            for (int j = 0; j < 8; j++) nt |= ((sc.nfMask >> j) & 1u) && aligned_with(sc.sa, sc.na[j], prec);
        }
        return nt;
    };
    /* the next (up to) four growing seeds of a chunk: their windows fetched together.  Always four fetches, the unused ones at the
     * first seed again (cache hits): no branch around a load, so nothing merges with a value in flight and the wavefront waits
     * only where a window is used */
    auto take_group = [&](unsigned long long& pend, int sxv, int syv, uint32_t& packed, int& cnt, Window& a0, Window& a1, Window& a2, Window& a3) {
        int lk[4];
        cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            lk[u] = pend ? __builtin_ctzll(pend) : (u ? lk[0] : 0);
            if (pend) { cnt++; pend &= pend - 1; }
        }
        packed = (uint32_t)lk[0] | (uint32_t)lk[1] << 8 | (uint32_t)lk[2] << 16 | (uint32_t)lk[3] << 24;
        a0 = load_window(w, rl_i32(sxv, lk[0]), rl_i32(syv, lk[0]));
        a1 = load_window(w, rl_i32(sxv, lk[1]), rl_i32(syv, lk[1]));
        a2 = load_window(w, rl_i32(sxv, lk[2]), rl_i32(syv, lk[2]));
        a3 = load_window(w, rl_i32(sxv, lk[3]), rl_i32(syv, lk[3]));
    };
    /* A chunk's growing seeds go in groups of four, a group's windows fetched while the group before it grows (the fields never
     * change, so nothing a region does can invalidate them) - across the chunk boundary too: while a chunk's LAST group grows,
     * the first group of the next chunk is on its way (haveFirst: it is in w0..w3 already when that chunk's turn comes). */
    Window w0, w1, w2, w3, v0, v1, v2, v3;
    uint32_t packed = 0, packedNext = 0;
    int cnt = 0, cntNext = 0;
    bool haveFirst = false, nontrivial = false, nontrivialNext = false;
    unsigned long long pend = 0, pendNext = 0;
    bool done = false;
    for (int base = 0; base < nOrder && !done; base += 64) {
        const unsigned long long ts0 = PROF_T();
        PROF_CNT(8, 1);
        const uint32_t keyAfter = scan_keys(base + 128);
        scan_fields(base + 64, keyNext, nxt);
        if (__ballot(!cur.valid)) done = true;                 /* bins descend: nothing seeds from the first low bin on */
        const int sx = cur.sx, sy = cur.sy;
        const uint32_t q = cur.q;
        const bool cand = cur.cand;
        if (!haveFirst) {
            nontrivial = false; pend = 0; cnt = 0;
            if (__ballot(cand)) {
                nontrivial = scan_nontrivial(cur);
                pend = __ballot(cand && nontrivial);
                PROF_ADD(1, ts0);
                const unsigned long long tw0 = PROF_T();
                if (pend) take_group(pend, sx, sy, packed, cnt, w0, w1, w2, w3);
                PROF_ADD(2, tw0);
            } else PROF_ADD(1, ts0);
        } else PROF_ADD(1, ts0);
        int from = 0;
        bool nextTaken = false;
        while (cnt > 0) {
            const unsigned long long tw0 = PROF_T();
            PROF_CNT(9, 1);
            if (pend) take_group(pend, sx, sy, packedNext, cntNext, v0, v1, v2, v3);
            else {
                /* this is the chunk's last group: the next chunk's first one */
                nontrivialNext = scan_nontrivial(nxt);
                pendNext = __ballot(nxt.cand && nontrivialNext);
                take_group(pendNext, nxt.sx, nxt.sy, packedNext, cntNext, v0, v1, v2, v3);
                nextTaken = true;
            }
#ifdef LSD_PROFILE
            if (w0.a + w1.a + w2.a + w3.a == 12345.678) w.status |= 4;         /* wait for this group's loads here */
#endif
            PROF_ADD(2, tw0);
            for (int k = 0; k < cnt; k++) {
                const int f = (int)(packed & 0xFFu);
                packed >>= 8;
                /* one-pixel regions between the previous growing seed and this one */
                if (cand && !nontrivial && lane >= from && lane < f && !w.bit(q)) atomicOr(&w.bm[q >> 5], 1u << (q & 31));
                from = f + 1;
                const int gx = rl_i32(sx, f), gy = rl_i32(sy, f);
                if (!uni(w.bit((uint32_t)(gy * W + gx)))) {
                    double regAngle;
                    int n = grow<true>(w, gx, gy, prec, regAngle, w0, T, minReg);
                    if (n >= minReg) {
                        Rect rec;
                        to_rect(w, n, regAngle, prec, rec, true);
                        const unsigned long long tf0 = PROF_T();
                        const bool okr = refine(w, n, regAngle, prec, rec, densityTh, w0, T, gx, gy);
                        PROF_ADD(6, tf0);
                        if (okr) {
                            if (nRects < rectCap) {
                                if (lane == 0) {
                                    const double o[12] = {rec.x1, rec.y1, rec.x2, rec.y2, rec.width, rec.x, rec.y, rec.theta, rec.dx, rec.dy, prec, p};   /* LsdRect */
                                    for (int k = 0; k < 12; k++) w.F.rects[(size_t)nRects * 12 + k] = o[k];
                                }
                                nRects++;
                            } else w.status |= DRFE_LSD_STATUS_OVERFLOW;
                        }
                    }
                }
                w0 = w1; w1 = w2; w2 = w3;
            }
            w0 = v0; w1 = v1; w2 = v2; w3 = v3; packed = packedNext;
            cnt = nextTaken ? 0 : cntNext;                     /* a group of the next chunk waits for that chunk's turn */
        }
        if (cand && !nontrivial && lane >= from && !w.bit(q)) atomicOr(&w.bm[q >> 5], 1u << (q & 31));
        haveFirst = nextTaken;
        if (nextTaken) { cnt = cntNext; nontrivial = nontrivialNext; pend = pendNext; }
        cur = nxt; keyNext = keyAfter;
    }
    if (lane == 0) { w.F.out[DRFE_LSD_OUT_NEXT_RECT] = 0; w.F.out[0] = nRects; w.F.out[1] = w.status; w.F.out[2] = 0; w.F.out[3] = 0; /* k_rect_improve's status word and its reasons */ }
#ifdef LSD_PROFILE
    PROF_ADD(7, tAll);
    if (lane == 0) for (int k = 0; k < 16; k++) ((GLOBAL_AS unsigned long long*)(w.F.out + 4))[k] = w.prof[k];
#endif
}

/* ================================================================================================================================
 * k_lsd_grow_mw - the same seed loop with MORE THAN ONE WAVEFRONT PER FRAME (round 5).
 *
 * The loop is order-defined only through the `used` map: a seed's processing (region_grow, region2rect, refine, reduce_region_
 * radius) reads the map and the constant fields, and leaves its final members marked.  So the seeds of the ordering are processed
 * SPECULATIVELY by the four wavefronts of the frame's workgroup and COMMITTED IN SEED ORDER:
 *   - the shared bitmap holds committed state only: nothing transient is ever written to it (a seed's own claims and releases live
 *     in an overlay table until its turn), so it only ever gains bits;
 *   - a speculation reads the bitmap at whatever moment it runs.  Every pixel it found claimed IS claimed at its turn (bits are
 *     never taken back); a pixel it found free and did not join fails the alignment test whatever the map says; so its run equals
 *     the sequential one iff every pixel it ever JOINED (J: the overlay's entries, released ones included) is still free when its
 *     turn comes.  That is the commit's validation; a seed that fails it is processed again at its turn, when the map is final
 *     for it;
 *   - every growing seed of the ordering and every chunk end is an ITEM with a ticket (its place in the commit order); a wave takes
 *     the next ticket with one atomic add and publishes the item's result in a ring of slots; whichever wave is between items
 *     drains the ring's head: one-pixel regions of the seeds without an aligned free neighbour (marks applied in passing), small
 *     regions (nineteen in twenty stay inside the seed's 7 x 7 window below the minimum size: a 49-bit mask of cells, no overlay),
 *     big ones (overlay table + rectangle, parked in a pool of tables so that their wave moves on at once);
 *   - a region that outgrows an overlay table (one in a hundred) is processed at its turn with the frame's overlay BITMAP in HBM:
 *     any size, and still nothing transient in the shared bitmap.
 * Accepted rectangles are emitted at commit: seed order, as the single-wave kernel's.  Every wait is bounded; a wave that waits
 * longer than the budget raises the abort flag and the frame is handed to the host (status overflow). */
#define MW_WAVES 4
#define MW_NCH 8                       /* chunk records (64 seeds of the ordering each) */
#ifndef MW_RING
#define MW_RING 64                     /* result slots */
#endif
#ifndef MW_TABS
#define MW_TABS 16                     /* overlay tables (128 entries each): one in each wave's hands, the rest for parked results (at most 31) */
#endif
#define MW_TAKEN 1
#define MW_SMALL 2
#define MW_END 3
#define MW_BIG 4
#define MW_DEFER 5
struct MwChunk { unsigned long long cand, grow; int base, last, idx, pad; uint32_t key[64]; };      /* base = ticket of its first item; its end item = base + popc(grow) */
struct MwSlot { int chunk, lane, tab, haveRect; unsigned long long mask; };
struct MwShared {
    int commitNext, ticketNext, commitLock, scanLock, abortFlag, nRects, status, pad1;
    int scanned, totalTickets, tabFree, pad0;
    unsigned long long prof[16];       /* LSD_PROFILE builds: 100 MHz ticks / counts summed over the four waves (see the host's print) */
    MwChunk ch[MW_NCH];
    int word[MW_RING];                 /* ticket << 3 | state: a slot's fields are valid once its word carries the ticket */
    MwSlot slot[MW_RING];
    double tabRect[MW_TABS][12];
};

namespace {

__device__ __forceinline__ int mw_ld(int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ int mw_ldr(int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
__device__ __forceinline__ void mw_st(int* p, int v, int lane) { if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ bool mw_trylock(int* p, int lane)
{
    int got = 0;
    if (lane == 0) got = __hip_atomic_exchange(p, 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
    return __builtin_amdgcn_readfirstlane(got) != 0;
}
__device__ __forceinline__ void mw_unlock(int* p, int lane) { mw_st(p, 0, lane); }
__device__ __forceinline__ int mw_add(int* p, int v, int lane)
{
    int r = 0;
    if (lane == 0) r = __hip_atomic_fetch_add(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(r);
}
#ifdef LSD_PROFILE
#define MWP_T() wall_clock64()
#define MWP_ADD(k, t0) do { if (C.lane == 0) atomicAdd(&C.S->prof[k], wall_clock64() - (t0)); } while (0)
#define MWP_CNT(k) do { if (C.lane == 0) atomicAdd(&C.S->prof[k], 1ull); } while (0)
#else
#define MWP_T() 0ull
#define MWP_ADD(k, t0) (void)(t0)
#define MWP_CNT(k) (void)0
#endif
#define MW_BUDGET 300000000ull          /* 3 s of the 100 MHz clock: a wait longer than this gives the frame to the host */

struct MwCtx {
    Wave w;
    MwShared* S;
    uint32_t* tabs;                     /* MW_TABS x OV_SLOTS */
    AlignTan T;
    double prec, p, densityTh;
    int minReg, rectCap, W, H, lane;
    int tab;                            /* the table in this wave's hands (w.ov points at it) */
    unsigned long long t00;
};

/* window cell t of seed (sx, sy) as a pixel index, or -1 */
__device__ __forceinline__ int mw_cell_pixel(int sx, int sy, int t, int W, int H)
{
    const int wy = (t * 37) >> 8, wx = t - 7 * wy;
    const int px = sx - 3 + wx, py = sy - 3 + wy;
    return (t < 49 && px >= 0 && py >= 0 && px < W && py < H) ? py * W + px : -1;
}

__device__ __forceinline__ void mw_tab_clear(uint32_t* tab, int lane)
{
    for (int k = lane; k < OV_SLOTS; k += 64) tab[k] = OV_EMPTY;
    wg_fence();
}
__device__ __forceinline__ void mw_ov_clear(Wave& w) { mw_tab_clear(w.ov, w.lane); w.ovCount = 0; }

/* the one-pixel regions of chunk record R between its growing seeds: lanes [from, to) that were candidates without an aligned
 * free neighbour at scan time and are still free */
__device__ __forceinline__ void mw_trivial_marks(Wave& w, unsigned long long cand, unsigned long long grow, const uint32_t* keys, int to)
{
    const unsigned long long below = to >= 64 ? ~0ull : (1ull << to) - 1ull;
    const unsigned long long g = grow & below;
    const int from = g ? 64 - __builtin_clzll(g) : 0;
    const unsigned long long todo = (cand & ~grow) & below & ~(from ? ((1ull << from) - 1ull) : 0ull);
    if (!todo) return;
    if ((todo >> w.lane) & 1ull) {
        const uint32_t key = keys[w.lane];
        const uint32_t q = ((key >> 11) & 0x7FFu) * (uint32_t)w.W + (key & 0x7FFu);
        if (!w.bit(q)) atomicOr(&w.bm[q >> 5], 1u << (q & 31));
    }
}

struct MwBig { Rect rec; int haveRect; };

/* One seed through region_grow / region2rect / refine in the mode the wave is in (w.ovl).  Returns 0 = the seed was claimed
 * already (nothing to do), 1 = small (w.smallMask holds its cells, nothing claimed), 2 = big (claims in the overlay or, in direct
 * mode, in the bitmap; B.haveRect / B.rec = the accepted rectangle).  LSD_STATUS_SPEC_OVERFLOW in w.status = gave up. */
__device__ __forceinline__ int mw_process(MwCtx& C, int gx, int gy, const Window& win, MwBig& B, bool allowSmall)
{
    Wave& w = C.w;
    B.haveRect = 0;
    if (uni(w.bit((uint32_t)(gy * C.W + gx)))) return 0;
    double regAngle;
    int n = grow<true, true>(w, gx, gy, C.prec, regAngle, win, C.T, allowSmall ? C.minReg : 0);
    if (uni(w.small != 0)) return 1;
    if (uni((w.status & LSD_STATUS_SPEC_OVERFLOW) != 0)) return 2;
    if (n >= C.minReg) {
        to_rect(w, n, regAngle, C.prec, B.rec, true);
        const bool okr = refine<true>(w, n, regAngle, C.prec, B.rec, C.densityTh, win, C.T, gx, gy);
        if (okr && !(w.status & LSD_STATUS_SPEC_OVERFLOW)) B.haveRect = 1;
    }
    return 2;
}

/* the holder of the commit lock appends an accepted rectangle (12 doubles, LsdRect) */
__device__ __forceinline__ void mw_emit_rect(MwCtx& C, const double* o12)
{
    MwShared* S = C.S;
    const int at = uni_i32(S->nRects);
    if (at < C.rectCap) {
        if (C.lane < 12) C.w.F.rects[(size_t)at * 12 + C.lane] = o12[C.lane];
        if (C.lane == 0) S->nRects = at + 1;
    } else C.w.status |= DRFE_LSD_STATUS_OVERFLOW;
    wg_fence();
}
__device__ __forceinline__ void mw_rect12(const MwCtx& C, const Rect& rec, double* o)
{
    o[0] = rec.x1; o[1] = rec.y1; o[2] = rec.x2; o[3] = rec.y2; o[4] = rec.width; o[5] = rec.x; o[6] = rec.y; o[7] = rec.theta; o[8] = rec.dx; o[9] = rec.dy;
    o[10] = C.prec; o[11] = C.p;
}

__device__ __forceinline__ void mw_emit_rec(MwCtx& C, const Rect& rec)
{
    double o[12];
    mw_rect12(C, rec, o);
    double mine = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) if (C.lane == k) mine = o[k];
    const int at = uni_i32(C.S->nRects);
    if (at < C.rectCap) {
        if (C.lane < 12) C.w.F.rects[(size_t)at * 12 + C.lane] = mine;
        if (C.lane == 0) C.S->nRects = at + 1;
    } else C.w.status |= DRFE_LSD_STATUS_OVERFLOW;
    wg_fence();
}

/* commit of a small region: its cells must still be free; returns false on a conflict (nothing written) */
__device__ __forceinline__ bool mw_commit_small(Wave& w, int gx, int gy, unsigned long long mask)
{
    const int q = mw_cell_pixel(gx, gy, w.lane, w.W, w.H);
    const bool mine = q >= 0 && ((mask >> w.lane) & 1ull);
    if (__ballot(mine && w.bit((uint32_t)q))) return false;
    if (mine) atomicOr(&w.bm[q >> 5], 1u << (q & 31));
    return true;
}

/* validation of an overlay table (every pixel ever joined still free) and, if it holds, its commit (the claimed ones marked) */
__device__ __forceinline__ bool mw_commit_table(Wave& w, const uint32_t* tab)
{
    uint32_t e[OV_SLOTS / 64];
    bool conflict = false;
#pragma unroll
    for (int k = 0; k < OV_SLOTS / 64; k++) e[k] = tab[k * 64 + w.lane];
#pragma unroll
    for (int k = 0; k < OV_SLOTS / 64; k++) if (e[k] != OV_EMPTY && w.bit(e[k] & OV_PIX)) conflict = true;
    if (__ballot(conflict)) return false;
#pragma unroll
    for (int k = 0; k < OV_SLOTS / 64; k++)
        if (e[k] != OV_EMPTY && (e[k] & OV_PRESENT)) { const uint32_t q = e[k] & OV_PIX; atomicOr(&w.bm[q >> 5], 1u << (q & 31)); }
    return true;
}

/* a seed at the head of the commit order, by the holder of the commit lock: the map is final for it.  With an empty table in its
 * hands (C.tab >= 0) the wave runs it with that table; without one - every table parked behind the head - or when the table runs
 * out, with the overlay bitmap in HBM */
__device__ __forceinline__ void mw_run_at_head(MwCtx& C, int gx, int gy)
{
    Wave& w = C.w;
    if (uni(w.bit((uint32_t)(gy * C.W + gx)))) return;
    const Window win = load_window(w, gx, gy);
    MwBig B;
    if (C.tab >= 0) {
        w.ovl = 1; w.status &= ~LSD_STATUS_SPEC_OVERFLOW;
        const int kind = mw_process(C, gx, gy, win, B, true);
        if (kind == 1) { (void)mw_commit_small(w, gx, gy, w.smallMask); return; }
        if (kind == 0) return;
        if (!(w.status & LSD_STATUS_SPEC_OVERFLOW)) {
            (void)mw_commit_table(w, w.ov);                 /* cannot conflict: nothing was committed since it started */
            if (B.haveRect) mw_emit_rec(C, B.rec);
            mw_ov_clear(w);
            return;
        }
        mw_ov_clear(w);
    }
    /* once more, with the frame's overlay bitmap in HBM: any size, and still nothing transient in the shared bitmap.  What is set
     * in the overlay at the end are the region's final members: OR-ed into the bitmap and cleared, word by word */
    w.status &= ~LSD_STATUS_SPEC_OVERFLOW;
    MWP_CNT(13);
    w.ovl = 2;
    (void)mw_process(C, gx, gy, win, B, false);
    const int nWords = (C.W * C.H + 31) >> 5;
    for (int base = 0; base < nWords; base += 64 * 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + 64 * u + C.lane; v[u] = i < nWords ? __hip_atomic_load(&w.gbm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = base + 64 * u + C.lane; if (v[u]) { atomicOr(&w.bm[i], v[u]); w.gbm[i] = 0u; } }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (w.status & LSD_STATUS_SPEC_OVERFLOW) { w.status = (w.status & ~LSD_STATUS_SPEC_OVERFLOW) | DRFE_LSD_STATUS_OVERFLOW; }     /* a wave's share of the member list ran out: the host takes the frame */
    else if (B.haveRect) mw_emit_rec(C, B.rec);
    w.ovl = 1;
    wg_fence();
}

/* the ring's head forward as far as finished results reach.  Called by a wave between items: the table in its hands is empty. */
__device__ __forceinline__ void mw_drain(MwCtx& C)
{
    MwShared* S = C.S;
    Wave& w = C.w;
    {   /* anything at the head?  (one read, no lock traffic, when there is not) */
        const int t = mw_ldr(&S->commitNext);
        const int wd = mw_ldr(&S->word[t & (MW_RING - 1)]);
        const int st = wd & 7;
        if ((wd >> 3) != t || st == MW_TAKEN) return;
    }
    if (!mw_trylock(&S->commitLock, C.lane)) return;
    const unsigned long long td0 = MWP_T();
    for (;;) {
        const int t = mw_ldr(&S->commitNext);
        const int wd = mw_ld(&S->word[t & (MW_RING - 1)]);
        const int st = wd & 7;
        if ((wd >> 3) != t || st == MW_TAKEN) break;
        const MwSlot sl = S->slot[t & (MW_RING - 1)];
        const int c = uni_i32(sl.chunk), f = uni_i32(sl.lane);
        const MwChunk& R = S->ch[c & (MW_NCH - 1)];
        mw_trivial_marks(w, uni_u64(R.cand), uni_u64(R.grow), R.key, st == MW_END ? 64 : f);
        if (st != MW_END) {
            const uint32_t key = uni_u32(R.key[f]);
            const int gx = (int)(key & 0x7FFu), gy = (int)((key >> 11) & 0x7FFu);
            bool done = false;
            if (st == MW_SMALL) { const unsigned long long mask = uni_u64(sl.mask); done = mask == 0 || mw_commit_small(w, gx, gy, mask); }
            else if (st == MW_BIG) {
                const int tb = uni_i32(sl.tab);
                uint32_t* tab = C.tabs + (size_t)tb * OV_SLOTS;
                if (mw_commit_table(w, tab)) {
                    if (uni_i32(sl.haveRect)) mw_emit_rect(C, S->tabRect[tb]);
                    done = true;
                }
                mw_tab_clear(tab, C.lane);
                if (C.lane == 0) atomicOr(&S->tabFree, 1 << tb);
            }
            if (!done) { const unsigned long long th0 = MWP_T(); MWP_CNT(st == MW_DEFER ? 10 : 11); mw_run_at_head(C, gx, gy); MWP_ADD(12, th0); }
        }
        wg_fence();
        mw_st(&S->commitNext, t + 1, C.lane);
        if (mw_ldr(&S->abortFlag)) break;
    }
    MWP_ADD(0, td0);
    mw_unlock(&S->commitLock, C.lane);
}

/* one chunk of the ordering into its record: candidates (free at this moment) and which of them have an aligned free neighbour.
 * Called under the scan lock, chunks in order. */
__device__ __forceinline__ bool mw_scan_chunk(MwCtx& C, int c)
{
    Wave& w = C.w;
    MwShared* S = C.S;
    const int lane = C.lane, W = C.W, H = C.H;
    const int pos = c * 64 + lane;
    const uint32_t key = pos < w.F.nOrder ? w.F.order[pos] : 0u;
    const bool valid = pos < w.F.nOrder && (key >> 22) >= w.F.minSeedBin;
    const int sx = (int)(key & 0x7FFu), sy = (int)((key >> 11) & 0x7FFu);
    const uint32_t q = (uint32_t)(sy * W + sx);
    const bool cand = valid && !w.bit(q);
    bool nt = false;
    if (cand) {
        const double sa = w.F.ang[q];
        double na[8];
        uint32_t nf = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int jj = j < 4 ? j : j + 1, dx = jj % 3 - 1, dy = jj / 3 - 1;
            const int nx = sx + dx, ny = sy + dy;
            const bool inb = nx >= 0 && ny >= 0 && nx < W && ny < H;
            const uint32_t nq = inb ? (uint32_t)(ny * W + nx) : 0u;
            na[j] = 0.0;
            if (inb && !w.bit(nq)) { nf |= 1u << j; na[j] = w.F.ang[nq]; }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) nt |= ((nf >> j) & 1u) && aligned_with(sa, na[j], C.prec);
    }
    const unsigned long long mc = __ballot(cand), mg = __ballot(cand && nt), inv = __ballot(!valid);
    MwChunk& R = S->ch[c & (MW_NCH - 1)];
    int base = 0;
    if (c > 0) { const MwChunk& P = S->ch[(c - 1) & (MW_NCH - 1)]; base = uni_i32(P.base) + __popcll(uni_u64(P.grow)) + 1; }
    const int last = (inv != 0 || (c + 1) * 64 >= w.F.nOrder) ? 1 : 0;
    /* a wave that is looking for its ticket's chunk may be reading this record as the (long committed) chunk c - MW_NCH: the index goes
     * invalid first and valid last, and the reader checks it on both sides of its reads */
    mw_st(&R.idx, -1, lane);
    R.key[lane] = key;
    if (lane == 0) { R.cand = mc; R.grow = mg; R.base = base; R.last = last; }
    wg_fence();
    mw_st(&R.idx, c, lane);
    if (last) mw_st(&S->totalTickets, base + __popcll(mg) + 1, lane);
    return true;
}

__device__ __forceinline__ bool mw_over_budget(MwCtx& C)
{
    if (wall_clock64() - C.t00 > MW_BUDGET) { mw_st(&C.S->abortFlag, 1, C.lane); return true; }
    return false;
}

}  // namespace

#ifndef MW_WAVES_PER_EU
#define MW_WAVES_PER_EU 2              /* measured: capping the kernel at 168 VGPRs for three workgroups per CU spills into its hot loops (38 -> 52 ms per 512 frames) */
#endif
extern "C" __global__ __launch_bounds__(64 * MW_WAVES) __attribute__((amdgpu_waves_per_eu(MW_WAVES_PER_EU, MW_WAVES_PER_EU))) void k_lsd_grow_mw(const LsdGrowFrame* __restrict__ frames, int W, int H, double prec, double p,
                                                                         int minReg, double densityTh, int rectCap, float tLo, float tHi, int regCap)
{
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int npx = W * H, nWords = (npx + 31) >> 5;
    MwCtx C;
    Wave& w = C.w;
    {
        const LsdGrowFrame f = frames[blockIdx.x];
        w.F.ang = (const GLOBAL_AS double*)f.ang; w.F.cs = (const GLOBAL_AS float*)f.cs; w.F.cs0 = (const GLOBAL_AS float*)f.cs0; w.F.mod = (const GLOBAL_AS double*)f.mod;
        w.F.order = (const GLOBAL_AS uint32_t*)f.order;
        w.F.reg = (GLOBAL_AS uint32_t*)f.regMw + (size_t)wave * regCap; w.F.tmp = (GLOBAL_AS uint32_t*)f.tmpMw + (size_t)wave * regCap;
        w.F.rects = (GLOBAL_AS double*)f.rects; w.F.out = (GLOBAL_AS int*)f.out; w.F.nOrder = f.nOrder;
        w.F.minSeedBin = f.meta ? 1024u - (uint32_t)(f.meta[1] & 0xFFFFFFFFull) : f.minSeedBin;
        w.gbm = (GLOBAL_AS uint32_t*)f.gbm;
    }
    w.W = W; w.H = H; w.lane = lane; w.status = 0;
    w.ovl = 1; w.ovCount = 0; w.regCap = regCap; w.small = 0; w.smallMask = 0;
#ifdef LSD_PROFILE
    for (int k = 0; k < 16; k++) w.prof[k] = 0;
#endif
    /* LDS: bitmap | shared control block | overlay tables | per wave: ring, ordered-sum columns */
    uint32_t* at = lds;
    w.bm = at; at += (nWords + 1) & ~1;
    MwShared* S = (MwShared*)at; at += (sizeof(MwShared) + 7) / 8 * 2;
    C.tabs = at; at += MW_TABS * OV_SLOTS;
    at += (size_t)wave * (LSD_RING + 64 * 3 * 2);
    w.ring = at; w.col = (double*)(at + LSD_RING);
    C.tab = wave; w.ov = C.tabs + (size_t)wave * OV_SLOTS;
    C.S = S; C.T.tLo = tLo; C.T.tHi = tHi; C.prec = prec; C.p = p; C.densityTh = densityTh; C.minReg = minReg; C.rectCap = rectCap; C.W = W; C.H = H; C.lane = lane;

    /* bitmap: pixels without a level-line angle never join (NOTDEF, incl. the last row and column); the four waves side by side */
    const GLOBAL_AS uint32_t* notdef = (const GLOBAL_AS uint32_t*)frames[blockIdx.x].notdef;
    if (notdef) { for (int k = tid; k < nWords; k += 64 * MW_WAVES) w.bm[k] = notdef[k]; }
    else
    for (int base = wave * 64; base < npx; base += 64 * MW_WAVES) {
        const int q = base + lane;
        const bool nd = q >= npx || w.F.ang[q] == -1024.0;
        const unsigned long long b = __ballot(nd);
        const int wd = base >> 5;
        if (lane == 0) { if (wd < nWords) w.bm[wd] = (uint32_t)b; if (wd + 1 < nWords) w.bm[wd + 1] = (uint32_t)(b >> 32); }
    }
    for (int k = tid; k < (int)(sizeof(MwShared) / 4); k += 64 * MW_WAVES) ((uint32_t*)S)[k] = 0;
    for (int k = tid; k < MW_TABS * OV_SLOTS; k += 64 * MW_WAVES) C.tabs[k] = OV_EMPTY;
    for (int k = tid; k < nWords; k += 64 * MW_WAVES) w.gbm[k] = 0u;                  /* the HBM overlay starts, and is left, all zero */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) { S->totalTickets = 0x7fffffff; S->tabFree = (int)(((1u << MW_TABS) - 1u) & ~((1u << MW_WAVES) - 1u)); for (int k = 0; k < MW_RING; k++) S->word[k] = -8; }
    __syncthreads();
    C.t00 = wall_clock64();

    int myChunk = 0;                       /* the chunk this wave's last ticket lay in: tickets only go up */
    for (;;) {
        if (mw_ldr(&S->abortFlag)) break;
        const int total = mw_ldr(&S->totalTickets);
        if (mw_ldr(&S->commitNext) >= total) break;
        mw_drain(C);
        /* the next item of the commit order */
        const unsigned long long tk0 = MWP_T();
        if (mw_ldr(&S->ticketNext) >= total) { const unsigned long long ti0 = MWP_T(); if (mw_over_budget(C)) break; __builtin_amdgcn_s_sleep(4); MWP_ADD(7, ti0); continue; }      /* all taken: drain until the end */
        const int ticket = mw_add(&S->ticketNext, 1, lane);
        /* which chunk / seed is it?  The chunk may have to be scanned first (chunks in order, one scanner at a time), and the
         * ring must have room for the ticket */
        int f = -1;
        bool stop = false;
        for (;;) {
            const int sc = mw_ld(&S->scanned);
            if (myChunk < sc) {
                MwChunk& R = S->ch[myChunk & (MW_NCH - 1)];
                const int idx1 = mw_ld(&R.idx);
                const int base = uni_i32(R.base), isLast = uni_i32(R.last);
                const unsigned long long grow = uni_u64(R.grow);
                wg_fence();
                const int idx2 = mw_ld(&R.idx);
                /* not this chunk (any more): it was committed to its end long ago and its record reused - the ticket lies further on */
                if (idx1 != myChunk || idx2 != myChunk) { myChunk++; continue; }
                const int cnt = __popcll(grow);
                if (ticket > base + cnt) { if (isLast) { stop = true; break; } myChunk++; continue; }
                if (ticket - mw_ldr(&S->commitNext) < MW_RING) {
                    f = 64;
                    if (ticket < base + cnt) { unsigned long long g = grow; for (int k = ticket - base; k > 0; k--) g &= g - 1; f = __builtin_ctzll(g); }
                    break;
                }
            } else if (sc - 1 >= 0 && uni_i32(S->ch[(sc - 1) & (MW_NCH - 1)].last)) { stop = true; break; }
            else if (mw_trylock(&S->scanLock, lane)) {
                const unsigned long long ts0 = MWP_T();
                const int sc2 = mw_ld(&S->scanned);
                /* the record's previous chunk (sc2 - MW_NCH) must be committed to its end */
                bool roomy = sc2 < MW_NCH;
                if (!roomy) { const MwChunk& O = S->ch[sc2 & (MW_NCH - 1)]; roomy = mw_ldr(&S->commitNext) > uni_i32(O.base) + __popcll(uni_u64(O.grow)); }
                const bool ended = sc2 > 0 && uni_i32(S->ch[(sc2 - 1) & (MW_NCH - 1)].last) != 0;
                if (roomy && !ended && mw_scan_chunk(C, sc2)) mw_st(&S->scanned, sc2 + 1, lane);
                mw_unlock(&S->scanLock, lane);
                MWP_ADD(1, ts0);
                if (roomy) continue;
            }
            /* waiting for the scanner, for a record to free up or for room in the ring: help the head along */
            if (mw_ldr(&S->abortFlag) || mw_over_budget(C)) { stop = true; break; }
            mw_drain(C);
            __builtin_amdgcn_s_sleep(1);
        }
        MWP_ADD(2, tk0);
        if (stop) { if (mw_ldr(&S->abortFlag)) break; continue; }      /* a ticket beyond the last item */
        MwSlot* sl = &S->slot[ticket & (MW_RING - 1)];
        int* word = &S->word[ticket & (MW_RING - 1)];
        if (lane == 0) { sl->chunk = myChunk; sl->lane = f; sl->mask = 0; sl->tab = -1; sl->haveRect = 0; }
        wg_fence();
        if (f == 64) { mw_st(word, ticket << 3 | MW_END, lane); continue; }
        mw_st(word, ticket << 3 | MW_TAKEN, lane);

        /* ---- speculate on the seed ---- */
        const unsigned long long tp0 = MWP_T();
        const uint32_t key = uni_u32(S->ch[myChunk & (MW_NCH - 1)].key[f]);
        const int gx = (int)(key & 0x7FFu), gy = (int)((key >> 11) & 0x7FFu);
        if (uni(w.bit((uint32_t)(gy * W + gx)))) { mw_st(word, ticket << 3 | MW_SMALL, lane); MWP_ADD(3, tp0); continue; }      /* mask 0: claimed already */
        const Window win = load_window(w, gx, gy);
        MwBig B;
        w.ovl = 1; w.status &= ~LSD_STATUS_SPEC_OVERFLOW;
        const int kind = mw_process(C, gx, gy, win, B, true);
        if (kind == 0) { mw_st(word, ticket << 3 | MW_SMALL, lane); MWP_ADD(3, tp0); continue; }
        if (kind == 1) {
            if (lane == 0) sl->mask = w.smallMask;
            wg_fence();
            mw_st(word, ticket << 3 | MW_SMALL, lane);
            MWP_ADD(3, tp0); MWP_CNT(8);
            continue;
        }
        MWP_CNT(9);
#ifdef LSD_PROFILE_JHIST
        if (w.ovCount <= 48) MWP_CNT(6); else if (w.ovCount <= 96) MWP_CNT(10); else if (w.ovCount <= 192) MWP_CNT(13); else MWP_CNT(7);
#endif
        /* big: finished speculatively in this wave's table - unless the table ran out: such a region waits for its turn */
        if ((w.status & LSD_STATUS_SPEC_OVERFLOW) != 0) {
            w.status &= ~LSD_STATUS_SPEC_OVERFLOW;
            mw_ov_clear(w);
            mw_st(word, ticket << 3 | MW_DEFER, lane);
            MWP_ADD(4, tp0); MWP_CNT(15);
            continue;
        }
        /* park the result: the table and the rectangle stay where they are until the head reaches the ticket */
        if (B.haveRect) { double o[12]; mw_rect12(C, B.rec, o); double mine = 0; for (int k = 0; k < 12; k++) if (lane == k) mine = o[k]; if (lane < 12) S->tabRect[C.tab][lane] = mine; }
        if (lane == 0) { sl->tab = C.tab; sl->haveRect = B.haveRect; }
        wg_fence();
        mw_st(word, ticket << 3 | MW_BIG, lane);
        MWP_ADD(4, tp0);
        /* another table */
        const unsigned long long tw0 = MWP_T();
        C.tab = -1;
        int got = -1;
        for (;;) {
            int fr = mw_ldr(&S->tabFree);
            if (fr) {
                const int b = __builtin_ctz(fr);
                int old = 0;
                if (lane == 0) old = atomicAnd(&S->tabFree, ~(1 << b));
                old = uni_i32(old);
                if (old & (1 << b)) { got = b; break; }
                continue;
            }
            if (mw_ldr(&S->abortFlag) || mw_over_budget(C)) break;
            /* none free: every table is in a wave's hands or parked.  The head may be a parked one: help it along (C.tab < 0: a
             * head that has to be run again goes the direct way) */
            mw_drain(C);
            __builtin_amdgcn_s_sleep(1);
        }
        MWP_ADD(5, tw0);
        if (got < 0) break;
        C.tab = got; w.ov = C.tabs + (size_t)got * OV_SLOTS; w.ovCount = 0;
    }
    if (lane == 0 && w.status) atomicOr(&S->status, w.status & ~LSD_STATUS_SPEC_OVERFLOW);
    __syncthreads();
    if (tid == 0) {
        int st = S->status;
        if (S->abortFlag || S->commitNext < S->totalTickets) st |= DRFE_LSD_STATUS_OVERFLOW;      /* did not finish: the host takes the frame */
        w.F.out[DRFE_LSD_OUT_NEXT_RECT] = 0; w.F.out[0] = S->nRects; w.F.out[1] = st; w.F.out[2] = 0; w.F.out[3] = 0;
#ifdef LSD_PROFILE
        S->prof[14] = wall_clock64() - C.t00;
        for (int k = 0; k < 16; k++) ((GLOBAL_AS unsigned long long*)(w.F.out + 4))[k] = S->prof[k];
#endif
    }
}

/* pseudo-ordering keys: gradient bin << 22 | y << 11 | x for the (W - 1) x (H - 1) pixels ll_angle visits, in raster order
 * (what the host sorts), the seed direction of every pixel with an angle, and the smallest bin of a pixel that has a level-line angle (kept as 1024 - bin under atomicMax in
 * the low half of the slot's second meta word, which the image passes zeroed).  blockIdx.z = frame slot. */
__global__ __launch_bounds__(256) void k_lsd_keys(const double* __restrict__ mod, const double* __restrict__ ang, int W, int H,
                                                  unsigned long long* __restrict__ meta, uint32_t* __restrict__ keys, float2* __restrict__ cs0)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    mod += (size_t)blockIdx.z * W * H; ang += (size_t)blockIdx.z * W * H; cs0 += (size_t)blockIdx.z * W * H;
    meta += 2 * (size_t)blockIdx.z; keys += (size_t)blockIdx.z * (W - 1) * (H - 1);
    uint32_t seedBin = 1024;
    if (x < W - 1) {
        const unsigned long long mb = meta[0];
        const double maxGrad = mb ? __longlong_as_double((long long)mb) : -1.0;
        const double binCoef = (maxGrad > 0) ? (double)(1024 - 1) / maxGrad : 0;
        const size_t o = (size_t)y * W + x;
        const uint32_t bin = (uint32_t)(int)(mod[o] * binCoef);
        keys[(size_t)y * (W - 1) + x] = (bin << 22) | ((uint32_t)y << 11) | (uint32_t)x;
        const double a = ang[o];
        if (a != -1024.0) {
            seedBin = bin;
            /* the direction a region starts with when this pixel seeds it: float(cos(a)), float(sin(a)), correctly rounded
             * (cr_sincos.h) - here, for every pixel side by side, instead of inside the frame's sequential wavefront */
            float sn, cn;
            const bool ok = drfe_cr_sincos_f(a, &sn, &cn) != 0;
            const float nanv = __int_as_float(0x7fc00000);
            cs0[o] = ok ? make_float2(cn, sn) : make_float2(nanv, nanv);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) seedBin = min(seedBin, (uint32_t)__shfl_xor((int)seedBin, o));
    if ((threadIdx.x & 63) == 0 && seedBin < 1024) atomicMax((uint32_t*)(meta + 1), 1024u - seedBin);
}

size_t drfe_lsd_grow_lds_bytes(int W, int H)
{
    const size_t nWords = ((size_t)W * H + 31) >> 5;
    return (((nWords + 1) & ~(size_t)1) + LSD_RING) * 4 + 64 * 3 * sizeof(double);
}

/* bit q of a frame's words = pixel q (row-major over the W x H field) has no level-line angle; bits past the field are set */
__global__ __launch_bounds__(256) void k_lsd_notdef(const double* __restrict__ ang, int npx, int nWords, uint32_t* __restrict__ words)
{
    ang += (size_t)blockIdx.y * npx; words += (size_t)blockIdx.y * nWords;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool nd = q >= npx || ang[q] == -1024.0;
    const unsigned long long b = __ballot(nd);
    const int wd = (q & ~63) >> 5;
    if ((threadIdx.x & 63) == 0) { if (wd < nWords) words[wd] = (uint32_t)b; if (wd + 1 < nWords) words[wd + 1] = (uint32_t)(b >> 32); }
}

hipError_t drfe_launch_lsd_keys(const double* d_mod, const double* d_ang, int W, int H, unsigned long long* d_meta, uint32_t* d_keys,
                                float2* d_cs0, uint32_t* d_notdef, int nframes, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    if (d_notdef) {
        const int npx = W * H, nWords = (npx + 31) >> 5;
        hipLaunchKernelGGL(k_lsd_notdef, dim3((npx + 255) / 256, nframes), dim3(256), 0, s, d_ang, npx, nWords, d_notdef);
    }
    hipLaunchKernelGGL(k_lsd_keys, dim3((W - 1 + 255) / 256, H - 1, nframes), dim3(256), 0, s, d_mod, d_ang, W, H, d_meta, d_keys, d_cs0);
    return hipGetLastError();
}

size_t drfe_lsd_grow_mw_lds_bytes(int W, int H)
{
    const size_t nWords = ((size_t)W * H + 31) >> 5;
    return (((nWords + 1) & ~(size_t)1) + (sizeof(MwShared) + 7) / 8 * 2 + (size_t)MW_TABS * OV_SLOTS + (size_t)MW_WAVES * (LSD_RING + 64 * 3 * 2)) * 4;
}

hipError_t drfe_launch_lsd_grow(const LsdGrowFrame* d_frames, int nframes, int W, int H, double prec, double p, int minReg,
                                double densityTh, int rectCap, hipStream_t s, int regCapMw)
{
    if (nframes <= 0) return hipSuccess;
    const size_t lds = regCapMw > 0 ? drfe_lsd_grow_mw_lds_bytes(W, H) : drfe_lsd_grow_lds_bytes(W, H);
    static size_t configured = 0, configuredMw = 0;
    if (lds > 64 * 1024 && lds > (regCapMw > 0 ? configuredMw : configured)) {
        hipError_t e = hipFuncSetAttribute(regCapMw > 0 ? (const void*)k_lsd_grow_mw : (const void*)k_lsd_grow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        (regCapMw > 0 ? configuredMw : configured) = lds;
    }
    /* thresholds of align_class for this precision: tan(prec -/+ 0.02 degrees), rounded away from the band; a tolerance that
     * reaches 90 degrees switches the shortcut off (tLo <= 0) */
    const double band = 0.02 * 3.14159265358979323846 / 180.0;
    float tLo = 0.f, tHi = 0.f;
    if (prec - band > 0 && prec + band < 1.5 && !std::getenv("DRFE_LSD_EXACT_ALIGN")) {
        tLo = nextafterf((float)tan(prec - band), 0.f);
        tHi = nextafterf((float)tan(prec + band), 1e30f);
    }
    if (regCapMw > 0)
        hipLaunchKernelGGL(k_lsd_grow_mw, dim3(nframes), dim3(64 * MW_WAVES), lds, s, d_frames, W, H, prec, p, minReg, densityTh, rectCap, tLo, tHi, regCapMw);
    else
        hipLaunchKernelGGL(k_lsd_grow, dim3(nframes), dim3(64), lds, s, d_frames, W, H, prec, p, minReg, densityTh, rectCap, tLo, tHi);
    return hipGetLastError();
}
