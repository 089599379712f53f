/* oracle/orb_oracle.cpp — TEST INFRASTRUCTURE (see oracle.h header note; parity unpinned vs OpenCV).
 *
 * CPU restatement of ORBextractor (reference src/ORBextractor.cc) and of the OpenCV 3.4.4 calls it
 * makes.  Each function cites the reference lines it follows.
 */
#include "oracle.h"
#include "oracle_math.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <list>
#include <stdexcept>

namespace orc {

static const int8_t kPattern[1024] = {
#include "../include/drfe_orb_pattern.inc"
};

/* ---------------------------------------------------------------------------------------------- */
/* OpenCV semantics                                                                */

/* copyMakeBorder(BORDER_REFLECT_101) index map, §10.3 */
int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

/* cv::resize(INTER_LINEAR) CV_8UC1, classic fixed-point path, §10.2 */
namespace {
struct LinCoef { int s; short w0, w1; bool edge; };
static short sat_short_round(float v)
{
    int r = round_he(v);
    return (short)std::min(32767, std::max(-32768, r));
}
static void build_axis_x(int src, int dst, std::vector<LinCoef>& t)
{
    const double inv = (double)dst / src;
    const double scale = 1. / inv;
    t.resize(dst);
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        bool edge = false;
        if (s < 0) { s = 0; f = 0; }
        if (s + 1 >= src) {
            /* xmax = min(xmax, dx): these columns copy S[s]*2048 */
            edge = true;
            if (s >= src - 1) { s = src - 1; f = 0; }
        }
        t[d].s = s;
        t[d].w0 = sat_short_round((1.f - f) * 2048.f);
        t[d].w1 = sat_short_round(f * 2048.f);
        t[d].edge = edge;
    }
}
} // namespace

void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                      size_t dstride)
{
    std::vector<LinCoef> cx;
    build_axis_x(sw, dw, cx);
    const double inv_y = (double)dh / sh;
    const double scale_y = 1. / inv_y;
    std::vector<int> h0(dw), h1(dw);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        const short b0 = sat_short_round((1.f - fy) * 2048.f);
        const short b1 = sat_short_round(fy * 2048.f);
        /* rows sy, sy+1 clipped to [0, sh-1] (clip() in resizeGeneric_Invoker) */
        const int r0 = std::min(std::max(sy, 0), sh - 1);
        const int r1 = std::min(std::max(sy + 1, 0), sh - 1);
        const uint8_t* S0 = src + (size_t)r0 * sstride;
        const uint8_t* S1 = src + (size_t)r1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            const LinCoef& c = cx[dx];
            if (c.edge) {
                h0[dx] = S0[c.s] * 2048;
                h1[dx] = S1[c.s] * 2048;
            } else {
                h0[dx] = S0[c.s] * c.w0 + S0[c.s + 1] * c.w1;
                h1[dx] = S1[c.s] * c.w0 + S1[c.s + 1] * c.w1;
            }
        }
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) {
            int v = (((b0 * (h0[dx] >> 4)) >> 16) + ((b1 * (h1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)std::min(255, std::max(0, v));
        }
    }
}

/* cv::FAST TYPE_9_16, §10.1 */
static const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                 {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* cornerScore<16>(ptr, pixel, threshold) of OpenCV fast_score.cpp */
static int corner_score16(const uint8_t* p, size_t stride, int threshold)
{
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) {
        const int* o = kRing[k & 15];
        d[k] = v - p[(ptrdiff_t)o[1] * (ptrdiff_t)stride + o[0]];
    }
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

int fast_score_9_16(const uint8_t* p, size_t stride) { return corner_score16(p, stride, 0); }

static bool is_fast_corner(const uint8_t* p, size_t stride, int t)
{
    const int v = p[0];
    int run_b = 0, run_d = 0;
    for (int k = 0; k < 25; k++) {
        const int* o = kRing[k & 15];
        const int x = p[(ptrdiff_t)o[1] * (ptrdiff_t)stride + o[0]];
        if (x > v + t) { if (++run_b > 8) return true; } else run_b = 0;
        if (x < v - t) { if (++run_d > 8) return true; } else run_d = 0;
    }
    return false;
}

/* cv::FAST(img, keypoints, threshold, nonmaxSuppression=true): rows/cols [3, n-3), scores of
 * non-corners and of pixels outside that region read as 0 in the 3x3 strict-maximum test. */
void fast_detect(const uint8_t* img, int w, int h, size_t stride, int threshold, std::vector<Candidate>& out)
{
    out.clear();
    if (w < 7 || h < 7) return;
    std::vector<int> sc((size_t)w * h, 0);
    std::vector<char> corner((size_t)w * h, 0);
    ptrdiff_t o[16];
    for (int k = 0; k < 16; k++) o[k] = (ptrdiff_t)kRing[k][1] * (ptrdiff_t)stride + kRing[k][0];
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            /* cv::FAST's early rejection: every 9-arc holds one pixel of each antipodal pair, so each
             * pair must have a darker (bit 1) / brighter (bit 2) member — a necessary condition only */
            const int v = p[0], lo = v - threshold, hi = v + threshold;
#define CLS(k) ((p[o[k]] < lo ? 1 : 0) | (p[o[k]] > hi ? 2 : 0))
            int d = CLS(0) | CLS(8);
            if (!d) continue;
            d &= CLS(2) | CLS(10); d &= CLS(4) | CLS(12); d &= CLS(6) | CLS(14);
            if (!d) continue;
            d &= CLS(1) | CLS(9); d &= CLS(3) | CLS(11); d &= CLS(5) | CLS(13); d &= CLS(7) | CLS(15);
#undef CLS
            if (!d) continue;
            if (is_fast_corner(p, stride, threshold)) {
                corner[(size_t)y * w + x] = 1;
                sc[(size_t)y * w + x] = corner_score16(p, stride, threshold);
            }
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            if (!corner[(size_t)y * w + x]) continue;
            const int s = sc[(size_t)y * w + x];
            bool keep = true;
            for (int dy = -1; dy <= 1 && keep; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    if (!(s > sc[(size_t)(y + dy) * w + (x + dx)])) { keep = false; break; }
                }
            if (keep) out.push_back({x, y, s});
        }
}

/* GaussianBlur(7x7, sigma 2) CV_8U fixed-point path with BORDER_REFLECT_101, §10.4 */
static const int kGauss7[7] = {18, 34, 49, 55, 49, 34, 18};

void gaussian_blur_7x7_s2_u8(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride)
{
    /* same arithmetic as the direct double loop; rows are padded once so the inner loops vectorise */
    std::vector<uint16_t> hbuf((size_t)w * h); /* 8.8 fixed point, <= 255*257 = 65535 */
    std::vector<uint8_t> pad((size_t)w + 6);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = src + (size_t)y * sstride;
        for (int x = -3; x < w + 3; x++) pad[x + 3] = s[reflect101(x, w)];
        uint16_t* hrow = &hbuf[(size_t)y * w];
        const uint8_t* p = pad.data();
        for (int x = 0; x < w; x++)
            hrow[x] = (uint16_t)(18u * (p[x] + p[x + 6]) + 34u * (p[x + 1] + p[x + 5]) + 49u * (p[x + 2] + p[x + 4]) +
                                 55u * p[x + 3]);
    }
    for (int y = 0; y < h; y++) {
        const uint16_t* r[7];
        for (int k = -3; k <= 3; k++) r[k + 3] = &hbuf[(size_t)reflect101(y + k, h) * w];
        uint8_t* d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; x++) {
            const uint32_t acc = 18u * ((uint32_t)r[0][x] + r[6][x]) + 34u * ((uint32_t)r[1][x] + r[5][x]) +
                                 49u * ((uint32_t)r[2][x] + r[4][x]) + 55u * (uint32_t)r[3][x];
            const uint32_t v = (acc + 32768u) >> 16;
            d[x] = (uint8_t)std::min<uint32_t>(255u, v);
        }
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* reference-owned logic                                                                           */

/* IC_Angle, src/ORBextractor.cc:77-104 */
float ic_angle(const uint8_t* center, size_t stride, const std::vector<int>& umax)
{
    int m_01 = 0, m_10 = 0;
    for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m_10 += u * center[u];
    const ptrdiff_t step = (ptrdiff_t)stride;
    for (int v = 1; v <= kHalfPatch; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2_deg((float)m_01, (float)m_10);
}

/* computeOrbDescriptor, src/ORBextractor.cc:108-147 (cos/sin canonicalised) */
void orb_descriptor(const uint8_t* center, size_t stride, float angle_deg, uint8_t* desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = angle_deg * factorPI;
    float a, b;
    sincos_f(angle, &b, &a);
    const ptrdiff_t step = (ptrdiff_t)stride;
    const int8_t* pat = kPattern;
    /* flat table: point j of byte i at pat[(16*i + j)*2 ..]; bit k compares points 2k and 2k+1 */
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const int8_t* q = pat + k * 4;
            const int t0 = center[round_he(q[0] * b + q[1] * a) * step +
                                  round_he(q[0] * a - q[1] * b)];
            const int t1 = center[round_he(q[2] * b + q[3] * a) * step +
                                  round_he(q[2] * a - q[3] * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* DescriptorDistance, src/ORBmatcher.cc:1712-1728 (Stanford bit-twiddling SWAR over 8 x u32) */
int descriptor_distance_swar(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ORBextractor::ORBextractor, src/ORBextractor.cc:410-470 */
OrbExtractor::OrbExtractor(int nf, float sf, int nl, int ini, int mn)
    : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), scaleFactorD((double)sf)
{
    scale.resize(nl); sigma2.resize(nl); invScale.resize(nl); invSigma2.resize(nl);
    scale[0] = 1.0f; sigma2[0] = 1.0f;
    for (int i = 1; i < nl; i++) {
        scale[i] = (float)(scale[i - 1] * scaleFactorD);
        sigma2[i] = scale[i] * scale[i];
    }
    for (int i = 0; i < nl; i++) {
        invScale[i] = 1.0f / scale[i];
        invSigma2[i] = 1.0f / sigma2[i];
    }
    quota.resize(nl);
    const float factor = (float)(1.0f / scaleFactorD);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        quota[l] = round_he(nDesired);
        sum += quota[l];
        nDesired *= factor;
    }
    quota[nl - 1] = std::max(nfeatures - sum, 0);

    umax.assign(kHalfPatch + 1, 0);
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) umax[v] = round_he_d(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* geometry part of ComputePyramid (:1111-1113) and ComputeKeyPointsOctTree (:773-787) */
void OrbExtractor::computeGeometry(int w, int h)
{
    geom.resize(nlevels);
    for (int l = 0; l < nlevels; l++) {
        LevelGeom& g = geom[l];
        g.w = round_he((float)w * invScale[l]);
        g.h = round_he((float)h * invScale[l]);
        g.quota = quota[l];
        g.minBX = kEdge - 3; g.minBY = kEdge - 3;
        g.maxBX = g.w - kEdge + 3; g.maxBY = g.h - kEdge + 3;
        const float width = (float)(g.maxBX - g.minBX), height = (float)(g.maxBY - g.minBY);
        g.nCols = (int)(width / 30.f);
        g.nRows = (int)(height / 30.f);
        if (g.nCols <= 0 || g.nRows <= 0) throw std::runtime_error("level too small for the 30-px cell grid");
        g.wCell = (int)std::ceil(width / g.nCols);
        g.hCell = (int)std::ceil(height / g.nRows);
    }
}

/* ComputePyramid, src/ORBextractor.cc:1107-1132 */
void OrbExtractor::computePyramid(const uint8_t* gray, int w, int h, size_t stride)
{
    pyramid.assign(nlevels, Image());
    for (int l = 0; l < nlevels; l++) {
        const LevelGeom& g = geom[l];
        Image& im = pyramid[l];
        im.w = g.w + 2 * kEdge; im.h = g.h + 2 * kEdge;
        im.px.assign((size_t)im.w * im.h, 0);
        uint8_t* interior = im.px.data() + (size_t)kEdge * im.w + kEdge;
        if (l == 0) {
            for (int y = 0; y < h; y++) std::memcpy(interior + (size_t)y * im.w, gray + (size_t)y * stride, w);
        } else {
            const Image& pr = pyramid[l - 1];
            resize_linear_u8(pr.px.data() + (size_t)kEdge * pr.w + kEdge, geom[l - 1].w, geom[l - 1].h, pr.w,
                             interior, g.w, g.h, im.w);
        }
        for (int y = 0; y < im.h; y++) {
            const int sy = reflect101(y - kEdge, g.h);
            for (int x = 0; x < im.w; x++) {
                if (y >= kEdge && y < kEdge + g.h && x >= kEdge && x < kEdge + g.w) continue;
                const int sx = reflect101(x - kEdge, g.w);
                im.px[(size_t)y * im.w + x] = interior[(size_t)sy * im.w + sx];
            }
        }
    }
}

/* FAST stage of ComputeKeyPointsOctTree, src/ORBextractor.cc:789-829 */
void OrbExtractor::computeCandidates(int level)
{
    const LevelGeom& g = geom[level];
    const Image& im = pyramid[level];
    const uint8_t* roi = im.px.data() + (size_t)kEdge * im.w + kEdge; /* mvImagePyramid[level] */
    std::vector<Candidate>& out = candidates[level];
    out.clear();
    std::vector<Candidate> cell;
    for (int i = 0; i < g.nRows; i++) {
        const float iniY = (float)(g.minBY + i * g.hCell);
        float maxY = iniY + g.hCell + 6;
        if (iniY >= g.maxBY - 3) continue;
        if (maxY > g.maxBY) maxY = (float)g.maxBY;
        for (int j = 0; j < g.nCols; j++) {
            const float iniX = (float)(g.minBX + j * g.wCell);
            float maxX = iniX + g.wCell + 6;
            if (iniX >= g.maxBX - 6) continue;
            if (maxX > g.maxBX) maxX = (float)g.maxBX;
            const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
            const uint8_t* sub = roi + (size_t)y0 * im.w + x0;
            fast_detect(sub, x1 - x0, y1 - y0, im.w, iniTh, cell);
            if (cell.empty()) fast_detect(sub, x1 - x0, y1 - y0, im.w, minTh, cell);
            for (const Candidate& c : cell) out.push_back({c.x + j * g.wCell, c.y + i * g.hCell, c.response});
        }
    }
}

/* DistributeOctTree + ExtractorNode::DivideNode, src/ORBextractor.cc:481-763.
 * Returns indices into `keys` in final list order.  Canonical tie-break for the pointer-tied sort at
 * :684: equal sizes order by node creation sequence number (a later-created node
 * compares greater, as with a monotonically growing heap). */
namespace {
struct Node {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::vector<int> keys;
    bool noMore = false;
    long seq = 0;
    std::list<Node>::iterator lit;
};
static void divide(const Node& n, const std::vector<Candidate>& K, Node c[4])
{
    const int halfX = (int)std::ceil((float)(n.URx - n.ULx) / 2);
    const int halfY = (int)std::ceil((float)(n.BRy - n.ULy) / 2);
    c[0].ULx = n.ULx; c[0].ULy = n.ULy;
    c[0].URx = n.ULx + halfX; c[0].URy = n.ULy;
    c[0].BLx = n.ULx; c[0].BLy = n.ULy + halfY;
    c[0].BRx = n.ULx + halfX; c[0].BRy = n.ULy + halfY;
    c[1].ULx = c[0].URx; c[1].ULy = c[0].URy;
    c[1].URx = n.URx; c[1].URy = n.URy;
    c[1].BLx = c[0].BRx; c[1].BLy = c[0].BRy;
    c[1].BRx = n.URx; c[1].BRy = n.ULy + halfY;
    c[2].ULx = c[0].BLx; c[2].ULy = c[0].BLy;
    c[2].URx = c[0].BRx; c[2].URy = c[0].BRy;
    c[2].BLx = n.BLx; c[2].BLy = n.BLy;
    c[2].BRx = c[0].BRx; c[2].BRy = n.BLy;
    c[3].ULx = c[2].URx; c[3].ULy = c[2].URy;
    c[3].URx = c[1].BRx; c[3].URy = c[1].BRy;
    c[3].BLx = c[2].BRx; c[3].BLy = c[2].BRy;
    c[3].BRx = n.BRx; c[3].BRy = n.BRy;
    for (int idx : n.keys) {
        const float px = (float)K[idx].x, py = (float)K[idx].y;
        if (px < c[0].URx) {
            if (py < c[0].BRy) c[0].keys.push_back(idx);
            else c[2].keys.push_back(idx);
        } else if (py < c[0].BRy) c[1].keys.push_back(idx);
        else c[3].keys.push_back(idx);
    }
    for (int k = 0; k < 4; k++)
        if (c[k].keys.size() == 1) c[k].noMore = true;
}
} // namespace

std::vector<int> OrbExtractor::distributeOctTree(const std::vector<Candidate>& K, int minX, int maxX, int minY,
                                                 int maxY, int N) const
{
    std::vector<int> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) throw std::runtime_error("DistributeOctTree: nIni < 1 (reference divides by zero)");
    const float hX = (float)(maxX - minX) / nIni;
    std::list<Node> L;
    std::vector<Node*> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        Node n;
        n.ULx = (int)(hX * (float)i); n.ULy = 0;
        n.URx = (int)(hX * (float)(i + 1)); n.URy = 0;
        n.BLx = n.ULx; n.BLy = maxY - minY;
        n.BRx = n.URx; n.BRy = maxY - minY;
        n.seq = seq++;
        L.push_back(n);
        ini[i] = &L.back();
    }
    for (size_t i = 0; i < K.size(); i++) {
        const size_t bin = (size_t)((float)K[i].x / hX);
        if (bin >= (size_t)nIni) throw std::runtime_error("DistributeOctTree: key outside the root nodes");
        ini[bin]->keys.push_back((int)i);
    }
    for (auto it = L.begin(); it != L.end();) {
        if (it->keys.size() == 1) { it->noMore = true; ++it; }
        else if (it->keys.empty()) it = L.erase(it);
        else ++it;
    }
    bool finish = false;
    typedef std::pair<int, long> SizeSeq;
    std::vector<std::pair<SizeSeq, Node*>> sizeAndNode;
    auto push_children = [&](Node c[4], int* nToExpand) {
        for (int k = 0; k < 4; k++) {
            if (c[k].keys.empty()) continue;
            c[k].seq = seq++;
            L.push_front(c[k]);
            if (c[k].keys.size() > 1) {
                if (nToExpand) (*nToExpand)++;
                sizeAndNode.push_back({{(int)c[k].keys.size(), L.front().seq}, &L.front()});
                L.front().lit = L.begin();
            }
        }
    };
    while (!finish) {
        const int prevSize = (int)L.size();
        auto it = L.begin();
        int nToExpand = 0;
        sizeAndNode.clear();
        while (it != L.end()) {
            if (it->noMore) { ++it; continue; }
            Node c[4];
            divide(*it, K, c);
            push_children(c, &nToExpand);
            it = L.erase(it);
        }
        if ((int)L.size() >= N || (int)L.size() == prevSize) {
            finish = true;
        } else if ((int)L.size() + nToExpand * 3 > N) {
            while (!finish) {
                const int prev2 = (int)L.size();
                std::vector<std::pair<SizeSeq, Node*>> prevNodes = sizeAndNode;
                sizeAndNode.clear();
                std::sort(prevNodes.begin(), prevNodes.end(),
                          [](const std::pair<SizeSeq, Node*>& a, const std::pair<SizeSeq, Node*>& b) {
                              return a.first < b.first;
                          });
                for (int j = (int)prevNodes.size() - 1; j >= 0; j--) {
                    Node c[4];
                    divide(*prevNodes[j].second, K, c);
                    push_children(c, nullptr);
                    L.erase(prevNodes[j].second->lit);
                    if ((int)L.size() >= N) break;
                }
                if ((int)L.size() >= N || (int)L.size() == prev2) finish = true;
            }
        }
    }
    result.reserve(L.size());
    for (const Node& n : L) {
        int best = n.keys[0];
        int maxR = K[best].response;
        for (size_t k = 1; k < n.keys.size(); k++)
            if (K[n.keys[k]].response > maxR) { best = n.keys[k]; maxR = K[best].response; }
        result.push_back(best);
    }
    return result;
}

void OrbExtractor::blurLevel(int level)
{
    const LevelGeom& g = geom[level];
    const Image& im = pyramid[level];
    Image& b = blurred[level];
    b.w = g.w; b.h = g.h;
    b.px.assign((size_t)g.w * g.h, 0);
    gaussian_blur_7x7_s2_u8(im.px.data() + (size_t)kEdge * im.w + kEdge, g.w, g.h, im.w, b.px.data(), g.w);
}

/* ORBextractor::operator(), src/ORBextractor.cc:1043-1105 (+ :830-853 keypoint finishing) */
int OrbExtractor::extract(const uint8_t* gray, int w, int h, size_t stride)
{
    keypoints.clear();
    descriptors.clear();
    if (!gray || w <= 0 || h <= 0) return 0;
    computeGeometry(w, h);
    computePyramid(gray, w, h, stride);
    candidates.assign(nlevels, {});
    blurred.assign(nlevels, Image());
    std::vector<std::vector<KeyPoint>> all(nlevels);
    for (int l = 0; l < nlevels; l++) {
        const LevelGeom& g = geom[l];
        computeCandidates(l);
        const std::vector<Candidate>& K = candidates[l];
        std::vector<int> sel = distributeOctTree(K, g.minBX, g.maxBX, g.minBY, g.maxBY, g.quota);
        const int scaledPatch = (int)(kPatch * scale[l]);
        for (int idx : sel) {
            KeyPoint kp;
            kp.x = (float)K[idx].x + g.minBX;
            kp.y = (float)K[idx].y + g.minBY;
            kp.size = (float)scaledPatch;
            kp.angle = -1.f;
            kp.response = (float)K[idx].response;
            kp.octave = l;
            kp.class_id = -1;
            all[l].push_back(kp);
        }
    }
    for (int l = 0; l < nlevels; l++) {
        const Image& im = pyramid[l];
        const uint8_t* roi = im.px.data() + (size_t)kEdge * im.w + kEdge;
        for (KeyPoint& kp : all[l]) {
            const uint8_t* c = roi + (ptrdiff_t)round_he(kp.y) * im.w + round_he(kp.x);
            kp.angle = ic_angle(c, im.w, umax);
        }
    }
    size_t total = 0;
    for (int l = 0; l < nlevels; l++) total += all[l].size();
    descriptors.assign(total * 32, 0);
    size_t off = 0;
    for (int l = 0; l < nlevels; l++) {
        if (all[l].empty()) continue;
        blurLevel(l);
        const Image& b = blurred[l];
        for (KeyPoint& kp : all[l]) {
            const uint8_t* c = b.px.data() + (ptrdiff_t)round_he(kp.y) * b.w + round_he(kp.x);
            orb_descriptor(c, b.w, kp.angle, &descriptors[off * 32]);
            off++;
        }
        if (l != 0) {
            const float s = scale[l];
            for (KeyPoint& kp : all[l]) { kp.x *= s; kp.y *= s; }
        }
        keypoints.insert(keypoints.end(), all[l].begin(), all[l].end());
    }
    return (int)keypoints.size();
}

} // namespace orc
