/* oracle/match_oracle.cpp — TEST INFRASTRUCTURE (see oracle.h; parity unpinned vs OpenCV BFMatcher/gemm).
 *
 * CPU restatement of the Frame glue and ORBmatcher paths on the hot path (SURVEY.md §8a a-9..a-12,
 * a-16, a-21).  Map points are modelled as plain records (world position, descriptor,
 * Observations()>0 flag); MapPoint* identity becomes an int id (-1 == NULL).
 */
#include "oracle.h"
#include "match_oracle.h"
#include "oracle_math.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

namespace orc {

/* Frame::ComputeStereoFromRGBD, src/Frame.cc:893-911 (float->int truncation of (v,u), §9.13) */
void Frame::computeStereoFromRGBD(const float* depth, int dw, int dh)
{
    uRight.assign(N, -1.f);
    this->depth.assign(N, -1.f);
    for (int i = 0; i < N; i++) {
        const int v = (int)keys[i].y, u = (int)keys[i].x;
        if (v < 0 || v >= dh || u < 0 || u >= dw) continue; /* reference would read out of bounds */
        const float d = depth[(size_t)v * dw + u];
        if (d > 0) {
            this->depth[i] = d;
            uRight[i] = keysUn[i].x - bf / d;
        }
    }
}

/* cvUndistortPointsInternal, OpenCV 3.4 modules/imgproc/src/undistort.cpp (R = identity, P = K, no tilt
 * terms, TermCriteria(MAX_ITER, 5, 0.01) -> exactly five iterations): K and the coefficients are converted
 * to double, the point is normalised, the distortion is inverted by fixed-point iteration
 *   r2 = x^2+y^2; icdist = (1 + ((k7 r2 + k6) r2 + k5) r2) / (1 + ((k4 r2 + k1) r2 + k0) r2)
 *   dX = 2 k2 x y + k3 (r2 + 2 x^2) + k8 r2 + k9 r2^2;  dY = k2 (r2 + 2 y^2) + 2 k3 x y + k10 r2 + k11 r2^2
 *   x = (x0 - dX) icdist; y = (y0 - dY) icdist
 * and re-projected with P: xx = P00 x + P01 y + P02, ..., x = xx * (1 / ww); stored as float.  Terms whose
 * coefficient is zero are kept (they add exact zeros), so the rounding sequence is the library's. */
void undistort_points(const float* xy, int n, const float K[4], const float* dist, int nd, float* out)
{
    double k[14] = {0};
    for (int i = 0; i < nd && i < 14; i++) k[i] = (double)dist[i];
    const double fx = (double)K[0], fy = (double)K[1], cx = (double)K[2], cy = (double)K[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};
    for (int i = 0; i < n; i++) {
        double x = (double)xy[2 * i], y = (double)xy[2 * i + 1];
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        out[2 * i] = (float)(xx * ww);
        out[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::ComputeImageBounds, src/Frame.cc:862-891 */
void image_bounds(int cols, int rows, const float K[4], const float* dist, int nd, float out[4])
{
    if (nd > 0 && dist[0] != 0.0f) {
        const float c[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
        float u[8];
        undistort_points(c, 4, K, dist, nd, u);
        out[0] = std::min(u[0], u[4]);
        out[1] = std::max(u[2], u[6]);
        out[2] = std::min(u[1], u[3]);
        out[3] = std::max(u[5], u[7]);
    } else {
        out[0] = 0.f; out[1] = (float)cols; out[2] = 0.f; out[3] = (float)rows;
    }
}

/* Frame::PosInGrid + AssignFeaturesToGrid, src/Frame.cc:224-237, 816-825 (C round(), §9.19) */
void Frame::assignFeaturesToGrid()
{
    for (int i = 0; i < kGridCols; i++)
        for (int j = 0; j < kGridRows; j++) grid[i][j].clear();
    for (int i = 0; i < N; i++) {
        const int px = (int)std::round((keysUn[i].x - minX) * gridInvW);
        const int py = (int)std::round((keysUn[i].y - minY) * gridInvH);
        if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) continue;
        grid[px][py].push_back(i);
    }
}

/* Frame::GetFeaturesInArea, src/Frame.cc:730-779 */
void Frame::getFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const
{
    out.clear();
    const int nMinCellX = std::max(0, (int)std::floor((x - minX - r) * gridInvW));
    if (nMinCellX >= kGridCols) return;
    const int nMaxCellX = std::min(kGridCols - 1, (int)std::ceil((x - minX + r) * gridInvW));
    if (nMaxCellX < 0) return;
    const int nMinCellY = std::max(0, (int)std::floor((y - minY - r) * gridInvH));
    if (nMinCellY >= kGridRows) return;
    const int nMaxCellY = std::min(kGridRows - 1, (int)std::ceil((y - minY + r) * gridInvH));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<int>& cell = grid[ix][iy];
            for (int idx : cell) {
                const KeyPoint& kp = keysUn[idx];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float dx = kp.x - x, dy = kp.y - y;
                if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(idx);
            }
        }
}

/* cv::Mat (3x3)*(3x1)+(3x1) CV_32F: OpenCV gemm small-matrix path: float dot, then
 * (float)(t*alpha + c*beta) in double (exact for two floats). */
static void mat3_mul_add(const float R[9], const float x[3], const float t[3], float out[3])
{
    for (int r = 0; r < 3; r++) {
        const float d = R[r * 3 + 0] * x[0] + R[r * 3 + 1] * x[1] + R[r * 3 + 2] * x[2];
        out[r] = (float)((double)d * 1.0 + (double)t[r] * 1.0);
    }
}

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1666-1707 */
static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

/* ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono), src/ORBmatcher.cc:1396-1535 */
int search_by_projection_last(const Frame& Cur, const Frame& Last, const float TcwCur[16], const float TcwLast[16],
                              const MapPointRec* lastMP, float th, bool bMono, bool checkOri,
                              const uint8_t* curClaimObsPositive, int* curMP)
{
    const int HISTO_LENGTH = 30, TH_HIGH = 100;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    float Rcw[9], tcw[3], Rlw[9], tlw[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { Rcw[r * 3 + c] = TcwCur[r * 4 + c]; Rlw[r * 3 + c] = TcwLast[r * 4 + c]; }
        tcw[r] = TcwCur[r * 4 + 3];
        tlw[r] = TcwLast[r * 4 + 3];
    }
    /* twc = -Rcw.t()*tcw : general gemm path (transposed operand), double accumulation, alpha = -1 */
    float twc[3];
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[k * 3 + i] * (double)tcw[k];
        twc[i] = (float)(s * -1.0);
    }
    float tlc[3];
    mat3_mul_add(Rlw, twc, tlw, tlc);
    const bool bForward = tlc[2] > Cur.mb && !bMono;
    const bool bBackward = -tlc[2] > Cur.mb && !bMono;
    /* claim bookkeeping: obs flag of whatever currently sits in curMP[i2] */
    std::vector<uint8_t> claimObs(Cur.N, 0);
    for (int i = 0; i < Cur.N; i++)
        if (curMP[i] >= 0) claimObs[i] = curClaimObsPositive ? curClaimObsPositive[i] : 1;
    std::vector<int> cand;
    for (int i = 0; i < Last.N; i++) {
        const MapPointRec& mp = lastMP[i];
        if (!mp.valid) continue; /* pMP && !mvbOutlier[i] */
        float x3Dc[3];
        mat3_mul_add(Rcw, mp.world, tcw, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / (double)x3Dc[2]);
        if (invzc < 0) continue;
        const float u = Cur.fx * xc * invzc + Cur.cx;
        const float v = Cur.fy * yc * invzc + Cur.cy;
        if (u < Cur.minX || u > Cur.maxX) continue;
        if (v < Cur.minY || v > Cur.maxY) continue;
        const int nLastOctave = Last.keys[i].octave;
        const float radius = th * Cur.scaleFactors[nLastOctave];
        if (bForward) Cur.getFeaturesInArea(u, v, radius, nLastOctave, -1, cand);
        else if (bBackward) Cur.getFeaturesInArea(u, v, radius, 0, nLastOctave, cand);
        else Cur.getFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1, cand);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : cand) {
            if (curMP[i2] >= 0 && claimObs[i2]) continue;
            if (Cur.uRight[i2] > 0) {
                const float ur = u - Cur.bf * invzc;
                const float er = std::fabs(ur - Cur.uRight[i2]);
                if (er > radius) continue;
            }
            const int dist = descriptor_distance_swar(mp.desc, &Cur.desc[(size_t)i2 * 32]);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            curMP[bestIdx2] = i;
            claimObs[bestIdx2] = mp.obsPositive;
            nmatches++;
            if (checkOri) {
                float rot = Last.keysUn[i].angle - Cur.keysUn[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { curMP[idx] = -1; nmatches--; }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), src/ORBmatcher.cc:46-138 */
int search_by_projection_map(const Frame& F, const TrackedPointRec* mps, int M, float th, float nnratio,
                             const uint8_t* claimObsPositive, int* frameMP)
{
    const int TH_HIGH = 100;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<uint8_t> claimObs(F.N, 0);
    for (int i = 0; i < F.N; i++)
        if (frameMP[i] >= 0) claimObs[i] = claimObsPositive ? claimObsPositive[i] : 1;
    std::vector<int> cand;
    for (int iMP = 0; iMP < M; iMP++) {
        const TrackedPointRec& mp = mps[iMP];
        if (!mp.trackInView) continue;
        if (mp.bad) continue;
        const int lvl = mp.level;
        float r = ((double)mp.viewCos > 0.998) ? 2.5f : 4.0f;
        if (bFactor) r *= th;
        F.getFeaturesInArea(mp.projX, mp.projY, r * F.scaleFactors[lvl], lvl - 1, lvl, cand);
        if (cand.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : cand) {
            if (frameMP[idx] >= 0 && claimObs[idx]) continue;
            if (F.uRight[idx] > 0) {
                const float er = std::fabs(mp.projXR - F.uRight[idx]);
                if (er > r * F.scaleFactors[lvl]) continue;
            }
            const int dist = descriptor_distance_swar(mp.desc, &F.desc[(size_t)idx * 32]);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = F.keysUn[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = F.keysUn[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            frameMP[bestIdx] = iMP;
            claimObs[bestIdx] = mp.obsPositive;
            nmatches++;
        }
    }
    return nmatches;
}

/* cv::BFMatcher(NORM_HAMMING).knnMatch / match, SURVEY.md §10.7: ascending distance, ties -> lower
 * train index. idx/dist are nq x k, missing entries -1. */
void bf_knn_hamming(const uint8_t* Q, int nq, const uint8_t* T, int nt, int k, int32_t* idx, int32_t* dist)
{
    for (int q = 0; q < nq; q++) {
        int bi[2] = {-1, -1}, bd[2] = {1 << 30, 1 << 30};
        for (int t = 0; t < nt; t++) {
            const int d = descriptor_distance_swar(Q + (size_t)q * 32, T + (size_t)t * 32);
            if (d < bd[0]) { bd[1] = bd[0]; bi[1] = bi[0]; bd[0] = d; bi[0] = t; }
            else if (d < bd[1]) { bd[1] = d; bi[1] = t; }
        }
        for (int j = 0; j < k; j++) {
            idx[(size_t)q * k + j] = bi[j];
            dist[(size_t)q * k + j] = bi[j] >= 0 ? bd[j] : -1;
        }
    }
}

/* ORBmatcher::MatchORBPoints, src/ORBmatcher.cc:1332-1394 (incl. the mvbOutlier[i] quirk, §9.12) */
int match_orb_points(const uint8_t* curDesc, int curN, const uint8_t* lastDesc, int lastN,
                     const int32_t* lastMP /* id or -1 */, const uint8_t* lastOutlier, int32_t* curMP)
{
    if (curN == 0 || lastN == 0) return 0;
    std::vector<int32_t> idx(curN), dist(curN);
    bf_knn_hamming(curDesc, curN, lastDesc, lastN, 1, idx.data(), dist.data());
    double min_dist = 1000;
    for (int i = 0; i < curN; i++)
        if ((float)dist[i] < min_dist) min_dist = (float)dist[i];
    std::vector<int> good;
    for (int i = 0; i < curN; i++)
        if ((float)dist[i] < std::max(2 * min_dist, 15.0)) good.push_back(i);
    const int NPair = (int)good.size();
    for (int i = 0; i < NPair; i++) {
        const int mp = lastMP[idx[good[i]]];
        if (mp >= 0 && i < lastN && !lastOutlier[i]) curMP[good[i]] = mp;
    }
    return NPair;
}

/* Frame::lineDescriptorMAD, src/Frame.cc:560-584: 1.4826 * median absolute deviation of the NN distance
 * and of the NN2-NN1 gap.  dist: nq x 2 knn distances (DMatch::distance is float). */
void line_descriptor_mad(const int32_t* dist, int nq, double& nn_mad, double& nn12_mad)
{
    std::vector<float> d0(nq), gap(nq);
    for (int i = 0; i < nq; i++) { d0[i] = (float)dist[2 * i]; gap[i] = (float)dist[2 * i + 1] - (float)dist[2 * i]; }
    std::vector<float> s = d0;
    std::sort(s.begin(), s.end());
    const double nn_median = s[nq / 2];
    for (int i = 0; i < nq; i++) s[i] = std::fabs((float)(d0[i] - nn_median));   /* fabsf(float - double) */
    std::sort(s.begin(), s.end());
    nn_mad = 1.4826 * s[nq / 2];
    std::vector<float> g = gap;
    std::sort(g.begin(), g.end(), [](float a, float b) { return a > b; });
    const double nn12_median = g[nq / 2];
    for (int i = 0; i < nq; i++) s[i] = std::fabs((float)(gap[i] - nn12_median));
    std::sort(s.begin(), s.end());
    nn12_mad = 1.4826 * s[nq / 2];
}

/* LSDmatcher::SearchByDescriptor(KeyFrame*, Frame&, vpMapLineMatches), src/LSDmatcher.cpp:242-279:
 * knnMatch(k=2) KF lines -> frame lines, accept d0/d1 < 1/1.5, later queries overwrite earlier ones.
 * out[tdx] = KF line index or -1. */
int lsd_search_by_descriptor(const uint8_t* descKF, int nKF, const uint8_t* kfHasLine, const uint8_t* descF, int nF,
                             int32_t* out)
{
    for (int i = 0; i < nF; i++) out[i] = -1;
    if (nKF == 0 || nF < 2) return 0;
    std::vector<int32_t> idx((size_t)nKF * 2), dist((size_t)nKF * 2);
    bf_knn_hamming(descKF, nKF, descF, nF, 2, idx.data(), dist.data());
    double nn_th, nn12_th;
    line_descriptor_mad(dist.data(), nKF, nn_th, nn12_th);   /* computed and unused by this overload */
    const float minRatio = 1.0f / 1.5f;
    int nmatches = 0;
    for (int q = 0; q < nKF; q++) {
        const double r = (float)dist[2 * q] / (float)dist[2 * q + 1];
        if (r < minRatio && kfHasLine[q]) { out[idx[2 * q]] = q; nmatches++; }
    }
    return nmatches;
}

/* LSDmatcher::SearchByDescriptor(KeyFrame*, KeyFrame*, ...), :281-314, and SerachForInitialize, :213-240:
 * accept when the NN2-NN1 gap exceeds half its MAD.  out[qdx] = train index or -1. */
int lsd_search_by_gap(const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, const uint8_t* trainHasLine, int32_t* out)
{
    for (int i = 0; i < nQ; i++) out[i] = -1;
    if (nQ == 0 || nT < 2) return 0;
    std::vector<int32_t> idx((size_t)nQ * 2), dist((size_t)nQ * 2);
    bf_knn_hamming(descQ, nQ, descT, nT, 2, idx.data(), dist.data());
    double nn_th, nn12_th;
    line_descriptor_mad(dist.data(), nQ, nn_th, nn12_th);
    nn12_th = nn12_th * 0.5;
    int nmatches = 0;
    for (int q = 0; q < nQ; q++) {
        const double gap = (float)dist[2 * q + 1] - (float)dist[2 * q];
        if (gap > nn12_th && (!trainHasLine || trainHasLine[idx[2 * q]])) { out[q] = idx[2 * q]; nmatches++; }
    }
    return nmatches;
}


/* LSDmatcher::SearchForTriangulation, :334-367 (LocalMapping::CreateNewMapLines): knnMatch(k = 2), accept the nearest
 * neighbour when the NN2-NN1 gap exceeds a TENTH of its MAD and neither line has a MapLine yet */
int lsd_search_for_triangulation(const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, const uint8_t* has1,
                                 const uint8_t* has2, int32_t* out12)
{
    for (int i = 0; i < n1; i++) out12[i] = -1;
    if (n1 == 0 || n2 < 2) return 0;
    std::vector<int32_t> idx((size_t)n1 * 2), dist((size_t)n1 * 2);
    bf_knn_hamming(desc1, n1, desc2, n2, 2, idx.data(), dist.data());
    double nn_th, nn12_th;
    line_descriptor_mad(dist.data(), n1, nn_th, nn12_th);
    nn12_th = nn12_th * 0.1;
    int nmatches = 0;
    for (int q = 0; q < n1; q++) {            /* sorted by queryIdx: knnMatch already returns that order */
        const int t = idx[2 * q];
        if (has1[q] || has2[t]) continue;
        const double gap = (float)dist[2 * q + 1] - (float)dist[2 * q];
        if (gap > nn12_th) { out12[q] = t; nmatches++; }
    }
    return nmatches;
}

/* ---------------------------------------------------------------------------------------------------- */
/* LSDmatcher::SearchByProjection (row a-15)                                                            */

/* Frame::GetLinesInArea, src/Frame.cc:781-813.  `0.5 * (x1 + x2)`: float sum, the rest of the distance in
 * double, stored to a float; slope in float, compared against the double `r * 0.01`; the level test
 * switches on with maxLevel > 0 (not >= 0 as in GetFeaturesInArea). */
void get_lines_in_area(const LineRec* lines, int n, float x1, float y1, float x2, float y2, float r, int minLevel,
                       int maxLevel, std::vector<int>& out)
{
    out.clear();
    const bool bCheckLevels = (minLevel > 0) || (maxLevel > 0);
    for (int i = 0; i < n; i++) {
        const LineRec& kl = lines[i];
        const double mx = 0.5 * (double)(x1 + x2) - (double)kl.ptX, my = 0.5 * (double)(y1 + y2) - (double)kl.ptY;
        const float distance = (float)(mx * mx + my * my);
        if (distance > r * r) continue;
        const float slope = (y1 - y2) / (x1 - x2) - kl.angle;
        if ((double)slope > (double)r * 0.01) continue;
        if (bCheckLevels) {
            if (kl.octave < minLevel) continue;
            if (maxLevel >= 0 && kl.octave > maxLevel) continue;
        }
        out.push_back(i);
    }
}

/* the candidate scan shared by both variants, src/LSDmatcher.cpp:95-136 / :168-209 */
static bool best_line(const std::vector<int>& cand, const LineRec* cur, const uint8_t* curDesc, const uint8_t* desc,
                      const int32_t* curML, const std::vector<uint8_t>& claimObs, float nnratio, int& bestIdx)
{
    const int TH_HIGH = 100;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1;
    bestIdx = -1;
    for (int idx : cand) {
        if (curML[idx] >= 0 && claimObs[idx]) continue;
        const int dist = descriptor_distance_swar(desc, curDesc + (size_t)idx * 32);
        if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = cur[idx].octave; bestIdx = idx;
        } else if (dist < bestDist2) {
            bestLevel2 = cur[idx].octave; bestDist2 = dist;
        }
    }
    if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) return false;
        return true;
    }
    return false;
}

int lsd_search_by_projection_last(const LineCamera& cam, const float TcwCur[16], const float TcwLast[16],
                                  const float* scaleFactors, const MapLineRec* last, int nLast, const LineRec* cur,
                                  const uint8_t* curDesc, int nCur, float th, bool bMono, float nnratio,
                                  const uint8_t* curObs, int32_t* curML)
{
    int nmatches = 0;
    float Rcw[9], tcw[3], Rlw[9], tlw[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { Rcw[r * 3 + c] = TcwCur[r * 4 + c]; Rlw[r * 3 + c] = TcwLast[r * 4 + c]; }
        tcw[r] = TcwCur[r * 4 + 3];
        tlw[r] = TcwLast[r * 4 + 3];
    }
    float twc[3], tlc[3];
    for (int i = 0; i < 3; i++) {   /* -Rcw.t()*tcw: general gemm path, double accumulation (as for points) */
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[k * 3 + i] * (double)tcw[k];
        twc[i] = (float)(s * -1.0);
    }
    mat3_mul_add(Rlw, twc, tlw, tlc);
    const bool bForward = tlc[2] > cam.mb && !bMono;
    const bool bBackward = -tlc[2] > cam.mb && !bMono;
    std::vector<uint8_t> claimObs(nCur, 0);
    for (int i = 0; i < nCur; i++)
        if (curML[i] >= 0) claimObs[i] = curObs ? curObs[i] : 1;
    std::vector<int> cand;
    for (int i = 0; i < nLast; i++) {
        const MapLineRec& ml = last[i];
        if (!ml.valid) continue;
        const float SP[3] = {(float)ml.world[0], (float)ml.world[1], (float)ml.world[2]};
        const float EP[3] = {(float)ml.world[3], (float)ml.world[4], (float)ml.world[5]};
        float SPc[3], EPc[3];
        mat3_mul_add(Rcw, SP, tcw, SPc);
        mat3_mul_add(Rcw, EP, tcw, EPc);
        if (SPc[2] < 0.0f || EPc[2] < 0.0f) continue;
        const float invz1 = 1.0f / SPc[2];
        const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
        if (u1 < cam.minX || u1 > cam.maxX) continue;
        if (v1 < cam.minY || v1 > cam.maxY) continue;
        const float invz2 = 1.0f / EPc[2];
        const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
        if (u2 < cam.minX || u2 > cam.maxX) continue;
        if (v2 < cam.minY || v2 > cam.maxY) continue;
        const int oct = ml.octave;
        const float radius = th * scaleFactors[oct];
        if (bForward) get_lines_in_area(cur, nCur, u1, v1, u2, v2, radius, oct, -1, cand);
        else if (bBackward) get_lines_in_area(cur, nCur, u1, v1, u2, v2, radius, 0, oct, cand);
        else get_lines_in_area(cur, nCur, u1, v1, u2, v2, radius, oct - 1, oct + 1, cand);
        if (cand.empty()) continue;
        int bestIdx;
        if (best_line(cand, cur, curDesc, ml.desc, curML, claimObs, nnratio, bestIdx)) {
            curML[bestIdx] = i;
            claimObs[bestIdx] = ml.obsPositive ? 1 : 0;
            nmatches++;
        }
    }
    return nmatches;
}

int lsd_search_by_projection_map(const float* scaleFactors, const TrackedLineRec* lines, int n, const LineRec* cur,
                                 const uint8_t* curDesc, int nCur, float th, float nnratio, const uint8_t* curObs,
                                 int32_t* curML)
{
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<uint8_t> claimObs(nCur, 0);
    for (int i = 0; i < nCur; i++)
        if (curML[i] >= 0) claimObs[i] = curObs ? curObs[i] : 1;
    std::vector<int> cand;
    for (int i = 0; i < n; i++) {
        const TrackedLineRec& tl = lines[i];
        if (!tl.inView) continue;
        float r = tl.viewCos > 0.998 ? 5.0f : 8.0f;   /* RadiusByViewingCos, :369-375 (double compare) */
        if (bFactor) r *= th;
        get_lines_in_area(cur, nCur, tl.x1, tl.y1, tl.x2, tl.y2, r * scaleFactors[tl.level], tl.level - 1, tl.level, cand);
        if (cand.empty()) continue;
        int bestIdx;
        if (best_line(cand, cur, curDesc, tl.desc, curML, claimObs, nnratio, bestIdx)) {
            curML[bestIdx] = i;
            claimObs[bestIdx] = tl.obsPositive ? 1 : 0;
            nmatches++;
        }
    }
    return nmatches;
}


/* ---------------------------------------------------------------------------------------------------- */
/* Frame::isInFrustum                                                                                   */

/* mOw = -mRcw.t()*mtcw (Frame::UpdatePoseMatrices, src/Frame.cc:592-600): general gemm path, double
 * accumulation, alpha = -1 */
static void camera_centre(const float Rcw[9], const float tcw[3], float Ow[3])
{
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)Rcw[k * 3 + i] * (double)tcw[k];
        Ow[i] = (float)(s * -1.0);
    }
}
/* KeyFrame::GetCameraCenter(): Ow = -Rwc*tcw with Rwc = Rcw.t() materialised first (KeyFrame::SetPose, src/KeyFrame.cc:153-154):
 * no transpose flag on the product, so cv::gemm takes its small-matrix float path (float dot, then * alpha = -1) */
static void camera_centre_kf(const float Rcw[9], const float tcw[3], float Ow[3])
{
    for (int i = 0; i < 3; i++) {
        const float d = Rcw[0 * 3 + i] * tcw[0] + Rcw[1 * 3 + i] * tcw[1] + Rcw[2 * 3 + i] * tcw[2];
        Ow[i] = (float)((double)d * -1.0);
    }
}
/* cv::norm(3x1 CV_32F): double accumulation, sqrt, stored to float; Mat::dot likewise accumulates in double */
static float norm3(const float v[3]) { return (float)std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); }
static double dot3(const float a[3], const float b[3]) { return (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]; }

void is_in_frustum(const LineCamera& cam, float bf, const float Tcw[16], float logScaleFactor, int nLevels,
                   const FrustumPointRec* pts, int n, float viewingCosLimit, FrustumOut* out)
{
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
    camera_centre(Rcw, tcw, Ow);
    for (int i = 0; i < n; i++) {
        FrustumOut& o = out[i];
        o.inView = 0; o.level = 0; o.projX = o.projY = o.projXR = o.viewCos = 0.f;
        const FrustumPointRec& p = pts[i];
        float Pc[3];
        mat3_mul_add(Rcw, p.world, tcw, Pc);
        if (Pc[2] < 0.0f) continue;
        const float invz = 1.0f / Pc[2];
        const float u = cam.fx * Pc[0] * invz + cam.cx, v = cam.fy * Pc[1] * invz + cam.cy;
        if (u < cam.minX || u > cam.maxX) continue;
        if (v < cam.minY || v > cam.maxY) continue;
        const float maxDistance = 1.2f * p.maxDistance, minDistance = 0.8f * p.minDistance;
        const float PO[3] = {p.world[0] - Ow[0], p.world[1] - Ow[1], p.world[2] - Ow[2]};
        const float dist = norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        const float viewCos = (float)(dot3(PO, p.normal) / (double)dist);
        if (viewCos < viewingCosLimit) continue;
        /* MapPoint::PredictScale(dist, Frame*), src/MapPoint.cc:448-463 */
        const float ratio = p.maxDistance / dist;
        int nScale = (int)std::ceil(log_f(ratio) / logScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= nLevels) nScale = nLevels - 1;
        o.inView = 1; o.projX = u; o.projXR = u - bf * invz; o.projY = v; o.level = nScale; o.viewCos = viewCos;
    }
}

void is_in_frustum_lines(const LineCamera& cam, const float Tcw[16], float logScaleFactor, const FrustumLineRec* lines,
                         int n, float viewingCosLimit, FrustumLineOut* out)
{
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
    camera_centre(Rcw, tcw, Ow);
    for (int i = 0; i < n; i++) {
        FrustumLineOut& o = out[i];
        o.inView = 0; o.level = 0; o.x1 = o.y1 = o.x2 = o.y2 = o.viewCos = 0.f;
        const FrustumLineRec& l = lines[i];
        const float SP[3] = {(float)l.world[0], (float)l.world[1], (float)l.world[2]};
        const float EP[3] = {(float)l.world[3], (float)l.world[4], (float)l.world[5]};
        float SPc[3], EPc[3];
        mat3_mul_add(Rcw, SP, tcw, SPc);
        mat3_mul_add(Rcw, EP, tcw, EPc);
        if (SPc[2] < 0.0f || EPc[2] < 0.0f) continue;
        const float invz1 = 1.0f / SPc[2];
        const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
        if (u1 < cam.minX || u1 > cam.maxX) continue;
        if (v1 < cam.minY || v1 > cam.maxY) continue;
        const float invz2 = 1.0f / EPc[2];
        const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
        if (u2 < cam.minX || u2 > cam.maxX) continue;
        if (v2 < cam.minY || v2 > cam.maxY) continue;
        const float maxDistance = 1.2f * l.maxDistance, minDistance = 0.8f * l.minDistance;
        /* OM = 0.5*(SP+EP) - mOw: float sum, exact halving, float difference (cv::addWeighted in float) */
        float OM[3];
        for (int k = 0; k < 3; k++) OM[k] = (SP[k] + EP[k]) * 0.5f - Ow[k];
        const float dist = norm3(OM);
        if (dist < minDistance || dist > maxDistance) continue;
        const float pn[3] = {(float)l.normal[0], (float)l.normal[1], (float)l.normal[2]};
        const float viewCos = (float)(dot3(OM, pn) / (double)dist);
        if (viewCos < viewingCosLimit) continue;
        /* MapLine::PredictScale, src/MapLine.cpp:381-390: no clamping */
        const float ratio = l.maxDistance / dist;
        o.inView = 1; o.x1 = u1; o.y1 = v1; o.x2 = u2; o.y2 = v2; o.viewCos = viewCos;
        o.level = (int)std::ceil(log_f(ratio) / logScaleFactor);
    }
}


/* Scw -> [Rcw | tcw], src/ORBmatcher.cc:989-993: Mat::dot accumulates in double, sqrt in double, stored to float;
 * Mat / float goes through convertTo with the float scale (float)(1.0 / s) */
void decompose_sim3(const float Scw[16], float Tcw[16])
{
    const double d = (double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2];
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) Tcw[r * 4 + c] = Scw[r * 4 + c] * inv;
        Tcw[r * 4 + 3] = Scw[r * 4 + 3] * inv;
    }
    Tcw[12] = Tcw[13] = Tcw[14] = 0.f; Tcw[15] = 1.f;
}

/* ---------------------------------------------------------------------------------------------------- */
/* ORBmatcher::Fuse (search part)                                                                       */

void fuse_search(const Frame& KF, const float Tcw[16], const float* invLevelSigma2, float logScaleFactor, int nLevels,
                 const FrustumPointRec* pts, const uint8_t* descs, const uint8_t* skip, int n, float th, int32_t* bestIdx,
                 int32_t* bestDist, bool sim3)
{
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
    if (sim3) camera_centre(Rcw, tcw, Ow);          /* `-Rcw.t()*tcw` at :993 */
    else camera_centre_kf(Rcw, tcw, Ow);            /* pKF->GetCameraCenter() at :840 */
    std::vector<int> cand;
    for (int i = 0; i < n; i++) {
        bestIdx[i] = -1; bestDist[i] = 256;
        if (skip && skip[i]) continue;
        const FrustumPointRec& p = pts[i];
        float p3Dc[3];
        mat3_mul_add(Rcw, p.world, tcw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = sim3 ? (float)(1.0 / (double)p3Dc[2]) : 1 / p3Dc[2];     /* `1.0/z` at :1023, `1/z` at :863 */
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = KF.fx * x + KF.cx, v = KF.fy * y + KF.cy;
        if (!(u >= KF.minX && u < KF.maxX && v >= KF.minY && v < KF.maxY)) continue;     /* KeyFrame::IsInImage */
        const float ur = u - KF.bf * invz;
        const float maxDistance = 1.2f * p.maxDistance, minDistance = 0.8f * p.minDistance;
        const float PO[3] = {p.world[0] - Ow[0], p.world[1] - Ow[1], p.world[2] - Ow[2]};
        const float dist3D = norm3(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (dot3(PO, p.normal) < 0.5 * (double)dist3D) continue;
        const float ratio = p.maxDistance / dist3D;
        int level = (int)std::ceil(log_f(ratio) / logScaleFactor);
        if (level < 0) level = 0;
        else if (level >= nLevels) level = nLevels - 1;
        const float radius = th * KF.scaleFactors[level];
        KF.getFeaturesInArea(u, v, radius, -1, -1, cand);          /* the KeyFrame version has no level filter */
        for (int idx : cand) {
            const KeyPoint& kp = KF.keysUn[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < level - 1 || kpLevel > level) continue;
            const float ex = u - kp.x, ey = v - kp.y;
            if (sim3) {
                /* the Sim3 overload has no reprojection gate */
            } else if (KF.uRight[idx] >= 0) {
                const float er = ur - KF.uRight[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if ((double)(e2 * invLevelSigma2[kpLevel]) > 7.8) continue;
            } else {
                const float e2 = ex * ex + ey * ey;
                if ((double)(e2 * invLevelSigma2[kpLevel]) > 5.99) continue;
            }
            const int dist = descriptor_distance_swar(descs + (size_t)i * 32, KF.desc.data() + (size_t)idx * 32);
            if (dist < bestDist[i]) { bestDist[i] = dist; bestIdx[i] = idx; }
        }
    }
}


/* ---------------------------------------------------------------------------------------------------- */
/* ORBmatcher::SearchBySim3                                                                             */

/* one direction (:1147-1203 / :1206-1262): points of `src` (camera pose Tsw) carried by (sR, t) into `dst` */
static void sim3_direction(const Frame& dst, const float Tsw[16], const float sR[9], const float t[3], float logScaleFactor,
                           int nLevels, const FrustumPointRec* pts, const uint8_t* descs, const uint8_t* skip, int n, float th,
                           std::vector<int>& match)
{
    const int TH_HIGH = 100;
    float Rsw[9], tsw[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rsw[r * 3 + c] = Tsw[r * 4 + c]; tsw[r] = Tsw[r * 4 + 3]; }
    match.assign(n, -1);
    std::vector<int> cand;
    for (int i = 0; i < n; i++) {
        if (skip[i]) continue;
        const FrustumPointRec& p = pts[i];
        float pa[3], pb[3];
        mat3_mul_add(Rsw, p.world, tsw, pa);
        mat3_mul_add(sR, pa, t, pb);
        if (pb[2] < 0.0) continue;
        const float invz = (float)(1.0 / (double)pb[2]);
        const float x = pb[0] * invz, y = pb[1] * invz;
        const float u = dst.fx * x + dst.cx, v = dst.fy * y + dst.cy;
        if (!(u >= dst.minX && u < dst.maxX && v >= dst.minY && v < dst.maxY)) continue;
        const float maxDistance = 1.2f * p.maxDistance, minDistance = 0.8f * p.minDistance;
        const float dist3D = norm3(pb);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float ratio = p.maxDistance / dist3D;
        int level = (int)std::ceil(log_f(ratio) / logScaleFactor);
        if (level < 0) level = 0;
        else if (level >= nLevels) level = nLevels - 1;
        const float radius = th * dst.scaleFactors[level];
        dst.getFeaturesInArea(u, v, radius, -1, -1, cand);
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : cand) {
            const int oct = dst.keysUn[idx].octave;
            if (oct < level - 1 || oct > level) continue;
            const int dist = descriptor_distance_swar(descs + (size_t)i * 32, dst.desc.data() + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_HIGH) match[i] = bestIdx;
    }
}

int search_by_sim3(const Frame& KF1, const Frame& KF2, const float T1w[16], const float T2w[16], float s12,
                   const float R12[9], const float t12[3], float logScaleFactor, int nLevels, const FrustumPointRec* pts1,
                   const uint8_t* descs1, const uint8_t* skip1, const FrustumPointRec* pts2, const uint8_t* descs2,
                   const uint8_t* skip2, float th, int32_t* out12)
{
    /* sR12 = s12*R12; sR21 = (1.0/s12)*R12.t(); t21 = -sR21*t12 (:1121-1123): scaling through convertTo with a float
     * factor, the product through the float small-matrix path with alpha = -1 */
    float sR12[9], sR21[9], t21[3];
    const float a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { sR12[r * 3 + c] = R12[r * 3 + c] * s12; sR21[r * 3 + c] = R12[c * 3 + r] * a21; }
    for (int r = 0; r < 3; r++) {
        const float d = sR21[r * 3] * t12[0] + sR21[r * 3 + 1] * t12[1] + sR21[r * 3 + 2] * t12[2];
        t21[r] = (float)((double)d * -1.0);
    }
    std::vector<int> m1, m2;
    sim3_direction(KF2, T1w, sR21, t21, logScaleFactor, nLevels, pts1, descs1, skip1, KF1.N, th, m1);
    sim3_direction(KF1, T2w, sR12, t12, logScaleFactor, nLevels, pts2, descs2, skip2, KF2.N, th, m2);
    int nFound = 0;
    for (int i1 = 0; i1 < KF1.N; i1++) {
        out12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { out12[i1] = idx2; nFound++; }
    }
    return nFound;
}


/* ---------------------------------------------------------------------------------------------------- */
/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, ...)                                                   */

int search_by_projection_kf(const Frame& KF, const float Scw[16], float logScaleFactor, int nLevels, const FrustumPointRec* pts,
                            const uint8_t* descs, const uint8_t* skip, int n, const uint8_t* matched, float th, int32_t* newMatch)
{
    const int TH_LOW = 50;
    float Tcw[16], Rcw[9], tcw[3], Ow[3];
    decompose_sim3(Scw, Tcw);                                                      /* :303-307 */
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
    camera_centre(Rcw, tcw, Ow);
    std::vector<uint8_t> taken(matched, matched + KF.N);
    for (int k = 0; k < KF.N; k++) newMatch[k] = -1;
    std::vector<int> cand;
    int nmatches = 0;
    for (int i = 0; i < n; i++) {
        if (skip && skip[i]) continue;                                             /* isBad() || spAlreadyFound.count(pMP) */
        const FrustumPointRec& p = pts[i];
        float p3Dc[3];
        mat3_mul_add(Rcw, p.world, tcw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = 1 / p3Dc[2];                                            /* :335 */
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = KF.fx * x + KF.cx, v = KF.fy * y + KF.cy;
        if (!(u >= KF.minX && u < KF.maxX && v >= KF.minY && v < KF.maxY)) continue;
        const float maxDistance = 1.2f * p.maxDistance, minDistance = 0.8f * p.minDistance;
        const float PO[3] = {p.world[0] - Ow[0], p.world[1] - Ow[1], p.world[2] - Ow[2]};
        const float dist = norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        if (dot3(PO, p.normal) < 0.5 * (double)dist) continue;
        const float ratio = p.maxDistance / dist;
        int level = (int)std::ceil(log_f(ratio) / logScaleFactor);
        if (level < 0) level = 0;
        else if (level >= nLevels) level = nLevels - 1;
        const float radius = th * KF.scaleFactors[level];
        KF.getFeaturesInArea(u, v, radius, -1, -1, cand);
        int bestDist = 256, bestIdx = -1;
        for (int idx : cand) {
            if (taken[idx]) continue;                                              /* vpMatched[idx] */
            const int kpLevel = KF.keysUn[idx].octave;
            if (kpLevel < level - 1 || kpLevel > level) continue;
            const int d = descriptor_distance_swar(descs + (size_t)i * 32, KF.desc.data() + (size_t)idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { taken[bestIdx] = 1; newMatch[bestIdx] = i; nmatches++; }
    }
    return nmatches;
}


/* ---------------------------------------------------------------------------------------------------- */
/* LSDmatcher::Fuse(KeyFrame*, vector<MapLine*>, th): the search, src/LSDmatcher.cpp:884-993               */

void lsd_fuse_search(const LineCamera& cam, const float Tcw[16], float logScaleFactor, const float* scaleFactors, int nLevels,
                     const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                     const uint8_t* kfDesc, int nKF, float th, int32_t* bestIdx, int32_t* bestDist)
{
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
    camera_centre_kf(Rcw, tcw, Ow);
    std::vector<int> cand;
    for (int i = 0; i < n; i++) {
        bestIdx[i] = -1; bestDist[i] = INT_MAX;
        if (skip && skip[i]) continue;                                            /* !pML || pML->isBad() */
        const FrustumLineRec& l = lines[i];
        const float SP[3] = {(float)l.world[0], (float)l.world[1], (float)l.world[2]};
        const float EP[3] = {(float)l.world[3], (float)l.world[4], (float)l.world[5]};
        float SPc[3], EPc[3];
        mat3_mul_add(Rcw, SP, tcw, SPc);
        mat3_mul_add(Rcw, EP, tcw, EPc);
        if (SPc[2] < 0.0f || EPc[2] < 0.0f) continue;
        const float invz1 = 1.0f / SPc[2];
        const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
        if (u1 < cam.minX || u1 > cam.maxX) continue;
        if (v1 < cam.minY || v1 > cam.maxY) continue;
        const float invz2 = 1.0f / EPc[2];
        const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
        if (u2 < cam.minX || u2 > cam.maxX) continue;
        if (v2 < cam.minY || v2 > cam.maxY) continue;
        const float maxDistance = 1.2f * l.maxDistance, minDistance = 0.8f * l.minDistance;
        float OM[3];
        for (int k = 0; k < 3; k++) OM[k] = (SP[k] + EP[k]) * 0.5f - Ow[k];
        const float dist = norm3(OM);
        if (dist < minDistance || dist > maxDistance) continue;
        const float pn[3] = {(float)l.normal[0], (float)l.normal[1], (float)l.normal[2]};
        if (dot3(OM, pn) < 0.5 * (double)dist) continue;                          /* :956 */
        const float ratio = l.maxDistance / dist;
        const int level = (int)std::ceil(log_f(ratio) / logScaleFactor);      /* MapLine::PredictScale: no clamp */
        if (level < 0 || level >= nLevels) { bestIdx[i] = -2; continue; }         /* mvScaleFactors[level] is out of bounds there */
        const float radius = th * scaleFactors[level];
        get_lines_in_area(kf, nKF, u1, v1, u2, v2, radius, -1, -1, cand);         /* KeyFrame::GetLinesInArea, src/KeyFrame.cc:749 */
        for (int idx : cand) {
            const int klLevel = kf[idx].octave;
            if (klLevel < level - 1 || klLevel > level) continue;
            const int d = descriptor_distance_swar(descs + (size_t)i * 32, kfDesc + (size_t)idx * 32);
            if (d < bestDist[i]) { bestDist[i] = d; bestIdx[i] = idx; }
        }
    }
}


/* ---------------------------------------------------------------------------------------------------- */
/* ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)                          */

int search_by_projection_reloc(const Frame& Cur, const float Tcw[16], float logScaleFactor, int nLevels, const FrustumPointRec* pts,
                               const uint8_t* descs, const float* kfAngles, const uint8_t* skip, int n, const uint8_t* matched,
                               float th, int orbDist, bool checkOri, int32_t* newMatch)
{
    const int HISTO_LENGTH = 30;
    float Rcw[9], tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
    camera_centre(Rcw, tcw, Ow);                                                  /* -Rcw.t()*tcw, :1543 */
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> taken(matched, matched + Cur.N);
    for (int k = 0; k < Cur.N; k++) newMatch[k] = -1;
    std::vector<int> cand;
    int nmatches = 0;
    for (int i = 0; i < n; i++) {
        if (skip && skip[i]) continue;
        const FrustumPointRec& p = pts[i];
        float x3Dc[3];
        mat3_mul_add(Rcw, p.world, tcw, x3Dc);
        if (!(x3Dc[2] != 0.0f)) continue;                                         /* documented deviation: z == 0 is skipped */
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / (double)x3Dc[2]);
        const float u = Cur.fx * xc * invzc + Cur.cx;
        const float v = Cur.fy * yc * invzc + Cur.cy;
        if (u < Cur.minX || u > Cur.maxX) continue;
        if (v < Cur.minY || v > Cur.maxY) continue;
        const float PO[3] = {p.world[0] - Ow[0], p.world[1] - Ow[1], p.world[2] - Ow[2]};
        const float dist3D = norm3(PO);
        const float maxDistance = 1.2f * p.maxDistance, minDistance = 0.8f * p.minDistance;
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float ratio = p.maxDistance / dist3D;
        int level = (int)std::ceil(log_f(ratio) / logScaleFactor);            /* MapPoint::PredictScale(dist, Frame*) */
        if (level < 0) level = 0;
        else if (level >= nLevels) level = nLevels - 1;
        const float radius = th * Cur.scaleFactors[level];
        Cur.getFeaturesInArea(u, v, radius, level - 1, level + 1, cand);
        if (cand.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int i2 : cand) {
            if (taken[i2]) continue;
            const int d = descriptor_distance_swar(descs + (size_t)i * 32, &Cur.desc[(size_t)i2 * 32]);
            if (d < bestDist) { bestDist = d; bestIdx2 = i2; }
        }
        if (bestDist <= orbDist) {
            taken[bestIdx2] = 1;
            newMatch[bestIdx2] = i;
            nmatches++;
            if (checkOri) {
                float rot = kfAngles[i] - Cur.keysUn[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int idx : rotHist[i]) { newMatch[idx] = -1; nmatches--; }
    }
    return nmatches;
}

/* ---------------------------------------------------------------------------------------------------- */
/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize), src/ORBmatcher.cc:409-524 */

int search_for_initialization(const Frame& F1, const Frame& F2, float* prevMatched /* F1.N x 2, in/out */, int windowSize,
                              float nnratio, bool checkOri, int32_t* matches12)
{
    const int HISTO_LENGTH = 30, TH_LOW = 50;
    int nmatches = 0;
    for (int i = 0; i < F1.N; i++) matches12[i] = -1;                             /* :412 */
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> matchedDistance(F2.N, INT_MAX), matches21(F2.N, -1);         /* :419-420 */
    std::vector<int> cand;
    for (int i1 = 0; i1 < F1.N; i1++) {
        const int level1 = F1.keysUn[i1].octave;
        if (level1 > 0) continue;                                                 /* :426 */
        F2.getFeaturesInArea(prevMatched[2 * i1], prevMatched[2 * i1 + 1], (float)windowSize, level1, level1, cand);   /* :429 */
        if (cand.empty()) continue;
        const uint8_t* d1 = &F1.desc[(size_t)i1 * 32];
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : cand) {
            const int dist = descriptor_distance_swar(d1, &F2.desc[(size_t)i2 * 32]);
            if (matchedDistance[i2] <= dist) continue;                            /* :448 */
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if ((float)bestDist < (float)bestDist2 * nnratio) {                   /* :465, int -> float as written */
                if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                matches21[bestIdx2] = i1;
                matchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = F1.keysUn[i1].angle - F2.keysUn[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }   /* :508-512: a replaced match is not counted twice */
        }
    }
    for (int i1 = 0; i1 < F1.N; i1++)                                             /* :519-521 */
        if (matches12[i1] >= 0) { prevMatched[2 * i1] = F2.keysUn[matches12[i1]].x; prevMatched[2 * i1 + 1] = F2.keysUn[matches12[i1]].y; }
    return nmatches;
}


/* ---------------------------------------------------------------------------------------------------- */
/* LSDmatcher's loop-closing variants (no caller in the reference; public API, src/LSDmatcher.cpp:377-882)  */

/* the common front of Fuse(KF, Scw, ...) / SearchByProjection(KF, Scw, ...): projection of both end points with the
 * decomposed similarity, image bounds, distance band, 60-degree cone, PredictScale, KeyFrame::GetLinesInArea and the
 * octave window (:408-470 / :783-845).  Returns false when the map line is dropped; level == -2 flags a predicted level
 * outside the pyramid (mvScaleFactors[level] is out of bounds in the reference). */
static bool lsd_scw_candidates(const LineCamera& cam, const float Rcw[9], const float tcw[3], const float Ow[3], float logScaleFactor,
                               const float* scaleFactors, int nLevels, const FrustumLineRec& l, const LineRec* kf, int nKF, float th,
                               int& level, std::vector<int>& cand)
{
    cand.clear();
    level = -1;
    const float SP[3] = {(float)l.world[0], (float)l.world[1], (float)l.world[2]};
    const float EP[3] = {(float)l.world[3], (float)l.world[4], (float)l.world[5]};
    float SPc[3], EPc[3];
    mat3_mul_add(Rcw, SP, tcw, SPc);
    mat3_mul_add(Rcw, EP, tcw, EPc);
    if (SPc[2] < 0.0f || EPc[2] < 0.0f) return false;
    const float invz1 = 1.0f / SPc[2];
    const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
    if (u1 < cam.minX || u1 > cam.maxX) return false;
    if (v1 < cam.minY || v1 > cam.maxY) return false;
    const float invz2 = 1.0f / EPc[2];
    const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
    if (u2 < cam.minX || u2 > cam.maxX) return false;
    if (v2 < cam.minY || v2 > cam.maxY) return false;
    const float maxDistance = 1.2f * l.maxDistance, minDistance = 0.8f * l.minDistance;
    float OM[3];
    for (int k = 0; k < 3; k++) OM[k] = (SP[k] + EP[k]) * 0.5f - Ow[k];
    const float dist = norm3(OM);
    if (dist < minDistance || dist > maxDistance) return false;
    const float pn[3] = {(float)l.normal[0], (float)l.normal[1], (float)l.normal[2]};
    if (dot3(OM, pn) < 0.5 * (double)dist) return false;
    const float ratio = l.maxDistance / dist;
    level = (int)std::ceil(log_f(ratio) / logScaleFactor);                    /* MapLine::PredictScale: no clamp */
    if (level < 0 || level >= nLevels) { level = -2; return true; }
    const float radius = th * scaleFactors[level];
    std::vector<int> area;
    get_lines_in_area(kf, nKF, u1, v1, u2, v2, radius, -1, -1, area);
    for (int idx : area) {
        const int klLevel = kf[idx].octave;
        if (klLevel < level - 1 || klLevel > level) continue;
        cand.push_back(idx);
    }
    return true;
}

/* LSDmatcher::Fuse(KeyFrame*, cv::Mat Scw, vpLines, th, vpReplaceLine), :750-882: the search.  bestIdx[i] = key line or -1
 * (-2: predicted level outside the pyramid), bestDist[i] its distance (INT_MAX if none); the caller applies TH_LOW */
void lsd_fuse_search_sim3(const LineCamera& cam, const float Scw[16], float logScaleFactor, const float* scaleFactors, int nLevels,
                          const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                          const uint8_t* kfDesc, int nKF, float th, int32_t* bestIdx, int32_t* bestDist)
{
    float T[16], Rcw[9], tcw[3], Ow[3];
    decompose_sim3(Scw, T);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = T[r * 4 + c]; tcw[r] = T[r * 4 + 3]; }
    camera_centre(Rcw, tcw, Ow);                                              /* -Rcw.t()*tcw, :763 */
    std::vector<int> cand;
    for (int i = 0; i < n; i++) {
        bestIdx[i] = -1; bestDist[i] = INT_MAX;
        if (skip && skip[i]) continue;                                        /* !pML || isBad() || spAlreadyFound.count(pML) */
        int level;
        if (!lsd_scw_candidates(cam, Rcw, tcw, Ow, logScaleFactor, scaleFactors, nLevels, lines[i], kf, nKF, th, level, cand)) continue;
        if (level == -2) { bestIdx[i] = -2; continue; }
        for (int idx : cand) {
            const int d = descriptor_distance_swar(descs + (size_t)i * 32, kfDesc + (size_t)idx * 32);
            if (d < bestDist[i]) { bestDist[i] = d; bestIdx[i] = idx; }
        }
    }
}

/* LSDmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, vpLines, vpMatched, th), :377-502: first come, first served over
 * the map lines - a key line whose vpMatched entry is set (on entry or by an earlier line) is no candidate.  matched[idx] != 0 on
 * entry = vpMatched[idx] != NULL; newMatch[idx] = the line that claimed key line idx in this call or -1.  A predicted level
 * outside the pyramid drops the line (the reference reads past mvScaleFactors there).  Returns nmatches. */
int lsd_search_by_projection_kf(const LineCamera& cam, const float Scw[16], float logScaleFactor, const float* scaleFactors, int nLevels,
                                const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                                const uint8_t* kfDesc, int nKF, const uint8_t* matched, int th, int32_t* newMatch)
{
    const int TH_LOW = 50;
    float T[16], Rcw[9], tcw[3], Ow[3];
    decompose_sim3(Scw, T);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = T[r * 4 + c]; tcw[r] = T[r * 4 + 3]; }
    camera_centre(Rcw, tcw, Ow);
    std::vector<uint8_t> taken(matched, matched + nKF);
    for (int k = 0; k < nKF; k++) newMatch[k] = -1;
    std::vector<int> cand;
    int nmatches = 0;
    for (int i = 0; i < n; i++) {
        if (skip && skip[i]) continue;
        int level;
        if (!lsd_scw_candidates(cam, Rcw, tcw, Ow, logScaleFactor, scaleFactors, nLevels, lines[i], kf, nKF, (float)th, level, cand)) continue;
        if (level == -2) continue;
        int bestDist = 256, bestIdx = -1;
        for (int idx : cand) {
            if (taken[idx]) continue;                                         /* :476 */
            const int d = descriptor_distance_swar(descs + (size_t)i * 32, kfDesc + (size_t)idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) { taken[bestIdx] = 1; newMatch[bestIdx] = i; nmatches++; }
    }
    return nmatches;
}

/* one direction of LSDmatcher::SearchBySim3 (:549-625 / :628-704): map lines of `src` (camera pose Tsw) carried by (sR, t)
 * into the other keyframe, whose key lines are kf / kfDesc */
static void lsd_sim3_direction(const LineCamera& cam, const float Tsw[16], const float sR[9], const float t[3], float logScaleFactor,
                               const float* scaleFactors, int nLevels, const FrustumLineRec* lines, const uint8_t* descs,
                               const uint8_t* skip, int n, const LineRec* kf, const uint8_t* kfDesc, int nKF, float th,
                               std::vector<int>& match)
{
    const int TH_HIGH = 100;
    float Rsw[9], tsw[3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rsw[r * 3 + c] = Tsw[r * 4 + c]; tsw[r] = Tsw[r * 4 + 3]; }
    match.assign(n, -1);
    std::vector<int> area;
    for (int i = 0; i < n; i++) {
        if (skip[i]) continue;
        const FrustumLineRec& l = lines[i];
        const float SP[3] = {(float)l.world[0], (float)l.world[1], (float)l.world[2]};
        const float EP[3] = {(float)l.world[3], (float)l.world[4], (float)l.world[5]};
        float a[3], SPc[3], EPc[3];
        mat3_mul_add(Rsw, SP, tsw, a);
        mat3_mul_add(sR, a, t, SPc);
        mat3_mul_add(Rsw, EP, tsw, a);
        mat3_mul_add(sR, a, t, EPc);
        if (SPc[2] < 0.0f || EPc[2] < 0.0f) continue;
        const float invz1 = 1.0f / SPc[2];
        const float u1 = cam.fx * SPc[0] * invz1 + cam.cx, v1 = cam.fy * SPc[1] * invz1 + cam.cy;
        if (!(u1 >= cam.minX && u1 < cam.maxX && v1 >= cam.minY && v1 < cam.maxY)) continue;          /* KeyFrame::IsInImage */
        const float invz2 = 1.0f / EPc[2];
        const float u2 = cam.fx * EPc[0] * invz2 + cam.cx, v2 = cam.fy * EPc[1] * invz2 + cam.cy;
        if (!(u2 >= cam.minX && u2 < cam.maxX && v2 >= cam.minY && v2 < cam.maxY)) continue;
        const float maxDistance = 1.2f * l.maxDistance, minDistance = 0.8f * l.minDistance;
        const float mid[3] = {(SPc[0] + EPc[0]) * 0.5f, (SPc[1] + EPc[1]) * 0.5f, (SPc[2] + EPc[2]) * 0.5f};
        const float dist3D = norm3(mid);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float ratio = l.maxDistance / dist3D;
        const int level = (int)std::ceil(log_f(ratio) / logScaleFactor);      /* MapLine::PredictScale: no clamp */
        if (level < 0 || level >= nLevels) continue;                              /* out-of-bounds read in the reference: dropped */
        const float radius = th * scaleFactors[level];
        get_lines_in_area(kf, nKF, u1, v1, u2, v2, radius, -1, -1, area);
        int bestDist = INT_MAX, bestIdx = -1;
        for (int idx : area) {
            const int klLevel = kf[idx].octave;
            if (klLevel < level - 1 || klLevel > level) continue;
            const int d = descriptor_distance_swar(descs + (size_t)i * 32, kfDesc + (size_t)idx * 32);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_HIGH) match[i] = bestIdx;
    }
}

/* LSDmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), :504-748.  lines / descs / skip per key line of the
 * keyframe (skip = !pML || isBad() || already matched); kf1 / kf2 = the key lines themselves.  out12[i1] = i2 or -1. */
int lsd_search_by_sim3(const LineCamera& cam, const float T1w[16], const float T2w[16], float s12, const float R12[9],
                       const float t12[3], float logScaleFactor, const float* scaleFactors, int nLevels, const FrustumLineRec* lines1,
                       const uint8_t* descs1, const uint8_t* skip1, const LineRec* kf1, const uint8_t* kf1Desc, int n1,
                       const FrustumLineRec* lines2, const uint8_t* descs2, const uint8_t* skip2, const LineRec* kf2,
                       const uint8_t* kf2Desc, int n2, float th, int32_t* out12)
{
    float sR12[9], sR21[9], t21[3];
    const float a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { sR12[r * 3 + c] = R12[r * 3 + c] * s12; sR21[r * 3 + c] = R12[c * 3 + r] * a21; }
    for (int r = 0; r < 3; r++) {
        const float d = sR21[r * 3] * t12[0] + sR21[r * 3 + 1] * t12[1] + sR21[r * 3 + 2] * t12[2];
        t21[r] = (float)((double)d * -1.0);
    }
    std::vector<int> m1, m2;
    lsd_sim3_direction(cam, T1w, sR21, t21, logScaleFactor, scaleFactors, nLevels, lines1, descs1, skip1, n1, kf2, kf2Desc, n2, th, m1);
    lsd_sim3_direction(cam, T2w, sR12, t12, logScaleFactor, scaleFactors, nLevels, lines2, descs2, skip2, n2, kf1, kf1Desc, n1, th, m2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        out12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { out12[i1] = idx2; nFound++; }
    }
    return nFound;
}

} // namespace orc
