/* matcher_caller.cpp - the thirteen methods of Planar_SLAM::ORBmatcher (reference include/ORBmatcher.h:41-84) called the way
 * src/Tracking.cc, src/LocalMapping.cc and src/LoopClosing.cc call them, on stand-in Frame / KeyFrame / MapPoint types that carry
 * the member names src/ORBmatcher.cc reads (mvKeys, mvKeysUn, mDescriptors, mvuRight, mvDepth, mvpMapPoints, mvbOutlier, mTcw,
 * GetMapPointMatches(), GetPose(), GetWorldPos(), Observations(), isBad(), mbTrackInView, mTrackProjX ...).  The frames live on
 * the HOST, as the reference's do: include/drfe_adaptor.hpp makes them resident (drfe_frame_load), flattens the pointer graph,
 * calls the C-ABI and writes pointers back.
 *
 *   matcher_caller scene.bin out.bin
 * scene.bin is written by tests/test_gpu_native.py (frames extracted through the ctypes path, a table of map points, a small
 * vocabulary); out.bin receives, per call, the return value and the resulting pointer vector as map-point ids, which the test
 * compares with what the ctypes path gives on device-resident slots of its own context. */
#include "drfe_adaptor.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>

using drfe_cv::KeyPoint;
using drfe_cv::Mat;

namespace {

Mat f32mat(int rows, int cols, const float* src)
{
    Mat m(rows, cols, 4);
    std::memcpy(m.data, src, sizeof(float) * (size_t)rows * cols);
    return m;
}

struct KeyFrame;

struct MapPoint {
    int id = -1;
    float world[3], normal[3], minDist = 0, maxDist = 0;
    uint8_t desc[32];
    int nObs = 0;
    bool mbBad = false;
    MapPoint* mpReplaced = nullptr;
    std::map<unsigned long, size_t> mObservations;           /* keyframe id -> index */
    /* fields Frame::isInFrustum leaves (include/MapPoint.h:88-94) */
    bool mbTrackInView = false; int mnTrackScaleLevel = 0; float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0;

    Mat GetWorldPos() const { return f32mat(3, 1, world); }
    Mat GetNormal() const { return f32mat(3, 1, normal); }
    Mat GetDescriptor() const { Mat m(1, 32); std::memcpy(m.data, desc, 32); return m; }
    float GetMinDistanceInvariance() const { return minDist; }
    float GetMaxDistanceInvariance() const { return maxDist; }
    int Observations() const { return nObs; }
    bool isBad() const { return mbBad; }
    template <class KF> bool IsInKeyFrame(KF* pKF) const { return mObservations.count(pKF->mnId) != 0; }
    template <class KF> int GetIndexInKeyFrame(KF* pKF) const { auto it = mObservations.find(pKF->mnId); return it == mObservations.end() ? -1 : (int)it->second; }
    template <class KF> void AddObservation(KF* pKF, size_t idx) { if (!mObservations.count(pKF->mnId)) { mObservations[pKF->mnId] = idx; nObs++; } }
    void Replace(MapPoint* pMP) { if (pMP->id == id) return; mbBad = true; mpReplaced = pMP; }      /* the observations' transfer is the map's business */
};

struct FrameBase {
    unsigned long mnId = 0;
    int N = 0;
    std::vector<KeyPoint> mvKeys, mvKeysUn;
    Mat mDescriptors;
    std::vector<float> mvuRight, mvDepth;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    Mat mTcw;
    float Ow[3];
    float fx, fy, cx, cy, mbf, mb;
    float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
struct Frame : FrameBase {};
struct KeyFrame : FrameBase {
    std::vector<MapPoint*> GetMapPointMatches() const { return mvpMapPoints; }
    MapPoint* GetMapPoint(size_t idx) const { return mvpMapPoints[idx]; }
    std::set<MapPoint*> GetMapPoints() const
    {
        std::set<MapPoint*> s;
        for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
        return s;
    }
    void AddMapPoint(MapPoint* p, size_t idx) { mvpMapPoints[idx] = p; }
    Mat GetPose() const { return mTcw; }
    Mat GetCameraCenter() const { return f32mat(3, 1, Ow); }
};

struct Reader {
    std::vector<uint8_t> buf; size_t at = 0;
    explicit Reader(const char* path)
    {
        FILE* f = std::fopen(path, "rb");
        if (!f) { std::perror(path); std::exit(2); }
        std::fseek(f, 0, SEEK_END);
        buf.resize((size_t)std::ftell(f));
        std::fseek(f, 0, SEEK_SET);
        if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fprintf(stderr, "short read\n"); std::exit(2); }
        std::fclose(f);
    }
    template <class T> T get() { T v; std::memcpy(&v, buf.data() + at, sizeof(T)); at += sizeof(T); return v; }
    void bytes(void* dst, size_t n) { if (at + n > buf.size()) { std::fprintf(stderr, "scene file too short\n"); std::exit(2); } std::memcpy(dst, buf.data() + at, n); at += n; }
};

struct Writer {
    FILE* f;
    void rec(int ret, const std::vector<int32_t>& v)
    {
        const int32_t h[2] = {ret, (int32_t)v.size()};
        std::fwrite(h, 4, 2, f);
        if (!v.empty()) std::fwrite(v.data(), 4, v.size(), f);
    }
};

std::vector<int32_t> ids(const std::vector<MapPoint*>& v)
{
    std::vector<int32_t> o(v.size());
    for (size_t i = 0; i < v.size(); i++) o[i] = v[i] ? v[i]->id : -1;
    return o;
}

template <class F> void fill_frame(F& f, const FrameBase& src, unsigned long id) { static_cast<FrameBase&>(f) = src; f.mnId = id; }

}  // namespace

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: matcher_caller scene.bin out.bin\n"); return 2; }
    Reader R(argv[1]);
    if (R.get<int32_t>() != 0x4d415443) { std::fprintf(stderr, "bad scene file\n"); return 2; }
    const int nframes = R.get<int32_t>(), M = R.get<int32_t>(), nVoc = R.get<int32_t>(), vocK = R.get<int32_t>(), vocL = R.get<int32_t>();
    float cam[9]; R.bytes(cam, sizeof(cam));
    const int nfeatures = R.get<int32_t>(), nlevels = R.get<int32_t>(), iniTh = R.get<int32_t>(), minTh = R.get<int32_t>();
    const float scaleFactor = R.get<float>();
    std::vector<FrameBase> fb((size_t)nframes);
    std::vector<std::vector<int32_t>> frameMp((size_t)nframes);
    for (int f = 0; f < nframes; f++) {
        FrameBase& b = fb[(size_t)f];
        const int N = R.get<int32_t>();
        b.N = N;
        b.mvKeys.resize((size_t)N); R.bytes(b.mvKeys.data(), sizeof(KeyPoint) * (size_t)N);
        b.mvKeysUn = b.mvKeys;
        b.mDescriptors = Mat(N, 32); R.bytes(b.mDescriptors.data, (size_t)N * 32);
        b.mvuRight.resize((size_t)N); R.bytes(b.mvuRight.data(), 4 * (size_t)N);
        b.mvDepth.resize((size_t)N); R.bytes(b.mvDepth.data(), 4 * (size_t)N);
        float T[16]; R.bytes(T, 64); b.mTcw = f32mat(4, 4, T);
        R.bytes(b.Ow, 12);
        frameMp[(size_t)f].resize((size_t)N); R.bytes(frameMp[(size_t)f].data(), 4 * (size_t)N);
        std::vector<uint8_t> o((size_t)N); R.bytes(o.data(), (size_t)N);
        b.mvbOutlier.assign(o.begin(), o.end());
        b.fx = cam[0]; b.fy = cam[1]; b.cx = cam[2]; b.cy = cam[3]; b.mbf = cam[4]; b.mb = cam[4] / cam[0];
        b.mnMinX = cam[5]; b.mnMaxX = cam[6]; b.mnMinY = cam[7]; b.mnMaxY = cam[8];
    }
    std::vector<MapPoint> mps((size_t)M);
    for (int i = 0; i < M; i++) {
        MapPoint& p = mps[(size_t)i];
        p.id = i;
        R.bytes(p.world, 12); R.bytes(p.normal, 12); p.minDist = R.get<float>(); p.maxDist = R.get<float>();
        R.bytes(p.desc, 32);
        p.nObs = R.get<int32_t>(); p.mbBad = R.get<int32_t>() != 0;
        const int kf = R.get<int32_t>(), idx = R.get<int32_t>();
        if (kf >= 0) p.mObservations[(unsigned long)kf] = (size_t)idx;
        p.mbTrackInView = R.get<int32_t>() != 0; p.mnTrackScaleLevel = R.get<int32_t>();
        p.mTrackProjX = R.get<float>(); p.mTrackProjY = R.get<float>(); p.mTrackProjXR = R.get<float>(); p.mTrackViewCos = R.get<float>();
    }
    std::vector<int32_t> parent((size_t)nVoc); R.bytes(parent.data(), 4 * (size_t)nVoc);
    std::vector<uint8_t> vdesc((size_t)nVoc * 32); R.bytes(vdesc.data(), vdesc.size());
    std::vector<double> vweight((size_t)nVoc); R.bytes(vweight.data(), 8 * (size_t)nVoc);
    std::vector<uint8_t> isLeaf((size_t)nVoc); R.bytes(isLeaf.data(), (size_t)nVoc);
    float F12[9], R12[9], t12[3], Scw8[16], Scw12[16];
    R.bytes(F12, 36); const float s12 = R.get<float>(); R.bytes(R12, 36); R.bytes(t12, 12); R.bytes(Scw8, 64); R.bytes(Scw12, 64);

    auto attach = [&](FrameBase& f, int src) {
        f.mvpMapPoints.assign((size_t)f.N, nullptr);
        for (int i = 0; i < f.N; i++) if (frameMp[(size_t)src][(size_t)i] >= 0) f.mvpMapPoints[(size_t)i] = &mps[(size_t)frameMp[(size_t)src][(size_t)i]];
    };
    /* Frames and KeyFrames of the same images: KeyFrame ids = frame index (the map points' observations name them) */
    std::vector<Frame> F((size_t)nframes);
    std::vector<KeyFrame> KF((size_t)nframes);
    for (int f = 0; f < nframes; f++) {
        fill_frame(F[(size_t)f], fb[(size_t)f], 100 + (unsigned long)f); attach(F[(size_t)f], f);
        fill_frame(KF[(size_t)f], fb[(size_t)f], (unsigned long)f); attach(KF[(size_t)f], f);
    }

    Writer W{std::fopen(argv[2], "wb")};
    if (!W.f) { std::perror(argv[2]); return 2; }
    try {
        using Planar_SLAM::MatcherDevice;
        using Planar_SLAM::ORBmatcher;
        /* three slots for four frames + four keyframes: residency is exercised, evictions included */
        MatcherDevice dev(nfeatures, scaleFactor, nlevels, iniTh, minTh, /*slots=*/3);
        dev.UploadVocabulary(vocK, vocL, 0, 0, nVoc, parent.data(), vdesc.data(), vweight.data(), isLeaf.data(), /*levelsup=*/vocL - 2);
        ORBmatcher::BindThread(&dev);

        /* 1  Tracking::TrackWithMotionModel, src/Tracking.cc:2181: frame 1 against frame 0 (frame 1 arrives with a few claims) */
        {
            Frame cur = F[1];
            ORBmatcher matcher(0.9f, true);
            const int n = matcher.SearchByProjection(cur, F[0], 15.f, false);
            W.rec(n, ids(cur.mvpMapPoints));
        }
        /* 2  the fallback, src/Tracking.cc:2198 */
        {
            Frame cur = F[2];
            ORBmatcher matcher(0.9f, true);
            const int n = matcher.MatchORBPoints(cur, F[0]);
            W.rec(n, ids(cur.mvpMapPoints));
        }
        /* 3  Tracking::SearchLocalPoints, src/Tracking.cc:3317: every map point carries the fields isInFrustum left for frame 1 */
        {
            Frame cur = F[1];
            std::vector<MapPoint*> local;
            for (MapPoint& p : mps) local.push_back(&p);
            ORBmatcher matcher(0.8f);
            const int n = matcher.SearchByProjection(cur, local, 3.f);
            W.rec(n, ids(cur.mvpMapPoints));
        }
        /* 4  TrackReferenceKeyFrame, src/Tracking.cc:2320 */
        {
            Frame cur = F[1];
            std::vector<MapPoint*> vpMapPointMatches;
            ORBmatcher matcher(0.7f, true);
            const int n = matcher.SearchByBoW(&KF[0], cur, vpMapPointMatches);
            W.rec(n, ids(vpMapPointMatches));
        }
        /* 5  LoopClosing::ComputeSim3, src/LoopClosing.cc:311 */
        std::vector<MapPoint*> vpMatches12;
        {
            ORBmatcher matcher(0.75f, true);
            const int n = matcher.SearchByBoW(&KF[0], &KF[3], vpMatches12);
            W.rec(n, ids(vpMatches12));
        }
        /* 6  LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:480 */
        {
            std::vector<std::pair<size_t, size_t>> pairs;
            ORBmatcher matcher(0.6f, false);
            const int n = matcher.SearchForTriangulation(&KF[1], &KF[2], f32mat(3, 3, F12), pairs, false);
            std::vector<int32_t> flat;
            for (auto& pr : pairs) { flat.push_back((int32_t)pr.first); flat.push_back((int32_t)pr.second); }
            W.rec(n, flat);
        }
        /* 7  ComputeSim3 again: SearchBySim3 on top of the BoW matches, src/LoopClosing.cc:395 */
        {
            std::vector<MapPoint*> m12 = vpMatches12;
            for (size_t i = 0; i < m12.size(); i += 3) m12[i] = nullptr;       /* keep two thirds as "already matched" */
            ORBmatcher matcher(0.75f, true);
            const int n = matcher.SearchBySim3(&KF[0], &KF[3], m12, s12, f32mat(3, 3, R12), f32mat(3, 1, t12), 7.5f);
            W.rec(n, ids(m12));
        }
        /* 8  ComputeSim3: SearchByProjection(pKF, Scw, vpLoopMapPoints, vpMatched, 10), src/LoopClosing.cc:438 */
        {
            std::vector<MapPoint*> pts;
            for (MapPoint* p : KF[0].mvpMapPoints) if (p) { pts.push_back(p); pts.push_back(p); }     /* twice: first come, first served */
            std::vector<MapPoint*> vpMatched((size_t)KF[3].N, nullptr);
            for (int k = 0; k < KF[3].N; k += 5) vpMatched[(size_t)k] = KF[3].mvpMapPoints[(size_t)k];
            ORBmatcher matcher(0.75f, true);
            const int n = matcher.SearchByProjection(&KF[3], f32mat(4, 4, Scw8), pts, vpMatched, 10);
            W.rec(n, ids(vpMatched));
        }
        /* 9  Tracking::Relocalization, src/Tracking.cc:3638 */
        {
            Frame cur = F[2];
            std::set<MapPoint*> sFound;
            for (int i = 0; i < cur.N; i++) if (cur.mvpMapPoints[(size_t)i]) sFound.insert(cur.mvpMapPoints[(size_t)i]);
            for (size_t i = 0; i < KF[0].mvpMapPoints.size(); i += 7) if (KF[0].mvpMapPoints[i]) sFound.insert(KF[0].mvpMapPoints[i]);
            ORBmatcher matcher(0.9f, true);
            const int n = matcher.SearchByProjection(cur, &KF[0], sFound, 10.f, 100);
            W.rec(n, ids(cur.mvpMapPoints));
        }
        /* 10  Tracking::MonocularInitialization, src/Tracking.cc:1120 */
        {
            std::vector<drfe_cv::Point2f> prev((size_t)F[0].N);
            for (int i = 0; i < F[0].N; i++) prev[(size_t)i] = F[0].mvKeysUn[(size_t)i].pt;
            std::vector<int> m12;
            ORBmatcher matcher(0.9f, true);
            const int n = matcher.SearchForInitialization(F[0], F[1], prev, m12, 100);
            std::vector<int32_t> out(m12.begin(), m12.end());
            for (auto& p : prev) { int32_t b; std::memcpy(&b, &p.x, 4); out.push_back(b); std::memcpy(&b, &p.y, 4); out.push_back(b); }
            W.rec(n, out);
        }
        /* 11  LocalMapping::SearchInNeighbors, src/LocalMapping.cc:1043: the points of keyframe 0 (with NULLs and repeats) into keyframe 1 */
        {
            std::vector<MapPoint*> pts = KF[0].mvpMapPoints;
            for (size_t i = 0; i + 1 < pts.size(); i += 11) pts[i + 1] = pts[i];              /* repeats: the second copy meets what the first left */
            ORBmatcher matcher;
            const int n = matcher.Fuse(&KF[1], pts, 3.0f);
            std::vector<int32_t> out = ids(KF[1].mvpMapPoints);
            for (const MapPoint& p : mps) { out.push_back(p.mbBad ? 1 : 0); out.push_back(p.mpReplaced ? p.mpReplaced->id : -1); out.push_back(p.nObs); }
            W.rec(n, out);
        }
        /* 12  LoopClosing::SearchAndFuse, src/LoopClosing.cc:612 */
        {
            std::vector<MapPoint*> pts;
            for (MapPoint* p : KF[3].mvpMapPoints) if (p) pts.push_back(p);
            std::vector<MapPoint*> vpReplace(pts.size(), nullptr);
            ORBmatcher matcher(0.8f);
            const int n = matcher.Fuse(&KF[2], f32mat(4, 4, Scw12), pts, 4.f, vpReplace);
            std::vector<int32_t> out = ids(vpReplace);
            const std::vector<int32_t> kf = ids(KF[2].mvpMapPoints);
            out.insert(out.end(), kf.begin(), kf.end());
            W.rec(n, out);
        }
        /* 13  DescriptorDistance(cv::Mat, cv::Mat) */
        {
            Mat a(1, 32), b(1, 32);
            std::memcpy(a.data, mps[0].desc, 32); std::memcpy(b.data, mps[(size_t)M - 1].desc, 32);
            W.rec(ORBmatcher::DescriptorDistance(a, b), {(int32_t)dev.loads()});
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "matcher_caller: %s\n", e.what());
        return 1;
    }
    std::fclose(W.f);
    std::printf("matcher ok\n");
    return 0;
}
