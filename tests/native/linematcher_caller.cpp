/* linematcher_caller.cpp - the ten methods of Planar_SLAM::LSDmatcher (reference include/LSDmatcher.h:21-36) called the way
 * src/Tracking.cc (:1697, :2189, :2323, :2445, :2572) and src/LocalMapping.cc (:606, :858, :1103, :1124) call them, on stand-in
 * Frame / KeyFrame / MapLine types that carry the member names src/LSDmatcher.cpp reads (NL, mvKeylinesUn, mLdesc, mvpMapLines,
 * mvbLineOutlier, mTcw; mvKeyLines, mLineDescriptors, GetMapLineMatches(), GetMapLine(), GetMapLines(), AddMapLine(), GetPose();
 * GetWorldPos() as six doubles, GetNormal(), GetDescriptor(), Observations(), isBad(), Replace(), AddObservation(),
 * GetIndexInKeyFrame(), mbTrackInView, mTrackProjX1 ...).  MapLine::Replace moves the replaced line's observations to its
 * successor as src/MapLine.cpp:178-214 does (ReplaceMapLineMatch / EraseMapLineMatch on the observing keyframes), so a later line
 * of the same Fuse loop meets what an earlier one left.  Also LineSegment through an UNINITIALISED-style use: a default-constructed
 * object and the process-wide binding (include/Frame.h:157 never initialises mpLineSegment).
 *
 *   linematcher_caller scene.bin out.bin
 * scene.bin is written by tests/test_gpu_native.py::test_cpp_lsdmatcher_ten_methods; out.bin receives, per call, the return value and
 * the resulting pointer vectors as map-line ids, which the test compares with the ctypes path + its own replay of the surgery. */
#include "drfe_adaptor.hpp"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <thread>

using drfe_cv::KeyLine;
using drfe_cv::Mat;

namespace {

Mat f32mat(int rows, int cols, const float* src)
{
    Mat m(rows, cols, 4);
    std::memcpy(m.data, src, sizeof(float) * (size_t)rows * cols);
    return m;
}

/* Eigen::Matrix<double, N, 1> as far as the matcher reads it: operator()(int) */
template <int N> struct VecNd { double v[N]; double operator()(int i) const { return v[i]; } };

struct KeyFrame;

struct MapLine {
    int mnId = -1;
    double world[6], normal[3];
    float minDist = 0, maxDist = 0;
    uint8_t desc[32];
    int nObs = 0;
    bool mbBad = false;
    MapLine* mpReplaced = nullptr;
    std::map<KeyFrame*, size_t> mObservations;
    /* fields Frame::isInFrustum(MapLine*) leaves (include/MapLine.h) */
    bool mbTrackInView = false; int mnTrackScaleLevel = 0; float mTrackProjX1 = 0, mTrackProjY1 = 0, mTrackProjX2 = 0, mTrackProjY2 = 0, mTrackViewCos = 0;

    VecNd<6> GetWorldPos() const { VecNd<6> p; std::memcpy(p.v, world, 48); return p; }
    VecNd<3> GetNormal() const { VecNd<3> p; std::memcpy(p.v, normal, 24); return p; }
    Mat GetDescriptor() const { Mat m(1, 32); std::memcpy(m.data, desc, 32); return m; }
    float GetMinDistanceInvariance() const { return minDist; }
    float GetMaxDistanceInvariance() const { return maxDist; }
    int Observations() const { return nObs; }
    bool isBad() const { return mbBad; }
    bool IsInKeyFrame(KeyFrame* pKF) const { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame* pKF) const { auto it = mObservations.find(pKF); return it == mObservations.end() ? -1 : (int)it->second; }
    void AddObservation(KeyFrame* pKF, size_t idx) { if (mObservations.count(pKF)) return; mObservations[pKF] = idx; nObs++; }      /* src/MapLine.cpp:91-100 */
    void Replace(MapLine* pML);                                                                                                  /* src/MapLine.cpp:178-214 */
};

struct LineFrameBase {
    unsigned long mnId = 0;
    int NL = 0;
    std::vector<KeyLine> mvKeylinesUn;       /* Frame's name */
    Mat mLdesc;
    std::vector<MapLine*> mvpMapLines;
    std::vector<bool> mvbLineOutlier;
    Mat mTcw;
    float fx, fy, cx, cy, mbf, mb;
    float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
struct Frame : LineFrameBase {};
struct KeyFrame : LineFrameBase {
    std::vector<KeyLine> mvKeyLines;         /* KeyFrame's names (include/KeyFrame.h:209-211) */
    Mat mLineDescriptors;
    std::vector<MapLine*> GetMapLineMatches() const { return mvpMapLines; }
    MapLine* GetMapLine(size_t idx) const { return mvpMapLines[idx]; }
    std::set<MapLine*> GetMapLines() const
    {
        std::set<MapLine*> s;
        for (MapLine* p : mvpMapLines) if (p && !p->isBad()) s.insert(p);
        return s;
    }
    void AddMapLine(MapLine* p, size_t idx) { mvpMapLines[idx] = p; }
    Mat GetPose() const { return mTcw; }
};

void MapLine::Replace(MapLine* pML)
{
    if (pML->mnId == mnId) return;
    std::map<KeyFrame*, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    mpReplaced = pML;
    for (auto& o : obs) {
        if (!pML->IsInKeyFrame(o.first)) { o.first->mvpMapLines[o.second] = pML; pML->AddObservation(o.first, o.second); }
        else o.first->mvpMapLines[o.second] = nullptr;
    }
}

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
    template <class T> T get() { T v; bytes(&v, sizeof(T)); return v; }
    void bytes(void* dst, size_t n) { if (at + n > buf.size()) { std::fprintf(stderr, "scene file too short\n"); std::exit(2); } std::memcpy(dst, buf.data() + at, n); at += n; }
    std::vector<int32_t> ints() { const int n = get<int32_t>(); std::vector<int32_t> v((size_t)n); bytes(v.data(), 4 * (size_t)n); return v; }
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

std::vector<int32_t> ids(const std::vector<MapLine*>& v)
{
    std::vector<int32_t> o(v.size());
    for (size_t i = 0; i < v.size(); i++) o[i] = v[i] ? v[i]->mnId : -1;
    return o;
}

KeyLine to_keyline(const drfe_keyline& k)
{
    KeyLine o;
    static_assert(sizeof(KeyLine) == sizeof(drfe_keyline), "stand-in layout");
    std::memcpy(&o, &k, sizeof(o));
    return o;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: linematcher_caller scene.bin out.bin\n"); return 2; }
    Reader R(argv[1]);
    if (R.get<int32_t>() != 0x4c494e45) { std::fprintf(stderr, "bad scene file\n"); return 2; }
    const int nML = R.get<int32_t>(), nFrames = R.get<int32_t>();
    float cam[9]; R.bytes(cam, sizeof(cam));
    const int nfeatures = R.get<int32_t>(), nlevels = R.get<int32_t>(), iniTh = R.get<int32_t>(), minTh = R.get<int32_t>();
    const float scaleFactor = R.get<float>();

    std::vector<MapLine> ml((size_t)nML);
    std::vector<std::pair<int, int>> mlObs((size_t)nML);            /* (keyframe index, key line) a line is observed at, or (-1, -) */
    for (int i = 0; i < nML; i++) {
        MapLine& p = ml[(size_t)i];
        p.mnId = i;
        R.bytes(p.world, 48); R.bytes(p.normal, 24); p.minDist = R.get<float>(); p.maxDist = R.get<float>();
        R.bytes(p.desc, 32);
        p.nObs = R.get<int32_t>(); p.mbBad = R.get<int32_t>() != 0;
        mlObs[(size_t)i].first = R.get<int32_t>(); mlObs[(size_t)i].second = R.get<int32_t>();
        p.mbTrackInView = R.get<int32_t>() != 0; p.mnTrackScaleLevel = R.get<int32_t>();
        p.mTrackProjX1 = R.get<float>(); p.mTrackProjY1 = R.get<float>(); p.mTrackProjX2 = R.get<float>(); p.mTrackProjY2 = R.get<float>();
        p.mTrackViewCos = R.get<float>();
    }
    std::vector<Frame> F((size_t)nFrames);
    std::vector<KeyFrame> KF((size_t)nFrames);
    for (int f = 0; f < nFrames; f++) {
        LineFrameBase b;
        const int n = R.get<int32_t>();
        b.NL = n; b.mnId = (unsigned long)f;
        std::vector<drfe_keyline> kl((size_t)n); R.bytes(kl.data(), sizeof(drfe_keyline) * (size_t)n);
        for (const drfe_keyline& k : kl) b.mvKeylinesUn.push_back(to_keyline(k));
        b.mLdesc = Mat(n, 32); if (n) R.bytes(b.mLdesc.data, (size_t)n * 32);
        b.mLdesc.rows = n;
        float T[16]; R.bytes(T, 64); b.mTcw = f32mat(4, 4, T);
        std::vector<int32_t> mid((size_t)n); R.bytes(mid.data(), 4 * (size_t)n);
        std::vector<uint8_t> o((size_t)n); R.bytes(o.data(), (size_t)n);
        b.mvbLineOutlier.assign(o.begin(), o.end());
        for (int i = 0; i < n; i++) b.mvpMapLines.push_back(mid[(size_t)i] >= 0 ? &ml[(size_t)mid[(size_t)i]] : nullptr);
        b.fx = cam[0]; b.fy = cam[1]; b.cx = cam[2]; b.cy = cam[3]; b.mbf = cam[4]; b.mb = cam[4] / cam[0];
        b.mnMinX = cam[5]; b.mnMaxX = cam[6]; b.mnMinY = cam[7]; b.mnMaxY = cam[8];
        static_cast<LineFrameBase&>(F[(size_t)f]) = b;
        static_cast<LineFrameBase&>(KF[(size_t)f]) = b;
        KF[(size_t)f].mvKeyLines = b.mvKeylinesUn; KF[(size_t)f].mLineDescriptors = b.mLdesc;
    }
    for (int i = 0; i < nML; i++) if (mlObs[(size_t)i].first >= 0) ml[(size_t)i].mObservations[&KF[(size_t)mlObs[(size_t)i].first]] = (size_t)mlObs[(size_t)i].second;

    /* scenario parameters */
    const std::vector<int32_t> localLines = R.ints();               /* call 2: vpMapLines (ids, -1 = NULL) */
    const float thLast = R.get<float>(), thMap = R.get<float>();
    const std::vector<int32_t> fuseLines = R.ints();                /* call 7 */
    const float thFuse = R.get<float>();
    float Scw9[16], Scw10[16]; R.bytes(Scw9, 64); R.bytes(Scw10, 64);
    const std::vector<int32_t> fuse2Lines = R.ints();               /* call 8 */
    const std::vector<int32_t> projLines = R.ints();                /* call 9: vpLines */
    const std::vector<int32_t> projMatched = R.ints();              /* call 9: vpMatched on entry */
    const std::vector<int32_t> sim3Matches = R.ints();              /* call 10: vpMatches12 on entry */
    const float s12 = R.get<float>(); float R12[9], t12[3]; R.bytes(R12, 36); R.bytes(t12, 12);
    const int w = R.get<int32_t>(), h = R.get<int32_t>();
    Mat image(h, w); R.bytes(image.data, (size_t)w * h);

    auto ptrs = [&](const std::vector<int32_t>& v) { std::vector<MapLine*> o; for (int32_t i : v) o.push_back(i >= 0 ? &ml[(size_t)i] : nullptr); return o; };

    Writer W{std::fopen(argv[2], "wb")};
    if (!W.f) { std::perror(argv[2]); return 2; }
    try {
        using Planar_SLAM::LSDmatcher;
        Planar_SLAM::ORBextractor ex(nfeatures, scaleFactor, nlevels, iniTh, minTh, w, h);
        LSDmatcher::BindThread(ex.context());

        /* 1  SearchByProjection(CurrentFrame = 0, LastFrame = 1, th, false) */
        {
            Frame cur = F[0];
            LSDmatcher lmatcher(0.9f, true);
            const int n = lmatcher.SearchByProjection(cur, F[1], thLast, false);
            W.rec(n, ids(cur.mvpMapLines));
        }
        /* 2  SearchByProjection(F = 0, local map lines, th) */
        {
            Frame cur = F[0];
            LSDmatcher lmatcher(0.9f, true);
            const int n = lmatcher.SearchByProjection(cur, ptrs(localLines), thMap);
            W.rec(n, ids(cur.mvpMapLines));
        }
        /* 3  SearchByDescriptor(pKF = 2, currentF = 3, vpMapLineMatches): src/Tracking.cc:2189 */
        {
            std::vector<MapLine*> vpMapLineMatches;
            LSDmatcher lmatcher;
            const int n = lmatcher.SearchByDescriptor(&KF[2], F[3], vpMapLineMatches);
            W.rec(n, ids(vpMapLineMatches));
        }
        /* 4  SearchByDescriptor(pKF = 2, pKF2 = 3, vpMapLineMatches) */
        {
            std::vector<MapLine*> vpMapLineMatches;
            LSDmatcher lmatcher;
            const int n = lmatcher.SearchByDescriptor(&KF[2], &KF[3], vpMapLineMatches);
            W.rec(n, ids(vpMapLineMatches));
        }
        /* 5  SerachForInitialize(InitialFrame = 2, CurrentFrame = 3, LineMatches): src/Tracking.cc:1697 */
        {
            std::vector<std::pair<int, int>> LineMatches;
            LSDmatcher lmatcher;
            const int n = lmatcher.SerachForInitialize(F[2], F[3], LineMatches);
            std::vector<int32_t> flat;
            for (auto& pr : LineMatches) { flat.push_back(pr.first); flat.push_back(pr.second); }
            W.rec(n, flat);
        }
        /* 6  SearchForTriangulation(pKF1 = 2, pKF2 = 3, vMatchedPairs): src/LocalMapping.cc:606 */
        {
            std::vector<std::pair<size_t, size_t>> vMatchedPairs;
            LSDmatcher lmatcher;
            const int n = lmatcher.SearchForTriangulation(&KF[2], &KF[3], vMatchedPairs);
            std::vector<int32_t> flat;
            for (auto& pr : vMatchedPairs) { flat.push_back((int32_t)pr.first); flat.push_back((int32_t)pr.second); }
            W.rec(n, flat);
        }
        /* 7  Fuse(pKF = 4, vpMapLines, th): src/LocalMapping.cc:1103 - lines with NULLs and repeats; afterwards the keyframe's
         *    assignment and every line's (bad, replaced-by, observations) */
        {
            LSDmatcher lmatcher;
            const int n = lmatcher.Fuse(&KF[4], ptrs(fuseLines), thFuse);
            std::vector<int32_t> out = ids(KF[4].mvpMapLines);
            for (const MapLine& p : ml) { out.push_back(p.mbBad ? 1 : 0); out.push_back(p.mpReplaced ? p.mpReplaced->mnId : -1); out.push_back(p.nObs); }
            W.rec(n, out);
        }
        /* 8  Fuse(pKF = 4, Scw, vpLines, 4.0, vpReplaceLine) on what call 7 left */
        {
            const std::vector<MapLine*> lines = ptrs(fuse2Lines);
            std::vector<MapLine*> vpReplaceLine(lines.size(), nullptr);
            LSDmatcher lmatcher;
            const int n = lmatcher.Fuse(&KF[4], f32mat(4, 4, Scw9), lines, 4.f, vpReplaceLine);
            std::vector<int32_t> out = ids(vpReplaceLine);
            const std::vector<int32_t> kf = ids(KF[4].mvpMapLines);
            out.insert(out.end(), kf.begin(), kf.end());
            W.rec(n, out);
        }
        /* 9  SearchByProjection(pKF = 4, Scw, vpLines, vpMatched, 10) */
        {
            std::vector<MapLine*> vpMatched = ptrs(projMatched);
            LSDmatcher lmatcher;
            const int n = lmatcher.SearchByProjection(&KF[4], f32mat(4, 4, Scw10), ptrs(projLines), vpMatched, 10);
            W.rec(n, ids(vpMatched));
        }
        /* 10  SearchBySim3(pKF1 = 5, pKF2 = 6, vpMatches12, s12, R12, t12, 7.5) */
        {
            std::vector<MapLine*> vpMatches12 = ptrs(sim3Matches);
            LSDmatcher lmatcher;
            const int n = lmatcher.SearchBySim3(&KF[5], &KF[6], vpMatches12, s12, f32mat(3, 3, R12), f32mat(3, 1, t12), 7.5f);
            W.rec(n, ids(vpMatches12));
        }
        /* 11  DescriptorDistance(cv::Mat, cv::Mat) */
        W.rec(LSDmatcher::DescriptorDistance(ml[0].GetDescriptor(), ml[(size_t)nML - 1].GetDescriptor()), {});
        /* 12  LineSegment as Frame::ExtractLSD meets it: an object nobody constructed with a context, on a fresh thread, the
         *     process-wide binding doing the work (include/Frame.h:157, src/Frame.cc:129) */
        {
            LineSegment::BindProcess(ex.context());
            std::vector<KeyLine> kl; Mat ldesc; std::vector<drfe_cv::Vector3d> lf;
            std::string err;
            std::thread t([&] {
                try { LineSegment ls; ls.ExtractLineSegment(image, kl, ldesc, lf); } catch (const std::exception& e) { err = e.what(); }
            });
            t.join();
            if (!err.empty()) throw std::runtime_error(err);
            std::vector<int32_t> out((size_t)ldesc.rows * 8);
            if (ldesc.rows) std::memcpy(out.data(), ldesc.data, (size_t)ldesc.rows * 32);
            for (const KeyLine& k : kl) { int32_t b; std::memcpy(&b, &k.startPointX, 4); out.push_back(b); std::memcpy(&b, &k.endPointY, 4); out.push_back(b); }
            W.rec((int)kl.size(), out);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "linematcher_caller: %s\n", e.what());
        return 1;
    }
    std::fclose(W.f);
    std::printf("linematcher ok\n");
    return 0;
}
