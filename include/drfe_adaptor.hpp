/* drfe_adaptor.hpp — header-only C++ adaptor: the reference's class interfaces over the C-ABI of drfe.h.
 *
 * What a DR-SLAM maintainer drops in place of src/ORBextractor.cc, src/LSDextractor.cpp and src/PlaneExtractor.cpp
 * (INTEGRATION.md): same class names, constructor arguments, methods and public members as
 *   Planar_SLAM::ORBextractor          include/ORBextractor.h:51-85
 *   LineSegment::ExtractLineSegment    include/LSDextractor.h:342-350
 *   Planar_SLAM::PlaneDetection        include/PlaneExtractor.h:61-82
 *   Planar_SLAM::PlaneDetection_CAPE   include/PlaneExtractor.h:84-115
 *   Planar_SLAM::ORBmatcher            include/ORBmatcher.h:41-84  - the reference's thirteen signatures (templates over Frame / KeyFrame / MapPoint)
 *   Planar_SLAM::LSDmatcher            include/LSDmatcher.h:21-36  - the reference's ten signatures (templates over Frame / KeyFrame / MapLine)
 * With -DDRFE_WITH_OPENCV the container types are OpenCV's (cv::Mat, cv::KeyPoint, cv::line_descriptor::KeyLine);
 * without it (this image has no OpenCV) minimal stand-ins with the same member names and memory layout are used, so the
 * header is compiled and exercised here (tests/native/adaptor_caller.cpp, run by tests/test_gpu_native.py).
 * Compiled against nothing but drfe.h; link with -ldrfe.  Errors of the C-ABI become std::runtime_error, as the
 * reference's constructors would throw; operator() keeps the reference's silent return on an empty image. */
#ifndef DRFE_ADAPTOR_HPP
#define DRFE_ADAPTOR_HPP

#include "drfe.h"

#include <cassert>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef DRFE_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
namespace drfe_cv {
using Mat = cv::Mat;
using KeyPoint = cv::KeyPoint;
using KeyLine = cv::line_descriptor::KeyLine;
inline const uint8_t* mat_data(const Mat& m) { return m.data; }
inline size_t mat_step(const Mat& m) { return m.step; }
inline Mat mat_u8(int rows, int cols) { return Mat(rows, cols, CV_8U); }
}  // namespace drfe_cv
#else
namespace drfe_cv {
struct Point2f { float x, y; };
/* cv::KeyPoint: pt, size, angle, response, octave, class_id (7 x 4 bytes) */
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
/* the fields of cv::line_descriptor::KeyLine, in its declaration order */
struct KeyLine {
    float angle; int class_id, octave; Point2f pt; float response, size;
    float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int numOfPixels;
};
/* continuous single-channel 8-bit matrix: the subset of cv::Mat the adaptor touches */
struct Mat {
    int rows = 0, cols = 0; size_t step = 0; uint8_t* data = nullptr; int elem = 1;
    std::shared_ptr<std::vector<uint8_t>> own;
    Mat() {}
    Mat(int r, int c, int elemSize = 1) : rows(r), cols(c), step((size_t)c * elemSize), elem(elemSize),
                                          own(std::make_shared<std::vector<uint8_t>>((size_t)r * c * elemSize)) { data = own->data(); }
    Mat(int r, int c, uint8_t* external, size_t stepBytes, int elemSize = 1) : rows(r), cols(c), step(stepBytes), data(external), elem(elemSize) {}
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    void release() { rows = cols = 0; step = 0; data = nullptr; own.reset(); }
    template <class T> T* ptr(int r) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <class T> const T* ptr(int r) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
};
inline const uint8_t* mat_data(const Mat& m) { return m.data; }
inline size_t mat_step(const Mat& m) { return m.step; }
inline Mat mat_u8(int rows, int cols) { return Mat(rows, cols); }
}  // namespace drfe_cv
#endif

/* Eigen::Vector3d where Eigen is present (VertexType of include/PlaneExtractor.h:30, keylineFunctions of include/LSDextractor.h:349);
 * a three-double stand-in with the same operator[] / operator() / layout otherwise (this image has no Eigen) */
#ifdef DRFE_WITH_EIGEN
#include <Eigen/Dense>
namespace drfe_cv { using Vector3d = Eigen::Vector3d; }
#else
namespace drfe_cv {
struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double x, double y, double z) : v{x, y, z} {}
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    const double& operator()(int i) const { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
};
}  // namespace drfe_cv
#endif
static_assert(sizeof(drfe_cv::Vector3d) == 3 * sizeof(double), "Vector3d is three doubles");

static_assert(sizeof(drfe_cv::KeyPoint) == sizeof(drfe_keypoint), "cv::KeyPoint and drfe_keypoint must share one layout");
#ifndef DRFE_WITH_OPENCV
static_assert(sizeof(drfe_cv::KeyLine) == sizeof(drfe_keyline), "KeyLine stand-in and drfe_keyline must share one layout");
#endif

namespace Planar_SLAM {

namespace drfe_detail {
inline void check(int rc, drfe_ctx* c, const char* what)
{
    if (rc != DRFE_OK) throw std::runtime_error(std::string(what) + ": " + drfe_last_error(c));
}
struct CtxDeleter { void operator()(drfe_ctx* c) const { drfe_destroy(c); } };
using CtxPtr = std::shared_ptr<drfe_ctx>;
inline CtxPtr make_ctx(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxW, int maxH, int maxBatch,
                       int device)
{
    drfe_config cfg = {device, maxW, maxH, maxBatch, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
    drfe_ctx* c = nullptr;
    if (drfe_create(&cfg, &c) != DRFE_OK) throw std::runtime_error(std::string("drfe_create: ") + drfe_last_error(nullptr));
    return CtxPtr(c, CtxDeleter());
}
/* the context the calling thread's matchers / line extractor work on (ORBmatcher::BindThread, LSDmatcher::BindThread) */
inline drfe_ctx*& thread_ctx() { static thread_local drfe_ctx* c = nullptr; return c; }
}  // namespace drfe_detail

/* include/ORBextractor.h:51-85 */
class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    /* slots: frame slots of the context - 1 for the plain operator(), >= 2 for the pipelined Submit / Collect flow in which
     * LastFrame stays on the device for the slot-pair matchers */
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxWidth = 640, int maxHeight = 480,
                 int device = 0, int slots = 1)
        : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST),
          mCtx(drfe_detail::make_ctx(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, maxWidth, maxHeight, slots, device))
    {
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        drfe_detail::check(drfe_orb_scale_tables(mCtx.get(), mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                                                 mvInvLevelSigma2.data()), mCtx.get(), "drfe_orb_scale_tables");
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() {}

    /* Compute the ORB features and descriptors on an image; the mask is ignored (include/ORBextractor.h:60-65) */
    void operator()(const drfe_cv::Mat& image, const drfe_cv::Mat& /*mask*/, std::vector<drfe_cv::KeyPoint>& keypoints,
                    drfe_cv::Mat& descriptors)
    {
        if (image.empty()) return;                                       /* src/ORBextractor.cc:1046-1047 */
        const int cap = drfe_orb_max_keypoints(mCtx.get());
        keypoints.resize(cap);
        drfe_cv::Mat desc = drfe_cv::mat_u8(cap, 32);
        int n = 0;
        drfe_detail::check(drfe_orb_extract(mCtx.get(), drfe_cv::mat_data(image), image.cols, image.rows, drfe_cv::mat_step(image),
                                            reinterpret_cast<drfe_keypoint*>(keypoints.data()), desc.data, cap, &n),
                           mCtx.get(), "drfe_orb_extract");
        keypoints.resize(n);
        if (n == 0) { descriptors.release(); return; }                   /* :1064-1065 */
        descriptors = drfe_cv::mat_u8(n, 32);
        std::memcpy(descriptors.data, desc.data, (size_t)n * 32);
        /* public member mvImagePyramid (include/ORBextractor.h:85): interior ROI views of the bordered levels */
        mvPyramidStore.resize(nlevels);
        for (int l = 0; l < nlevels; ++l) {
            int bw = 0, bh = 0;
            drfe_detail::check(drfe_orb_pyramid_level(mCtx.get(), 0, l, nullptr, &bw, &bh), mCtx.get(), "drfe_orb_pyramid_level");
            mvPyramidStore[l] = drfe_cv::mat_u8(bh, bw);
            drfe_detail::check(drfe_orb_pyramid_level(mCtx.get(), 0, l, mvPyramidStore[l].data, &bw, &bh), mCtx.get(),
                               "drfe_orb_pyramid_level");
#ifdef DRFE_WITH_OPENCV
            mvImagePyramid[l] = mvPyramidStore[l](cv::Rect(19, 19, bw - 38, bh - 38));
#else
            mvImagePyramid[l] = drfe_cv::Mat(bh - 38, bw - 38, mvPyramidStore[l].data + 19 * (size_t)bw + 19, (size_t)bw);
#endif
        }
    }

    /* The same extraction without waiting for it (drfe_frame_submit / drfe_frame_collect): Frame::Frame calls Submit where it
     * started the ExtractORB thread (src/Frame.cc:124), runs ExtractLSD / ComputePlanes on the calling thread, then Collect where
     * it joined.  With a depth image (CV_16U, as Tracking hands imDepth before its convertTo) and the camera, the glue
     * (UndistortKeyPoints, ComputeStereoFromRGBD, AssignFeaturesToGrid) runs in the same submission and mvuRight / mvDepth
     * come back with the keypoints. */
    void Submit(int slot, const drfe_cv::Mat& image, const uint16_t* depth16 = nullptr, size_t depthStrideElems = 0,
                const drfe_camera* cam = nullptr)
    {
        drfe_detail::check(drfe_frame_submit(mCtx.get(), slot, drfe_cv::mat_data(image), image.cols, image.rows, drfe_cv::mat_step(image),
                                             depth16, depthStrideElems, cam), mCtx.get(), "drfe_frame_submit");
    }
    void Collect(int slot, std::vector<drfe_cv::KeyPoint>& keypoints, drfe_cv::Mat& descriptors, std::vector<float>* mvuRight = nullptr,
                 std::vector<float>* mvDepth = nullptr)
    {
        const int cap = drfe_orb_max_keypoints(mCtx.get());
        keypoints.resize(cap);
        drfe_cv::Mat desc = drfe_cv::mat_u8(cap, 32);
        if (mvuRight) mvuRight->resize(cap);
        if (mvDepth) mvDepth->resize(cap);
        int n = 0;
        drfe_detail::check(drfe_frame_collect(mCtx.get(), slot, reinterpret_cast<drfe_keypoint*>(keypoints.data()), desc.data,
                                              mvuRight ? mvuRight->data() : nullptr, mvDepth ? mvDepth->data() : nullptr, cap, &n),
                           mCtx.get(), "drfe_frame_collect");
        keypoints.resize(n);
        if (mvuRight) mvuRight->resize(n);
        if (mvDepth) mvDepth->resize(n);
        if (n == 0) { descriptors.release(); return; }
        descriptors = drfe_cv::mat_u8(n, 32);
        std::memcpy(descriptors.data, desc.data, (size_t)n * 32);
    }

    /* One submission per tracked frame (drfe_frame_submit_tracked): what Frame::Frame extracts AND what TrackWithMotionModel's
     * ORBmatcher(0.9, true).SearchByProjection(mCurrentFrame, mLastFrame, th, mono) returns (src/Tracking.cc:2181-2202), in one
     * captured graph.  TcwCur = mVelocity * mLastFrame.mTcw; lastMapPoints = mLastFrame.mvpMapPoints flattened to drfe_map_point
     * records (NULL: the RGB-D temporal points, built on the device from LastFrame's depth and TwcLast).  CollectTracked fills
     * matches[i] = index into LastFrame of the map point current keypoint i received, -1 = none, and returns nmatches. */
    void SubmitTracked(int slot, const drfe_cv::Mat& image, const uint16_t* depth16, size_t depthStrideElems, const drfe_camera& cam,
                       int lastSlot, const float* TcwCur, const float* TcwLast, const float* TwcLast, const drfe_map_point* lastMapPoints,
                       int nLast, float th, bool mono, bool checkOrientation = true)
    {
        drfe_detail::check(drfe_frame_submit_tracked(mCtx.get(), slot, drfe_cv::mat_data(image), image.cols, image.rows, drfe_cv::mat_step(image),
                                                     depth16, depthStrideElems, &cam, lastSlot, TcwCur, TcwLast, TwcLast, lastMapPoints, nLast,
                                                     th, mono ? 1 : 0, checkOrientation ? 1 : 0),
                           mCtx.get(), "drfe_frame_submit_tracked");
    }
    int CollectTracked(int slot, std::vector<drfe_cv::KeyPoint>& keypoints, drfe_cv::Mat& descriptors, std::vector<float>& mvuRight,
                       std::vector<float>& mvDepth, std::vector<int32_t>& matches)
    {
        const int cap = drfe_orb_max_keypoints(mCtx.get());
        keypoints.resize(cap); mvuRight.resize(cap); mvDepth.resize(cap); matches.assign(cap, -1);
        drfe_cv::Mat desc = drfe_cv::mat_u8(cap, 32);
        int n = 0, nmatches = 0;
        drfe_detail::check(drfe_frame_collect_tracked(mCtx.get(), slot, reinterpret_cast<drfe_keypoint*>(keypoints.data()), desc.data,
                                                      mvuRight.data(), mvDepth.data(), cap, &n, matches.data(), &nmatches),
                           mCtx.get(), "drfe_frame_collect_tracked");
        keypoints.resize(n); mvuRight.resize(n); mvDepth.resize(n); matches.resize(n);
        if (n == 0) { descriptors.release(); return 0; }
        descriptors = drfe_cv::mat_u8(n, 32);
        std::memcpy(descriptors.data, desc.data, (size_t)n * 32);
        return nmatches;
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    std::vector<drfe_cv::Mat> mvImagePyramid;

    drfe_ctx* context() { return mCtx.get(); }        /* for the Frame glue / matcher adaptors that share the device state */

protected:
    int nfeatures; double scaleFactor; int nlevels; int iniThFAST; int minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<drfe_cv::Mat> mvPyramidStore;
    drfe_detail::CtxPtr mCtx;
};

/* ---------------------------------------------------------------------------------------------------------------------------
 * ORBmatcher (include/ORBmatcher.h:41-84, src/ORBmatcher.cc) with the reference's thirteen signatures.
 *
 * The reference's matchers take Frame& / KeyFrame* / MapPoint* - objects of the map, on the host.  Every method here
 *   1. makes its frames resident in a device slot (MatcherDevice: a small LRU over the slots of one drfe_ctx; a frame that is
 *      not there yet is uploaded from its own members - mvKeys, mvKeysUn, mDescriptors, mvuRight, mvDepth - by drfe_frame_load,
 *      which rebuilds its 64 x 48 grid on the device; a frame is recognised by its type and mnId),
 *   2. flattens what the reference's loop reads through the pointers (GetWorldPos, GetDescriptor, Observations, isBad, the
 *      fields Frame::isInFrustum left on the point, ...) into the records of drfe.h,
 *   3. calls the C entry point (window gathers, Hamming distances, claim order, rotation histogram: on the device),
 *   4. turns the returned indices back into pointers and applies the graph surgery (Replace / AddObservation / AddMapPoint)
 *      in the reference's order on the host - that part touches mutex-protected map objects and stays where they live.
 * The methods are templates over the frame / keyframe / map point types and only use the member names src/ORBmatcher.cc uses, so
 * they bind to Planar_SLAM::Frame, KeyFrame and MapPoint as they are (cv::Mat members are read through .data: continuous
 * CV_32F / CV_8U matrices, as the reference creates them) and to the stand-ins of tests/native/matcher_caller.cpp.
 *
 * Threading: the reference constructs matchers on the stack of three threads (Tracking, LocalMapping, LoopClosing).  Each thread
 * binds its own MatcherDevice once (ORBmatcher::BindThread); a matcher object itself holds two scalars, as in the reference. */

namespace drfe_detail {
template <class M> inline const float* f32(const M& m) { return reinterpret_cast<const float*>(m.data); }
template <class M> inline const uint8_t* u8(const M& m) { return reinterpret_cast<const uint8_t*>(m.data); }
template <class T> struct TypeTag { static const char v; };
template <class T> const char TypeTag<T>::v = 0;
/* camera block of a Frame or KeyFrame: fx fy cx cy mbf and the image bounds (Frame: static floats; KeyFrame: const members, the
 * bounds as ints - include/KeyFrame.h:230-233) */
template <class F> inline drfe_camera camera_of(const F& f)
{
    drfe_camera c;
    c.fx = f.fx; c.fy = f.fy; c.cx = f.cx; c.cy = f.cy; c.bf = f.mbf; c.depth_factor = 0.f;
    c.min_x = (float)f.mnMinX; c.max_x = (float)f.mnMaxX; c.min_y = (float)f.mnMinY; c.max_y = (float)f.mnMaxY;
    return c;
}
template <class MP> inline void frustum_of(MP* p, drfe_frustum_point& o, uint8_t* desc32)
{
    const auto w = p->GetWorldPos(); std::memcpy(o.world, w.data, 12);
    const auto n = p->GetNormal(); std::memcpy(o.normal, n.data, 12);
    o.min_distance = p->GetMinDistanceInvariance(); o.max_distance = p->GetMaxDistanceInvariance();
    const auto d = p->GetDescriptor(); std::memcpy(desc32, d.data, 32);
}
}  // namespace drfe_detail

/* The slots of one drfe_ctx as a cache of host frames. */
class MatcherDevice {
public:
    /* ORB parameters as the extractor's (they size a slot: drfe_orb_max_keypoints, and give the scale tables the searches use) */
    MatcherDevice(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int slots = 8, int maxWidth = 640,
                  int maxHeight = 480, int device = 0)
        : mCtx(drfe_detail::make_ctx(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, maxWidth, maxHeight, slots, device)), mSlots((size_t)slots)
    {
    }
    drfe_ctx* ctx() const { return mCtx.get(); }
    /* ORBVocabulary for SearchByBoW / SearchForTriangulation: the nodes as TemplatedVocabulary::loadFromTextFile holds them */
    void UploadVocabulary(int k, int L, int scoring, int weighting, int nNodes, const int32_t* parent, const uint8_t* desc, const double* weight,
                          const uint8_t* isLeaf, int levelsup = 4)
    {
        drfe_detail::check(drfe_voc_upload(mCtx.get(), k, L, scoring, weighting, nNodes, parent, desc, weight, isLeaf), mCtx.get(), "drfe_voc_upload");
        mLevelsUp = levelsup; mHaveVoc = true;
        for (Slot& s : mSlots) s.bow = false;
    }
    /* slot of a frame, loading it when absent; `keep` (another frame's slot of the same call, or -1) is never evicted */
    template <class F> int Resident(const F& f, bool needBow = false, int keep = -1)
    {
        const void* tag = &drfe_detail::TypeTag<F>::v;
        int at = -1;
        for (size_t i = 0; i < mSlots.size(); i++) if (mSlots[i].tag == tag && mSlots[i].id == (unsigned long)f.mnId) { at = (int)i; break; }
        if (at < 0) {
            unsigned long oldest = ~0ul;
            for (size_t i = 0; i < mSlots.size(); i++) if ((int)i != keep && mSlots[i].stamp < oldest) { oldest = mSlots[i].stamp; at = (int)i; }
            if (at < 0) throw std::runtime_error("MatcherDevice: no free slot");
            const drfe_camera cam = drfe_detail::camera_of(f);
            const int n = (int)f.mvKeys.size();
            static_assert(sizeof(f.mvKeys[0]) == sizeof(drfe_keypoint), "cv::KeyPoint layout");
            drfe_detail::check(drfe_frame_load(mCtx.get(), at, reinterpret_cast<const drfe_keypoint*>(f.mvKeys.data()),
                                               reinterpret_cast<const drfe_keypoint*>(f.mvKeysUn.data()), drfe_detail::u8(f.mDescriptors),
                                               f.mvuRight.empty() ? nullptr : f.mvuRight.data(), f.mvDepth.empty() ? nullptr : f.mvDepth.data(), n, &cam),
                               mCtx.get(), "drfe_frame_load");
            mSlots[(size_t)at].tag = tag; mSlots[(size_t)at].id = (unsigned long)f.mnId; mSlots[(size_t)at].bow = false;
            mLoads++;
        }
        Slot& s = mSlots[(size_t)at];
        s.stamp = ++mClock;
        if (needBow && !s.bow) {
            if (!mHaveVoc) throw std::runtime_error("MatcherDevice: this matcher needs the vocabulary (UploadVocabulary)");
            drfe_detail::check(drfe_bow_transform_slot(mCtx.get(), mLevelsUp, at, nullptr), mCtx.get(), "drfe_bow_transform_slot");
            s.bow = true;
        }
        return at;
    }
    /* a frame whose keypoints changed in place (never in the reference; a test may) */
    void Forget() { for (Slot& s : mSlots) { s.tag = nullptr; s.stamp = 0; s.bow = false; } }
    unsigned long loads() const { return mLoads; }
private:
    struct Slot { const void* tag = nullptr; unsigned long id = 0, stamp = 0; bool bow = false; };
    drfe_detail::CtxPtr mCtx;
    std::vector<Slot> mSlots;
    unsigned long mClock = 0, mLoads = 0;
    int mLevelsUp = 4; bool mHaveVoc = false;
};

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    /* the calling thread's device (Tracking / LocalMapping / LoopClosing bind one each at start-up) */
    static void BindThread(MatcherDevice* dev) { tls() = dev; drfe_detail::thread_ctx() = dev ? dev->ctx() : nullptr; }
    static MatcherDevice& Device()
    {
        if (!tls()) throw std::runtime_error("ORBmatcher: no MatcherDevice bound to this thread (ORBmatcher::BindThread)");
        return *tls();
    }

    /* src/ORBmatcher.cc:1712-1728 */
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b)
    {
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t pa, pb;
            std::memcpy(&pa, a + 4 * i, 4); std::memcpy(&pb, b + 4 * i, 4);
            uint32_t v = pa ^ pb;
            v = v - ((v >> 1) & 0x55555555);
            v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
            dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
        }
        return dist;
    }
    static int DescriptorDistance(const drfe_cv::Mat& a, const drfe_cv::Mat& b) { return DescriptorDistance(drfe_detail::u8(a), drfe_detail::u8(b)); }

    /* ---- src/ORBmatcher.cc:46-130 - Tracking::SearchLocalPoints: the local map into the frame ---- */
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3)
    {
        MatcherDevice& D = Device();
        const int slot = D.Resident(F), m = (int)vpMapPoints.size(), N = (int)F.mvpMapPoints.size();
        std::vector<drfe_tracked_point> tp((size_t)m);
        for (int i = 0; i < m; i++) {
            MapPointT* p = vpMapPoints[(size_t)i];
            drfe_tracked_point& t = tp[(size_t)i];
            std::memset(&t, 0, sizeof(t));
            t.track_in_view = p->mbTrackInView ? 1 : 0;
            if (!t.track_in_view) continue;
            t.bad = p->isBad() ? 1 : 0;
            if (t.bad) continue;
            t.obs_positive = p->Observations() > 0 ? 1 : 0;
            t.level = p->mnTrackScaleLevel; t.proj_x = p->mTrackProjX; t.proj_y = p->mTrackProjY; t.proj_xr = p->mTrackProjXR; t.view_cos = p->mTrackViewCos;
            const auto d = p->GetDescriptor(); std::memcpy(t.desc, d.data, 32);
        }
        /* claims the frame already holds: any index >= m (never confused with a new match, which is an index into vpMapPoints) */
        std::vector<int32_t> claim((size_t)N);
        std::vector<uint8_t> obs((size_t)N);
        for (int i = 0; i < N; i++) { MapPointT* p = F.mvpMapPoints[(size_t)i]; claim[(size_t)i] = p ? m + i : -1; obs[(size_t)i] = p && p->Observations() > 0; }
        int n = 0;
        drfe_detail::check(drfe_search_by_projection_map(D.ctx(), slot, tp.data(), m, th, mfNNratio, obs.data(), claim.data(), N, &n), D.ctx(),
                           "drfe_search_by_projection_map");
        for (int i = 0; i < N; i++) if (claim[(size_t)i] >= 0 && claim[(size_t)i] < m) F.mvpMapPoints[(size_t)i] = vpMapPoints[(size_t)claim[(size_t)i]];
        return n;
    }

    /* ---- src/ORBmatcher.cc:1396-1535 - Tracking::TrackWithMotionModel ---- */
    template <class FrameT, class LastT>
    typename std::enable_if<std::is_class<LastT>::value, int>::type
    SearchByProjection(FrameT& CurrentFrame, const LastT& LastFrame, const float th, const bool bMono)
    {
        MatcherDevice& D = Device();
        const int cur = D.Resident(CurrentFrame), last = D.Resident(LastFrame, false, cur);
        const int nl = (int)LastFrame.mvpMapPoints.size(), nc = (int)CurrentFrame.mvpMapPoints.size();
        std::vector<drfe_map_point> mp((size_t)nl);
        for (int i = 0; i < nl; i++) {
            auto* p = LastFrame.mvpMapPoints[(size_t)i];
            drfe_map_point& r = mp[(size_t)i];
            std::memset(&r, 0, sizeof(r));
            r.valid = p && !LastFrame.mvbOutlier[(size_t)i];
            if (!r.valid) continue;
            r.obs_positive = p->Observations() > 0;
            const auto w = p->GetWorldPos(); std::memcpy(r.world, w.data, 12);
            const auto d = p->GetDescriptor(); std::memcpy(r.desc, d.data, 32);
        }
        std::vector<int32_t> claim((size_t)nc);
        std::vector<uint8_t> obs((size_t)nc);
        for (int i = 0; i < nc; i++) { auto* p = CurrentFrame.mvpMapPoints[(size_t)i]; claim[(size_t)i] = p ? nl + i : -1; obs[(size_t)i] = p && p->Observations() > 0; }
        const drfe_camera cam = drfe_detail::camera_of(CurrentFrame);
        int n = 0;
        drfe_detail::check(drfe_search_by_projection_last(D.ctx(), cur, last, drfe_detail::f32(CurrentFrame.mTcw), drfe_detail::f32(LastFrame.mTcw), &cam, mp.data(), nl,
                                                          th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, obs.data(), claim.data(), nc, &n), D.ctx(),
                           "drfe_search_by_projection_last");
        for (int i = 0; i < nc; i++) {
            const int32_t v = claim[(size_t)i];
            if (v < 0) CurrentFrame.mvpMapPoints[(size_t)i] = nullptr;        /* also a match the rotation histogram took back (:1524) */
            else if (v < nl) CurrentFrame.mvpMapPoints[(size_t)i] = LastFrame.mvpMapPoints[(size_t)v];
        }
        return n;
    }

    /* ---- src/ORBmatcher.cc:1332-1394 - the brute-force fallback of TrackWithMotionModel (src/Tracking.cc:2195-2200) ---- */
    template <class FrameT, class LastT>
    int MatchORBPoints(FrameT& CurrentFrame, const LastT& LastFrame)
    {
        MatcherDevice& D = Device();
        const int cur = D.Resident(CurrentFrame), last = D.Resident(LastFrame, false, cur);
        const int nl = (int)LastFrame.mvpMapPoints.size(), nc = (int)CurrentFrame.mvpMapPoints.size();
        std::vector<int32_t> lastMp((size_t)nl), curMp((size_t)nc);
        std::vector<uint8_t> outl((size_t)nl);
        for (int i = 0; i < nl; i++) { lastMp[(size_t)i] = LastFrame.mvpMapPoints[(size_t)i] ? i : -1; outl[(size_t)i] = LastFrame.mvbOutlier[(size_t)i] ? 1 : 0; }
        for (int i = 0; i < nc; i++) curMp[(size_t)i] = CurrentFrame.mvpMapPoints[(size_t)i] ? nl + i : -1;
        int nPair = 0;
        drfe_detail::check(drfe_match_orb_points(D.ctx(), cur, last, lastMp.data(), outl.data(), nl, curMp.data(), nc, &nPair), D.ctx(), "drfe_match_orb_points");
        for (int i = 0; i < nc; i++) if (curMp[(size_t)i] >= 0 && curMp[(size_t)i] < nl) CurrentFrame.mvpMapPoints[(size_t)i] = LastFrame.mvpMapPoints[(size_t)curMp[(size_t)i]];
        return nPair;
    }

    /* ---- src/ORBmatcher.cc:1537-1664 - Tracking::Relocalization (src/Tracking.cc:3638, :3651) ---- */
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th, const int ORBdist)
    {
        MatcherDevice& D = Device();
        const int slot = D.Resident(CurrentFrame);
        const std::vector<MapPointT*> vpMPs = pKF->GetMapPointMatches();
        const int n = (int)vpMPs.size(), nc = (int)CurrentFrame.mvpMapPoints.size();
        std::vector<drfe_frustum_point> fp((size_t)n);
        std::vector<uint8_t> descs((size_t)n * 32), skip((size_t)n), matched((size_t)nc);
        std::vector<float> angles((size_t)n);
        for (int i = 0; i < n; i++) {
            MapPointT* p = vpMPs[(size_t)i];
            std::memset(&fp[(size_t)i], 0, sizeof(drfe_frustum_point));
            skip[(size_t)i] = !p || p->isBad() || sAlreadyFound.count(p);
            angles[(size_t)i] = pKF->mvKeysUn[(size_t)i].angle;
            if (!skip[(size_t)i]) drfe_detail::frustum_of(p, fp[(size_t)i], descs.data() + 32 * (size_t)i);
        }
        for (int k = 0; k < nc; k++) matched[(size_t)k] = CurrentFrame.mvpMapPoints[(size_t)k] != nullptr;
        std::vector<int32_t> nm((size_t)nc, -1);
        int cnt = 0;
        drfe_detail::check(drfe_search_by_projection_reloc(D.ctx(), slot, drfe_detail::f32(CurrentFrame.mTcw), fp.data(), descs.data(), angles.data(), skip.data(), n,
                                                           matched.data(), nc, th, ORBdist, mbCheckOrientation ? 1 : 0, nm.data(), &cnt), D.ctx(),
                           "drfe_search_by_projection_reloc");
        for (int k = 0; k < nc; k++) if (nm[(size_t)k] >= 0) CurrentFrame.mvpMapPoints[(size_t)k] = vpMPs[(size_t)nm[(size_t)k]];
        return cnt;
    }

    /* ---- src/ORBmatcher.cc:294-407 - LoopClosing::ComputeSim3 after the Sim3 optimisation ---- */
    template <class KeyFrameT, class MatT, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, MatT Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th)
    {
        MatcherDevice& D = Device();
        const int slot = D.Resident(*pKF);
        const int n = (int)vpPoints.size(), nk = (int)vpMatched.size();
        std::set<MapPointT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPointT*>(nullptr));
        std::vector<drfe_frustum_point> fp((size_t)n);
        std::vector<uint8_t> descs((size_t)n * 32), skip((size_t)n), matched((size_t)nk);
        for (int i = 0; i < n; i++) {
            MapPointT* p = vpPoints[(size_t)i];
            std::memset(&fp[(size_t)i], 0, sizeof(drfe_frustum_point));
            skip[(size_t)i] = p->isBad() || spAlreadyFound.count(p);
            if (!skip[(size_t)i]) drfe_detail::frustum_of(p, fp[(size_t)i], descs.data() + 32 * (size_t)i);
        }
        for (int k = 0; k < nk; k++) matched[(size_t)k] = vpMatched[(size_t)k] != nullptr;
        std::vector<int32_t> nm((size_t)nk, -1);
        int cnt = 0;
        drfe_detail::check(drfe_search_by_projection_kf(D.ctx(), slot, drfe_detail::f32(Scw), fp.data(), descs.data(), skip.data(), n, matched.data(), nk, (float)th,
                                                        nm.data(), &cnt), D.ctx(), "drfe_search_by_projection_kf");
        for (int k = 0; k < nk; k++) if (nm[(size_t)k] >= 0) vpMatched[(size_t)k] = vpPoints[(size_t)nm[(size_t)k]];
        return cnt;
    }

    /* ---- src/ORBmatcher.cc:160-292 - TrackReferenceKeyFrame / Relocalization ---- */
    template <class KeyFrameT, class FrameT, class MapPointT>
    typename std::enable_if<std::is_class<FrameT>::value, int>::type
    SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches)
    {
        MatcherDevice& D = Device();
        const int kf = D.Resident(*pKF, true), fs = D.Resident(F, true, kf);
        const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
        const int nk = (int)vpMapPointsKF.size(), nf = (int)F.mvKeys.size();
        vpMapPointMatches = std::vector<MapPointT*>((size_t)nf, static_cast<MapPointT*>(nullptr));
        std::vector<int32_t> kfMp((size_t)nk), match((size_t)nf, -1);
        for (int i = 0; i < nk; i++) { MapPointT* p = vpMapPointsKF[(size_t)i]; kfMp[(size_t)i] = (p && !p->isBad()) ? i : -1; }
        int n = 0;
        drfe_detail::check(drfe_search_by_bow(D.ctx(), kf, fs, kfMp.data(), nk, mfNNratio, mbCheckOrientation ? 1 : 0, match.data(), nf, &n), D.ctx(),
                           "drfe_search_by_bow");
        for (int j = 0; j < nf; j++) if (match[(size_t)j] >= 0) vpMapPointMatches[(size_t)j] = vpMapPointsKF[(size_t)match[(size_t)j]];
        return n;
    }

    /* ---- src/ORBmatcher.cc:526-660 - LoopClosing::ComputeSim3 ---- */
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12)
    {
        MatcherDevice& D = Device();
        const int s1 = D.Resident(*pKF1, true), s2 = D.Resident(*pKF2, true, s1);
        const std::vector<MapPointT*> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
        const int n1 = (int)vp1.size(), n2 = (int)vp2.size();
        vpMatches12 = std::vector<MapPointT*>((size_t)n1, static_cast<MapPointT*>(nullptr));
        std::vector<int32_t> mp1((size_t)n1), mp2((size_t)n2), match2((size_t)n2, -1);
        for (int i = 0; i < n1; i++) mp1[(size_t)i] = (vp1[(size_t)i] && !vp1[(size_t)i]->isBad()) ? i : -1;
        for (int i = 0; i < n2; i++) mp2[(size_t)i] = (vp2[(size_t)i] && !vp2[(size_t)i]->isBad()) ? i : -1;
        int n = 0;
        drfe_detail::check(drfe_search_by_bow_kf(D.ctx(), s1, s2, mp1.data(), n1, mp2.data(), n2, mfNNratio, mbCheckOrientation ? 1 : 0, match2.data(), &n), D.ctx(),
                           "drfe_search_by_bow_kf");
        for (int i2 = 0; i2 < n2; i2++) if (match2[(size_t)i2] >= 0) vpMatches12[(size_t)match2[(size_t)i2]] = vp2[(size_t)i2];
        return n;
    }

    /* ---- src/ORBmatcher.cc:409-524 - Tracking::MonocularInitialization; PointT = cv::Point2f (two floats) ---- */
    template <class FrameT, class PointT>
    int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<PointT>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10)
    {
        static_assert(sizeof(PointT) == 8, "cv::Point2f is two floats");
        MatcherDevice& D = Device();
        const int s1 = D.Resident(F1), s2 = D.Resident(F2, false, s1);
        const int n1 = (int)F1.mvKeysUn.size();
        vnMatches12.assign((size_t)n1, -1);
        if ((int)vbPrevMatched.size() != n1) throw std::runtime_error("SearchForInitialization: vbPrevMatched must hold one point per keypoint of F1");
        int n = 0;
        drfe_detail::check(drfe_search_for_initialization(D.ctx(), s1, s2, reinterpret_cast<float*>(vbPrevMatched.data()), n1, windowSize, mfNNratio,
                                                          mbCheckOrientation ? 1 : 0, vnMatches12.data(), &n), D.ctx(), "drfe_search_for_initialization");
        return n;
    }

    /* ---- src/ORBmatcher.cc:661-827 - LocalMapping::CreateNewMapPoints ---- */
    template <class KeyFrameT, class MatT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, MatT F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo)
    {
        MatcherDevice& D = Device();
        const int s1 = D.Resident(*pKF1, true), s2 = D.Resident(*pKF2, true, s1);
        const int n1 = (int)pKF1->mvKeysUn.size(), n2 = (int)pKF2->mvKeysUn.size();
        std::vector<int32_t> mp1((size_t)n1), mp2((size_t)n2), m12((size_t)n1, -1);
        for (int i = 0; i < n1; i++) mp1[(size_t)i] = pKF1->GetMapPoint((size_t)i) ? i : -1;
        for (int i = 0; i < n2; i++) mp2[(size_t)i] = pKF2->GetMapPoint((size_t)i) ? i : -1;
        const auto Cw = pKF1->GetCameraCenter();
        const auto T2w = pKF2->GetPose();
        const drfe_camera cam2 = drfe_detail::camera_of(*pKF2);
        int n = 0;
        drfe_detail::check(drfe_search_for_triangulation(D.ctx(), s1, s2, mp1.data(), n1, mp2.data(), n2, drfe_detail::f32(F12), drfe_detail::f32(Cw), drfe_detail::f32(T2w),
                                                         &cam2, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), &n), D.ctx(),
                           "drfe_search_for_triangulation");
        vMatchedPairs.clear();
        vMatchedPairs.reserve((size_t)n);
        for (int i = 0; i < n1; i++) if (m12[(size_t)i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[(size_t)i]));
        return n;
    }

    /* ---- src/ORBmatcher.cc:1106-1330 - LoopClosing::ComputeSim3 ---- */
    template <class KeyFrameT, class MapPointT, class MatT>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const MatT& R12, const MatT& t12, const float th)
    {
        MatcherDevice& D = Device();
        const int s1 = D.Resident(*pKF1), s2 = D.Resident(*pKF2, false, s1);
        const std::vector<MapPointT*> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vp1.size(), N2 = (int)vp2.size();
        std::vector<uint8_t> skip1((size_t)N1, 0), skip2((size_t)N2, 0);
        for (int i = 0; i < N1; i++) {                                     /* :1131-1142 */
            MapPointT* p = vpMatches12[(size_t)i];
            if (p) {
                skip1[(size_t)i] = 1;
                const int idx2 = p->GetIndexInKeyFrame(pKF2);
                if (idx2 >= 0 && idx2 < N2) skip2[(size_t)idx2] = 1;
            }
        }
        std::vector<drfe_frustum_point> f1((size_t)N1), f2((size_t)N2);
        std::vector<uint8_t> d1((size_t)N1 * 32), d2((size_t)N2 * 32);
        for (int i = 0; i < N1; i++) {
            MapPointT* p = vp1[(size_t)i];
            std::memset(&f1[(size_t)i], 0, sizeof(drfe_frustum_point));
            if (!p || p->isBad()) skip1[(size_t)i] = 1;
            if (!skip1[(size_t)i]) drfe_detail::frustum_of(p, f1[(size_t)i], d1.data() + 32 * (size_t)i);
        }
        for (int i = 0; i < N2; i++) {
            MapPointT* p = vp2[(size_t)i];
            std::memset(&f2[(size_t)i], 0, sizeof(drfe_frustum_point));
            if (!p || p->isBad()) skip2[(size_t)i] = 1;
            if (!skip2[(size_t)i]) drfe_detail::frustum_of(p, f2[(size_t)i], d2.data() + 32 * (size_t)i);
        }
        const auto T1w = pKF1->GetPose();
        const auto T2w = pKF2->GetPose();
        std::vector<int32_t> m12((size_t)N1, -1);
        int nFound = 0;
        drfe_detail::check(drfe_search_by_sim3(D.ctx(), s1, s2, drfe_detail::f32(T1w), drfe_detail::f32(T2w), s12, drfe_detail::f32(R12), drfe_detail::f32(t12), f1.data(),
                                               d1.data(), skip1.data(), N1, f2.data(), d2.data(), skip2.data(), N2, th, m12.data(), &nFound), D.ctx(),
                           "drfe_search_by_sim3");
        for (int i1 = 0; i1 < N1; i1++) if (m12[(size_t)i1] >= 0) vpMatches12[(size_t)i1] = vp2[(size_t)m12[(size_t)i1]];
        return nFound;
    }

    /* ---- src/ORBmatcher.cc:829-985 - LocalMapping::SearchInNeighbors.  The search of every point on the device; the loop that
     * applies it runs here in the reference's order, re-reading isBad() / IsInKeyFrame() / GetMapPoint() at each step exactly where the
     * reference reads them: an earlier Replace or AddMapPoint of this very loop changes what a later point meets (:845, :957) ---- */
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th = 3.0)
    {
        MatcherDevice& D = Device();
        const int slot = D.Resident(*pKF);
        const int n = (int)vpMapPoints.size();
        std::vector<drfe_frustum_point> fp((size_t)n);
        std::vector<uint8_t> descs((size_t)n * 32), skip((size_t)n);
        for (int i = 0; i < n; i++) {
            MapPointT* p = vpMapPoints[(size_t)i];
            std::memset(&fp[(size_t)i], 0, sizeof(drfe_frustum_point));
            skip[(size_t)i] = !p || p->isBad() || p->IsInKeyFrame(pKF);
            if (!skip[(size_t)i]) drfe_detail::frustum_of(p, fp[(size_t)i], descs.data() + 32 * (size_t)i);
        }
        std::vector<int32_t> bestIdx((size_t)n, -1), bestDist((size_t)n, 256);
        const auto Tcw = pKF->GetPose();
        drfe_detail::check(drfe_fuse_search(D.ctx(), slot, drfe_detail::f32(Tcw), fp.data(), descs.data(), skip.data(), n, th, bestIdx.data(), bestDist.data()), D.ctx(),
                           "drfe_fuse_search");
        int nFused = 0;
        for (int i = 0; i < n; i++) {
            MapPointT* pMP = vpMapPoints[(size_t)i];
            if (!pMP) continue;
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            if (bestIdx[(size_t)i] < 0 || bestDist[(size_t)i] > TH_LOW) continue;
            MapPointT* pMPinKF = pKF->GetMapPoint((size_t)bestIdx[(size_t)i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, (size_t)bestIdx[(size_t)i]);
                pKF->AddMapPoint(pMP, (size_t)bestIdx[(size_t)i]);
            }
            nFused++;
        }
        return nFused;
    }

    /* ---- src/ORBmatcher.cc:981-1105 - LoopClosing::SearchAndFuse ---- */
    template <class KeyFrameT, class MatT, class MapPointT>
    int Fuse(KeyFrameT* pKF, MatT Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint)
    {
        MatcherDevice& D = Device();
        const int slot = D.Resident(*pKF);
        const std::set<MapPointT*> spAlreadyFound = pKF->GetMapPoints();
        const int n = (int)vpPoints.size();
        std::vector<drfe_frustum_point> fp((size_t)n);
        std::vector<uint8_t> descs((size_t)n * 32), skip((size_t)n);
        for (int i = 0; i < n; i++) {
            MapPointT* p = vpPoints[(size_t)i];
            std::memset(&fp[(size_t)i], 0, sizeof(drfe_frustum_point));
            skip[(size_t)i] = p->isBad() || spAlreadyFound.count(p);
            if (!skip[(size_t)i]) drfe_detail::frustum_of(p, fp[(size_t)i], descs.data() + 32 * (size_t)i);
        }
        std::vector<int32_t> bestIdx((size_t)n, -1), bestDist((size_t)n, 256);
        drfe_detail::check(drfe_fuse_search_sim3(D.ctx(), slot, drfe_detail::f32(Scw), fp.data(), descs.data(), skip.data(), n, th, bestIdx.data(), bestDist.data()), D.ctx(),
                           "drfe_fuse_search_sim3");
        int nFused = 0;
        for (int i = 0; i < n; i++) {
            if (skip[(size_t)i] || bestIdx[(size_t)i] < 0 || bestDist[(size_t)i] > TH_LOW) continue;
            MapPointT* pMP = vpPoints[(size_t)i];
            MapPointT* pMPinKF = pKF->GetMapPoint((size_t)bestIdx[(size_t)i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[(size_t)i] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, (size_t)bestIdx[(size_t)i]);
                pKF->AddMapPoint(pMP, (size_t)bestIdx[(size_t)i]);
            }
            nFused++;
        }
        return nFused;
    }

    /* ---- slot-level forms (the frames already live in slots of `ctx`: the per-frame flow of ORBextractor::Submit / Collect) ---- */
    /* SearchByProjection(CurrentFrame, LastFrame, th, bMono) on the slots `cur` / `last` of ctx (extract + glue done):
     * lastPoints[i] = what the loop reads of LastFrame.mvpMapPoints[i]; curClaims in/out = index into lastPoints or -1 */
    int SearchByProjection(drfe_ctx* ctx, int cur, int last, const float* TcwCur, const float* TcwLast, const drfe_camera& cam,
                           const std::vector<drfe_map_point>& lastPoints, std::vector<int32_t>& curClaims, float th, bool bMono)
    {
        int n = 0;
        drfe_detail::check(drfe_search_by_projection_last(ctx, cur, last, TcwCur, TcwLast, &cam, lastPoints.data(), (int)lastPoints.size(),
                                                          th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, nullptr, curClaims.data(),
                                                          (int)curClaims.size(), &n), ctx, "drfe_search_by_projection_last");
        return n;
    }
    /* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) on the slots f1 / f2 of ctx (src/ORBmatcher.cc:409-524):
     * vbPrevMatched = (x, y) per F1 keypoint, updated in place; vnMatches12[i1] = index into F2's keypoints or -1 */
    int SearchForInitialization(drfe_ctx* ctx, int f1, int f2, std::vector<float>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10)
    {
        int n = 0;
        vnMatches12.assign(vbPrevMatched.size() / 2, -1);
        drfe_detail::check(drfe_search_for_initialization(ctx, f1, f2, vbPrevMatched.data(), (int)vnMatches12.size(), windowSize, mfNNratio,
                                                          mbCheckOrientation ? 1 : 0, vnMatches12.data(), &n), ctx,
                           "drfe_search_for_initialization");
        return n;
    }
protected:
    static MatcherDevice*& tls() { static thread_local MatcherDevice* d = nullptr; return d; }
    float mfNNratio; bool mbCheckOrientation;
};

/* include/PlaneExtractor.h:30-59 */
typedef drfe_cv::Vector3d VertexType;
const int kDepthWidth = 640;
const int kDepthHeight = 480;

/* ImagePointCloud (include/PlaneExtractor.h:42-59).  `vertices[j]` is what Frame::ComputePlanes reads
 * (planeDetector.cloud.vertices[j][0..2], src/Frame.cc:959-961) for the members of every plane - a few ten thousand of the
 * 307 200 pixels - so the container computes a vertex when it is asked for it, with PlaneDetection::readDepthImage's
 * arithmetic (src/PlaneExtractor.cpp:36-53: doubles, K's floats promoted, z > 5 -> (0, 0, 0)), instead of filling 7.4 MB per
 * frame on the host: the extractor itself builds its points on the device (k_ahc_blocks) and never reads this array. */
struct ImagePointCloud {
    struct Vertices {
        const uint16_t* depth = nullptr; size_t strideElems = 0; int w = 0, h = 0;
        float fx = 1, fy = 1, cx = 0, cy = 0, factor = 0;
        size_t size() const { return (size_t)w * h; }
        VertexType operator[](size_t pixIdx) const
        {
            const int i = (int)(pixIdx / (size_t)w), j = (int)(pixIdx % (size_t)w);
            const double z = (double)depth[(size_t)i * strideElems + j] * factor;
            if (std::isnan(z)) return VertexType(0, 0, z);
            if (z > 5.0) return VertexType(0, 0, 0);
            const double x = ((double)j - cx) * z / fx, y = ((double)i - cy) * z / fy;
            return VertexType(x, y, z);
        }
    } vertices;
    int w = kDepthWidth, h = kDepthHeight;
    inline int width() const { return w; }
    inline int height() const { return h; }
    inline bool get(const int row, const int col, double& x, double& y, double& z) const
    {
        const VertexType p = vertices[(size_t)row * w + col];
        z = p[2];
        if (z == 0 || std::isnan(z)) return false;
        x = p[0]; y = p[1];
        return true;
    }
};

/* the members of ahc::PlaneSeg Frame::ComputePlanes reads through plane_filter.extractedPlanes[i] (include/peac/AHCPlaneSeg.hpp:
 * normal, center, mse, curvature, N) */
struct ExtractedPlane { double normal[3], center[3], mse, curvature; int N; };
/* the part of ahc::PlaneFitter<ImagePointCloud> the callers touch (include/peac/AHCPlaneFitter.hpp:118): extractedPlanes */
struct PlaneFitterView { std::vector<std::shared_ptr<ExtractedPlane>> extractedPlanes; };

/* include/PlaneExtractor.h:61-82 (the live AHC extractor).  Member shape as Frame::ComputePlanes uses it (src/Frame.cc:947-979):
 *   planeDetector.readColorImage(img); planeDetector.readDepthImage(depth, K, factor); planeDetector.runPlaneDetection();
 *   planeDetector.plane_num_, .plane_vertices_[i], .cloud.vertices[j][k], .plane_filter.extractedPlanes[i]->normal / ->center,
 *   .seg_output */
class PlaneDetection {
public:
    typedef Planar_SLAM::ExtractedPlane ExtractedPlane;
    static const int kDepthWidth = 640, kDepthHeight = 480;
    ImagePointCloud cloud;
    PlaneFitterView plane_filter;
    std::vector<std::vector<int>> plane_vertices_;     /* vertex indices each plane contains */
    std::vector<std::shared_ptr<ExtractedPlane>>& extractedPlanes = plane_filter.extractedPlanes;   /* round-2 spelling, same object */
    drfe_cv::Mat seg_output;
    drfe_cv::Mat color_img_;
    int plane_num_ = 0;

    explicit PlaneDetection(drfe_ctx* ctx) : mCtx(ctx) {}
    PlaneDetection(const PlaneDetection&) = delete;
    PlaneDetection& operator=(const PlaneDetection&) = delete;

    bool readColorImage(const drfe_cv::Mat& RGBImg) { color_img_ = RGBImg; return !color_img_.empty(); }   /* kept for the caller; the extractor does not read it */

    bool readDepthImage(const drfe_cv::Mat& depthImg, const float K[9] /* mK row-major */, float depthfactor)
    {
#ifdef DRFE_WITH_OPENCV
        if (depthImg.empty() || depthImg.depth() != CV_16U) return false;
#else
        if (depthImg.empty() || depthImg.elem != 2) return false;       /* "cannot read depth image": CV_16U only */
#endif
        mDepth = depthImg; mFactor = depthfactor;
        mK4[0] = K[0]; mK4[1] = K[4]; mK4[2] = K[2]; mK4[3] = K[5];
        cloud.w = depthImg.cols; cloud.h = depthImg.rows;
        cloud.vertices.depth = mDepth.ptr<uint16_t>(0); cloud.vertices.strideElems = drfe_cv::mat_step(mDepth) / 2;
        cloud.vertices.w = depthImg.cols; cloud.vertices.h = depthImg.rows;
        cloud.vertices.fx = K[0]; cloud.vertices.fy = K[4]; cloud.vertices.cx = K[2]; cloud.vertices.cy = K[5];
        cloud.vertices.factor = depthfactor;
        return true;
    }
#ifdef DRFE_WITH_OPENCV
    bool readDepthImage(cv::Mat depthImg, cv::Mat& K, const float depthfactor)        /* the reference's signature: CV_32F 3 x 3 K */
    {
        const float k9[9] = {K.at<float>(0, 0), 0, K.at<float>(0, 2), 0, K.at<float>(1, 1), K.at<float>(1, 2), 0, 0, 1};
        return readDepthImage(static_cast<const drfe_cv::Mat&>(depthImg), k9, depthfactor);
    }
#endif
    void runPlaneDetection()
    {
        std::vector<drfe_plane> pl(64);
        std::vector<int32_t> off(65), idx((size_t)mDepth.cols * mDepth.rows);
        seg_output = drfe_cv::mat_u8(mDepth.rows, mDepth.cols);
        int np = 0;
        drfe_detail::check(drfe_planes_ahc(mCtx, mDepth.ptr<uint16_t>(0), mDepth.cols, mDepth.rows, drfe_cv::mat_step(mDepth) / 2, mK4, mFactor,
                                           pl.data(), 64, &np, seg_output.data, off.data(), idx.data()), mCtx, "drfe_planes_ahc");
        plane_num_ = np;
        plane_vertices_.assign(np, std::vector<int>());
        plane_filter.extractedPlanes.clear();
        for (int i = 0; i < np; i++) {
            plane_vertices_[i].assign(idx.begin() + off[i], idx.begin() + off[i + 1]);
            auto e = std::make_shared<ExtractedPlane>();
            std::memcpy(e->normal, pl[i].normal, 24); std::memcpy(e->center, pl[i].center, 24);
            e->mse = pl[i].mse; e->curvature = pl[i].curvature; e->N = pl[i].n_points;
            plane_filter.extractedPlanes.push_back(e);
        }
    }
private:
    drfe_ctx* mCtx; drfe_cv::Mat mDepth; float mFactor = 0; float mK4[4] = {0, 0, 0, 0};
};

/* PlaneSeg of src/CAPE/PlaneSeg.h as far as Frame::ComputePlanes_CAPE reads it (src/Frame.cc:1118-1121: normal[0..2], d), plus
 * the fields CAPE::process fills beside them */
struct PlaneSeg { double normal[3], mean[3], d; float MSE, score; int nr_pts; };
struct CylinderSeg { int nr_segments = 0; };             /* cylinder_detection is false in the reference (include/PlaneExtractor.h:112) */

/* include/PlaneExtractor.h:84-115.  Member shape as Frame::ComputePlanes_CAPE uses it (src/Frame.cc:1096-1141):
 *   readColorImage(imGrey); readDepthImage(depth /+ CV_32F metres +/, K); runPlaneDetection();
 *   nr_planes, plane_cloud[i] (the plane's points in raster order), plane_params[i].normal / .d, seg_output
 * PointCloud::Ptr of the reference is a pcl::PointCloud<pcl::PointXYZRGB>::Ptr; here plane_cloud[i] is a shared pointer to a
 * vector of float xyz triples (PCL is absent) with the same points in the same order. */
class PlaneDetection_CAPE {
public:
    struct PointT { float x, y, z; };
    struct PointCloud { std::vector<PointT> points; size_t size() const { return points.size(); } typedef std::shared_ptr<PointCloud> Ptr; };

    explicit PlaneDetection_CAPE(drfe_ctx* ctx) : mCtx(ctx) {}
    ~PlaneDetection_CAPE() {}

    bool readColorImage(const drfe_cv::Mat& RGBImg) { color_img_ = RGBImg; return !color_img_.empty(); }
    bool readDepthImage(const drfe_cv::Mat& depthImg, const float K[9] /* row-major */)
    {
#ifdef DRFE_WITH_OPENCV
        if (depthImg.empty() || depthImg.depth() != CV_32F) return false;
#else
        if (depthImg.empty() || depthImg.elem != 4) return false;        /* CV_32F metres (src/PlaneExtractor.cpp:104-112) */
#endif
        depth_img = depthImg;
        mK4[0] = K[0]; mK4[1] = K[4]; mK4[2] = K[2]; mK4[3] = K[5];
        return true;
    }
#ifdef DRFE_WITH_OPENCV
    bool readDepthImage(cv::Mat depthImg, cv::Mat& K)
    {
        K_ = K;
        const float k9[9] = {K.at<float>(0, 0), 0, K.at<float>(0, 2), 0, K.at<float>(1, 1), K.at<float>(1, 2), 0, 0, 1};
        return readDepthImage(static_cast<const drfe_cv::Mat&>(depthImg), k9);
    }
#endif
    void runPlaneDetection()
    {
        const int rows = depth_img.rows, cols = depth_img.cols;
        std::vector<drfe_cape_plane> pl(64);
        seg_output = drfe_cv::mat_u8(rows, cols);
        int np = 0;
        const size_t strideElems = drfe_cv::mat_step(depth_img) / 4;
        drfe_detail::check(drfe_planes_cape(mCtx, depth_img.ptr<float>(0), cols, rows, strideElems, mK4, PATCH_SIZE, COS_ANGLE_MAX,
                                            MAX_MERGE_DIST, pl.data(), 64, &np, seg_output.data, nullptr, nullptr, nullptr), mCtx, "drfe_planes_cape");
        nr_planes = np; nr_cylinders = 0;
        plane_params.resize(np);
        for (int i = 0; i < np; i++) {
            PlaneSeg& o = plane_params[i];
            std::memcpy(o.normal, pl[i].normal, 24); std::memcpy(o.mean, pl[i].mean, 24);
            o.d = pl[i].d; o.MSE = pl[i].mse; o.score = pl[i].score; o.nr_pts = pl[i].n_points;
        }
        /* plane_cloud: the reference APPENDS nr_planes new clouds per call and indexes them from 0 (src/PlaneExtractor.cpp:165-189
         * with plane_cloud a member that is never cleared) - frame 2's points land in frame 1's clouds.  Reproduced literally. */
        for (int i = 0; i < np; ++i) plane_cloud.push_back(std::make_shared<PointCloud>());
        for (int i = 0; i < rows; i++) {
            const uint8_t* sCode = seg_output.ptr<uint8_t>(i);
            const float* drow = depth_img.ptr<float>(i);
            for (int j = 0; j < cols; j++) {
                const int code = sCode[j];
                if (code > 0) {
                    const double z = (double)drow[j];
                    const double x = ((double)j - mK4[2]) * z / mK4[0], y = ((double)i - mK4[3]) * z / mK4[1];
                    plane_cloud[code - 1]->points.push_back(PointT{(float)(float)x, (float)(float)y, (float)(float)z});   /* double -> MatrixXf -> PointT */
                }
            }
        }
    }

    std::vector<PointCloud::Ptr> plane_cloud;
    std::vector<PlaneSeg> plane_params;
    std::vector<CylinderSeg> cylinder_params;
    int nr_planes = 0, nr_cylinders = 0;
    drfe_cv::Mat seg_output;
    drfe_cv::Mat color_img_, depth_img;
#ifdef DRFE_WITH_OPENCV
    cv::Mat K_;
#endif
    int PATCH_SIZE = 20;
    float COS_ANGLE_MAX = (float)std::cos(3.14159265358979323846 / 12);
    float MAX_MERGE_DIST = 0;
    bool cylinder_detection = false;
private:
    drfe_ctx* mCtx; float mK4[4] = {0, 0, 0, 0};
};

/* ---------------------------------------------------------------------------------------------------------------------------
 * LSDmatcher (include/LSDmatcher.h:21-36, src/LSDmatcher.cpp) with the reference's ten signatures, the same way as ORBmatcher above:
 * member templates over the frame / keyframe / map-line types that use only the member names src/LSDmatcher.cpp reads
 *   Frame:    mTcw, fx fy cx cy mbf, mnMinX .. mnMaxY, NL, mvKeylinesUn, mLdesc, mvpMapLines, mvbLineOutlier
 *   KeyFrame: fx fy cx cy mbf, mnMinX .. mnMaxY, mvKeyLines, mLineDescriptors, GetMapLineMatches(), GetMapLine(idx), GetMapLines(),
 *             AddMapLine(pML, idx), GetPose(), mnId (through MapLine::GetIndexInKeyFrame)
 *   MapLine:  isBad(), Observations(), GetWorldPos() (six doubles through operator()), GetNormal() (three), GetDescriptor(),
 *             GetMinDistanceInvariance(), GetMaxDistanceInvariance(), GetIndexInKeyFrame(pKF), AddObservation(pKF, idx), Replace(pML),
 *             mbTrackInView, mnTrackScaleLevel, mTrackViewCos, mTrackProjX1 / Y1 / X2 / Y2
 * so they bind to Planar_SLAM::Frame, KeyFrame and MapLine as they are and to the stand-ins of tests/native/linematcher_caller.cpp.
 * A line frame needs no device residency (a frame holds at most 40 key lines: they travel with the call), only a context: the one
 * bound to the calling thread (LSDmatcher::BindThread, or ORBmatcher::BindThread's device).  mvScaleFactors / mfLogScaleFactor are
 * the context's own tables (same ORB parameters as the frames': include/drfe.h drfe_orb_scale_tables).
 * Every method flattens what the reference's loop reads through the pointers, calls the C entry point (projection, GetLinesInArea,
 * Hamming distances, claim order: on the device), writes MapLine* back, and - Fuse - applies Replace / AddObservation / AddMapLine
 * on the host in the reference's order, re-reading isBad() / GetMapLine() where the reference reads them. */
namespace drfe_detail {
template <class KL> inline drfe_keyline keyline_of(const KL& k)
{
    drfe_keyline o;
    o.angle = k.angle; o.class_id = k.class_id; o.octave = k.octave; o.pt_x = k.pt.x; o.pt_y = k.pt.y; o.response = k.response; o.size = k.size;
    o.start_point_x = k.startPointX; o.start_point_y = k.startPointY; o.end_point_x = k.endPointX; o.end_point_y = k.endPointY;
    o.s_point_in_octave_x = k.sPointInOctaveX; o.s_point_in_octave_y = k.sPointInOctaveY;
    o.e_point_in_octave_x = k.ePointInOctaveX; o.e_point_in_octave_y = k.ePointInOctaveY;
    o.line_length = k.lineLength; o.num_of_pixels = k.numOfPixels;
    return o;
}
template <class V> inline std::vector<drfe_keyline> keylines_of(const V& v)
{
    std::vector<drfe_keyline> o(v.size());
    for (size_t i = 0; i < v.size(); i++) o[i] = keyline_of(v[i]);
    return o;
}
/* what the projection loops read of a MapLine: GetWorldPos (Vector6d), GetNormal (Vector3d), the distance band, GetDescriptor */
template <class ML> inline void frustum_line_of(ML* p, drfe_frustum_line& o, uint8_t* desc32)
{
    const auto P = p->GetWorldPos();
    for (int k = 0; k < 6; k++) o.world[k] = P(k);
    const auto Pn = p->GetNormal();
    for (int k = 0; k < 3; k++) o.normal[k] = Pn(k);
    o.min_distance = p->GetMinDistanceInvariance(); o.max_distance = p->GetMaxDistanceInvariance();
    const auto d = p->GetDescriptor(); std::memcpy(desc32, d.data, 32);
}
}  // namespace drfe_detail

class LSDmatcher {
public:
    static const int TH_HIGH = 100, TH_LOW = 50;                          /* src/LSDmatcher.cpp:13-14 */
    LSDmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    /* the calling thread's context: once per thread that constructs line matchers (ORBmatcher::BindThread does it too) */
    static void BindThread(drfe_ctx* ctx) { drfe_detail::thread_ctx() = ctx; }
    static drfe_ctx* Context()
    {
        drfe_ctx* c = drfe_detail::thread_ctx();
        if (!c) throw std::runtime_error("LSDmatcher: no context bound to this thread (LSDmatcher::BindThread / ORBmatcher::BindThread)");
        return c;
    }

    /* ---- src/LSDmatcher.cpp:242-279 - Tracking::TrackReferenceKeyFrame / TrackWithMotionModel / Relocalization
     * (src/Tracking.cc:2189, :2323, :2445, :2572) ---- */
    template <class KeyFrameT, class FrameT, class MapLineT>
    typename std::enable_if<std::is_class<FrameT>::value, int>::type
    SearchByDescriptor(KeyFrameT* pKF, FrameT& currentF, std::vector<MapLineT*>& vpMapLineMatches)
    {
        drfe_ctx* c = Context();
        const std::vector<MapLineT*> vpMapLinesKF = pKF->GetMapLineMatches();
        vpMapLineMatches = std::vector<MapLineT*>((size_t)currentF.NL, static_cast<MapLineT*>(nullptr));
        const int nq = pKF->mLineDescriptors.rows, nt = currentF.mLdesc.rows;
        std::vector<uint8_t> has((size_t)nq, 0);
        for (int q = 0; q < nq && q < (int)vpMapLinesKF.size(); q++) has[(size_t)q] = vpMapLinesKF[(size_t)q] != nullptr;
        std::vector<int32_t> m((size_t)nt, -1);
        int n = 0;
        drfe_detail::check(drfe_lsd_search_by_descriptor(c, drfe_detail::u8(pKF->mLineDescriptors), nq, drfe_detail::u8(currentF.mLdesc), nt, has.data(), 0,
                                                         m.data(), &n), c, "drfe_lsd_search_by_descriptor");
        for (int t = 0; t < nt && t < (int)vpMapLineMatches.size(); t++) if (m[(size_t)t] >= 0) vpMapLineMatches[(size_t)t] = vpMapLinesKF[(size_t)m[(size_t)t]];
        return n;
    }

    /* ---- src/LSDmatcher.cpp:281-314 ---- */
    template <class KeyFrameT, class MapLineT>
    int SearchByDescriptor(KeyFrameT* pKF, KeyFrameT* pKF2, std::vector<MapLineT*>& vpMapLineMatches)
    {
        drfe_ctx* c = Context();
        const std::vector<MapLineT*> vpMapLinesKF = pKF->GetMapLineMatches();
        const std::vector<MapLineT*> vpMapLinesKF2 = pKF2->GetMapLineMatches();
        vpMapLineMatches = std::vector<MapLineT*>(vpMapLinesKF.size(), static_cast<MapLineT*>(nullptr));
        const int nq = pKF->mLineDescriptors.rows, nt = pKF2->mLineDescriptors.rows;
        std::vector<uint8_t> has((size_t)nt, 0);
        for (int t = 0; t < nt && t < (int)vpMapLinesKF2.size(); t++) has[(size_t)t] = vpMapLinesKF2[(size_t)t] != nullptr;
        std::vector<int32_t> m((size_t)nq, -1);
        int n = 0;
        drfe_detail::check(drfe_lsd_search_by_descriptor(c, drfe_detail::u8(pKF->mLineDescriptors), nq, drfe_detail::u8(pKF2->mLineDescriptors), nt, has.data(), 1,
                                                         m.data(), &n), c, "drfe_lsd_search_by_descriptor");
        for (int q = 0; q < nq && q < (int)vpMapLineMatches.size(); q++) if (m[(size_t)q] >= 0) vpMapLineMatches[(size_t)q] = vpMapLinesKF2[(size_t)m[(size_t)q]];
        return n;
    }

    /* ---- src/LSDmatcher.cpp:20-139 - the motion-model search of the line tracker ---- */
    template <class FrameT, class LastT>
    typename std::enable_if<std::is_class<LastT>::value, int>::type
    SearchByProjection(FrameT& CurrentFrame, const LastT& LastFrame, const float th, const bool bMono)
    {
        drfe_ctx* c = Context();
        const int nl = (int)LastFrame.NL, nc = (int)CurrentFrame.mvKeylinesUn.size();
        std::vector<drfe_map_line> ml((size_t)nl);
        for (int i = 0; i < nl; i++) {
            auto* p = LastFrame.mvpMapLines[(size_t)i];
            drfe_map_line& r = ml[(size_t)i];
            std::memset(&r, 0, sizeof(r));
            r.valid = p && !p->isBad() && !LastFrame.mvbLineOutlier[(size_t)i];
            if (!r.valid) continue;
            r.octave = LastFrame.mvKeylinesUn[(size_t)i].octave;
            r.obs_positive = p->Observations() > 0;
            const auto P = p->GetWorldPos();
            for (int k = 0; k < 6; k++) r.world[k] = P(k);
            const auto d = p->GetDescriptor(); std::memcpy(r.desc, d.data, 32);
        }
        const std::vector<drfe_keyline> kl = drfe_detail::keylines_of(CurrentFrame.mvKeylinesUn);
        /* claims the frame already holds: any index >= nl (a new match is an index into LastFrame) */
        std::vector<int32_t> claim((size_t)nc);
        std::vector<uint8_t> obs((size_t)nc);
        for (int i = 0; i < nc; i++) { auto* p = CurrentFrame.mvpMapLines[(size_t)i]; claim[(size_t)i] = p ? nl + i : -1; obs[(size_t)i] = p && p->Observations() > 0; }
        const drfe_camera cam = drfe_detail::camera_of(CurrentFrame);
        int n = 0;
        drfe_detail::check(drfe_lsd_search_by_projection_last(c, drfe_detail::f32(CurrentFrame.mTcw), drfe_detail::f32(LastFrame.mTcw), &cam, ml.data(), nl, kl.data(),
                                                              drfe_detail::u8(CurrentFrame.mLdesc), nc, th, bMono ? 1 : 0, mfNNratio, obs.data(), claim.data(), &n), c,
                           "drfe_lsd_search_by_projection_last");
        for (int i = 0; i < nc; i++) if (claim[(size_t)i] >= 0 && claim[(size_t)i] < nl) CurrentFrame.mvpMapLines[(size_t)i] = LastFrame.mvpMapLines[(size_t)claim[(size_t)i]];
        return n;
    }

    /* ---- src/LSDmatcher.cpp:141-211 - the local map's lines into the frame, after Frame::isInFrustum(MapLine*) ---- */
    template <class FrameT, class MapLineT>
    int SearchByProjection(FrameT& F, const std::vector<MapLineT*>& vpMapLines, const float th = 3)
    {
        drfe_ctx* c = Context();
        const int m = (int)vpMapLines.size(), nc = (int)F.mvKeylinesUn.size();
        std::vector<drfe_tracked_line> tl((size_t)m);
        for (int i = 0; i < m; i++) {
            MapLineT* p = vpMapLines[(size_t)i];
            drfe_tracked_line& t = tl[(size_t)i];
            std::memset(&t, 0, sizeof(t));
            t.in_view = p && !p->isBad() && p->mbTrackInView;
            if (!t.in_view) continue;
            t.level = p->mnTrackScaleLevel; t.obs_positive = p->Observations() > 0;
            t.x1 = p->mTrackProjX1; t.y1 = p->mTrackProjY1; t.x2 = p->mTrackProjX2; t.y2 = p->mTrackProjY2; t.view_cos = p->mTrackViewCos;
            const auto d = p->GetDescriptor(); std::memcpy(t.desc, d.data, 32);
        }
        const std::vector<drfe_keyline> kl = drfe_detail::keylines_of(F.mvKeylinesUn);
        std::vector<int32_t> claim((size_t)nc);
        std::vector<uint8_t> obs((size_t)nc);
        for (int i = 0; i < nc; i++) { auto* p = F.mvpMapLines[(size_t)i]; claim[(size_t)i] = p ? m + i : -1; obs[(size_t)i] = p && p->Observations() > 0; }
        int n = 0;
        drfe_detail::check(drfe_lsd_search_by_projection_map(c, tl.data(), m, kl.data(), drfe_detail::u8(F.mLdesc), nc, th, mfNNratio, obs.data(), claim.data(), &n), c,
                           "drfe_lsd_search_by_projection_map");
        for (int i = 0; i < nc; i++) if (claim[(size_t)i] >= 0 && claim[(size_t)i] < m) F.mvpMapLines[(size_t)i] = vpMapLines[(size_t)claim[(size_t)i]];
        return n;
    }

    /* ---- src/LSDmatcher.cpp:377-502 - a keyframe against the lines of a loop candidate's neighbourhood ---- */
    template <class KeyFrameT, class MatT, class MapLineT>
    int SearchByProjection(KeyFrameT* pKF, MatT Scw, const std::vector<MapLineT*>& vpLines, std::vector<MapLineT*>& vpMatched, int th)
    {
        drfe_ctx* c = Context();
        const int n = (int)vpLines.size(), nk = (int)pKF->mvKeyLines.size();
        if ((int)vpMatched.size() < nk) throw std::runtime_error("LSDmatcher::SearchByProjection: vpMatched must hold one entry per key line of pKF");
        std::set<MapLineT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapLineT*>(nullptr));
        std::vector<drfe_frustum_line> fl((size_t)n);
        std::vector<uint8_t> descs((size_t)n * 32), skip((size_t)n), matched((size_t)nk);
        for (int i = 0; i < n; i++) {
            MapLineT* p = vpLines[(size_t)i];
            std::memset(&fl[(size_t)i], 0, sizeof(drfe_frustum_line));
            skip[(size_t)i] = !p || p->isBad() || spAlreadyFound.count(p);
            if (!skip[(size_t)i]) drfe_detail::frustum_line_of(p, fl[(size_t)i], descs.data() + 32 * (size_t)i);
        }
        for (int k = 0; k < nk; k++) matched[(size_t)k] = vpMatched[(size_t)k] != nullptr;
        const std::vector<drfe_keyline> kl = drfe_detail::keylines_of(pKF->mvKeyLines);
        const drfe_camera cam = drfe_detail::camera_of(*pKF);
        std::vector<int32_t> nm((size_t)nk, -1);
        int cnt = 0;
        drfe_detail::check(drfe_lsd_search_by_projection_kf(c, drfe_detail::f32(Scw), &cam, fl.data(), descs.data(), skip.data(), n, kl.data(),
                                                            drfe_detail::u8(pKF->mLineDescriptors), nk, matched.data(), th, nm.data(), &cnt), c,
                           "drfe_lsd_search_by_projection_kf");
        for (int k = 0; k < nk; k++) if (nm[(size_t)k] >= 0) vpMatched[(size_t)k] = vpLines[(size_t)nm[(size_t)k]];
        return cnt;
    }

    /* ---- src/LSDmatcher.cpp:504-748 ---- */
    template <class KeyFrameT, class MapLineT, class MatT>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapLineT*>& vpMatches12, const float& s12, const MatT& R12, const MatT& t12, const float th)
    {
        drfe_ctx* c = Context();
        const std::vector<MapLineT*> vp1 = pKF1->GetMapLineMatches(), vp2 = pKF2->GetMapLineMatches();
        const int N1 = (int)vp1.size(), N2 = (int)vp2.size();
        if ((int)pKF1->mvKeyLines.size() != N1 || (int)pKF2->mvKeyLines.size() != N2)
            throw std::runtime_error("LSDmatcher::SearchBySim3: GetMapLineMatches() must hold one entry per key line");
        std::vector<uint8_t> skip1((size_t)N1, 0), skip2((size_t)N2, 0);
        for (int i = 0; i < N1; i++) {                                     /* :533-543 */
            MapLineT* p = vpMatches12[(size_t)i];
            if (p) {
                skip1[(size_t)i] = 1;
                const int idx2 = p->GetIndexInKeyFrame(pKF2);
                if (idx2 >= 0 && idx2 < N2) skip2[(size_t)idx2] = 1;
            }
        }
        std::vector<drfe_frustum_line> f1((size_t)N1), f2((size_t)N2);
        std::vector<uint8_t> d1((size_t)N1 * 32), d2((size_t)N2 * 32);
        for (int i = 0; i < N1; i++) {
            MapLineT* p = vp1[(size_t)i];
            std::memset(&f1[(size_t)i], 0, sizeof(drfe_frustum_line));
            if (!p || p->isBad()) skip1[(size_t)i] = 1;
            if (!skip1[(size_t)i]) drfe_detail::frustum_line_of(p, f1[(size_t)i], d1.data() + 32 * (size_t)i);
        }
        for (int i = 0; i < N2; i++) {
            MapLineT* p = vp2[(size_t)i];
            std::memset(&f2[(size_t)i], 0, sizeof(drfe_frustum_line));
            if (!p || p->isBad()) skip2[(size_t)i] = 1;
            if (!skip2[(size_t)i]) drfe_detail::frustum_line_of(p, f2[(size_t)i], d2.data() + 32 * (size_t)i);
        }
        const std::vector<drfe_keyline> kl1 = drfe_detail::keylines_of(pKF1->mvKeyLines), kl2 = drfe_detail::keylines_of(pKF2->mvKeyLines);
        const auto T1w = pKF1->GetPose();
        const auto T2w = pKF2->GetPose();
        const drfe_camera cam = drfe_detail::camera_of(*pKF1);
        std::vector<int32_t> m12((size_t)N1, -1);
        int nFound = 0;
        drfe_detail::check(drfe_lsd_search_by_sim3(c, &cam, drfe_detail::f32(T1w), drfe_detail::f32(T2w), s12, drfe_detail::f32(R12), drfe_detail::f32(t12), f1.data(),
                                                   d1.data(), skip1.data(), kl1.data(), drfe_detail::u8(pKF1->mLineDescriptors), N1, f2.data(), d2.data(), skip2.data(),
                                                   kl2.data(), drfe_detail::u8(pKF2->mLineDescriptors), N2, th, m12.data(), &nFound), c, "drfe_lsd_search_by_sim3");
        for (int i1 = 0; i1 < N1; i1++) if (m12[(size_t)i1] >= 0) vpMatches12[(size_t)i1] = vp2[(size_t)m12[(size_t)i1]];
        return nFound;
    }

    /* ---- src/LSDmatcher.cpp:213-240 - Tracking::MonocularInitialization (src/Tracking.cc:1697); the reference spells it "Serach" ---- */
    template <class FrameT>
    typename std::enable_if<std::is_class<FrameT>::value, int>::type
    SerachForInitialize(FrameT& InitialFrame, FrameT& CurrentFrame, std::vector<std::pair<int, int>>& LineMatches)
    {
        drfe_ctx* c = Context();
        LineMatches.clear();
        const int nq = InitialFrame.mLdesc.rows, nt = CurrentFrame.mLdesc.rows;
        std::vector<int32_t> m((size_t)nq, -1);
        int n = 0;
        drfe_detail::check(drfe_lsd_search_by_descriptor(c, drfe_detail::u8(InitialFrame.mLdesc), nq, drfe_detail::u8(CurrentFrame.mLdesc), nt, nullptr, 1, m.data(), &n), c,
                           "drfe_lsd_search_by_descriptor");
        for (int q = 0; q < nq; q++) if (m[(size_t)q] >= 0) LineMatches.push_back(std::make_pair(q, (int)m[(size_t)q]));
        return n;
    }

    /* ---- src/LSDmatcher.cpp:334-367 - LocalMapping::CreateNewMapLines (src/LocalMapping.cc:606, :858) ---- */
    template <class KeyFrameT>
    typename std::enable_if<std::is_class<KeyFrameT>::value, int>::type
    SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs)
    {
        drfe_ctx* c = Context();
        vMatchedPairs.clear();
        const int n1 = pKF1->mLineDescriptors.rows, n2 = pKF2->mLineDescriptors.rows;
        std::vector<uint8_t> has1((size_t)n1), has2((size_t)n2);
        for (int i = 0; i < n1; i++) has1[(size_t)i] = pKF1->GetMapLine((size_t)i) != nullptr;
        for (int i = 0; i < n2; i++) has2[(size_t)i] = pKF2->GetMapLine((size_t)i) != nullptr;
        std::vector<int32_t> m((size_t)n1, -1);
        int n = 0;
        drfe_detail::check(drfe_lsd_search_for_triangulation(c, drfe_detail::u8(pKF1->mLineDescriptors), n1, drfe_detail::u8(pKF2->mLineDescriptors), n2, has1.data(),
                                                             has2.data(), m.data(), &n), c, "drfe_lsd_search_for_triangulation");
        for (int i = 0; i < n1; i++) if (m[(size_t)i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m[(size_t)i]));
        return n;
    }

    /* ---- src/LSDmatcher.cpp:884-1015 - LocalMapping::SearchInNeighbors (src/LocalMapping.cc:1103, :1124).  The search of every line on
     * the device (it reads the keyframe's key lines and descriptors, never its MapLine assignments); the loop that applies it runs here
     * in the reference's order and re-reads isBad() / GetMapLine() at each step: an earlier Replace or AddMapLine of this very loop
     * changes what a later line meets (:907, :995).  A predicted level outside the pyramid (the reference reads mvScaleFactors out of
     * bounds there) fuses nothing. ---- */
    template <class KeyFrameT, class MapLineT>
    int Fuse(KeyFrameT* pKF, const std::vector<MapLineT*>& vpMapLines, const float th = 3.0)
    {
        drfe_ctx* c = Context();
        const int n = (int)vpMapLines.size(), nk = (int)pKF->mvKeyLines.size();
        std::vector<drfe_frustum_line> fl((size_t)n);
        std::vector<uint8_t> descs((size_t)n * 32), skip((size_t)n);
        for (int i = 0; i < n; i++) {
            MapLineT* p = vpMapLines[(size_t)i];
            std::memset(&fl[(size_t)i], 0, sizeof(drfe_frustum_line));
            skip[(size_t)i] = !p || p->isBad();
            if (!skip[(size_t)i]) drfe_detail::frustum_line_of(p, fl[(size_t)i], descs.data() + 32 * (size_t)i);
        }
        const std::vector<drfe_keyline> kl = drfe_detail::keylines_of(pKF->mvKeyLines);
        const drfe_camera cam = drfe_detail::camera_of(*pKF);
        const auto Tcw = pKF->GetPose();
        std::vector<int32_t> bestIdx((size_t)n, -1), bestDist((size_t)n, 0);
        drfe_detail::check(drfe_lsd_fuse_search(c, drfe_detail::f32(Tcw), &cam, fl.data(), descs.data(), skip.data(), n, kl.data(), drfe_detail::u8(pKF->mLineDescriptors),
                                                nk, th, bestIdx.data(), bestDist.data()), c, "drfe_lsd_fuse_search");
        int nFused = 0;
        for (int i = 0; i < n; i++) {
            MapLineT* pML = vpMapLines[(size_t)i];
            if (!pML || pML->isBad()) continue;
            if (bestIdx[(size_t)i] < 0 || bestDist[(size_t)i] > TH_LOW) continue;
            MapLineT* pMLinKF = pKF->GetMapLine((size_t)bestIdx[(size_t)i]);
            if (pMLinKF) {
                if (!pMLinKF->isBad()) {
                    if (pMLinKF->Observations() > pML->Observations()) pML->Replace(pMLinKF);
                    else pMLinKF->Replace(pML);
                }
            } else {
                pML->AddObservation(pKF, (size_t)bestIdx[(size_t)i]);
                pKF->AddMapLine(pML, (size_t)bestIdx[(size_t)i]);
            }
            nFused++;
        }
        return nFused;
    }

    /* ---- src/LSDmatcher.cpp:750-882 ---- */
    template <class KeyFrameT, class MatT, class MapLineT>
    int Fuse(KeyFrameT* pKF, MatT Scw, const std::vector<MapLineT*>& vpLines, float th, std::vector<MapLineT*>& vpReplaceLine)
    {
        drfe_ctx* c = Context();
        const std::set<MapLineT*> spAlreadyFound = pKF->GetMapLines();
        const int n = (int)vpLines.size(), nk = (int)pKF->mvKeyLines.size();
        std::vector<drfe_frustum_line> fl((size_t)n);
        std::vector<uint8_t> descs((size_t)n * 32), skip((size_t)n);
        for (int i = 0; i < n; i++) {
            MapLineT* p = vpLines[(size_t)i];
            std::memset(&fl[(size_t)i], 0, sizeof(drfe_frustum_line));
            skip[(size_t)i] = !p || p->isBad() || spAlreadyFound.count(p);
            if (!skip[(size_t)i]) drfe_detail::frustum_line_of(p, fl[(size_t)i], descs.data() + 32 * (size_t)i);
        }
        const std::vector<drfe_keyline> kl = drfe_detail::keylines_of(pKF->mvKeyLines);
        const drfe_camera cam = drfe_detail::camera_of(*pKF);
        std::vector<int32_t> bestIdx((size_t)n, -1), bestDist((size_t)n, 0);
        drfe_detail::check(drfe_lsd_fuse_search_sim3(c, drfe_detail::f32(Scw), &cam, fl.data(), descs.data(), skip.data(), n, kl.data(), drfe_detail::u8(pKF->mLineDescriptors),
                                                     nk, th, bestIdx.data(), bestDist.data()), c, "drfe_lsd_fuse_search_sim3");
        int nFused = 0;
        for (int i = 0; i < n; i++) {
            if (skip[(size_t)i] || bestIdx[(size_t)i] < 0 || bestDist[(size_t)i] > TH_LOW) continue;
            MapLineT* pML = vpLines[(size_t)i];
            MapLineT* pMLinKF = pKF->GetMapLine((size_t)bestIdx[(size_t)i]);
            if (pMLinKF) {
                if (!pMLinKF->isBad()) vpReplaceLine[(size_t)i] = pMLinKF;
            } else {
                pML->AddObservation(pKF, (size_t)bestIdx[(size_t)i]);
                pKF->AddMapLine(pML, (size_t)bestIdx[(size_t)i]);
            }
            nFused++;
        }
        return nFused;
    }

    /* src/LSDmatcher.cpp:316-332 on two 1 x 32 CV_8U rows */
    template <class MatT>
    static typename std::enable_if<std::is_class<MatT>::value, int>::type
    DescriptorDistance(const MatT& a, const MatT& b) { return DescriptorDistance(drfe_detail::u8(a), drfe_detail::u8(b)); }

    /* ---- flat forms (descriptor rows / key lines / records handed over directly: what the templates above call, for callers that
     * hold the flattened data already - tests/native/members_caller.cpp) ---- */
    /* SearchByDescriptor(KeyFrame* pKF, Frame& currentF, vpMapLineMatches), src/LSDmatcher.cpp:242-279: descKF / descF = the LBD rows,
     * kfHasLine[i] = pKF's line i has a MapLine; matches[line of currentF] = line of pKF or -1 */
    int SearchByDescriptor(drfe_ctx* ctx, const drfe_cv::Mat& descKF, const std::vector<uint8_t>& kfHasLine, const drfe_cv::Mat& descF,
                           std::vector<int32_t>& matches)
    {
        int n = 0;
        matches.assign(descF.rows, -1);
        drfe_detail::check(drfe_lsd_search_by_descriptor(ctx, descKF.data, descKF.rows, descF.data, descF.rows, kfHasLine.data(), 0, matches.data(), &n),
                           ctx, "drfe_lsd_search_by_descriptor");
        return n;
    }
    /* SearchByDescriptor(KeyFrame*, KeyFrame*, ...) (:281-314) and SerachForInitialize(InitialFrame, CurrentFrame, LineMatches) (:213-240):
     * matches[line of the first] = line of the second or -1; trainHasLine NULL = all */
    int SearchByDescriptorKF(drfe_ctx* ctx, const drfe_cv::Mat& desc1, const drfe_cv::Mat& desc2, const std::vector<uint8_t>* trainHasLine,
                             std::vector<int32_t>& matches)
    {
        int n = 0;
        matches.assign(desc1.rows, -1);
        drfe_detail::check(drfe_lsd_search_by_descriptor(ctx, desc1.data, desc1.rows, desc2.data, desc2.rows, trainHasLine ? trainHasLine->data() : nullptr,
                                                         1, matches.data(), &n), ctx, "drfe_lsd_search_by_descriptor");
        return n;
    }
    int SerachForInitialize(drfe_ctx* ctx, const drfe_cv::Mat& descInitial, const drfe_cv::Mat& descCurrent, std::vector<std::pair<int, int>>& LineMatches)
    {
        std::vector<int32_t> m;
        const int n = SearchByDescriptorKF(ctx, descInitial, descCurrent, nullptr, m);
        LineMatches.clear();
        for (size_t i = 0; i < m.size(); i++) if (m[i] >= 0) LineMatches.push_back(std::make_pair((int)i, (int)m[i]));
        return n;
    }
    /* SearchForTriangulation(pKF1, pKF2, vMatchedPairs), :334-367 */
    int SearchForTriangulation(drfe_ctx* ctx, const drfe_cv::Mat& desc1, const drfe_cv::Mat& desc2, const std::vector<uint8_t>& has1,
                               const std::vector<uint8_t>& has2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs)
    {
        int n = 0;
        std::vector<int32_t> m(desc1.rows, -1);
        drfe_detail::check(drfe_lsd_search_for_triangulation(ctx, desc1.data, desc1.rows, desc2.data, desc2.rows, has1.data(), has2.data(), m.data(), &n),
                           ctx, "drfe_lsd_search_for_triangulation");
        vMatchedPairs.clear();
        for (size_t i = 0; i < m.size(); i++) if (m[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m[i]));
        return n;
    }
    /* SearchByProjection(CurrentFrame, LastFrame, th, bMono), :20-108: lastLines[i] = what the loop reads of LastFrame.mvpMapLines[i];
     * curMapLine in/out = index into lastLines or -1, curObs[i] = the MapLine current line i already holds has Observations() > 0 */
    int SearchByProjection(drfe_ctx* ctx, const float* TcwCur, const float* TcwLast, const drfe_camera& cam, const std::vector<drfe_map_line>& lastLines,
                           const std::vector<drfe_cv::KeyLine>& curLines, const drfe_cv::Mat& curDesc, const std::vector<uint8_t>* curObs,
                           std::vector<int32_t>& curMapLine, float th, bool bMono)
    {
        int n = 0;
        drfe_detail::check(drfe_lsd_search_by_projection_last(ctx, TcwCur, TcwLast, &cam, lastLines.data(), (int)lastLines.size(),
                                                              reinterpret_cast<const drfe_keyline*>(curLines.data()), curDesc.data, (int)curLines.size(), th,
                                                              bMono ? 1 : 0, mfNNratio, curObs ? curObs->data() : nullptr, curMapLine.data(), &n),
                           ctx, "drfe_lsd_search_by_projection_last");
        return n;
    }
    /* SearchByProjection(F, vpMapLines, th), :110-211 after Frame::isInFrustum(MapLine*) left its fields in `tracked` */
    int SearchByProjection(drfe_ctx* ctx, const std::vector<drfe_tracked_line>& tracked, const std::vector<drfe_cv::KeyLine>& curLines,
                           const drfe_cv::Mat& curDesc, const std::vector<uint8_t>* curObs, std::vector<int32_t>& curMapLine, float th = 3)
    {
        int n = 0;
        drfe_detail::check(drfe_lsd_search_by_projection_map(ctx, tracked.data(), (int)tracked.size(), reinterpret_cast<const drfe_keyline*>(curLines.data()),
                                                             curDesc.data, (int)curLines.size(), th, mfNNratio, curObs ? curObs->data() : nullptr,
                                                             curMapLine.data(), &n), ctx, "drfe_lsd_search_by_projection_map");
        return n;
    }
    /* the search of Fuse(pKF, vpMapLines, th), :884-1015: bestIdx / bestDist per map line; the caller applies `<= TH_LOW` and the
     * Replace / AddObservation surgery on its map graph */
    void FuseSearch(drfe_ctx* ctx, const float* Tcw, const drfe_camera& cam, const std::vector<drfe_frustum_line>& lines, const drfe_cv::Mat& descs,
                    const std::vector<uint8_t>& skip, const std::vector<drfe_cv::KeyLine>& kfLines, const drfe_cv::Mat& kfDesc, float th,
                    std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist)
    {
        bestIdx.assign(lines.size(), -1); bestDist.assign(lines.size(), 0);
        drfe_detail::check(drfe_lsd_fuse_search(ctx, Tcw, &cam, lines.data(), descs.data, skip.data(), (int)lines.size(),
                                                reinterpret_cast<const drfe_keyline*>(kfLines.data()), kfDesc.data, (int)kfLines.size(), th, bestIdx.data(),
                                                bestDist.data()), ctx, "drfe_lsd_fuse_search");
    }
    /* src/LSDmatcher.cpp:316-332: cv::norm(a, b, NORM_HAMMING) of two 32-byte LBD rows */
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b)
    {
        int dist = 0;
        for (int i = 0; i < 32; i++) dist += __builtin_popcount((unsigned)(a[i] ^ b[i]));
        return dist;
    }
protected:
    float mfNNratio; bool mbCheckOrientation;
};

}  // namespace Planar_SLAM

/* include/LSDextractor.h:342-350 (global namespace in the reference).
 * The reference reaches this class through `LineSegment* mpLineSegment` (include/Frame.h:157), a member no constructor ever
 * initialises: `mpLineSegment->ExtractLineSegment(...)` (src/Frame.cc:241) works there only because the method touches no member.
 * The same holds here: ExtractLineSegment reads NO member of `this` - the object is default-constructible and empty - and takes its
 * context from the calling thread's binding (LineSegment::BindThread, LSDmatcher::BindThread or ORBmatcher::BindThread), else from
 * the process-wide one (LineSegment::BindProcess: Frame::Frame starts a fresh thread for ExtractLSD on every frame, src/Frame.cc:129,
 * which has no binding of its own; calls through the process-wide context are serialised). */
class LineSegment {
public:
    LineSegment() {}
    explicit LineSegment(drfe_ctx* ctx) { BindThread(ctx); }
    static void BindThread(drfe_ctx* ctx) { Planar_SLAM::drfe_detail::thread_ctx() = ctx; }
    static void BindProcess(drfe_ctx* ctx) { std::lock_guard<std::mutex> g(process_mutex()); process_ctx() = ctx; }
    /* keylineFunctions[i] = normalised sp x ep: std::vector<Eigen::Vector3d> as in the reference (include/LSDextractor.h:349) when
     * built with -DDRFE_WITH_EIGEN, the three-double stand-in with the same element access otherwise */
    void ExtractLineSegment(const drfe_cv::Mat& img, std::vector<drfe_cv::KeyLine>& keylines, drfe_cv::Mat& ldesc,
                            std::vector<drfe_cv::Vector3d>& keylineFunctions, float /*scale*/ = 1.2f, int /*numOctaves*/ = 1)
    {
        drfe_ctx* ctx = Planar_SLAM::drfe_detail::thread_ctx();
        std::unique_lock<std::mutex> guard;
        if (!ctx) {
            guard = std::unique_lock<std::mutex>(process_mutex());
            ctx = process_ctx();
            if (!ctx) throw std::runtime_error("LineSegment: no context bound (LineSegment::BindThread / BindProcess)");
        }
        const int cap = 40;                                            /* lsdNFeatures, src/LSDextractor.cpp:20-28 */
        std::vector<drfe_keyline> kl(cap);
        drfe_cv::Mat desc = drfe_cv::mat_u8(cap, 32);
        std::vector<double> lf(3 * cap);
        int n = 0, found = 0;
        Planar_SLAM::drfe_detail::check(drfe_lsd_extract(ctx, drfe_cv::mat_data(img), img.cols, img.rows, drfe_cv::mat_step(img), cap,
                                                         kl.data(), desc.data, lf.data(), cap, &n, &found), ctx, "drfe_lsd_extract");
        keylines.resize(n);
        keylineFunctions.clear();
        for (int i = 0; i < n; i++) {
            drfe_cv::KeyLine& k = keylines[i];
            k.angle = kl[i].angle; k.class_id = kl[i].class_id; k.octave = kl[i].octave;
            k.pt.x = kl[i].pt_x; k.pt.y = kl[i].pt_y; k.response = kl[i].response; k.size = kl[i].size;
            k.startPointX = kl[i].start_point_x; k.startPointY = kl[i].start_point_y; k.endPointX = kl[i].end_point_x; k.endPointY = kl[i].end_point_y;
            k.sPointInOctaveX = kl[i].s_point_in_octave_x; k.sPointInOctaveY = kl[i].s_point_in_octave_y;
            k.ePointInOctaveX = kl[i].e_point_in_octave_x; k.ePointInOctaveY = kl[i].e_point_in_octave_y;
            k.lineLength = kl[i].line_length; k.numOfPixels = kl[i].num_of_pixels;
            keylineFunctions.push_back(drfe_cv::Vector3d(lf[3 * i], lf[3 * i + 1], lf[3 * i + 2]));
        }
        ldesc = drfe_cv::mat_u8(n, 32);
        if (n) std::memcpy(ldesc.data, desc.data, (size_t)n * 32);
    }
private:
    static drfe_ctx*& process_ctx() { static drfe_ctx* c = nullptr; return c; }
    static std::mutex& process_mutex() { static std::mutex m; return m; }
};

#endif /* DRFE_ADAPTOR_HPP */
