/* drfe_adaptor.hpp — header-only C++ adaptor: the reference's class interfaces over the C-ABI of drfe.h.
 *
 * What a DR-SLAM maintainer drops in place of src/ORBextractor.cc, src/LSDextractor.cpp and src/PlaneExtractor.cpp
 * (INTEGRATION.md): same class names, constructor arguments, methods and public members as
 *   Planar_SLAM::ORBextractor          include/ORBextractor.h:51-85
 *   LineSegment::ExtractLineSegment    include/LSDextractor.h:342-350
 *   Planar_SLAM::PlaneDetection        include/PlaneExtractor.h:61-82
 *   Planar_SLAM::PlaneDetection_CAPE   include/PlaneExtractor.h:84-115
 *   Planar_SLAM::ORBmatcher (DescriptorDistance + the index-level SearchByProjection the MapPoint* overloads wrap)
 *   Planar_SLAM::LSDmatcher            include/LSDmatcher.h:19-50 (index level, like ORBmatcher)
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

/* include/ORBmatcher.h:41-84: the parts that do not touch the MapPoint graph.  The MapPoint* overloads of the reference
 * flatten what their loops read into drfe_map_point / drfe_tracked_point records and call these. */
class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

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

/* include/LSDmatcher.h:19-50: the parts that do not touch the MapLine graph.  The MapLine* / KeyFrame* overloads of the reference
 * flatten what their loops read (descriptor rows, key lines, `has a MapLine` flags, drfe_map_line / drfe_tracked_line /
 * drfe_frustum_line records - INTEGRATION.md section 3b) and call these; results come back as index arrays in the reference's
 * conventions. */
class LSDmatcher {
public:
    static const int TH_HIGH = 100, TH_LOW = 50;                          /* src/LSDmatcher.cpp:13-14 */
    LSDmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

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

/* include/LSDextractor.h:342-350 (global namespace in the reference) */
class LineSegment {
public:
    explicit LineSegment(drfe_ctx* ctx) : mCtx(ctx) {}
    /* keylineFunctions[i] = normalised sp x ep: std::vector<Eigen::Vector3d> as in the reference (include/LSDextractor.h:349) when
     * built with -DDRFE_WITH_EIGEN, the three-double stand-in with the same element access otherwise */
    void ExtractLineSegment(const drfe_cv::Mat& img, std::vector<drfe_cv::KeyLine>& keylines, drfe_cv::Mat& ldesc,
                            std::vector<drfe_cv::Vector3d>& keylineFunctions, float /*scale*/ = 1.2f, int /*numOctaves*/ = 1)
    {
        const int cap = 40;                                            /* lsdNFeatures, src/LSDextractor.cpp:20-28 */
        std::vector<drfe_keyline> kl(cap);
        drfe_cv::Mat desc = drfe_cv::mat_u8(cap, 32);
        std::vector<double> lf(3 * cap);
        int n = 0, found = 0;
        Planar_SLAM::drfe_detail::check(drfe_lsd_extract(mCtx, drfe_cv::mat_data(img), img.cols, img.rows, drfe_cv::mat_step(img), cap,
                                                         kl.data(), desc.data, lf.data(), cap, &n, &found), mCtx, "drfe_lsd_extract");
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
    drfe_ctx* mCtx;
};

#endif /* DRFE_ADAPTOR_HPP */
