/* members_caller.cpp — replays, through include/drfe_adaptor.hpp, the member accesses the reference makes on its extractor and
 * matcher objects, spelled as in the reference:
 *   Frame::ComputePlanes           src/Frame.cc:947-979    planeDetector.plane_num_ / .plane_vertices_[i] / .cloud.vertices[j][k] /
 *                                                          .plane_filter.extractedPlanes[i]->normal / ->center / .seg_output
 *   Frame::ComputePlanes_CAPE      src/Frame.cc:1096-1121  planeDetectionCape.nr_planes / .plane_cloud[i] / .plane_params[i].normal / .d
 *   LineSegment::ExtractLineSegment include/LSDextractor.h:349 with std::vector<Vector3d> keylineFunctions
 *   LSDmatcher                     include/LSDmatcher.h:19-50: SearchByDescriptor (KF, Frame), SerachForInitialize, SearchForTriangulation
 * usage: members_caller <gray0.raw> <gray1.raw> <depth16.raw> <w> <h> <out.bin>; tests/test_gpu_native.py compares out.bin with the
 * ctypes path. */
#include "drfe_adaptor.hpp"

#include <cstdio>
#include <cstdlib>

static std::vector<uint8_t> slurp(const char* path, size_t n)
{
    std::vector<uint8_t> b(n);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(b.data(), 1, n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::fclose(f);
    return b;
}

struct PointT { float x, y, z; };

int main(int argc, char** argv)
{
    if (argc != 7) return 2;
    const int w = std::atoi(argv[4]), h = std::atoi(argv[5]);
    std::vector<uint8_t> gray0 = slurp(argv[1], (size_t)w * h), gray1 = slurp(argv[2], (size_t)w * h), depth = slurp(argv[3], (size_t)w * h * 2);
    const float mMax_point_dist = 9.0f;
    try {
        Planar_SLAM::ORBextractor ex(1000, 1.2f, 8, 20, 7, w, h);
        FILE* f = std::fopen(argv[6], "wb");
        const float K[9] = {535.4f, 0, 320.1f, 0, 539.2f, 247.6f, 0, 0, 1};
        const float depthFactor = 1.0f / 5000.0f;
        /* ---- Frame::ComputePlanes, src/Frame.cc:947-979 ---- */
        Planar_SLAM::PlaneDetection planeDetector(ex.context());
        drfe_cv::Mat imGrey(h, w, gray0.data(), (size_t)w), dimg(h, w, depth.data(), (size_t)w * 2, 2);
        planeDetector.readColorImage(imGrey);
        if (!planeDetector.readDepthImage(dimg, K, depthFactor)) return 3;
        planeDetector.runPlaneDetection();
        drfe_cv::Mat seg_out = planeDetector.seg_output;
        int32_t np = planeDetector.plane_num_;
        std::fwrite(&np, 4, 1, f);
        for (int i = 0; i < planeDetector.plane_num_; i++) {
            auto& indices = planeDetector.plane_vertices_[i];
            std::vector<PointT> inputCloud;
            for (int j : indices) {
                PointT p;
                p.x = (float)planeDetector.cloud.vertices[j][0];
                p.y = (float)planeDetector.cloud.vertices[j][1];
                p.z = (float)planeDetector.cloud.vertices[j][2];
                if (p.z > mMax_point_dist) continue;
                inputCloud.push_back(p);
            }
            auto extractedPlane = planeDetector.plane_filter.extractedPlanes[i];
            double nx = extractedPlane->normal[0];
            double ny = extractedPlane->normal[1];
            double nz = extractedPlane->normal[2];
            double cx = extractedPlane->center[0];
            double cy = extractedPlane->center[1];
            double cz = extractedPlane->center[2];
            float d = (float)-(nx * cx + ny * cy + nz * cz);
            const int32_t n = (int32_t)inputCloud.size();
            std::fwrite(&n, 4, 1, f);
            std::fwrite(&d, 4, 1, f);
            std::fwrite(inputCloud.data(), sizeof(PointT), inputCloud.size(), f);
        }
        double gx, gy, gz;
        const int32_t got = planeDetector.cloud.get(h / 2, w / 2, gx, gy, gz) ? 1 : 0;       /* ImagePointCloud::get, include/PlaneExtractor.h:50-58 */
        std::fwrite(&got, 4, 1, f);
        std::fwrite(&gz, 8, 1, f);
        /* ---- Frame::ComputePlanes_CAPE, src/Frame.cc:1096-1121 ---- */
        std::vector<float> depthM((size_t)w * h);
        for (size_t i = 0; i < depthM.size(); i++) depthM[i] = (float)reinterpret_cast<const uint16_t*>(depth.data())[i] * depthFactor;   /* imDepth.convertTo(CV_32F, factor) */
        drfe_cv::Mat dm(h, w, reinterpret_cast<uint8_t*>(depthM.data()), (size_t)w * 4, 4);
        Planar_SLAM::PlaneDetection_CAPE planeDetectionCape(ex.context());
        planeDetectionCape.PATCH_SIZE = 20; planeDetectionCape.MAX_MERGE_DIST = 50.0f;
        planeDetectionCape.readColorImage(imGrey);
        if (!planeDetectionCape.readDepthImage(dm, K)) return 4;
        planeDetectionCape.runPlaneDetection();
        int32_t nc = planeDetectionCape.nr_planes;
        std::fwrite(&nc, 4, 1, f);
        for (int i = 0; i < planeDetectionCape.nr_planes; i++) {
            auto inputCloud = planeDetectionCape.plane_cloud[i];
            double nx = planeDetectionCape.plane_params[i].normal[0];
            double ny = planeDetectionCape.plane_params[i].normal[1];
            double nz = planeDetectionCape.plane_params[i].normal[2];
            double d = planeDetectionCape.plane_params[i].d;
            const double rec[4] = {nx, ny, nz, d};
            const int32_t n = (int32_t)inputCloud->points.size();
            std::fwrite(rec, 8, 4, f);
            std::fwrite(&n, 4, 1, f);
            std::fwrite(inputCloud->points.data(), 12, inputCloud->points.size(), f);
        }
        std::fwrite(planeDetectionCape.seg_output.data, 1, (size_t)w * h, f);
        /* ---- lines of two frames + LSDmatcher ---- */
        LineSegment ls(ex.context());
        drfe_cv::Mat image1(h, w, gray1.data(), (size_t)w);
        std::vector<drfe_cv::KeyLine> kl0, kl1; drfe_cv::Mat ld0, ld1; std::vector<drfe_cv::Vector3d> lf0, lf1;
        ls.ExtractLineSegment(imGrey, kl0, ld0, lf0);
        ls.ExtractLineSegment(image1, kl1, ld1, lf1);
        const int32_t nl[2] = {(int32_t)kl0.size(), (int32_t)kl1.size()};
        std::fwrite(nl, 4, 2, f);
        for (const drfe_cv::Vector3d& v : lf0) { const double t[3] = {v[0], v[1], v[2]}; std::fwrite(t, 8, 3, f); }
        Planar_SLAM::LSDmatcher lmatcher(0.6f, true);
        std::vector<uint8_t> has0(kl0.size()), has1(kl1.size());
        for (size_t i = 0; i < has0.size(); i++) has0[i] = (uint8_t)(i % 5 != 0);
        for (size_t i = 0; i < has1.size(); i++) has1[i] = (uint8_t)(i % 3 == 0);
        std::vector<int32_t> m;
        int32_t n = lmatcher.SearchByDescriptor(ex.context(), ld0, has0, ld1, m);
        std::fwrite(&n, 4, 1, f); std::fwrite(m.data(), 4, m.size(), f);
        std::vector<std::pair<int, int>> LineMatches;
        n = lmatcher.SerachForInitialize(ex.context(), ld0, ld1, LineMatches);
        const int32_t nlm = (int32_t)LineMatches.size();
        std::fwrite(&n, 4, 1, f); std::fwrite(&nlm, 4, 1, f);
        for (auto& pr : LineMatches) { const int32_t t[2] = {pr.first, pr.second}; std::fwrite(t, 4, 2, f); }
        std::vector<uint8_t> none0(kl0.size(), 0);
        std::vector<std::pair<size_t, size_t>> vMatchedPairs;
        n = lmatcher.SearchForTriangulation(ex.context(), ld0, ld1, none0, has1, vMatchedPairs);
        const int32_t nmp = (int32_t)vMatchedPairs.size();
        std::fwrite(&n, 4, 1, f); std::fwrite(&nmp, 4, 1, f);
        for (auto& pr : vMatchedPairs) { const int32_t t[2] = {(int32_t)pr.first, (int32_t)pr.second}; std::fwrite(t, 4, 2, f); }
        const int32_t dd = Planar_SLAM::LSDmatcher::DescriptorDistance(ld0.data, ld1.data);
        std::fwrite(&dd, 4, 1, f);
        std::fclose(f);
        std::printf("members ok: %d AHC planes, %d CAPE planes, %d + %d lines\n", np, nc, nl[0], nl[1]);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "members_caller: %s\n", e.what());
        return 1;
    }
    return 0;
}
