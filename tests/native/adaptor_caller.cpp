/* adaptor_caller.cpp — compiles include/drfe_adaptor.hpp (no OpenCV: the stand-in container types) and drives one frame
 * through the reference's class interfaces: ORBextractor(...)(image, mask, keypoints, descriptors), the getters, the public
 * pyramid member, LineSegment::ExtractLineSegment, PlaneDetection::readDepthImage + runPlaneDetection.
 * usage: adaptor_caller <gray.raw> <depth16.raw> <w> <h> <out.bin>; tests/test_gpu_native.py compares out.bin with the ctypes path. */
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

int main(int argc, char** argv)
{
    if (argc != 6) return 2;
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    std::vector<uint8_t> gray = slurp(argv[1], (size_t)w * h), depth = slurp(argv[2], (size_t)w * h * 2);
    try {
        Planar_SLAM::ORBextractor ex(1000, 1.2f, 8, 20, 7, w, h);
        drfe_cv::Mat image(h, w, gray.data(), (size_t)w), mask, desc;
        std::vector<drfe_cv::KeyPoint> kps;
        ex(image, mask, kps, desc);
        drfe_cv::Mat empty;
        std::vector<drfe_cv::KeyPoint> none;
        ex(empty, mask, none, desc);                                      /* empty image: silent return, outputs untouched */
        if (!none.empty() || desc.rows != (int)kps.size()) return 3;
        if (ex.GetLevels() != 8 || ex.GetScaleFactors().size() != 8 || ex.mvImagePyramid[0].cols != w) return 4;
        LineSegment ls(ex.context());
        std::vector<drfe_cv::KeyLine> kl; drfe_cv::Mat ldesc; std::vector<drfe_cv::Vector3d> lf;
        ls.ExtractLineSegment(image, kl, ldesc, lf);
        Planar_SLAM::PlaneDetection pd(ex.context());
        const float K[9] = {535.4f, 0, 320.1f, 0, 539.2f, 247.6f, 0, 0, 1};
        drfe_cv::Mat dimg(h, w, depth.data(), (size_t)w * 2, 2);
        if (!pd.readDepthImage(dimg, K, 1.0f / 5000.0f)) return 5;
        pd.runPlaneDetection();
        {   /* the pipelined flow through the adaptor: Submit returns at once, Collect brings the same keypoints + the glue's stereo */
            Planar_SLAM::ORBextractor ex2(1000, 1.2f, 8, 20, 7, w, h, 0, 2);
            const drfe_camera cam = {535.4f, 539.2f, 320.1f, 247.6f, 40.0f, 1.0f / 5000.0f, 0.0f, (float)w, 0.0f, (float)h};
            std::vector<drfe_cv::KeyPoint> k2; drfe_cv::Mat d2; std::vector<float> ur, z;
            ex2.Submit(1, image, reinterpret_cast<const uint16_t*>(depth.data()), (size_t)w, &cam);
            ex2.Collect(1, k2, d2, &ur, &z);
            if (k2.size() != kps.size() || std::memcmp(k2.data(), kps.data(), kps.size() * sizeof(drfe_cv::KeyPoint)) != 0 ||
                std::memcmp(d2.data, desc.data, kps.size() * 32) != 0 || ur.size() != kps.size() || z.size() != kps.size()) return 6;
        }
        {   /* one submission per tracked frame: the same image as LastFrame (slot 0) and CurrentFrame (slot 1) under the identity pose -
             * Frame::Frame's outputs and TrackWithMotionModel's SearchByProjection(Cur, Last) from one captured graph */
            Planar_SLAM::ORBextractor ex3(1000, 1.2f, 8, 20, 7, w, h, 0, 3);
            const drfe_camera cam = {535.4f, 539.2f, 320.1f, 247.6f, 40.0f, 1.0f / 5000.0f, 0.0f, (float)w, 0.0f, (float)h};
            const uint16_t* d16 = reinterpret_cast<const uint16_t*>(depth.data());
            const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            std::vector<drfe_cv::KeyPoint> k0, k1; drfe_cv::Mat d0, d1; std::vector<float> ur0, z0, ur1, z1; std::vector<int32_t> m;
            ex3.Submit(0, image, d16, (size_t)w, &cam);
            ex3.Collect(0, k0, d0, &ur0, &z0);
            ex3.SubmitTracked(1, image, d16, (size_t)w, cam, 0, I, I, I, nullptr, 0, 15.0f, false, true);
            const int nm = ex3.CollectTracked(1, k1, d1, ur1, z1, m);
            if (k1.size() != kps.size() || std::memcmp(k1.data(), kps.data(), kps.size() * sizeof(drfe_cv::KeyPoint)) != 0) return 7;
            int self = 0, valid = 0;
            for (size_t i = 0; i < m.size(); i++) { if (m[i] >= (int)k0.size()) return 8; if (m[i] >= 0) { valid++; if (m[i] == (int)i) self++; } }
            if (nm != valid || nm < 100 || self * 10 < nm * 9) return 9;          /* a keypoint under the identity pose finds itself */
        }
        FILE* f = std::fopen(argv[5], "wb");
        const int32_t hdr[4] = {(int32_t)kps.size(), (int32_t)kl.size(), pd.plane_num_, Planar_SLAM::ORBmatcher::DescriptorDistance(desc.data, desc.data + 32)};
        std::fwrite(hdr, 4, 4, f);
        std::fwrite(kps.data(), sizeof(drfe_cv::KeyPoint), kps.size(), f);
        std::fwrite(desc.data, 32, kps.size(), f);
        std::fwrite(kl.data(), sizeof(drfe_cv::KeyLine), kl.size(), f);
        std::fwrite(ldesc.data, 32, kl.size(), f);
        for (int i = 0; i < pd.plane_num_; i++) { std::fwrite(pd.extractedPlanes[i]->normal, 8, 3, f); std::fwrite(pd.extractedPlanes[i]->center, 8, 3, f); }
        std::fwrite(pd.seg_output.data, 1, (size_t)w * h, f);
        std::fwrite(ex.mvImagePyramid[1].ptr<uint8_t>(7), 1, ex.mvImagePyramid[1].cols, f);   /* one interior row of level 1 */
        std::fclose(f);
        std::printf("adaptor ok: %zu keypoints, %zu lines, %d planes\n", kps.size(), kl.size(), pd.plane_num_);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "adaptor_caller: %s\n", e.what());
        return 1;
    }
    return 0;
}
