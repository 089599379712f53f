// dump_eigen_pcl_reference.cpp - pins the oracle's restatements of Eigen 3.3.7 and PCL 1.9 to the REAL libraries.
// Build and run OUTSIDE the build container, where the reference's dependencies exist (README.md:25-31 of the reference: Eigen
// 3.3.7, PCL 1.9):
//
//   g++ -O2 -std=c++14 tools/dump_eigen_pcl_reference.cpp -o /tmp/dump_pins $(pkg-config --cflags --libs eigen3 pcl_filters-1.9 \
//       pcl_features-1.9 pcl_segmentation-1.9 pcl_sample_consensus-1.9 pcl_common-1.9)
//   /tmp/dump_pins tests/golden/eigen_pcl_pins.bin
//
// The inputs are generated here from SplitMix64 (integer arithmetic only, the same generator tests/test_eigen_pcl_pins.py runs in
// numpy), so nothing but this file and the libraries decides the outputs.  tests/test_eigen_pcl_pins.py compares the oracle with the
// file when it exists and SKIPS ("parity unpinned") when it does not - the state of this repository.
//
// Sections of the file (all little-endian; counts as int32, payloads as float64 / float32):
//   "EIG3"  n, then per matrix: 9 doubles K (row-major, symmetric), 3 eigenvalues, 9 eigenvector entries (column j = vector j)
//           from Eigen::SelfAdjointEigenSolver<Matrix3d>(K) - compute(), the iterative QL path LA::eig33sym takes
//           (reference include/peac/eig33sym.hpp:70-74)
//   "VOXG"  n points (float32 xyz), m, m centroids (float32 xyz) in pcl::VoxelGrid<PointXYZRGB> output order, leaf 0.05
//           (reference src/Frame.cc:981-986)
//   "SACP"  the cloud of VOXG's output, 4 start coefficients, valid flag, 4 refitted coefficients from the exact sequence of
//           Frame::MaxPointDistanceFromPlane (src/Frame.cc:1222-1307: SACSegmentation, SACMODEL_PLANE, SAC_RANSAC, 50 iterations,
//           probability 0.99, optimize on, threshold 0.10)
//   "NORM"  w, h, organized cloud (float32 xyz), normals (float32 xyz) from pcl::IntegralImageNormalEstimation
//           (AVERAGE_3D_GRADIENT, MaxDepthChangeFactor 0.05, NormalSmoothingSize 10), src/Frame.cc:1051-1067
#include <Eigen/Dense>
#include <pcl/features/integral_image_normal.h>
#include <pcl/filters/voxel_grid.h>
#include <pcl/point_types.h>
#include <pcl/segmentation/sac_segmentation.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static uint64_t g_state = 0;
static uint64_t splitmix64()
{
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double unit() { return (double)(splitmix64() >> 11) * (1.0 / 9007199254740992.0); }      // [0, 1): 53 bits
static void tag(FILE* f, const char* t) { std::fwrite(t, 1, 4, f); }
static void i32(FILE* f, int32_t v) { std::fwrite(&v, 4, 1, f); }

int main(int argc, char** argv)
{
    if (argc != 2) { std::fprintf(stderr, "usage: dump_pins out.bin\n"); return 2; }
    FILE* f = std::fopen(argv[1], "wb");
    if (!f) { std::perror(argv[1]); return 2; }

    // ---- Eigen: 4096 symmetric matrices, a few of them degenerate ----
    g_state = 0x0123456789ABCDEFull;
    const int NE = 4096;
    tag(f, "EIG3"); i32(f, NE);
    for (int i = 0; i < NE; i++) {
        double d[3], o[3];
        for (int k = 0; k < 3; k++) d[k] = 1.0 + (2.0 * unit() - 1.0);
        for (int k = 0; k < 3; k++) o[k] = 0.3 * (2.0 * unit() - 1.0);
        if (i % 64 == 1) { d[1] = d[0]; o[0] = 0; }                 // repeated diagonal
        if (i % 64 == 2) { o[0] = o[1] = o[2] = 0; }               // already diagonal
        if (i % 64 == 3) { o[1] = 0; }                             // m20 == 0: the tridiagonalisation's shortcut
        const double sc = (i % 7 == 0) ? 1e-6 : (i % 7 == 1) ? 1e4 : 1.0;
        Eigen::Matrix3d K;
        K << d[0] * sc, o[0] * sc, o[1] * sc, o[0] * sc, d[1] * sc, o[2] * sc, o[1] * sc, o[2] * sc, d[2] * sc;
        Eigen::SelfAdjointEigenSolver<Eigen::Matrix3d> es(K);
        double rec[21];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rec[3 * r + c] = K(r, c);
        for (int k = 0; k < 3; k++) rec[9 + k] = es.eigenvalues()(k);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rec[12 + 3 * r + c] = es.eigenvectors()(r, c);
        std::fwrite(rec, 8, 21, f);
    }

    // ---- PCL: a noisy plane patch seen from the origin, then its voxel grid, refit and the normals of an organized view ----
    g_state = 0x0FEDCBA987654321ull;
    const int NP = 20000;
    pcl::PointCloud<pcl::PointXYZRGB>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZRGB>());
    std::vector<float> xyz;
    for (int i = 0; i < NP; i++) {
        const float x = (float)(2.0 * unit() - 1.0), y = (float)(1.5 * unit() - 0.75);
        const float z = 2.0f + 0.3f * x - 0.2f * y + (float)(0.01 * (2.0 * unit() - 1.0));
        pcl::PointXYZRGB p; p.x = x; p.y = y; p.z = z; p.r = p.g = p.b = 0;
        cloud->points.push_back(p);
        xyz.push_back(x); xyz.push_back(y); xyz.push_back(z);
    }
    pcl::VoxelGrid<pcl::PointXYZRGB> voxel;
    voxel.setLeafSize(0.05f, 0.05f, 0.05f);
    pcl::PointCloud<pcl::PointXYZRGB>::Ptr coarse(new pcl::PointCloud<pcl::PointXYZRGB>());
    voxel.setInputCloud(cloud);
    voxel.filter(*coarse);
    tag(f, "VOXG"); i32(f, NP); std::fwrite(xyz.data(), 4, xyz.size(), f);
    i32(f, (int32_t)coarse->points.size());
    for (const auto& p : coarse->points) { const float v[3] = {p.x, p.y, p.z}; std::fwrite(v, 4, 3, f); }

    {
        // Frame::MaxPointDistanceFromPlane's refit (src/Frame.cc:1268-1303)
        pcl::SACSegmentation<pcl::PointXYZRGB> seg;
        pcl::ModelCoefficients::Ptr coefficients(new pcl::ModelCoefficients);
        pcl::PointIndices::Ptr inliers(new pcl::PointIndices);
        seg.setOptimizeCoefficients(true);
        seg.setModelType(pcl::SACMODEL_PLANE);
        seg.setMethodType(pcl::SAC_RANSAC);
        seg.setDistanceThreshold(0.10);
        seg.setInputCloud(coarse);
        seg.segment(*inliers, *coefficients);
        const float start[4] = {0.27f, -0.18f, -0.94f, 1.9f};
        tag(f, "SACP"); std::fwrite(start, 4, 4, f);
        i32(f, inliers->indices.empty() ? 0 : 1);
        float out[4] = {0, 0, 0, 0};
        for (size_t k = 0; k < 4 && k < coefficients->values.size(); k++) out[k] = coefficients->values[k];
        std::fwrite(out, 4, 4, f);
    }

    // organized cloud 107 x 80 (a 320 x 240 depth image subsampled by 3): two planes meeting at a crease, a hole, a far band
    g_state = 0x1122334455667788ull;
    const int W = 107, H = 80;
    pcl::PointCloud<pcl::PointXYZ>::Ptr org(new pcl::PointCloud<pcl::PointXYZ>());
    org->width = W; org->height = H; org->is_dense = false; org->points.resize((size_t)W * H);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            float z = c < 60 ? 1.5f + 0.004f * c + 0.002f * r : 1.74f + 0.02f * (c - 60);
            z += (float)(0.001 * (2.0 * unit() - 1.0));
            if (r > 30 && r < 36 && c > 20 && c < 30) z = 0.f;            // hole
            pcl::PointXYZ& p = org->points[(size_t)r * W + c];
            p.x = (c * 3 - 160.f) * z / 260.f; p.y = (r * 3 - 120.f) * z / 260.f; p.z = z;
        }
    pcl::IntegralImageNormalEstimation<pcl::PointXYZ, pcl::Normal> ne;
    ne.setNormalEstimationMethod(ne.AVERAGE_3D_GRADIENT);
    ne.setMaxDepthChangeFactor(0.05f);
    ne.setNormalSmoothingSize(10.0f);
    pcl::PointCloud<pcl::Normal>::Ptr normals(new pcl::PointCloud<pcl::Normal>());
    ne.setInputCloud(org);
    ne.compute(*normals);
    tag(f, "NORM"); i32(f, W); i32(f, H);
    for (const auto& p : org->points) { const float v[3] = {p.x, p.y, p.z}; std::fwrite(v, 4, 3, f); }
    for (const auto& n : normals->points) { const float v[3] = {n.normal_x, n.normal_y, n.normal_z}; std::fwrite(v, 4, 3, f); }
    std::fclose(f);
    std::printf("wrote %s\n", argv[1]);
    return 0;
}
