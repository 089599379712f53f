/* post_internal.h — records shared by normals_kernels.hip and planes_post.cpp (Frame::ComputePlanes post-processing) */
#ifndef DRFE_POST_INTERNAL_H
#define DRFE_POST_INTERNAL_H
#include "drfe_internal.h"

/* the 3x-subsampled organized cloud of src/Frame.cc:1027-1048: ceil(rows/3.0) x ceil(cols/3.0) */
static inline int drfe_sn_w(int w) { return (w + 2) / 3; }
static inline int drfe_sn_h(int h) { return (h + 2) / 3; }

struct SnBuffers {            /* device scratch of the surface-normal pass, [frame][...] */
    float* d_cloud;           /* W*H x 3 */
    float* d_dist;            /* W*H: depth-change seeds, then the chamfer distance */
    double* d_integ;          /* (W+1)*(H+1) x 6 */
    unsigned* d_cnt;          /* (W+1)*(H+1) x 2 */
    float* d_normals;         /* W*H x 3 */
    drfe_surface_normal* d_recs;   /* (W/2)*(H/2) */
    void* d_depth;            /* staging of the host-buffer entry point */
    size_t frames, w, h;      /* capacity */
};

hipError_t drfe_launch_surface_normals(const void* d_depth, int isU16, float factor, size_t frameStride, size_t rowStride, int w,
                                       int h, const float K4[4], float maxDist, int nframes, const SnBuffers& b, hipStream_t s);
void drfe_post_free(drfe_ctx* c);
#include <string>
/* pcl::VoxelGrid for njobs point clouds at once (voxel_kernels.hip): job j = points [jobs[j].x, jobs[j].x + jobs[j].y) of d_pts
 * (xyz packed); d_list: njobs + 2 ints of scratch (the job order); scratch arrays span all points; centroids of job j to d_out at the job's offset, their number to d_counts[j]
 * (-1: grid overflows int32, PCL keeps the input cloud; -2: the sort needs the heap-sort branch: run the job on the host) */
hipError_t drfe_launch_voxel_grid(const float* d_pts, const int2* d_jobs, int njobs, int* d_list, unsigned long long* d_recs, unsigned long long* d_tmp,
                                  uint32_t* d_posL, uint32_t* d_posR, float* d_out, int* d_counts, float leafSize, hipStream_t s);
/* Gates + Frame::MaxPointDistanceFromPlane (RANSAC + least-squares refit) of njobs planes on the centroids k_voxel_grid left
 * (refit_kernels.hip): job j = plane j % planeCap of frame j / planeCap of the extractor's frame table; d_post[j] / d_status[j]
 * (0 final, 1 not certified: refit on the host, 2 the voxel grid came back: grid + refit on the host, -1 no such plane).
 * d_mtState: std::mt19937(12345)'s 624 state words after seeding; logP = log(1 - 0.99) by the host's libm. */
struct AhcDevFrame;
hipError_t drfe_launch_plane_refit(const AhcDevFrame* d_frames, const int2* d_jobs, const int* d_vcounts, const float* d_vout, const uint32_t* d_mtState,
                                   int njobs, int planeCap, float maxPointDist, double distThreshold, double logP, drfe_plane_post* d_post, int* d_status,
                                   hipStream_t s);
/* a lane's device voxel grid (planes_post.cpp): buffers + stream; NULL = the host voxel grid */
struct VoxelDevice;
VoxelDevice* drfe_voxel_device_create(int device, std::string* err);
void drfe_voxel_device_free(VoxelDevice* v);
int drfe_ahc_post_core(std::string* err, const uint16_t* depth, int w, int h, size_t stride, const float* K4, float depth_factor,
                       const drfe_plane* planes, int n_planes, const int32_t* member_offsets, const int32_t* member_idx,
                       float max_point_dist, double dist_threshold, drfe_plane_post* post, float* voxel_xyz, int32_t* voxel_offsets,
                       int cap_voxels, int* n_accepted, int* plane_num, VoxelDevice* vox = nullptr);
int drfe_ahc_post_from_coarse(std::string* err, const drfe_plane* planes, int n_planes, const float* const* coarse_xyz, const int* coarse_n,
                              float max_point_dist, double dist_threshold, drfe_plane_post* post, float* voxel_xyz, int32_t* voxel_offsets,
                              int cap_voxels, int* n_accepted, int* plane_num);
#endif
