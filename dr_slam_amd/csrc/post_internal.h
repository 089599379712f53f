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
int drfe_ahc_post_core(std::string* err, const uint16_t* depth, int w, int h, size_t stride, const float* K4, float depth_factor,
                       const drfe_plane* planes, int n_planes, const int32_t* member_offsets, const int32_t* member_idx,
                       float max_point_dist, double dist_threshold, drfe_plane_post* post, float* voxel_xyz, int32_t* voxel_offsets,
                       int cap_voxels, int* n_accepted, int* plane_num);
#endif
