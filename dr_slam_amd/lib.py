"""ctypes binding of libdrfe.so (include/drfe.h) — the only way Python reaches the HIP kernels.

There is NO CPU fallback: if the shared library is missing or no HIP device is present the import /
context creation raises.  PyTorch is used by callers only for device buffers and streams; nothing
here depends on it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRFE_LIB") or os.path.join(_HERE, "csrc", "libdrfe.so")   # DRFE_LIB: variant builds (experiments)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
MAPPOINT_DTYPE = np.dtype([("valid", "u1"), ("obs_positive", "u1"), ("pad", "u1", (2,)), ("world", "<f4", (3,)),
                           ("desc", "u1", (32,))])
TRACKED_DTYPE = np.dtype([("track_in_view", "u1"), ("bad", "u1"), ("obs_positive", "u1"), ("pad", "u1"),
                          ("level", "<i4"), ("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"),
                          ("view_cos", "<f4"), ("desc", "u1", (32,))])

STAGES = ("pyramid", "fast", "quadtree", "blur", "desc", "glue", "match", "fast_b")

# every symbol include/drfe.h declares (tests check the library exports all of them)
SYMBOLS = (
    "drfe_create", "drfe_destroy", "drfe_last_error", "drfe_version", "drfe_orb_scale_tables",
    "drfe_orb_max_keypoints", "drfe_orb_extract", "drfe_orb_extract_batch", "drfe_orb_download", "drfe_orb_counts",
    "drfe_orb_pyramid_level", "drfe_orb_blurred_level", "drfe_orb_candidates", "drfe_frame_stereo_grid_batch",
    "drfe_fuse_search", "drfe_fuse_search_sim3", "drfe_search_by_sim3", "drfe_search_by_projection_kf", "drfe_search_by_projection_reloc", "drfe_lsd_fuse_search", "drfe_frame_is_in_frustum", "drfe_frame_is_in_frustum_lines", "drfe_frame_set_distortion", "drfe_frame_image_bounds", "drfe_frame_download_keys_un",
    "drfe_frame_download_stereo", "drfe_frame_download_grid", "drfe_match_consecutive_batch", "drfe_match_download",
    "drfe_search_by_projection_last", "drfe_search_by_projection_map", "drfe_match_bf_knn", "drfe_profile_enable",
    "drfe_profile_stage_ms", "drfe_stream_sync", "drfe_planes_ahc", "drfe_planes_ahc_batch", "drfe_planes_ahc_blocks",
    "drfe_match_orb_points", "drfe_planes_cape", "drfe_voc_upload", "drfe_bow_transform_batch", "drfe_bow_download",
    "drfe_search_by_bow", "drfe_search_by_bow_kf", "drfe_search_for_triangulation", "drfe_lsd_extract", "drfe_lsd_extract_batch", "drfe_lsd_stages", "drfe_lines_is_good", "drfe_lsd_search_by_descriptor", "drfe_lsd_search_for_triangulation", "drfe_lsd_search_by_projection_last",
    "drfe_lsd_search_by_projection_map", "drfe_plane_voxel_grid", "drfe_plane_refit", "drfe_planes_ahc_postprocess",
    "drfe_planes_cape_postprocess", "drfe_surface_normals", "drfe_surface_normals_batch", "drfe_surface_normals_download", "drfe_batch_download_async", "drfe_orb_fast_partition", "drfe_lsd_segments_host", "drfe_orb_keypoint_pixels_async", "drfe_gather_keypoint_depth",
    "drfe_frame_stereo_grid_batch_kpdepth", "drfe_planes_ahc_post_batch", "drfe_planes_ahc_from_blocks", "drfe_debug_ahc_trials", "drfe_debug_order_sort", "drfe_search_for_initialization", "drfe_lsd_fuse_search_sim3", "drfe_lsd_search_by_projection_kf",
    "drfe_lsd_search_by_sim3", "drfe_frame_submit", "drfe_frame_collect", "drfe_pipeline_create", "drfe_pipeline_destroy",
    "drfe_pipeline_depth", "drfe_pipeline_context", "drfe_pipeline_last_error", "drfe_pipeline_submit", "drfe_pipeline_sync",
    "drfe_lsd_configure", "drfe_lsd_configure_rect", "drfe_shard_unique_id", "drfe_shard_create", "drfe_shard_destroy", "drfe_shard_broadcast", "drfe_shard_reduce_report", "drfe_shard_sequences_of_rank", "drfe_shard_last_error", "drfe_planes_configure_cape", "drfe_planes_cape_stats", "drfe_lsd_configure_nfa", "drfe_lsd_stats", "drfe_lsd_segments_host_mode", "drfe_debug_cr_sincos", "drfe_debug_device_order_sort", "drfe_debug_device_order_sort_depth", "drfe_batch_status_async", "drfe_batch_check", "drfe_frame_submit_tracked", "drfe_frame_collect_tracked", "drfe_planes_cape_batch", "drfe_planes_configure", "drfe_planes_configure_extractor", "drfe_planes_ahc_stats", "drfe_planes_configure_refit", "drfe_planes_refit_stats", "drfe_frame_load", "drfe_bow_transform_slot", "drfe_long_kernel_clock", "drfe_long_kernel_ms",
)

FRUSTUM_POINT_DTYPE = np.dtype([("world", "<f4", (3,)), ("normal", "<f4", (3,)), ("min_distance", "<f4"),
                                ("max_distance", "<f4")])                            # drfe_frustum_point, 32 B
FRUSTUM_LINE_DTYPE = np.dtype([("world", "<f8", (6,)), ("normal", "<f8", (3,)), ("min_distance", "<f4"),
                               ("max_distance", "<f4")])                             # drfe_frustum_line, 80 B
MAPLINE_DTYPE = np.dtype([("valid", "<i4"), ("octave", "<i4"), ("obs_positive", "<i4"), ("pad", "<i4"),
                          ("world", "<f8", (6,)), ("desc", "u1", (32,))])            # drfe_map_line, 96 B
TRACKED_LINE_DTYPE = np.dtype([("in_view", "<i4"), ("level", "<i4"), ("obs_positive", "<i4"), ("x1", "<f4"),
                               ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("view_cos", "<f4"),
                               ("desc", "u1", (32,))])                                # drfe_tracked_line, 64 B
KEYLINE_DTYPE = np.dtype([("angle", "<f4"), ("class_id", "<i4"), ("octave", "<i4"), ("pt_x", "<f4"), ("pt_y", "<f4"),
                          ("response", "<f4"), ("size", "<f4"), ("start_point_x", "<f4"), ("start_point_y", "<f4"),
                          ("end_point_x", "<f4"), ("end_point_y", "<f4"), ("s_point_in_octave_x", "<f4"),
                          ("s_point_in_octave_y", "<f4"), ("e_point_in_octave_x", "<f4"), ("e_point_in_octave_y", "<f4"),
                          ("line_length", "<f4"), ("num_of_pixels", "<i4")])

CAPE_PLANE_DTYPE = np.dtype([("normal", "<f8", (3,)), ("mean", "<f8", (3,)), ("d", "<f8"), ("mse", "<f4"),
                             ("score", "<f4"), ("n_points", "<i4"), ("pad", "<i4")])

PLANE_POST_DTYPE = np.dtype([("coef", "<f4", (4,)), ("accepted", "<i4"), ("n_voxels", "<i4")])      # drfe_plane_post, 24 B
SURFACE_NORMAL_DTYPE = np.dtype([("normal", "<f4", (3,)), ("camera_position", "<f4", (3,)), ("frame_x", "<i4"),
                                 ("frame_y", "<i4")])                                               # drfe_surface_normal, 32 B

PLANE_DTYPE = np.dtype([("normal", "<f8", (3,)), ("center", "<f8", (3,)), ("mse", "<f8"), ("curvature", "<f8"),
                        ("n_points", "<i4"), ("rid", "<i4")])


class DrfeError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32),
                ("max_batch", C.c_int32), ("nfeatures", C.c_int32), ("scale_factor", C.c_float),
                ("nlevels", C.c_int32), ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
                ("depth_factor", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float)]


_lib = None


def load() -> C.CDLL:
    """Load libdrfe.so; raises if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DrfeError(f"{LIB_PATH} not found: build it with `make -C dr_slam_amd/csrc` "
                        "(there is no CPU fallback for the feature path)")
    # One HIP runtime per process: PyTorch ships its own libamdhip64; if libdrfe pulled in the system
    # copy first, a later `import torch` would bring a second runtime that cannot see the device.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.drfe_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.drfe_destroy.argtypes = [vp]
    L.drfe_destroy.restype = None
    L.drfe_last_error.argtypes = [vp]
    L.drfe_last_error.restype = C.c_char_p
    L.drfe_version.restype = C.c_char_p
    L.drfe_orb_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.drfe_orb_max_keypoints.argtypes = [vp]
    L.drfe_orb_extract.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, C.POINTER(i32)]
    L.drfe_orb_extract_batch.argtypes = [vp, vp, sz, sz, i32, i32, i32, vp]
    L.drfe_pipeline_create.argtypes = [C.POINTER(Config), i32, C.POINTER(vp)]
    L.drfe_pipeline_destroy.argtypes = [vp]
    L.drfe_pipeline_destroy.restype = None
    L.drfe_pipeline_depth.argtypes = [vp]
    L.drfe_pipeline_context.argtypes = [vp, i32]
    L.drfe_pipeline_context.restype = vp
    L.drfe_pipeline_last_error.argtypes = [vp]
    L.drfe_pipeline_last_error.restype = C.c_char_p
    L.drfe_pipeline_submit.argtypes = [vp, vp, vp, sz, sz, i32, i32, vp, vp, C.POINTER(Camera), C.c_float, i32, i32, i32]
    L.drfe_pipeline_sync.argtypes = [vp, i32]
    L.drfe_frame_submit.argtypes = [vp, i32, vp, i32, i32, sz, vp, sz, C.POINTER(Camera)]
    L.drfe_frame_collect.argtypes = [vp, i32, vp, vp, vp, vp, i32, C.POINTER(i32)]
    L.drfe_frame_submit_tracked.argtypes = [vp, i32, vp, i32, i32, sz, vp, sz, C.POINTER(Camera), i32, vp, vp, vp, vp, i32, f32, i32, i32]
    L.drfe_frame_collect_tracked.argtypes = [vp, i32, vp, vp, vp, vp, i32, C.POINTER(i32), vp, C.POINTER(i32)]
    L.drfe_orb_download.argtypes = [vp, i32, vp, vp, i32, C.POINTER(i32)]
    L.drfe_orb_counts.argtypes = [vp, i32, vp]
    L.drfe_orb_fast_partition.argtypes = [vp, i32, i32, vp, vp]
    L.drfe_orb_keypoint_pixels_async.argtypes = [vp, i32, vp, vp, vp]
    L.drfe_gather_keypoint_depth.argtypes = [vp, sz, sz, i32, vp, vp, i32, vp, i32]
    L.drfe_frame_stereo_grid_batch_kpdepth.argtypes = [vp, vp, i32, C.POINTER(Camera), i32, vp]
    L.drfe_batch_download_async.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.drfe_orb_pyramid_level.argtypes = [vp, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
    L.drfe_orb_blurred_level.argtypes = [vp, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
    L.drfe_orb_candidates.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.drfe_frame_stereo_grid_batch.argtypes = [vp, vp, sz, sz, C.POINTER(Camera), i32, vp]
    L.drfe_frame_download_stereo.argtypes = [vp, i32, vp, vp, i32]
    L.drfe_frame_download_grid.argtypes = [vp, i32, vp, vp, i32]
    L.drfe_match_consecutive_batch.argtypes = [vp, vp, vp, C.POINTER(Camera), f32, i32, i32, i32, vp]
    L.drfe_match_download.argtypes = [vp, i32, vp, i32, C.POINTER(i32)]
    L.drfe_search_by_projection_last.argtypes = [vp, i32, i32, vp, vp, C.POINTER(Camera), vp, i32, f32, i32, i32,
                                                 vp, vp, i32, C.POINTER(i32)]
    L.drfe_search_by_projection_map.argtypes = [vp, i32, vp, i32, f32, f32, vp, vp, i32, C.POINTER(i32)]
    L.drfe_match_bf_knn.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp]
    L.drfe_match_orb_points.argtypes = [vp, i32, i32, vp, vp, i32, vp, i32, C.POINTER(i32)]
    L.drfe_search_by_bow_kf.argtypes = [vp, i32, i32, vp, i32, vp, i32, C.c_float, i32, vp, C.POINTER(i32)]
    L.drfe_search_for_triangulation.argtypes = [vp, i32, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, vp, C.POINTER(i32)]
    L.drfe_planes_ahc_batch.argtypes = [vp, vp, C.c_size_t, i32, i32, C.c_size_t, i32, vp, C.c_float, vp, i32, vp, vp, vp, vp, i32]
    L.drfe_lsd_extract_batch.argtypes = [vp, vp, C.c_size_t, i32, i32, C.c_size_t, i32, i32, vp, vp, vp, i32, vp, vp, i32]
    L.drfe_lines_is_good.argtypes = [vp, i32, vp, i32, i32, C.c_size_t, vp, i32, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_uint32, vp, vp, vp, C.POINTER(i32)]
    L.drfe_lsd_fuse_search.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, C.c_float, vp, vp]
    L.drfe_lsd_fuse_search_sim3.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, C.c_float, vp, vp]
    L.drfe_lsd_search_by_projection_kf.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, C.POINTER(i32)]
    L.drfe_lsd_search_by_sim3.argtypes = [vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, C.c_float,
                                          vp, C.POINTER(i32)]
    L.drfe_search_by_projection_reloc.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, C.c_float, i32, i32, vp, C.POINTER(i32)]
    L.drfe_search_for_initialization.argtypes = [vp, i32, i32, vp, i32, i32, C.c_float, i32, vp, C.POINTER(i32)]
    L.drfe_search_by_projection_kf.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, i32, C.c_float, vp, C.POINTER(i32)]
    L.drfe_search_by_sim3.argtypes = [vp, i32, i32, vp, vp, C.c_float, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, C.c_float, vp,
                                      C.POINTER(i32)]
    L.drfe_fuse_search_sim3.argtypes = [vp, i32, vp, vp, vp, vp, i32, C.c_float, vp, vp]
    L.drfe_fuse_search.argtypes = [vp, i32, vp, vp, vp, vp, i32, C.c_float, vp, vp]
    L.drfe_frame_is_in_frustum.argtypes = [vp, vp, vp, vp, i32, C.c_float, vp]
    L.drfe_frame_is_in_frustum_lines.argtypes = [vp, vp, vp, vp, i32, C.c_float, vp]
    L.drfe_frame_set_distortion.argtypes = [vp, vp, vp, i32]
    L.drfe_frame_image_bounds.argtypes = [vp, vp, i32, i32, i32, vp]
    L.drfe_frame_download_keys_un.argtypes = [vp, i32, vp, i32]
    L.drfe_lsd_search_for_triangulation.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, C.POINTER(i32)]
    L.drfe_lsd_search_by_descriptor.argtypes = [vp, vp, i32, vp, i32, vp, i32, vp, C.POINTER(i32)]
    L.drfe_lsd_search_by_projection_last.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, i32, C.c_float, i32, C.c_float, vp, vp,
                                                     C.POINTER(i32)]
    L.drfe_lsd_search_by_projection_map.argtypes = [vp, vp, i32, vp, vp, i32, C.c_float, C.c_float, vp, vp, C.POINTER(i32)]
    L.drfe_lsd_extract.argtypes = [vp, vp, i32, i32, sz, i32, vp, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.drfe_lsd_stages.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.drfe_voc_upload.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.drfe_bow_transform_batch.argtypes = [vp, i32, i32, vp]
    L.drfe_bow_download.argtypes = [vp, i32, vp, vp, vp, i32]
    L.drfe_search_by_bow.argtypes = [vp, i32, i32, vp, i32, f32, i32, vp, i32, C.POINTER(i32)]
    L.drfe_planes_ahc.argtypes = [vp, vp, i32, i32, sz, vp, f32, vp, i32, C.POINTER(i32), vp, vp, vp]
    L.drfe_planes_ahc_blocks.argtypes = [vp, vp, i32, i32, sz, vp, f32, vp, vp, i32]
    L.drfe_planes_cape.argtypes = [vp, vp, i32, i32, sz, vp, i32, f32, f32, vp, i32, C.POINTER(i32), vp, vp, vp, vp]
    f64 = C.c_double
    L.drfe_plane_voxel_grid.argtypes = [vp, i32, f32, vp, i32, C.POINTER(i32)]
    L.drfe_plane_refit.argtypes = [vp, vp, i32, f64, C.POINTER(i32)]
    L.drfe_planes_ahc_postprocess.argtypes = [vp, vp, i32, i32, sz, vp, f32, vp, i32, vp, vp, f32, f64, vp, vp, vp, i32,
                                              C.POINTER(i32), C.POINTER(i32)]
    L.drfe_planes_ahc_post_batch.argtypes = [vp, vp, sz, i32, i32, sz, i32, vp, f32, f32, f64, vp, i32, vp, vp, vp, vp, vp, i32]
    L.drfe_debug_ahc_trials.argtypes = [vp, vp, i32, i32, vp]
    L.drfe_debug_order_sort.argtypes = [vp, sz, i32, i32, i32, C.c_uint32]
    L.drfe_planes_ahc_from_blocks.argtypes = [vp, vp, vp, i32, i32, sz, vp, f32, vp, i32, C.POINTER(i32), vp, vp, vp]
    L.drfe_planes_cape_postprocess.argtypes = [vp, vp, i32, i32, sz, vp, vp, vp, i32, f32, f64, vp, vp, vp, i32,
                                               C.POINTER(i32), C.POINTER(i32)]
    L.drfe_surface_normals.argtypes = [vp, vp, i32, i32, sz, vp, f32, vp, i32, C.POINTER(i32), vp, vp, vp]
    L.drfe_surface_normals_batch.argtypes = [vp, vp, sz, sz, i32, i32, vp, f32, f32, i32, vp]
    L.drfe_surface_normals_download.argtypes = [vp, i32, vp, i32, C.POINTER(i32)]
    L.drfe_lsd_segments_host.argtypes = [vp, vp, vp, i32, i32, f64, vp, i32, C.POINTER(i32)]
    L.drfe_lsd_configure.argtypes = [vp, i32]
    L.drfe_lsd_configure_rect.argtypes = [vp, i32]
    L.drfe_shard_unique_id.argtypes = [vp]
    L.drfe_shard_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.drfe_shard_destroy.argtypes = [vp]
    L.drfe_shard_destroy.restype = None
    L.drfe_shard_broadcast.argtypes = [vp, vp, sz, i32]
    L.drfe_shard_reduce_report.argtypes = [vp, vp, i32, vp, i32]
    L.drfe_shard_sequences_of_rank.argtypes = [i32, i32, i32, vp, i32]
    L.drfe_shard_last_error.argtypes = [vp]
    L.drfe_shard_last_error.restype = C.c_char_p
    L.drfe_planes_configure_cape.argtypes = [vp, i32]
    L.drfe_planes_cape_stats.argtypes = [vp, vp]
    L.drfe_planes_ahc_stats.argtypes = [vp, vp]
    L.drfe_planes_configure_refit.argtypes = [vp, i32]
    L.drfe_planes_refit_stats.argtypes = [vp, vp]
    L.drfe_frame_load.argtypes = [vp, i32, vp, vp, vp, vp, vp, i32, vp]
    L.drfe_bow_transform_slot.argtypes = [vp, i32, i32, vp]
    L.drfe_lsd_configure_nfa.argtypes = [vp, i32]
    L.drfe_lsd_stats.argtypes = [vp, vp]
    L.drfe_long_kernel_clock.argtypes = [vp, i32]
    L.drfe_long_kernel_ms.argtypes = [vp, vp]
    L.drfe_lsd_segments_host_mode.argtypes = [vp, vp, vp, i32, i32, f64, i32, vp, i32, C.POINTER(i32)]
    L.drfe_planes_configure.argtypes = [vp, i32]
    L.drfe_planes_configure_extractor.argtypes = [vp, i32]
    L.drfe_planes_cape_batch.argtypes = [vp, vp, sz, i32, i32, sz, i32, vp, i32, f32, f32, vp, i32, vp, vp, i32]
    L.drfe_batch_status_async.argtypes = [vp, vp, vp]
    L.drfe_batch_check.argtypes = [vp]
    L.drfe_debug_cr_sincos.argtypes = [vp, i32, vp, vp, vp]
    L.drfe_debug_device_order_sort.argtypes = [vp, vp, sz, C.POINTER(i32)]
    L.drfe_debug_device_order_sort_depth.argtypes = [vp, vp, sz, i32, C.POINTER(i32)]
    L.drfe_profile_enable.argtypes = [vp, i32]
    L.drfe_profile_stage_ms.argtypes = [vp, vp]
    L.drfe_stream_sync.argtypes = [vp]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def lsd_segments_host(modgrad, angles, cs, max_grad, rect_mode=0):
    """The sequential half of LSD on given level-line fields (host code, no device): [n, 4] float32 segments.
    rect_mode: rect_nfa's reading (0 literal OpenCV 3.4, 1 real-valued)."""
    L = load()
    m = np.ascontiguousarray(modgrad, np.float64)
    a = np.ascontiguousarray(angles, np.float64)
    c = np.ascontiguousarray(cs, np.float32)
    H, W = m.shape
    out = np.zeros((20000, 4), np.float32)
    n = C.c_int()
    rc = L.drfe_lsd_segments_host_mode(_p(m), _p(a), _p(c), W, H, float(max_grad), int(rect_mode), _p(out), len(out), C.byref(n))
    if rc != 0:
        raise DrfeError(f"drfe_lsd_segments_host failed ({rc})")
    return out[:n.value].copy()


def planes_ahc_from_blocks(blocks17, valid, N, depth16, K4, depth_factor, cap=64):
    """The host half of the AHC extractor on given block fits (no device) -> dict(planes, seg, members)."""
    L = load()
    d = np.ascontiguousarray(depth16, np.uint16)
    h, w = d.shape
    b = np.ascontiguousarray(blocks17, np.float64)
    vn = np.ascontiguousarray(np.stack([valid, N], 1), np.int32)
    planes = np.zeros(cap, PLANE_DTYPE)
    n = C.c_int()
    seg = np.zeros((h, w), np.uint8)
    off = np.zeros(cap + 1, np.int32)
    idx = np.zeros(h * w, np.int32)
    rc = L.drfe_planes_ahc_from_blocks(_p(b), _p(vn), _p(d), w, h, w, _p(np.ascontiguousarray(K4, np.float32)), np.float32(depth_factor),
                                       _p(planes), cap, C.byref(n), _p(seg), _p(off), _p(idx))
    if rc != 0:
        raise DrfeError(f"drfe_planes_ahc_from_blocks failed ({rc})")
    return dict(planes=planes[:n.value].copy(), seg=seg, members=[idx[off[i]:off[i + 1]].copy() for i in range(n.value)])


def plane_voxel_grid(xyz, leaf=0.05):
    """pcl::VoxelGrid(leaf) of an [n, 3] float32 point list (inputCloud order) -> [m, 3] centroids. Host code."""
    L = load()
    p = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.zeros_like(p)
    n = C.c_int()
    rc = L.drfe_plane_voxel_grid(_p(p), len(p), np.float32(leaf), _p(out), len(p), C.byref(n))
    if rc != 0:
        raise DrfeError(f"drfe_plane_voxel_grid failed ({rc})")
    return out[:n.value].copy()


def plane_refit(coef4, xyz, dist_threshold):
    """Frame::MaxPointDistanceFromPlane(coef, cloud) -> (valid, coef after the refit). Host code."""
    L = load()
    c = np.ascontiguousarray(coef4, np.float32).copy()
    p = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    v = C.c_int()
    rc = L.drfe_plane_refit(_p(c), _p(p), len(p), float(dist_threshold), C.byref(v))
    if rc != 0:
        raise DrfeError(f"drfe_plane_refit failed ({rc})")
    return bool(v.value), c


def make_camera(fx, fy, cx, cy, bf, depth_map_factor, width, height) -> Camera:
    """Frame constants for the no-distortion path (bounds = image, reference src/Frame.cc:884-889).
    depth_map_factor is the yaml DepthMapFactor; the reference inverts it (src/Tracking.cc:144-148)."""
    inv = np.float32(1.0) if abs(depth_map_factor) < 1e-5 else np.float32(1.0) / np.float32(depth_map_factor)
    return Camera(fx, fy, cx, cy, bf, float(inv), 0.0, float(width), 0.0, float(height))


def lines_is_good(lines, depth_f32, K9, cx, cy, invfx, invfy, k_as_f64=False, seed=1):
    """Frame::isLineGood (host entry, no context / GPU needed): returns (mvDepthLine, mvLines3D[n,6], inliers, n_good).
    k_as_f64=False is the reference as shipped (CV_32F mK read as double -> nothing is accepted)."""
    L = load()
    kl = np.ascontiguousarray(lines, KEYLINE_DTYPE)
    d = np.ascontiguousarray(depth_f32, np.float32)
    K = np.ascontiguousarray(K9, np.float32).reshape(9)
    n = len(kl)
    dl = np.zeros(max(n, 1), np.float32)
    l3 = np.zeros((max(n, 1), 6), np.float64)
    ni = np.zeros(max(n, 1), np.int32)
    good = C.c_int()
    rc = L.drfe_lines_is_good(_p(kl), n, _p(d), d.shape[1], d.shape[0], d.shape[1], _p(K), int(bool(k_as_f64)),
                              C.c_float(cx), C.c_float(cy), C.c_float(invfx), C.c_float(invfy), int(seed), _p(dl), _p(l3),
                              _p(ni), C.byref(good))
    if rc != 0:
        raise RuntimeError(f"drfe_lines_is_good failed ({rc})")
    return dl[:n], l3[:n], ni[:n], good.value


class Shard:
    """drfe_shard_*: the native (RCCL) side of the batched-sequence mode - what a C++ host calls where bench.py uses
    torch.distributed: unique id on rank 0, communicator per rank, broadcast of host buffers (the vocabulary), end-of-run
    MAX / SUM reduction."""

    @staticmethod
    def unique_id():
        L = load()
        ident = np.zeros(128, np.uint8)
        rc = L.drfe_shard_unique_id(_p(ident))
        if rc != 0:
            raise DrfeError(f"drfe_shard_unique_id failed ({rc}): {L.drfe_shard_last_error(None).decode()}")
        return ident

    @staticmethod
    def sequences_of_rank(n_sequences, nranks, rank):
        L = load()
        out = np.zeros(max(1, n_sequences), np.int32)
        n = L.drfe_shard_sequences_of_rank(n_sequences, nranks, rank, _p(out), len(out))
        if n < 0:
            raise DrfeError(f"drfe_shard_sequences_of_rank failed ({n})")
        return out[:n].copy()

    def __init__(self, ident, nranks, rank, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.drfe_shard_create(_p(np.ascontiguousarray(ident, np.uint8)), nranks, rank, device, C.byref(h))
        if rc != 0:
            raise DrfeError(f"drfe_shard_create failed ({rc}): {self.L.drfe_shard_last_error(None).decode()}")
        self.h = h

    def _chk(self, rc, what):
        if rc != 0:
            raise DrfeError(f"{what} failed ({rc}): {self.L.drfe_shard_last_error(self.h).decode()}")

    def broadcast(self, arr: np.ndarray, root=0):
        """in place: a C-contiguous host array from rank `root` to every rank"""
        assert arr.flags.c_contiguous
        self._chk(self.L.drfe_shard_broadcast(self.h, _p(arr), arr.nbytes, root), "drfe_shard_broadcast")
        return arr

    def reduce_report(self, maxima, sums):
        m = np.ascontiguousarray(maxima, np.float64).copy()
        s = np.ascontiguousarray(sums, np.int64).copy()
        self._chk(self.L.drfe_shard_reduce_report(self.h, _p(m), len(m), _p(s), len(s)), "drfe_shard_reduce_report")
        return m, s

    def close(self):
        if getattr(self, "h", None):
            self.L.drfe_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """drfe_pipeline: `depth` contexts used round robin, each on its own stream (batches in flight)."""

    def __init__(self, depth, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7, max_width=640, max_height=480,
                 max_batch=1, device=0):
        self.L = load()
        self.cfg = Config(device, max_width, max_height, max_batch, nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast)
        h = C.c_void_p()
        rc = self.L.drfe_pipeline_create(C.byref(self.cfg), depth, C.byref(h))
        if rc != 0:
            raise DrfeError(f"drfe_pipeline_create failed ({rc}): {self.L.drfe_last_error(None).decode()}")
        self.h = h
        self.depth = self.L.drfe_pipeline_depth(self.h)
        self.contexts = [Context.view(self.L.drfe_pipeline_context(self.h, k), nlevels, max_batch) for k in range(self.depth)]

    def submit(self, d_gray: int, d_depth: int, frame_stride: int, row_stride: int, w: int, h: int, Tcw, Twc, cam, th=15.0, mono=False,
               check_ori=True, nframes=1) -> int:
        """-> index of the context that holds this batch's results"""
        t1 = np.ascontiguousarray(Tcw, np.float32) if Tcw is not None else None
        t2 = np.ascontiguousarray(Twc, np.float32) if Twc is not None else None
        k = self.L.drfe_pipeline_submit(self.h, C.c_void_p(d_gray), C.c_void_p(d_depth) if d_depth else None, frame_stride, row_stride, w, h,
                                        _p(t1) if t1 is not None else None, _p(t2) if t2 is not None else None,
                                        C.byref(cam) if cam is not None else None, th, int(mono), int(check_ori), nframes)
        if k < 0:
            raise DrfeError(f"drfe_pipeline_submit failed ({k}): {self.L.drfe_pipeline_last_error(self.h).decode()}")
        return k

    def sync(self, k=-1):
        rc = self.L.drfe_pipeline_sync(self.h, k)
        if rc != 0:
            raise DrfeError(f"drfe_pipeline_sync failed ({rc}): {self.L.drfe_pipeline_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            for c in self.contexts:
                c.close()
            self.L.drfe_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Owns one drfe_ctx (one HIP device, one batch arena)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7,
                 max_width=640, max_height=480, max_batch=1, device=0):
        self.L = load()
        self.cfg = Config(device, max_width, max_height, max_batch, nfeatures, scale_factor, nlevels, ini_th_fast,
                          min_th_fast)
        h = C.c_void_p()
        rc = self.L.drfe_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise DrfeError(f"drfe_create failed ({rc}): {self.L.drfe_last_error(None).decode()}")
        self.h = h
        self.nlevels = nlevels
        self.max_kp = self.L.drfe_orb_max_keypoints(self.h)
        self.max_batch = max_batch

    @classmethod
    def view(cls, handle, nlevels, max_batch):
        """A Context over a drfe_ctx somebody else owns (a pipeline's): same methods, close() does not destroy it."""
        self = cls.__new__(cls)
        self.L = load()
        self.h = C.c_void_p(handle)
        self.owned = False
        self.nlevels = nlevels
        self.max_kp = self.L.drfe_orb_max_keypoints(self.h)
        self.max_batch = max_batch
        return self

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "owned", True):
                self.L.drfe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise DrfeError(f"{what} failed ({rc}): {self.L.drfe_last_error(self.h).decode()}")

    # --- ORB ---------------------------------------------------------------------------------------
    def scale_tables(self):
        t = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        self._chk(self.L.drfe_orb_scale_tables(self.h, *[_p(a) for a in t]), "drfe_orb_scale_tables")
        return t

    def orb_extract(self, gray: np.ndarray):
        """Single host frame -> (keypoints structured array, descriptors [N,32] uint8)."""
        if gray is None or gray.size == 0:
            n = C.c_int(0)
            self._chk(self.L.drfe_orb_extract(self.h, None, 0, 0, 0, None, None, 0, C.byref(n)), "drfe_orb_extract")
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert gray.dtype == np.uint8 and gray.ndim == 2 and gray.strides[1] == 1
        kps = np.zeros(self.max_kp, KP_DTYPE)
        desc = np.zeros((self.max_kp, 32), np.uint8)
        n = C.c_int(0)
        self._chk(self.L.drfe_orb_extract(self.h, _p(gray), gray.shape[1], gray.shape[0], gray.strides[0], _p(kps),
                                          _p(desc), self.max_kp, C.byref(n)), "drfe_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def frame_submit(self, slot: int, gray: np.ndarray, depth16: np.ndarray = None, cam: Camera = None):
        """Per-frame pipelined flow: enqueue ORB (+ the Frame glue when a raw depth image is given) of one host frame into
        `slot` and return without waiting."""
        assert gray.dtype == np.uint8 and gray.ndim == 2 and gray.strides[1] == 1
        dp, ds = None, 0
        if depth16 is not None:
            assert depth16.dtype == np.uint16 and depth16.shape == gray.shape and depth16.strides[1] == 2
            dp, ds = _p(depth16), depth16.strides[0] // 2
        self._chk(self.L.drfe_frame_submit(self.h, slot, _p(gray), gray.shape[1], gray.shape[0], gray.strides[0], dp,
                                           C.c_size_t(ds), C.byref(cam) if cam is not None else None), "drfe_frame_submit")

    def frame_collect(self, slot: int, stereo=False):
        """-> (mvKeys, mDescriptors[, mvuRight, mvDepth]) of the slot's outstanding submission."""
        kps = np.zeros(self.max_kp, KP_DTYPE)
        desc = np.zeros((self.max_kp, 32), np.uint8)
        ur = np.zeros(self.max_kp, np.float32) if stereo else None
        z = np.zeros(self.max_kp, np.float32) if stereo else None
        n = C.c_int(0)
        self._chk(self.L.drfe_frame_collect(self.h, slot, _p(kps), _p(desc), _p(ur) if stereo else None,
                                            _p(z) if stereo else None, self.max_kp, C.byref(n)), "drfe_frame_collect")
        out = (kps[:n.value].copy(), desc[:n.value].copy())
        return out + (ur[:n.value].copy(), z[:n.value].copy()) if stereo else out

    def frame_submit_tracked(self, slot: int, gray: np.ndarray, depth16: np.ndarray, cam: Camera, last_slot: int, Tcw_cur, Tcw_last,
                             Twc_last=None, last_mp=None, th=15.0, mono=False, check_ori=True):
        """One submission per tracked frame: frame_submit + SearchByProjection(this frame, the frame in last_slot) in the same
        captured graph.  last_mp (MAPPOINT_DTYPE array) = LastFrame.mvpMapPoints; None = its keypoints with depth, unprojected
        with Twc_last on the device."""
        assert gray.dtype == np.uint8 and depth16.dtype == np.uint16 and depth16.shape == gray.shape
        tc = np.ascontiguousarray(Tcw_cur, np.float32); tl = np.ascontiguousarray(Tcw_last, np.float32)
        tw = np.ascontiguousarray(Twc_last, np.float32) if Twc_last is not None else None
        mp = np.ascontiguousarray(last_mp, MAPPOINT_DTYPE) if last_mp is not None else None
        self._chk(self.L.drfe_frame_submit_tracked(self.h, slot, _p(gray), gray.shape[1], gray.shape[0], gray.strides[0], _p(depth16),
                                                   C.c_size_t(depth16.strides[0] // 2), C.byref(cam), last_slot, _p(tc), _p(tl), _p(tw),
                                                   _p(mp), 0 if mp is None else len(mp), float(th), int(mono), int(check_ori)),
                  "drfe_frame_submit_tracked")

    def frame_collect_tracked(self, slot: int):
        """-> (mvKeys, mDescriptors, mvuRight, mvDepth, cur_to_last, nmatches) of the slot's tracked submission."""
        kps = np.zeros(self.max_kp, KP_DTYPE)
        desc = np.zeros((self.max_kp, 32), np.uint8)
        ur = np.zeros(self.max_kp, np.float32); z = np.zeros(self.max_kp, np.float32)
        m = np.full(self.max_kp, -1, np.int32)
        n, nm = C.c_int(0), C.c_int(0)
        self._chk(self.L.drfe_frame_collect_tracked(self.h, slot, _p(kps), _p(desc), _p(ur), _p(z), self.max_kp, C.byref(n), _p(m),
                                                    C.byref(nm)), "drfe_frame_collect_tracked")
        k = n.value
        return kps[:k].copy(), desc[:k].copy(), ur[:k].copy(), z[:k].copy(), m[:k].copy(), nm.value

    def orb_extract_batch_ptr(self, d_gray: int, frame_stride: int, row_stride: int, w: int, h: int, nframes: int,
                              stream: int = 0):
        self._chk(self.L.drfe_orb_extract_batch(self.h, C.c_void_p(d_gray), frame_stride, row_stride, w, h, nframes,
                                                C.c_void_p(stream)), "drfe_orb_extract_batch")

    def orb_download(self, slot: int):
        kps = np.zeros(self.max_kp, KP_DTYPE)
        desc = np.zeros((self.max_kp, 32), np.uint8)
        n = C.c_int(0)
        self._chk(self.L.drfe_orb_download(self.h, slot, _p(kps), _p(desc), self.max_kp, C.byref(n)), "drfe_orb_download")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def batch_download_async_ptr(self, nframes: int, kps: int, desc: int, kp_counts: int, matches: int, match_counts: int,
                                 stream: int = 0):
        """Raw host pointers (pinned torch tensors): see drfe_batch_download_async."""
        self._chk(self.L.drfe_batch_download_async(self.h, nframes, C.c_void_p(kps), C.c_void_p(desc), C.c_void_p(kp_counts),
                                                   C.c_void_p(matches), C.c_void_p(match_counts), C.c_void_p(stream)),
                  "drfe_batch_download_async")

    def fast_partition(self, w: int, h: int):
        """-> (cells[2], evaluated pixels[2]) of the two FAST launches for a w x h frame."""
        cells = np.zeros(2, np.int32)
        px = np.zeros(2, np.int64)
        self._chk(self.L.drfe_orb_fast_partition(self.h, w, h, _p(cells), _p(px)), "drfe_orb_fast_partition")
        return cells, px

    def orb_counts(self, nframes: int):
        c = np.zeros(nframes, np.int32)
        self._chk(self.L.drfe_orb_counts(self.h, nframes, _p(c)), "drfe_orb_counts")
        return c

    def pyramid_level(self, slot, level):
        w, h = C.c_int(), C.c_int()
        self._chk(self.L.drfe_orb_pyramid_level(self.h, slot, level, None, C.byref(w), C.byref(h)), "pyramid_level")
        out = np.zeros((h.value, w.value), np.uint8)
        self._chk(self.L.drfe_orb_pyramid_level(self.h, slot, level, _p(out), C.byref(w), C.byref(h)), "pyramid_level")
        return out

    def blurred_level(self, slot, level):
        w, h = C.c_int(), C.c_int()
        self._chk(self.L.drfe_orb_blurred_level(self.h, slot, level, None, C.byref(w), C.byref(h)), "blurred_level")
        out = np.zeros((h.value, w.value), np.uint8)
        self._chk(self.L.drfe_orb_blurred_level(self.h, slot, level, _p(out), C.byref(w), C.byref(h)), "blurred_level")
        return out

    def candidates(self, slot, level):
        n = C.c_int()
        self._chk(self.L.drfe_orb_candidates(self.h, slot, level, None, 0, C.byref(n)), "candidates")
        out = np.zeros((max(n.value, 1), 3), np.int32)
        self._chk(self.L.drfe_orb_candidates(self.h, slot, level, _p(out), len(out), C.byref(n)), "candidates")
        return out[:n.value]

    # --- Frame glue --------------------------------------------------------------------------------
    def stereo_grid_batch_ptr(self, d_depth: int, frame_stride_elems: int, row_stride_elems: int, cam: Camera,
                              nframes: int, stream: int = 0):
        self._chk(self.L.drfe_frame_stereo_grid_batch(self.h, C.c_void_p(d_depth), frame_stride_elems,
                                                      row_stride_elems, C.byref(cam), nframes, C.c_void_p(stream)),
                  "drfe_frame_stereo_grid_batch")

    def keypoint_pixels_async_ptr(self, nframes: int, uv: int, counts: int, stream: int = 0):
        """uv / counts: raw host pointers (pinned) of [nframes][max_kp] uint32 and [nframes] int32."""
        self._chk(self.L.drfe_orb_keypoint_pixels_async(self.h, nframes, C.c_void_p(uv), C.c_void_p(counts), C.c_void_p(stream)),
                  "drfe_orb_keypoint_pixels_async")

    def gather_keypoint_depth(self, depth16: np.ndarray, uv: np.ndarray, counts: np.ndarray, out: np.ndarray, n_threads=0):
        """Host gather: out[f, i] = depth16[f, v, u] for the pixels drfe_orb_keypoint_pixels_async reported."""
        B, h, w = depth16.shape
        rc = self.L.drfe_gather_keypoint_depth(_p(depth16), w * h, w, B, _p(uv), _p(counts), self.max_kp, _p(out), int(n_threads))
        if rc != 0:
            raise DrfeError(f"drfe_gather_keypoint_depth failed ({rc})")

    def stereo_grid_batch_kpdepth_ptr(self, kp_depth: int, on_host: bool, cam: Camera, nframes: int, stream: int = 0):
        self._chk(self.L.drfe_frame_stereo_grid_batch_kpdepth(self.h, C.c_void_p(kp_depth), int(on_host), C.byref(cam), nframes,
                                                               C.c_void_p(stream)), "drfe_frame_stereo_grid_batch_kpdepth")

    def download_stereo(self, slot):
        ur = np.zeros(self.max_kp, np.float32)
        z = np.zeros(self.max_kp, np.float32)
        self._chk(self.L.drfe_frame_download_stereo(self.h, slot, _p(ur), _p(z), self.max_kp), "download_stereo")
        return ur, z

    def frame_load(self, slot, kps, desc, cam: Camera, kps_un=None, u_right=None, depth_m=None):
        """drfe_frame_load: a host-held frame (KeyFrame members mvKeys / mvKeysUn / mDescriptors / mvuRight / mvDepth) into `slot`;
        the 64 x 48 grid is rebuilt on the device."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(kps)
        ku = None if kps_un is None else np.ascontiguousarray(kps_un, KP_DTYPE)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        z = None if depth_m is None else np.ascontiguousarray(depth_m, np.float32)
        self._chk(self.L.drfe_frame_load(self.h, slot, _p(kps), _p(ku), _p(desc), _p(ur), _p(z), n, C.byref(cam)), "drfe_frame_load")

    def download_grid(self, slot):
        off = np.zeros(64 * 48 + 1, np.int32)
        idx = np.zeros(self.max_kp, np.int32)
        self._chk(self.L.drfe_frame_download_grid(self.h, slot, _p(off), _p(idx), self.max_kp), "download_grid")
        return off, idx[:off[-1]].copy()

    # --- matchers ----------------------------------------------------------------------------------
    def match_consecutive_batch(self, Tcw: np.ndarray, Twc: np.ndarray, cam: Camera, th=15.0, mono=False,
                                check_ori=True, nframes=None, stream: int = 0):
        Tcw = np.ascontiguousarray(Tcw, np.float32)
        Twc = np.ascontiguousarray(Twc, np.float32)
        nframes = len(Tcw) if nframes is None else nframes
        self._chk(self.L.drfe_match_consecutive_batch(self.h, _p(Tcw), _p(Twc), C.byref(cam), th, int(mono),
                                                      int(check_ori), nframes, C.c_void_p(stream)),
                  "drfe_match_consecutive_batch")

    def match_download(self, slot):
        m = np.zeros(self.max_kp, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_match_download(self.h, slot, _p(m), self.max_kp, C.byref(n)), "drfe_match_download")
        return m, n.value

    def search_by_projection_last(self, cur_slot, last_slot, Tcw_cur, Tcw_last, cam, last_mp, n_cur, th=15.0,
                                  mono=False, check_ori=True, cur_mp=None, cur_obs=None):
        last_mp = np.ascontiguousarray(last_mp, MAPPOINT_DTYPE)
        out = np.full(n_cur, -1, np.int32) if cur_mp is None else np.ascontiguousarray(cur_mp, np.int32).copy()
        obs = None if cur_obs is None else np.ascontiguousarray(cur_obs, np.uint8)
        n = C.c_int()
        self._chk(self.L.drfe_search_by_projection_last(
            self.h, cur_slot, last_slot, _p(np.ascontiguousarray(Tcw_cur, np.float32)),
            _p(np.ascontiguousarray(Tcw_last, np.float32)), C.byref(cam), _p(last_mp), len(last_mp), th, int(mono),
            int(check_ori), _p(obs), _p(out), n_cur, C.byref(n)), "drfe_search_by_projection_last")
        return n.value, out

    def search_by_projection_map(self, slot, mps, n, th, nnratio, frame_mp=None, claim_obs=None):
        mps = np.ascontiguousarray(mps, TRACKED_DTYPE)
        out = np.full(n, -1, np.int32) if frame_mp is None else np.ascontiguousarray(frame_mp, np.int32).copy()
        obs = None if claim_obs is None else np.ascontiguousarray(claim_obs, np.uint8)
        nm = C.c_int()
        self._chk(self.L.drfe_search_by_projection_map(self.h, slot, _p(mps), len(mps), th, nnratio, _p(obs), _p(out),
                                                       n, C.byref(nm)), "drfe_search_by_projection_map")
        return nm.value, out

    def match_orb_points(self, cur_slot, last_slot, last_mp, last_outlier, n_cur, cur_mp=None):
        last_mp = np.ascontiguousarray(last_mp, np.int32)
        last_outlier = np.ascontiguousarray(last_outlier, np.uint8)
        out = np.full(n_cur, -1, np.int32) if cur_mp is None else np.ascontiguousarray(cur_mp, np.int32).copy()
        n = C.c_int()
        self._chk(self.L.drfe_match_orb_points(self.h, cur_slot, last_slot, _p(last_mp), _p(last_outlier), len(last_mp),
                                               _p(out), n_cur, C.byref(n)), "drfe_match_orb_points")
        return n.value, out

    def is_in_frustum(self, Tcw, cam, pts, viewing_cos_limit, out=None):
        """Frame::isInFrustum(MapPoint*, limit) for an array of map points; returns the drfe_tracked_point array with
        the tracking fields filled (bad / obs_positive / desc of `out` are preserved)."""
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
        o = np.zeros(len(p), TRACKED_DTYPE) if out is None else np.ascontiguousarray(out, TRACKED_DTYPE).copy()
        self._chk(self.L.drfe_frame_is_in_frustum(self.h, _p(T), C.byref(cam), _p(p), len(p), C.c_float(viewing_cos_limit),
                                                  _p(o)), "drfe_frame_is_in_frustum")
        return o

    def fuse_search(self, slot, Tcw, pts, descs, skip, th):
        """Search part of ORBmatcher::Fuse(pKF, vpMapPoints, th); returns (best_idx, best_dist) per map point."""
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
        d = np.ascontiguousarray(descs, np.uint8)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        bi = np.zeros(len(p), np.int32)
        bd = np.zeros(len(p), np.int32)
        self._chk(self.L.drfe_fuse_search(self.h, slot, _p(T), _p(p), _p(d), _p(sk), len(p), C.c_float(th), _p(bi), _p(bd)),
                  "drfe_fuse_search")
        return bi, bd

    def fuse_search_sim3(self, slot, Scw, pts, descs, skip, th):
        """Search part of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)."""
        T = np.ascontiguousarray(Scw, np.float32).reshape(16)
        p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
        d = np.ascontiguousarray(descs, np.uint8)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8)
        bi = np.zeros(len(p), np.int32)
        bd = np.zeros(len(p), np.int32)
        self._chk(self.L.drfe_fuse_search_sim3(self.h, slot, _p(T), _p(p), _p(d), _p(sk), len(p), C.c_float(th), _p(bi), _p(bd)),
                  "drfe_fuse_search_sim3")
        return bi, bd

    def lsd_fuse_search(self, Tcw, cam, lines, descs, skip, kf_lines, kf_desc, th):
        """Search part of LSDmatcher::Fuse(pKF, vpMapLines, th); returns (best_idx, best_dist) per map line."""
        l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
        kl = np.ascontiguousarray(kf_lines, KEYLINE_DTYPE)
        bi = np.zeros(len(l), np.int32)
        bd = np.zeros(len(l), np.int32)
        self._chk(self.L.drfe_lsd_fuse_search(self.h, _p(np.ascontiguousarray(Tcw, np.float32).reshape(16)), C.byref(cam), _p(l),
                                              _p(np.ascontiguousarray(descs, np.uint8)), _p(np.ascontiguousarray(skip, np.uint8)),
                                              len(l), _p(kl), _p(np.ascontiguousarray(kf_desc, np.uint8)), len(kl), C.c_float(th),
                                              _p(bi), _p(bd)), "drfe_lsd_fuse_search")
        return bi, bd

    def lsd_fuse_search_sim3(self, Scw, cam, lines, descs, skip, kf_lines, kf_desc, th):
        """Search part of LSDmatcher::Fuse(pKF, Scw, vpLines, th, vpReplaceLine); returns (best_idx, best_dist) per map line."""
        l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
        kl = np.ascontiguousarray(kf_lines, KEYLINE_DTYPE)
        bi = np.zeros(len(l), np.int32)
        bd = np.zeros(len(l), np.int32)
        self._chk(self.L.drfe_lsd_fuse_search_sim3(self.h, _p(np.ascontiguousarray(Scw, np.float32).reshape(16)), C.byref(cam), _p(l),
                                                   _p(np.ascontiguousarray(descs, np.uint8)), _p(np.ascontiguousarray(skip, np.uint8)),
                                                   len(l), _p(kl), _p(np.ascontiguousarray(kf_desc, np.uint8)), len(kl), C.c_float(th),
                                                   _p(bi), _p(bd)), "drfe_lsd_fuse_search_sim3")
        return bi, bd

    def lsd_search_by_projection_kf(self, Scw, cam, lines, descs, skip, kf_lines, kf_desc, matched, th):
        """LSDmatcher::SearchByProjection(pKF, Scw, vpLines, vpMatched, th); returns (nmatches, new_match per key line)."""
        l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
        kl = np.ascontiguousarray(kf_lines, KEYLINE_DTYPE)
        m = np.ascontiguousarray(matched, np.uint8)
        out = np.full(len(kl), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_lsd_search_by_projection_kf(self.h, _p(np.ascontiguousarray(Scw, np.float32).reshape(16)), C.byref(cam),
                                                          _p(l), _p(np.ascontiguousarray(descs, np.uint8)),
                                                          _p(np.ascontiguousarray(skip, np.uint8)), len(l), _p(kl),
                                                          _p(np.ascontiguousarray(kf_desc, np.uint8)), len(kl), _p(m), int(th), _p(out),
                                                          C.byref(n)), "drfe_lsd_search_by_projection_kf")
        return n.value, out

    def lsd_search_by_sim3(self, cam, T1w, T2w, s12, R12, t12, lines1, descs1, skip1, kf1_lines, kf1_desc, lines2, descs2, skip2,
                           kf2_lines, kf2_desc, th):
        """LSDmatcher::SearchBySim3; returns (nFound, matches12[i1] = key line of KF2 or -1)."""
        f = lambda a, n: _p(np.ascontiguousarray(a, np.float32).reshape(n))
        u8 = lambda a: _p(np.ascontiguousarray(a, np.uint8))
        l1, l2 = np.ascontiguousarray(lines1, FRUSTUM_LINE_DTYPE), np.ascontiguousarray(lines2, FRUSTUM_LINE_DTYPE)
        k1, k2 = np.ascontiguousarray(kf1_lines, KEYLINE_DTYPE), np.ascontiguousarray(kf2_lines, KEYLINE_DTYPE)
        assert len(l1) == len(k1) and len(l2) == len(k2)
        out = np.full(len(l1), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_lsd_search_by_sim3(self.h, C.byref(cam), f(T1w, 16), f(T2w, 16), C.c_float(s12), f(R12, 9), f(t12, 3),
                                                 _p(l1), u8(descs1), u8(skip1), _p(k1), u8(kf1_desc), len(l1), _p(l2), u8(descs2),
                                                 u8(skip2), _p(k2), u8(kf2_desc), len(l2), C.c_float(th), _p(out), C.byref(n)),
                  "drfe_lsd_search_by_sim3")
        return n.value, out

    def search_by_projection_kf(self, slot, Scw, pts, descs, skip, matched, th):
        """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th); returns (nmatches, new_match per keypoint)."""
        p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
        m = np.ascontiguousarray(matched, np.uint8)
        out = np.full(len(m), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_search_by_projection_kf(self.h, slot, _p(np.ascontiguousarray(Scw, np.float32).reshape(16)), _p(p),
                                                      _p(np.ascontiguousarray(descs, np.uint8)),
                                                      _p(np.ascontiguousarray(skip, np.uint8)), len(p), _p(m), len(m),
                                                      C.c_float(th), _p(out), C.byref(n)), "drfe_search_by_projection_kf")
        return n.value, out

    def search_by_projection_reloc(self, slot, Tcw, pts, descs, kf_angles, skip, matched, th, orb_dist, check_orientation=True):
        """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist); returns (nmatches, new_match)."""
        p = np.ascontiguousarray(pts, FRUSTUM_POINT_DTYPE)
        m = np.ascontiguousarray(matched, np.uint8)
        out = np.full(len(m), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_search_by_projection_reloc(self.h, slot, _p(np.ascontiguousarray(Tcw, np.float32).reshape(16)), _p(p),
                                                         _p(np.ascontiguousarray(descs, np.uint8)),
                                                         _p(np.ascontiguousarray(kf_angles, np.float32)),
                                                         _p(np.ascontiguousarray(skip, np.uint8)), len(p), _p(m), len(m),
                                                         C.c_float(th), int(orb_dist), int(bool(check_orientation)), _p(out),
                                                         C.byref(n)), "drfe_search_by_projection_reloc")
        return n.value, out

    def search_for_initialization(self, slot1, slot2, prev_matched, window_size=100, nnratio=0.9, check_orientation=True):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize); returns (nmatches, matches12,
        prev_matched after the update)."""
        pm = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
        out = np.full(len(pm), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_search_for_initialization(self.h, slot1, slot2, _p(pm), len(pm), int(window_size), C.c_float(nnratio),
                                                        int(bool(check_orientation)), _p(out), C.byref(n)),
                  "drfe_search_for_initialization")
        return n.value, out, pm

    def search_by_sim3(self, slot1, slot2, T1w, T2w, s12, R12, t12, pts1, descs1, skip1, pts2, descs2, skip2, th):
        """ORBmatcher::SearchBySim3; returns (nFound, matches12[i1] = i2 or -1)."""
        f = lambda a, n: np.ascontiguousarray(a, np.float32).reshape(n)
        p1, p2 = np.ascontiguousarray(pts1, FRUSTUM_POINT_DTYPE), np.ascontiguousarray(pts2, FRUSTUM_POINT_DTYPE)
        out = np.full(len(p1), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_search_by_sim3(self.h, slot1, slot2, _p(f(T1w, 16)), _p(f(T2w, 16)), C.c_float(s12), _p(f(R12, 9)),
                                             _p(f(t12, 3)), _p(p1), _p(np.ascontiguousarray(descs1, np.uint8)),
                                             _p(np.ascontiguousarray(skip1, np.uint8)), len(p1), _p(p2),
                                             _p(np.ascontiguousarray(descs2, np.uint8)), _p(np.ascontiguousarray(skip2, np.uint8)),
                                             len(p2), C.c_float(th), _p(out), C.byref(n)), "drfe_search_by_sim3")
        return n.value, out

    def is_in_frustum_lines(self, Tcw, cam, lines, viewing_cos_limit, out=None):
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        l = np.ascontiguousarray(lines, FRUSTUM_LINE_DTYPE)
        o = np.zeros(len(l), TRACKED_LINE_DTYPE) if out is None else np.ascontiguousarray(out, TRACKED_LINE_DTYPE).copy()
        self._chk(self.L.drfe_frame_is_in_frustum_lines(self.h, _p(T), C.byref(cam), _p(l), len(l),
                                                        C.c_float(viewing_cos_limit), _p(o)), "drfe_frame_is_in_frustum_lines")
        return o

    def set_distortion(self, cam, dist):
        """Frame::UndistortKeyPoints model (k1, k2, p1, p2[, k3]); None / k1 == 0 switches it off."""
        d = np.zeros(0, np.float32) if dist is None else np.ascontiguousarray(dist, np.float32)
        self._chk(self.L.drfe_frame_set_distortion(self.h, C.byref(cam), _p(d), len(d)), "drfe_frame_set_distortion")

    def image_bounds(self, cam, dist, cols, rows):
        d = np.zeros(0, np.float32) if dist is None else np.ascontiguousarray(dist, np.float32)
        out = np.zeros(4, np.float32)
        self._chk(self.L.drfe_frame_image_bounds(C.byref(cam), _p(d), len(d), int(cols), int(rows), _p(out)),
                  "drfe_frame_image_bounds")
        return out

    def download_keys_un(self, slot, n):
        out = np.zeros(max(n, 1), KP_DTYPE)
        self._chk(self.L.drfe_frame_download_keys_un(self.h, slot, _p(out), len(out)), "drfe_frame_download_keys_un")
        return out[:n]

    def lsd_search_by_descriptor(self, desc_q, desc_t, has_line=None, mode=0):
        dq = np.ascontiguousarray(desc_q, np.uint8)
        dt = np.ascontiguousarray(desc_t, np.uint8)
        has = None if has_line is None else np.ascontiguousarray(has_line, np.uint8)
        out = np.full(len(dt) if mode == 0 else len(dq), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_lsd_search_by_descriptor(self.h, _p(dq), len(dq), _p(dt), len(dt), _p(has), mode, _p(out),
                                                       C.byref(n)), "drfe_lsd_search_by_descriptor")
        return n.value, out

    def lsd_search_by_projection_last(self, Tcw_cur, Tcw_last, cam, last_lines, cur_lines, cur_desc, th, mono, nnratio,
                                      cur_ml, cur_obs=None):
        """LSDmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono); returns (nmatches, cur_ml)."""
        tc = np.ascontiguousarray(Tcw_cur, np.float32).reshape(16)
        tl = np.ascontiguousarray(Tcw_last, np.float32).reshape(16)
        ll = np.ascontiguousarray(last_lines, MAPLINE_DTYPE)
        kl = np.ascontiguousarray(cur_lines, KEYLINE_DTYPE)
        cd = np.ascontiguousarray(cur_desc, np.uint8)
        out = np.ascontiguousarray(cur_ml, np.int32).copy()
        obs = None if cur_obs is None else np.ascontiguousarray(cur_obs, np.uint8)
        n = C.c_int()
        self._chk(self.L.drfe_lsd_search_by_projection_last(self.h, _p(tc), _p(tl), C.byref(cam), _p(ll), len(ll), _p(kl),
                                                            _p(cd), len(kl), C.c_float(th), int(mono), C.c_float(nnratio),
                                                            _p(obs), _p(out), C.byref(n)),
                  "drfe_lsd_search_by_projection_last")
        return n.value, out

    def lsd_search_by_projection_map(self, lines, cur_lines, cur_desc, th, nnratio, cur_ml, cur_obs=None):
        """LSDmatcher::SearchByProjection(F, vpMapLines, th); returns (nmatches, cur_ml)."""
        tl = np.ascontiguousarray(lines, TRACKED_LINE_DTYPE)
        kl = np.ascontiguousarray(cur_lines, KEYLINE_DTYPE)
        cd = np.ascontiguousarray(cur_desc, np.uint8)
        out = np.ascontiguousarray(cur_ml, np.int32).copy()
        obs = None if cur_obs is None else np.ascontiguousarray(cur_obs, np.uint8)
        n = C.c_int()
        self._chk(self.L.drfe_lsd_search_by_projection_map(self.h, _p(tl), len(tl), _p(kl), _p(cd), len(kl), C.c_float(th),
                                                           C.c_float(nnratio), _p(obs), _p(out), C.byref(n)),
                  "drfe_lsd_search_by_projection_map")
        return n.value, out

    def lsd_search_for_triangulation(self, desc1, desc2, has1, has2):
        d1 = np.ascontiguousarray(desc1, np.uint8)
        d2 = np.ascontiguousarray(desc2, np.uint8)
        out = np.full(len(d1), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_lsd_search_for_triangulation(self.h, _p(d1), len(d1), _p(d2), len(d2),
                                                           _p(np.ascontiguousarray(has1, np.uint8)),
                                                           _p(np.ascontiguousarray(has2, np.uint8)), _p(out), C.byref(n)),
                  "drfe_lsd_search_for_triangulation")
        return n.value, out

    def bf_knn(self, Q, T, k):
        Q = np.ascontiguousarray(Q, np.uint8)
        T = np.ascontiguousarray(T, np.uint8)
        idx = np.zeros((len(Q), k), np.int32)
        dist = np.zeros((len(Q), k), np.int32)
        self._chk(self.L.drfe_match_bf_knn(self.h, _p(Q), len(Q), _p(T), len(T), k, _p(idx), _p(dist)), "drfe_match_bf_knn")
        return idx, dist

    # --- lines -------------------------------------------------------------------------------------
    def lsd_extract(self, gray: np.ndarray, max_lines=40, stages=False):
        """LineSegment::ExtractLineSegment -> dict(lines, desc, lineF, detected[, stage images])."""
        g = np.ascontiguousarray(gray, np.uint8)
        h, w = g.shape
        cap = max_lines
        lines = np.zeros(cap, KEYLINE_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        lf = np.zeros((cap, 3))
        n, nd = C.c_int(), C.c_int()
        self._chk(self.L.drfe_lsd_extract(self.h, _p(g), w, h, w, max_lines, _p(lines), _p(desc), _p(lf), cap,
                                          C.byref(n), C.byref(nd)), "drfe_lsd_extract")
        out = dict(lines=lines[:n.value].copy(), desc=desc[:n.value].copy(), lineF=lf[:n.value].copy(), detected=nd.value)
        if stages:
            sw, sh = C.c_int(), C.c_int()
            self._chk(self.L.drfe_lsd_stages(self.h, None, None, None, None, None, C.byref(sw), C.byref(sh)), "lsd_stages")
            scaled = np.zeros((sh.value, sw.value), np.uint8)
            modgrad, angles = np.zeros((sh.value, sw.value)), np.zeros((sh.value, sw.value))
            gx, gy = np.zeros((h, w), np.int16), np.zeros((h, w), np.int16)
            self._chk(self.L.drfe_lsd_stages(self.h, _p(scaled), _p(modgrad), _p(angles), _p(gx), _p(gy), C.byref(sw),
                                             C.byref(sh)), "lsd_stages")
            out.update(scaled=scaled, modgrad=modgrad, angles=angles, gx=gx, gy=gy)
        return out

    def lsd_configure(self, device_grow=True):
        """Where lsd_extract_batch grows its regions: False / 0 the host pool; True / 1 the device, kernel chosen by the size of the
        call (default); 2 the device with one wavefront per frame; 3 with four (speculation + in-order commit)."""
        self._chk(self.L.drfe_lsd_configure(self.h, int(device_grow)), "drfe_lsd_configure")

    def planes_configure_cape(self, on_device=True):
        """Where planes_cape_batch runs CAPE::process (histogram seeding, cell growing, merging, masks): device or host pool."""
        self._chk(self.L.drfe_planes_configure_cape(self.h, 1 if on_device else 0), "drfe_planes_configure_cape")

    def planes_ahc_stats(self):
        """dict(frames, to_host, voxel_grids, voxel_grids_to_host): counters of the device extractor behind planes_ahc_batch /
        planes_ahc_post_batch since the context was created"""
        out = np.zeros(4, np.int64)
        self._chk(self.L.drfe_planes_ahc_stats(self.h, _p(out)), "drfe_planes_ahc_stats")
        return dict(frames=int(out[0]), to_host=int(out[1]), voxel_grids=int(out[2]), voxel_grids_to_host=int(out[3]))

    def planes_configure_refit(self, on_device=True):
        """Where planes_ahc_post_batch runs gates + RANSAC refit: the device behind the device voxel grids (default) or the host pool."""
        self._chk(self.L.drfe_planes_configure_refit(self.h, 1 if on_device else 0), "drfe_planes_configure_refit")

    def planes_refit_stats(self):
        out = np.zeros(2, np.int64)
        self._chk(self.L.drfe_planes_refit_stats(self.h, _p(out)), "drfe_planes_refit_stats")
        return dict(frames=int(out[0]), to_host=int(out[1]))

    def planes_cape_stats(self):
        out = np.zeros(2, np.int64)
        self._chk(self.L.drfe_planes_cape_stats(self.h, _p(out)), "drfe_planes_cape_stats")
        return dict(frames=int(out[0]), to_host=int(out[1]))

    def lsd_configure_nfa(self, device_nfa=True):
        """Where lsd_extract_batch takes rect_improve's NFA decisions: on the device (certified comparisons, default) or on the
        host pool with the caller's libm."""
        self._chk(self.L.drfe_lsd_configure_nfa(self.h, 1 if device_nfa else 0), "drfe_lsd_configure_nfa")

    def long_kernel_clock(self, on=True):
        """lsd_extract_batch / planes_ahc_post_batch bracket every kernel of their first chunk with HIP events (long_kernel_ms)"""
        self._chk(self.L.drfe_long_kernel_clock(self.h, int(bool(on))), "drfe_long_kernel_clock")

    def long_kernel_ms(self):
        """ms of the last clocked batch call's first chunk, by kernel (0 where the stage ran elsewhere)"""
        out = np.zeros(16, np.float32)
        self._chk(self.L.drfe_long_kernel_ms(self.h, _p(out)), "drfe_long_kernel_ms")
        names = ("lines_upload", "lines_image_passes", "k_lsd_keys", "k_lsd_order", "k_lsd_grow", "k_rect_improve", "k_lsd_keylines+k_lbd", None,
                 "planes_upload", "k_ahc_blocks", "k_ahc_cluster", "k_ahc_refine", "k_ahc_labels", "k_voxel_grid", "k_plane_refit", None)
        return {n: float(v) for n, v in zip(names, out) if n}

    def lsd_stats(self):
        """dict(frames, grow_to_host, nfa_to_host, keylines_to_host): counters of lsd_extract_batch's device path since the context was created"""
        out = np.zeros(4, np.int64)
        self._chk(self.L.drfe_lsd_stats(self.h, _p(out)), "drfe_lsd_stats")
        return dict(frames=int(out[0]), grow_to_host=int(out[1]), nfa_to_host=int(out[2]), keylines_to_host=int(out[3]))

    def lsd_configure_rect(self, rect_mode=0):
        """The reading of OpenCV 3.4's lsd.cpp: 0 the source text (rect_nfa's integer corners and step quotients, nfa()'s
        `double(n) + 1` first term; default), 1 the LSD paper's reading of both (rounds 2-3), 2 integer corners with
        log_gamma(n + 1) (round 4's default)."""
        self._chk(self.L.drfe_lsd_configure_rect(self.h, int(rect_mode)), "drfe_lsd_configure_rect")

    def lsd_extract_batch(self, gray_batch: np.ndarray, max_lines=40, n_threads=0):
        """LineSegment::ExtractLineSegment for a [B, H, W] uint8 host array: region growing on the device (or, after
        lsd_configure(False), on the host pool), ordering + NFA on a pool of host threads; returns a list of dicts like
        lsd_extract."""
        g = np.ascontiguousarray(gray_batch, np.uint8)
        B, h, w = g.shape
        cap = max_lines
        lines = np.zeros((B, cap), KEYLINE_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        lf = np.zeros((B, cap, 3))
        n = np.zeros(B, np.int32)
        nd = np.zeros(B, np.int32)
        self._chk(self.L.drfe_lsd_extract_batch(self.h, _p(g), w * h, w, h, w, B, max_lines, _p(lines), _p(desc), _p(lf), cap,
                                                _p(n), _p(nd), int(n_threads)), "drfe_lsd_extract_batch")
        return [dict(lines=lines[f, :n[f]].copy(), desc=desc[f, :n[f]].copy(), lineF=lf[f, :n[f]].copy(), detected=int(nd[f]))
                for f in range(B)]

    # --- bag of words ------------------------------------------------------------------------------
    def voc_upload(self, k, L, scoring, weighting, parent, desc, weight, is_leaf):
        parent = np.ascontiguousarray(parent, np.int32)
        desc = np.ascontiguousarray(desc, np.uint8)
        weight = np.ascontiguousarray(weight, np.float64)
        is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        self._chk(self.L.drfe_voc_upload(self.h, k, L, scoring, weighting, len(parent), _p(parent), _p(desc), _p(weight),
                                         _p(is_leaf)), "drfe_voc_upload")

    def bow_transform_batch(self, levelsup, nframes, stream: int = 0):
        self._chk(self.L.drfe_bow_transform_batch(self.h, levelsup, nframes, C.c_void_p(stream)), "drfe_bow_transform_batch")

    def bow_transform_slot(self, levelsup, slot, stream: int = 0):
        self._chk(self.L.drfe_bow_transform_slot(self.h, levelsup, slot, C.c_void_p(stream)), "drfe_bow_transform_slot")

    def bow_download(self, slot):
        word = np.zeros(self.max_kp, np.int32)
        weight = np.zeros(self.max_kp, np.float64)
        nid = np.zeros(self.max_kp, np.int32)
        self._chk(self.L.drfe_bow_download(self.h, slot, _p(word), _p(weight), _p(nid), self.max_kp), "drfe_bow_download")
        return word, weight, nid

    def search_by_bow(self, kf_slot, f_slot, kf_mp, n_f, nnratio, check_ori=True):
        kf_mp = np.ascontiguousarray(kf_mp, np.int32)
        out = np.full(n_f, -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_search_by_bow(self.h, kf_slot, f_slot, _p(kf_mp), len(kf_mp), nnratio, int(check_ori),
                                            _p(out), n_f, C.byref(n)), "drfe_search_by_bow")
        return n.value, out

    def search_by_bow_kf(self, slot1, slot2, mp1, mp2, nnratio, check_ori=True):
        """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12); returns (nmatches, match2) with match2[i2] = i1 or -1."""
        mp1 = np.ascontiguousarray(mp1, np.int32)
        mp2 = np.ascontiguousarray(mp2, np.int32)
        out = np.full(len(mp2), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_search_by_bow_kf(self.h, slot1, slot2, _p(mp1), len(mp1), _p(mp2), len(mp2), C.c_float(nnratio),
                                               int(check_ori), _p(out), C.byref(n)), "drfe_search_by_bow_kf")
        return n.value, out

    def search_for_triangulation(self, slot1, slot2, mp1, mp2, F12, Cw1, T2w, cam2, only_stereo=False, check_ori=True):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, pairs, bOnlyStereo); returns (nmatches, matches12)."""
        mp1 = np.ascontiguousarray(mp1, np.int32)
        mp2 = np.ascontiguousarray(mp2, np.int32)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        Cw = np.ascontiguousarray(Cw1, np.float32).reshape(3)
        T = np.ascontiguousarray(T2w, np.float32).reshape(16)
        out = np.full(len(mp1), -1, np.int32)
        n = C.c_int()
        self._chk(self.L.drfe_search_for_triangulation(self.h, slot1, slot2, _p(mp1), len(mp1), _p(mp2), len(mp2), _p(F), _p(Cw),
                                                       _p(T), C.byref(cam2), int(only_stereo), int(check_ori), _p(out),
                                                       C.byref(n)), "drfe_search_for_triangulation")
        return n.value, out

    # --- planes ------------------------------------------------------------------------------------
    def planes_ahc(self, depth16: np.ndarray, K4, depth_factor, cap=64):
        """PlaneDetection::readDepthImage + runPlaneDetection -> dict(planes, seg, members)."""
        d = np.ascontiguousarray(depth16, np.uint16)
        h, w = d.shape
        K4 = np.ascontiguousarray(K4, np.float32)
        planes = np.zeros(cap, PLANE_DTYPE)
        n = C.c_int()
        seg = np.zeros((h, w), np.uint8)
        off = np.zeros(cap + 1, np.int32)
        idx = np.zeros(h * w, np.int32)
        self._chk(self.L.drfe_planes_ahc(self.h, _p(d), w, h, w, _p(K4), np.float32(depth_factor), _p(planes), cap,
                                         C.byref(n), _p(seg), _p(off), _p(idx)), "drfe_planes_ahc")
        n = n.value
        return dict(planes=planes[:n].copy(), seg=seg, members=[idx[off[i]:off[i + 1]].copy() for i in range(n)])

    def planes_ahc_batch(self, depth16_batch: np.ndarray, K4, depth_factor, cap=64, n_threads=0, members=True):
        """drfe_planes_ahc for a [B, H, W] uint16 host array on a pool of host threads; list of dicts like planes_ahc."""
        d = np.ascontiguousarray(depth16_batch, np.uint16)
        B, h, w = d.shape
        K4 = np.ascontiguousarray(K4, np.float32)
        planes = np.zeros((B, cap), PLANE_DTYPE)
        n = np.zeros(B, np.int32)
        seg = np.zeros((B, h, w), np.uint8)
        off = np.zeros((B, cap + 1), np.int32) if members else None
        idx = np.zeros((B, h * w), np.int32) if members else None
        self._chk(self.L.drfe_planes_ahc_batch(self.h, _p(d), w * h, w, h, w, B, _p(K4), np.float32(depth_factor), _p(planes),
                                               cap, _p(n), _p(seg), _p(off), _p(idx), int(n_threads)), "drfe_planes_ahc_batch")
        out = []
        for f in range(B):
            r = dict(planes=planes[f, :n[f]].copy(), seg=seg[f])
            if members:
                r["members"] = [idx[f, off[f, i]:off[f, i + 1]].copy() for i in range(n[f])]
            out.append(r)
        return out

    def planes_configure(self, device_voxel_grid=1):
        """Where planes_ahc_post_batch runs the per-plane voxel grids: 1 (default) the device behind the device extractor, 2 the
        device in the host-extractor mode too, 0 the pool's host threads."""
        self._chk(self.L.drfe_planes_configure(self.h, int(device_voxel_grid)), "drfe_planes_configure")

    def planes_configure_extractor(self, on_device=True):
        """Where planes_ahc_post_batch runs PEAC's extractor: the device (one wavefront per frame, default) or the host pool."""
        self._chk(self.L.drfe_planes_configure_extractor(self.h, 1 if on_device else 0), "drfe_planes_configure_extractor")

    def planes_cape_batch(self, depth_m: np.ndarray, K4, patch=20, cos_angle_max=None, max_merge_dist=50.0, cap=64, n_threads=0, seg=False):
        """PlaneDetection_CAPE for a [B, H, W] float32 batch on a pool of host threads -> (planes [B, cap], n [B][, seg [B, H, W]])."""
        d = np.ascontiguousarray(depth_m, np.float32)
        B, h, w = d.shape
        if cos_angle_max is None:
            cos_angle_max = np.float32(np.cos(np.pi / 12))
        planes = np.zeros((B, cap), CAPE_PLANE_DTYPE)
        n = np.zeros(B, np.int32)
        sg = np.zeros((B, h, w), np.uint8) if seg else None
        self._chk(self.L.drfe_planes_cape_batch(self.h, _p(d), w * h, w, h, w, B, _p(np.ascontiguousarray(K4, np.float32)), patch,
                                                np.float32(cos_angle_max), np.float32(max_merge_dist), _p(planes), cap, _p(n), _p(sg),
                                                int(n_threads)), "drfe_planes_cape_batch")
        return (planes, n, sg) if seg else (planes, n)

    def planes_cape(self, depth_m: np.ndarray, K4, patch=20, cos_angle_max=None, max_merge_dist=50.0, cap=64):
        """PlaneDetection_CAPE::readDepthImage + runPlaneDetection -> dict(planes, seg, cell taps)."""
        d = np.ascontiguousarray(depth_m, np.float32)
        h, w = d.shape
        if cos_angle_max is None:
            cos_angle_max = np.float32(np.cos(np.pi / 12))
        nc = (w // patch) * (h // patch)
        planes = np.zeros(cap, CAPE_PLANE_DTYPE)
        n = C.c_int()
        seg = np.zeros((h, w), np.uint8)
        cells = np.zeros((nc, 16))
        mst = np.zeros((nc, 3), np.float32)
        pn = np.zeros((nc, 2), np.int32)
        self._chk(self.L.drfe_planes_cape(self.h, _p(d), w, h, w, _p(np.ascontiguousarray(K4, np.float32)), patch,
                                          np.float32(cos_angle_max), np.float32(max_merge_dist), _p(planes), cap,
                                          C.byref(n), _p(seg), _p(cells), _p(mst), _p(pn)), "drfe_planes_cape")
        return dict(planes=planes[:n.value].copy(), seg=seg, cells=cells, cell_mst=mst, cell_planar=pn[:, 0],
                    cell_npts=pn[:, 1])

    def planes_ahc_blocks(self, depth16: np.ndarray, K4, depth_factor):
        d = np.ascontiguousarray(depth16, np.uint16)
        h, w = d.shape
        nb = (w // 10) * (h // 10)
        blocks = np.zeros((nb, 17))
        vn = np.zeros((nb, 2), np.int32)
        self._chk(self.L.drfe_planes_ahc_blocks(self.h, _p(d), w, h, w, _p(np.ascontiguousarray(K4, np.float32)),
                                                np.float32(depth_factor), _p(blocks), _p(vn), nb), "drfe_planes_ahc_blocks")
        return blocks, vn[:, 0], vn[:, 1]

    # --- plane post-processing + surface normals (Frame::ComputePlanes after the extractor) ------------
    def planes_ahc_postprocess(self, depth16, K4, depth_factor, ahc, max_point_dist, dist_threshold):
        """ahc = the dict planes_ahc returned. -> dict(post, voxels=[per plane [m,3]], n_accepted, plane_num)."""
        d = np.ascontiguousarray(depth16, np.uint16)
        h, w = d.shape
        planes = np.ascontiguousarray(ahc["planes"], PLANE_DTYPE)
        n = len(planes)
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum([len(m) for m in ahc["members"]])
        idx = np.concatenate([np.asarray(m, np.int32) for m in ahc["members"]]) if n else np.zeros(0, np.int32)
        idx = np.ascontiguousarray(idx, np.int32)
        post = np.zeros(n, PLANE_POST_DTYPE)
        vox = np.zeros((max(len(idx), 1), 3), np.float32)
        voff = np.zeros(n + 1, np.int32)
        na, pn = C.c_int(), C.c_int()
        self._chk(self.L.drfe_planes_ahc_postprocess(self.h, _p(d), w, h, w, _p(np.ascontiguousarray(K4, np.float32)),
                                                     np.float32(depth_factor), _p(planes), n, _p(off), _p(idx),
                                                     np.float32(max_point_dist), float(dist_threshold), _p(post), _p(vox),
                                                     _p(voff), len(vox), C.byref(na), C.byref(pn)),
                  "drfe_planes_ahc_postprocess")
        return dict(post=post, voxels=[vox[voff[i]:voff[i + 1]].copy() for i in range(n)], n_accepted=na.value,
                    plane_num=pn.value)

    def planes_ahc_post_batch(self, depth16_batch, K4, depth_factor, max_point_dist, dist_threshold, cap=64, n_threads=0, seg=False):
        """planes_ahc_batch + planes_ahc_postprocess per frame inside the C++ thread pool -> (planes [B,cap], n_planes [B],
        post [B,cap], n_accepted [B], plane_num [B][, seg [B,H,W]])."""
        d = np.ascontiguousarray(depth16_batch, np.uint16)
        B, h, w = d.shape
        planes = np.zeros((B, cap), PLANE_DTYPE)
        post = np.zeros((B, cap), PLANE_POST_DTYPE)
        n, na, pn = (np.zeros(B, np.int32) for _ in range(3))
        sg = np.zeros((B, h, w), np.uint8) if seg else None
        self._chk(self.L.drfe_planes_ahc_post_batch(self.h, _p(d), w * h, w, h, w, B, _p(np.ascontiguousarray(K4, np.float32)),
                                                    np.float32(depth_factor), np.float32(max_point_dist), float(dist_threshold),
                                                    _p(planes), cap, _p(n), _p(sg), _p(post), _p(na), _p(pn), int(n_threads)),
                  "drfe_planes_ahc_post_batch")
        return (planes, n, post, na, pn, sg) if seg else (planes, n, post, na, pn)

    def planes_cape_postprocess(self, depth_m, K4, cape, max_point_dist, dist_threshold):
        """cape = the dict planes_cape returned. -> dict(post, voxels, n_accepted, plane_num)."""
        d = np.ascontiguousarray(depth_m, np.float32)
        h, w = d.shape
        planes = np.ascontiguousarray(cape["planes"], CAPE_PLANE_DTYPE)
        seg = np.ascontiguousarray(cape["seg"], np.uint8)
        n = len(planes)
        post = np.zeros(n, PLANE_POST_DTYPE)
        vox = np.zeros((h * w, 3), np.float32)
        voff = np.zeros(n + 1, np.int32)
        na, pn = C.c_int(), C.c_int()
        self._chk(self.L.drfe_planes_cape_postprocess(self.h, _p(d), w, h, w, _p(np.ascontiguousarray(K4, np.float32)),
                                                      _p(seg), _p(planes), n, np.float32(max_point_dist),
                                                      float(dist_threshold), _p(post), _p(vox), _p(voff), len(vox),
                                                      C.byref(na), C.byref(pn)), "drfe_planes_cape_postprocess")
        return dict(post=post, voxels=[vox[voff[i]:voff[i + 1]].copy() for i in range(n)], n_accepted=na.value,
                    plane_num=pn.value)

    def surface_normals(self, depth_m, K4, max_point_dist, taps=False):
        """vSurfaceNormal of Frame::ComputePlanes for one CV_32F depth image (metres)."""
        d = np.ascontiguousarray(depth_m, np.float32)
        h, w = d.shape
        W, H = (w + 2) // 3, (h + 2) // 3
        out = np.zeros((H // 2) * (W // 2), SURFACE_NORMAL_DTYPE)
        n = C.c_int()
        cloud = np.zeros((H, W, 3), np.float32) if taps else None
        nrm = np.zeros((H, W, 3), np.float32) if taps else None
        dist = np.zeros((H, W), np.float32) if taps else None
        self._chk(self.L.drfe_surface_normals(self.h, _p(d), w, h, w, _p(np.ascontiguousarray(K4, np.float32)),
                                              np.float32(max_point_dist), _p(out), len(out), C.byref(n), _p(cloud), _p(nrm),
                                              _p(dist)), "drfe_surface_normals")
        return (out[:n.value], cloud, nrm, dist) if taps else out[:n.value]

    def surface_normals_batch_ptr(self, d_depth: int, frame_stride: int, row_stride: int, w: int, h: int, K4, depth_factor,
                                  max_point_dist, nframes: int, stream: int = 0):
        self._chk(self.L.drfe_surface_normals_batch(self.h, C.c_void_p(d_depth), frame_stride, row_stride, w, h,
                                                    _p(np.ascontiguousarray(K4, np.float32)), np.float32(depth_factor),
                                                    np.float32(max_point_dist), nframes, C.c_void_p(stream)),
                  "drfe_surface_normals_batch")

    def surface_normals_download(self, slot: int):
        n = C.c_int()
        self._chk(self.L.drfe_surface_normals_download(self.h, slot, None, 0, C.byref(n)), "drfe_surface_normals_download")
        out = np.zeros(n.value, SURFACE_NORMAL_DTYPE)
        self._chk(self.L.drfe_surface_normals_download(self.h, slot, _p(out), len(out), C.byref(n)),
                  "drfe_surface_normals_download")
        return out

    # --- measurement -------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self.L.drfe_profile_enable(self.h, int(on)), "drfe_profile_enable")

    def profile_stage_ms(self):
        ms = np.zeros(len(STAGES), np.float32)
        self._chk(self.L.drfe_profile_stage_ms(self.h, _p(ms)), "drfe_profile_stage_ms")
        return dict(zip(STAGES, ms.tolist()))

    def sync(self):
        self._chk(self.L.drfe_stream_sync(self.h), "drfe_stream_sync")
