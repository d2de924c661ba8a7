"""ctypes binding of the C ABI in include/mmf_hip.h (libmmf_hip.so).

This module is plumbing: it declares argument types and turns a non-zero status into a Python
exception carrying mmf_last_error().  There is NO fallback: if the library is missing or fails
to load the import raises, so a GPU box can never silently run something else.
"""
import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# MMF_HIP_LIB: an instrumented build of the same library (tools/rgb_step_probe.py), never a fallback
LIB_PATH = os.environ.get("MMF_HIP_LIB") or os.path.join(_PKG_DIR, "libmmf_hip.so")

MMF_NUM_PYRS = 3


class MmfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"mmf status {status}: {message}")
        self.status = status


class mmf_camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class mmf_dataterm(C.Structure):
    _fields_ = [("zero_x", C.c_int16), ("zero_y", C.c_int16), ("one_x", C.c_int16), ("one_y", C.c_int16),
                ("diff", C.c_float), ("valid", C.c_uint8), ("pad_", C.c_uint8 * 3)]


class mmf_odom_stats(C.Structure):
    _fields_ = [("lastICPError", C.c_float), ("lastICPCount", C.c_float), ("lastRGBError", C.c_float),
                ("lastRGBCount", C.c_float), ("lastSO3Error", C.c_float), ("lastSO3Count", C.c_float),
                ("lastA", C.c_double * 36), ("lastb", C.c_double * 6), ("iterations_run", C.c_int),
                ("so3_iterations_run", C.c_int)]


class mmf_fusion_config(C.Structure):
    _fields_ = [("time_delta", C.c_int), ("conf_global_init", C.c_float), ("icp_weight", C.c_float),
                ("depth_cutoff", C.c_float), ("max_depth_processed", C.c_float), ("rgb_only", C.c_int),
                ("pyramid", C.c_int), ("fast_odom", C.c_int), ("so3", C.c_int), ("frame_to_frame_rgb", C.c_int),
                ("outlier_coeff", C.c_float), ("fill_in", C.c_int), ("max_surfels", C.c_int),
                ("conf_object_init", C.c_float), ("enable_multiple_models", C.c_int), ("preallocated_models", C.c_int),
                ("error_recording", C.c_int), ("pose_logging", C.c_int), ("max_object_surfels", C.c_int), ("batch_tracking", C.c_int)]


class mmf_odom_timing(C.Structure):
    _fields_ = [("producer_us_sum", C.c_double * 3), ("producer_us_min", C.c_double * 3), ("rgb_step_us_sum", C.c_double * 3),
                ("rgb_step_us_min", C.c_double * 3), ("producer_launches", C.c_int * 3), ("rgb_step_launches", C.c_int * 3),
                ("chain_us_sum", C.c_double), ("chains", C.c_int)]


class mmf_segmentation_model(C.Structure):
    _fields_ = [("id", C.c_uint), ("super_pixel_count", C.c_uint), ("avg_confidence", C.c_float),
                ("depth_mean", C.c_float), ("depth_std", C.c_float)]


class mmf_segmentation(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("has_new_label", C.c_int), ("n_models", C.c_int),
                ("model_data", C.POINTER(mmf_segmentation_model))]


class mmf_frame(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("depth", C.c_void_p), ("timestamp", C.c_longlong),
                ("in_pose", C.POINTER(C.c_float)), ("weight_multiplier", C.c_float), ("bootstrap", C.c_int),
                ("init_transforms", C.POINTER(C.c_float)), ("n_init_transforms", C.c_int), ("icp_refine", C.c_int),
                ("segmentation", C.POINTER(mmf_segmentation)), ("next_rgb", C.c_void_p), ("next_depth", C.c_void_p)]


SEGMENTATION_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(mmf_frame), C.POINTER(mmf_segmentation))


_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_cam = C.POINTER(mmf_camera)

# name -> (restype, argtypes); must list every function declared in include/mmf_hip.h
SIGNATURES = {
    "mmf_abi_version": (_i, []),
    "mmf_last_error": (C.c_char_p, []),
    "mmf_ctx_create": (_i, [_i, _vp, _i, C.POINTER(_vp)]),
    "mmf_ctx_destroy": (None, [_vp]),
    "mmf_ctx_synchronize": (_i, [_vp]),
    "mmf_ctx_stream": (_vp, [_vp]),
    "mmf_ctx_device_name": (_i, [_vp, C.c_char_p, _sz]),
    "mmf_icp_step": (_i, [_vp, _fp, _fp, _vp, _sz, _vp, _sz, _fp, _fp, _cam, _vp, _sz, _vp, _sz, _f, _f, _i, _i,
                          _fp, _fp, _fp, _vp, _sz]),
    "mmf_compute_rgb_residual": (_i, [_vp, _f, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _f,
                                      _fp, _fp, _i, _i, _ip, _ip, _vp, _sz]),
    "mmf_rgb_step": (_i, [_vp, _vp, _f, _vp, _f, _f, _vp, _sz, _vp, _sz, _f, _i, _i, _fp, _fp]),
    "mmf_so3_step": (_i, [_vp, _vp, _sz, _vp, _sz, _fp, _fp, _fp, _i, _i, _fp, _fp, _fp]),
    "mmf_create_vmap": (_i, [_vp, _cam, _vp, _sz, _i, _i, _vp, _sz, _f]),
    "mmf_create_nmap": (_i, [_vp, _vp, _sz, _i, _i, _vp, _sz]),
    "mmf_transform_maps": (_i, [_vp, _vp, _vp, _sz, _i, _i, _fp, _fp, _vp, _vp, _sz]),
    "mmf_copy_maps": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz]),
    "mmf_resize_vmap": (_i, [_vp, _vp, _sz, _i, _i, _vp, _sz]),
    "mmf_resize_nmap": (_i, [_vp, _vp, _sz, _i, _i, _vp, _sz]),
    "mmf_image_bgr_to_intensity": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp, _sz]),
    "mmf_vertices_to_depth": (_i, [_vp, _vp, _i, _i, _f, _vp, _sz]),
    "mmf_project_to_point_cloud": (_i, [_vp, _vp, _sz, _i, _i, _cam, _i, _vp]),
    "mmf_pyr_down_gauss_f": (_i, [_vp, _vp, _sz, _i, _i, _vp, _sz]),
    "mmf_pyr_down_uchar_gauss": (_i, [_vp, _vp, _sz, _i, _i, _vp, _sz]),
    "mmf_compute_derivative_images": (_i, [_vp, _vp, _sz, _i, _i, _vp, _sz, _vp, _sz]),
    "mmf_odom_create": (_i, [_vp, _i, _i, _f, _f, _f, _f, _f, _f, C.POINTER(_vp)]),
    "mmf_odom_destroy": (None, [_vp]),
    "mmf_odom_build_depth_pyramid": (_i, [_vp, _vp, _sz]),
    "mmf_odom_init_icp": (_i, [_vp, C.POINTER(_vp), C.POINTER(_sz), _f]),
    "mmf_odom_init_icp_from_prediction": (_i, [_vp, _vp, _vp, _f]),
    "mmf_odom_init_icp_model": (_i, [_vp, _vp, _vp, _f, _fp]),
    "mmf_odom_init_rgb": (_i, [_vp, _vp, _sz, _i]),
    "mmf_odom_init_rgb_model": (_i, [_vp, _vp, _sz, _i]),
    "mmf_odom_init_first_rgb": (_i, [_vp, _vp, _sz, _i]),
    "mmf_odom_get_incremental_transformation": (_i, [_vp, _fp, _fp, _i, _f, _i, _i, _i, _vp, _vp]),
    "mmf_odom_get_stats": (_i, [_vp, C.POINTER(mmf_odom_stats)]),
    "mmf_odom_get_covariance": (_i, [_vp, C.POINTER(C.c_double)]),
    "mmf_odom_buffer": (_i, [_vp, C.c_char_p, _i, C.POINTER(_vp), C.POINTER(_sz)]),
    "mmf_odom_download": (_i, [_vp, C.c_char_p, _i, _vp, _sz]),
    "mmf_odom_time_icp_kernel": (_i, [_vp, _i, _i, _i, _fp]),
    "mmf_model_create": (_i, [_vp, _i, _i, _f, _f, _f, _f, C.c_ubyte, _f, _i, C.POINTER(_vp)]),
    "mmf_model_destroy": (None, [_vp]),
    "mmf_model_set_pose": (_i, [_vp, _fp]),
    "mmf_model_get_pose": (_i, [_vp, _fp]),
    "mmf_model_count": (_i, [_vp, C.POINTER(C.c_uint)]),
    "mmf_filter_depth": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "mmf_model_initialise": (_i, [_vp, _vp, _vp, _vp, _i, _f]),
    "mmf_model_predict_indices": (_i, [_vp, _i, _f, _i]),
    "mmf_model_combined_predict": (_i, [_vp, _f, _i, _i, _i]),
    "mmf_model_synthesize_depth": (_i, [_vp, _f, _f, _i, _i, _i]),
    "mmf_match_descriptors": (_i, [_vp, _vp, _i, _vp, _i, _i, _f, _vp, _vp]),
    "mmf_superpoint_create": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_vp)]),
    "mmf_superpoint_destroy": (None, [_vp]),
    "mmf_superpoint_forward": (_i, [_vp, _vp, _i, _i, _i]),
    "mmf_superpoint_download": (_i, [_vp, _i, _vp, _sz]),
    "mmf_superpoint_get_features": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _ip]),
    "mmf_superpoint_conv": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mmf_slic_downsample": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp]),
    "mmf_slic_downsample_rgb": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "mmf_slic_upsample_u8": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp]),
    "mmf_rigid_fit": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mmf_rigid_apply": (_i, [_vp, _vp, _vp, _i, _vp]),
    "mmf_ransac_create": (_i, [_i, _f, _f, _vp]),
    "mmf_ransac_destroy": (None, [_vp]),
    "mmf_ransac_estimate": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "mmf_model_fuse": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _f, _f]),
    "mmf_model_clean": (_i, [_vp, _i, _i, _f, _vp, _vp, _f]),
    "mmf_model_perform_fill_in": (_i, [_vp, _vp, _vp, _i, _i]),
    "mmf_model_requires_fill_in": (_i, [_vp, _f, _ip]),
    "mmf_model_download_map": (_i, [_vp, _fp, C.c_uint, C.POINTER(C.c_uint)]),
    "mmf_model_upload_map": (_i, [_vp, _fp, C.c_uint]),
    "mmf_model_texture": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)]),
    "mmf_fusion_default_config": (_i, [C.POINTER(mmf_fusion_config)]),
    "mmf_fusion_create": (_i, [_vp, _i, _i, _f, _f, _f, _f, C.POINTER(mmf_fusion_config), C.POINTER(_vp)]),
    "mmf_fusion_destroy": (None, [_vp]),
    "mmf_fusion_process_frame": (_i, [_vp, _vp, _vp, C.c_longlong, _fp, _f, _i]),
    "mmf_fusion_process_frame_init": (_i, [_vp, _vp, _vp, C.c_longlong, _fp, _i, _f]),
    "mmf_fusion_prefetch_frame": (_i, [_vp, _vp, _vp]),
    "mmf_fusion_reset": (_i, [_vp]),
    "mmf_fusion_get_pose": (_i, [_vp, _fp]),
    "mmf_fusion_tick": (_i, [_vp]),
    "mmf_fusion_model": (_vp, [_vp]),
    "mmf_fusion_odometry": (_vp, [_vp]),
    "mmf_fusion_depth_filtered": (_vp, [_vp]),
    "mmf_fusion_process_frame_ex": (_i, [_vp, C.POINTER(mmf_frame)]),
    "mmf_fusion_process_frame_host": (_i, [_vp, _vp, _vp, _vp, _i, C.c_longlong, _fp, _f, _i]),
    "mmf_fusion_process_frame_host_next": (_i, [_vp, _vp, _vp, _vp, _i, C.c_longlong, _fp, _f, _i, _vp, _vp]),
    "mmf_fusion_predict": (_i, [_vp]),
    "mmf_fusion_set_tick": (_i, [_vp, _i]),
    "mmf_fusion_num_models": (_i, [_vp]),
    "mmf_fusion_model_at": (_vp, [_vp, _i]),
    "mmf_fusion_odometry_at": (_vp, [_vp, _i]),
    "mmf_fusion_num_inactive_models": (_i, [_vp]),
    "mmf_fusion_inactive_model_at": (_vp, [_vp, _i]),
    "mmf_fusion_next_model_id": (_i, [_vp]),
    "mmf_fusion_schedule_deactivation": (_i, [_vp, _i]),
    "mmf_fusion_error_texture": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "mmf_fusion_texture": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_sz)]),
    "mmf_fusion_set_rgb_only": (_i, [_vp, _i]),
    "mmf_fusion_set_icp_weight": (_i, [_vp, _f]),
    "mmf_fusion_set_outlier_coefficient": (_i, [_vp, _f]),
    "mmf_fusion_set_pyramid": (_i, [_vp, _i]),
    "mmf_fusion_set_fast_odom": (_i, [_vp, _i]),
    "mmf_fusion_set_so3": (_i, [_vp, _i]),
    "mmf_fusion_set_frame_to_frame_rgb": (_i, [_vp, _i]),
    "mmf_fusion_set_depth_cutoff": (_i, [_vp, _f]),
    "mmf_fusion_set_confidence_threshold": (_i, [_vp, _f]),
    "mmf_fusion_set_enable_multiple_models": (_i, [_vp, _i]),
    "mmf_fusion_get_config": (_i, [_vp, C.POINTER(mmf_fusion_config)]),
    "mmf_fusion_set_segmentation_callback": (_i, [_vp, _vp, _vp]),
    "mmf_fusion_export_poses": (_i, [_vp, C.c_char_p]),
    "mmf_fusion_pose_log": (_i, [_vp, _i, C.POINTER(C.c_longlong), _fp, _i, _ip]),
    "mmf_compute_fusion_weight": (_i, [_fp, _fp, _f, _fp]),
    "mmf_fusion_preallocate_models": (_i, [_vp, C.c_uint]),
    "mmf_shard_unique_id": (_i, [C.c_char_p]),
    "mmf_shard_create": (_i, [_vp, _i, _i, C.c_char_p, C.POINTER(_vp)]),
    "mmf_shard_attach": (_i, [_vp, _i, _i, _vp, C.POINTER(_vp)]),
    "mmf_shard_destroy": (None, [_vp]),
    "mmf_shard_broadcast_frame": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i]),
    "mmf_shard_post_frame": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "mmf_shard_wait_frame": (_i, [_vp, _i]),
    "mmf_shard_gather_poses": (_i, [_vp, _vp]),
    "mmf_shard_gather_poses_begin": (_i, [_vp, _vp]),
    "mmf_shard_gather_poses_end": (_i, [_vp, _vp]),
    "mmf_shard_gather_maps": (_i, [_vp, _vp, _vp, _i, _vp]),
    "mmf_debug_expf": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mmf_model_surfel_arrays": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_uint)]),
    "mmf_debug_set_gn_fused": (_i, [_i]),
    "mmf_debug_set_mid_predict": (_i, [_i]),
    "mmf_debug_force_gn_fault": (_i, [_i]),
    "mmf_debug_set_sparse_check": (_i, [_i]),
    "mmf_debug_set_pass_batch": (_i, [_i]),
    "mmf_debug_set_prep_rect": (_i, [_i]),
    "mmf_debug_set_xcd": (_i, [_i]),
    "mmf_debug_xcd_block": (C.c_uint, [C.c_uint, C.c_uint]),
    "mmf_debug_set_begin_rider": (_i, [_i]),
    "mmf_debug_begin_rider_count": (_i, []),
    "mmf_debug_set_sparse_groups": (_i, [_i]),
    "mmf_debug_odom_sparse_outside": (_i, [_vp, C.POINTER(C.c_uint), C.POINTER(_i), C.POINTER(_i)]),
    "mmf_gn_chain_status": (_i, [C.POINTER(_i), C.POINTER(_i)]),
    "mmf_debug_set_splat_bound": (_i, [_i]),
    "mmf_debug_set_track_cull": (_i, [_i]),
    "mmf_debug_depth_keys": (_i, [_vp, _vp, _i, _f, _vp, _vp]),
    "mmf_fusion_set_shard": (_i, [_vp, _i, _i]),
    "mmf_fusion_owns_model": (_i, [_vp, _i]),
    "mmf_fusion_set_model_pose": (_i, [_vp, _i, _fp]),
    "mmf_fusion_last_timings": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mmf_odom_enable_timing": (_i, [_vp, _i]),
    "mmf_odom_get_timing": (_i, [_vp, C.POINTER(mmf_odom_timing)]),
    "mmf_model_set_max_depth": (_i, [_vp, _f]),
    "mmf_model_set_confidence_threshold": (_i, [_vp, _f]),
    "mmf_model_confidence_threshold": (_f, [_vp]),
    "mmf_model_id": (_i, [_vp]),
}

_lib = None


def load():
    """Load libmmf_hip.so (once).  Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m multimotionfusion_amd.build` "
            "(hipcc, gfx950). multimotionfusion_amd has no CPU fallback.")
    # torch bundles its own libamdhip64; it must be the FIRST HIP runtime mapped into the process
    # (ours then binds to the same SONAME).  Two runtimes in one process do not see the device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        if os.environ.get("MMF_HIP_LIB") and name.startswith("mmf_debug_") and not hasattr(lib, name):
            continue  # (an older build of the library in a same-box A/B, tools/ab_libs.sh: it lacks the newer test hooks only)
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise MmfError(status, load().mmf_last_error().decode("utf-8", "replace"))


def fptr(arr):
    """float32 numpy array -> POINTER(c_float) (no copy; caller keeps the array alive)."""
    return arr.ctypes.data_as(_fp)
