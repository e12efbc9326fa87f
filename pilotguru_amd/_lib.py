"""ctypes binding of libpgorb.so (the C ABI declared in include/pgorb.h).

The HIP library is the product: importing this module without a built
pilotguru_amd/libpgorb.so raises immediately -- there is no Python or CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGORB_LIBRARY: a developer build of the same ABI (e.g. a timing build made with make EXTRA=...), for the scripts under tools/experiments
LIB_PATH = os.environ.get("PGORB_LIBRARY") or os.path.join(_HERE, "libpgorb.so")

PGORB_MAX_LEVELS = 16
PGORB_OK, PGORB_E_ARG, PGORB_E_TOOSMALL, PGORB_E_CAP = 0, -1, -2, -3
PGORB_E_NODEVICE, PGORB_E_HIP, PGORB_E_LIMIT, PGORB_E_OVERFLOW = -4, -5, -6, -7

# every symbol include/pgorb.h declares (tests check the .so exports all of them)
SYMBOLS = (
    "pgorb_create", "pgorb_destroy", "pgorb_last_error", "pgorb_levels", "pgorb_scale_tables",
    "pgorb_features_per_level", "pgorb_max_keypoints", "pgorb_extract", "pgorb_extract_batch",
    "pgorb_extract_batch_device", "pgorb_check_async", "pgorb_descriptor_distance",
    "pgorb_hamming_matrix", "pgorb_hamming_best2", "pgorb_match_batch_device",
    "pgorb_debug_level_size", "pgorb_debug_level_image", "pgorb_debug_level_candidates",
    "pgorb_debug_level_keypoints", "pgorb_debug_sincos_checksum", "pgorb_profile_begin", "pgorb_profile_read", "pgorb_profile_host",
    "pgorb_vocab_load_text", "pgorb_vocab_load_cached", "pgorb_vocab_from_blob", "pgorb_vocab_blob", "pgorb_vocab_info",
    "pgorb_vocab_free", "pgorb_vocab_upload", "pgorb_vocab_upload_device", "pgorb_bow_transform",
    "pgorb_bow_transform_device", "pgorb_bow_vectors", "pgorb_bow_score_l1",
    "pgorb_frame_grid", "pgorb_frame_grid_batch_device", "pgorb_search_for_initialization",
    "pgorb_search_for_initialization_batch_device", "pgorb_extract_batch_color_device",
    "pgorb_extract_batch_ingest_device",
    "pgorb_search_by_projection_points", "pgorb_search_by_projection_frame", "pgorb_search_by_bow",
    "pgorb_search_by_projection_points_batch_device", "pgorb_search_by_projection_frame_batch_device",
    "pgorb_feature_vectors_batch_device", "pgorb_search_by_bow_batch_device",
    "pgorb_search_by_projection_keyframe", "pgorb_search_by_projection_keyframe_batch_device",
    "pgorb_log_f", "pgorb_log_scale_factor", "pgorb_predict_scale",
    "pgorb_undistort_keypoints", "pgorb_undistort_keypoints_batch_device", "pgorb_image_bounds",
    "pgorb_host_alloc", "pgorb_host_free", "pgorb_host_register", "pgorb_host_unregister", "pgorb_stream_create", "pgorb_stream_create_ingest", "pgorb_stream_create_device", "pgorb_stream_submit_device", "pgorb_stream_wait_device", "pgorb_stream_lanes", "pgorb_stream_destroy", "pgorb_stream_input", "pgorb_stream_reset", "pgorb_stream_submit",
    "pgorb_stream_wait", "pgorb_stream_frontend", "pgorb_stream_frontend_results", "pgorb_set_option", "pgorb_get_option", "pgorb_matcher_is_popcount",
    "pgorb_smooth_heading_directions", "pgorb_smooth_time_series", "pgorb_trajectory_pca",
    "pgorb_project_directions", "pgorb_project_translations", "pgorb_turn_angles",
    "pgorb_principal_rotation_axes", "pgorb_angular_velocities_around_axis",
    "pgorb_comm_library", "pgorb_comm_unique_id", "pgorb_comm_create_local", "pgorb_comm_create_rank", "pgorb_comm_ranks", "pgorb_vocab_broadcast", "pgorb_comm_destroy",
    "pgorb_kahan_sum", "pgorb_fit_num_windows", "pgorb_fit_velocity_windows", "pgorb_calibrator_eval", "pgorb_fit_motion_velocities",
)


class PgorbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
                ("blur_tie_mode", C.c_int32)]


class PgorbError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("pgorb error %d: %s" % (code, text))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C pilotguru_amd/csrc` "
                          "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (+ HSA).  If
    # libpgorb.so pulled in /opt/rocm's copy first, torch would later load a SECOND runtime and
    # see no GPUs.  Importing torch first makes the loader satisfy our DT_NEEDED
    # libamdhip64.so.7 with the already-loaded copy (same SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32p, fp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)
    L.pgorb_create.restype = C.c_int
    L.pgorb_create.argtypes = [C.POINTER(PgorbParams), C.POINTER(vp)]
    L.pgorb_destroy.restype = None
    L.pgorb_destroy.argtypes = [vp]
    L.pgorb_last_error.restype = C.c_char_p
    L.pgorb_last_error.argtypes = [vp]
    L.pgorb_levels.argtypes = [vp]
    L.pgorb_scale_tables.argtypes = [vp, fp, fp, fp, fp]
    L.pgorb_features_per_level.argtypes = [vp, i32p]
    L.pgorb_max_keypoints.argtypes = [vp, C.c_int, C.c_int]
    L.pgorb_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, i32p]
    L.pgorb_extract_batch.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int,
                                      vp, vp, C.c_int, i32p]
    L.pgorb_extract_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                             vp, vp, C.c_int, vp, vp]
    L.pgorb_check_async.argtypes = [vp, vp]
    L.pgorb_extract_batch_color_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                                   vp, vp, C.c_int, vp, vp]
    L.pgorb_extract_batch_ingest_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                                    C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]
    L.pgorb_descriptor_distance.argtypes = [vp, vp]
    L.pgorb_hamming_matrix.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.pgorb_hamming_best2.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, vp, vp]
    L.pgorb_match_batch_device.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp]
    L.pgorb_debug_level_size.argtypes = [vp, C.c_int, i32p, i32p]
    L.pgorb_debug_level_image.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pgorb_debug_level_candidates.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int]
    L.pgorb_debug_level_keypoints.argtypes = [vp, C.c_int, C.c_int]
    L.pgorb_stream_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.pgorb_stream_create_ingest.argtypes = [vp] + [C.c_int] * 9 + [C.POINTER(vp)]
    L.pgorb_stream_create_device.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.pgorb_stream_submit_device.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int64, vp]
    L.pgorb_stream_wait_device.argtypes = [vp, C.c_int, C.c_int, vp] + [C.POINTER(vp)] * 6 + [i32p]
    L.pgorb_stream_lanes.argtypes = [vp]
    L.pgorb_stream_destroy.restype = None
    L.pgorb_stream_destroy.argtypes = [vp]
    L.pgorb_stream_input.restype = vp
    L.pgorb_stream_input.argtypes = [vp, C.c_int]
    L.pgorb_stream_reset.argtypes = [vp]
    L.pgorb_stream_submit.argtypes = [vp, C.c_int, C.c_int]
    L.pgorb_stream_wait.argtypes = [vp, C.c_int] + [C.POINTER(vp)] * 6 + [i32p]
    L.pgorb_stream_frontend.argtypes = [vp] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_int, C.c_int]
    L.pgorb_stream_frontend_results.argtypes = [vp, C.c_int] + [C.POINTER(vp)] * 5
    L.pgorb_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.pgorb_get_option.argtypes = [vp, C.c_char_p]
    L.pgorb_matcher_is_popcount.argtypes = [vp, C.c_int]
    L.pgorb_profile_begin.argtypes = [vp, C.c_int]
    L.pgorb_profile_read.argtypes = [vp, C.POINTER(C.c_double)]
    L.pgorb_profile_host.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
    f4 = [C.c_float] * 4
    L.pgorb_frame_grid.argtypes = [vp, vp, C.c_int] + f4 + [vp, vp]
    L.pgorb_frame_grid_batch_device.argtypes = [vp, vp, vp, C.c_int, C.c_int] + f4 + [vp, vp, vp]
    L.pgorb_search_for_initialization.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int] + f4 + [vp, vp, C.c_int, C.c_float, C.c_int]
    L.pgorb_search_for_initialization_batch_device.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int] + f4 + \
        [vp, vp, vp, C.c_int, C.c_float, C.c_int, vp]
    L.pgorb_search_by_projection_points.argtypes = [vp, vp, vp, C.c_int] + f4 + [vp, C.c_int] + [vp] * 7 + [C.c_float, C.c_float, vp]
    L.pgorb_search_by_projection_frame.argtypes = [vp, vp, vp, C.c_int] + f4 + [vp, C.c_int] + [vp] * 7 + [C.c_float, C.c_int, vp]
    L.pgorb_search_by_bow.argtypes = [vp] + [vp] * 3 + [C.c_int] + [vp] * 3 + [C.c_int] + [vp] * 2 + [C.c_int] + [vp] * 3 + \
        [C.c_int, C.c_float, C.c_int, vp]
    # (ctx, kps, desc, n, cap, grid_start, grid_idx, pair_frame, npairs, bounds x 4, kp_has_point, qcap, nq, 7 query arrays, th, ratio | check, assigned, nmatches, stream)
    L.pgorb_search_by_projection_points_batch_device.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int] + f4 + [vp, C.c_int, vp] + [vp] * 7 + \
        [C.c_float, C.c_float, vp, vp, vp]
    L.pgorb_search_by_projection_frame_batch_device.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int] + f4 + [vp, C.c_int, vp] + [vp] * 7 + \
        [C.c_float, C.c_int, vp, vp, vp]
    L.pgorb_search_by_projection_keyframe.argtypes = [vp, vp, vp, C.c_int] + f4 + [vp, C.c_int] + [vp] * 9 + [C.c_float, C.c_float, C.c_int, C.c_int, vp]
    L.pgorb_search_by_projection_keyframe_batch_device.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int] + f4 + [vp, C.c_int, vp] + [vp] * 9 + \
        [C.c_float, C.c_float, C.c_int, C.c_int, vp, vp, vp]
    L.pgorb_log_f.argtypes = [C.c_float]
    L.pgorb_log_scale_factor.argtypes = [vp]
    L.pgorb_predict_scale.argtypes = [vp, C.c_float, C.c_float]
    L.pgorb_feature_vectors_batch_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.pgorb_search_by_bow_batch_device.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_float, C.c_int, vp, vp, vp]
    L.pgorb_undistort_keypoints.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.pgorb_undistort_keypoints_batch_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.pgorb_image_bounds.argtypes = [C.c_int, C.c_int, vp, vp, vp]
    L.pgorb_smooth_heading_directions.argtypes = [vp, C.c_int, C.c_int]
    L.pgorb_smooth_time_series.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_double, vp]
    L.pgorb_trajectory_pca.argtypes = [vp, C.c_int, vp, vp, vp]
    L.pgorb_project_directions.argtypes = [vp, C.c_int, vp, vp]
    L.pgorb_project_translations.argtypes = [vp, C.c_int, vp]
    L.pgorb_turn_angles.argtypes = [vp, C.c_int, vp]
    series = [vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int]
    L.pgorb_principal_rotation_axes.argtypes = [vp, vp, C.c_int, C.c_int64, vp]
    L.pgorb_angular_velocities_around_axis.argtypes = [vp, C.c_int, vp, vp]
    L.pgorb_kahan_sum.argtypes = [vp, C.c_int, C.c_int, vp]
    L.pgorb_fit_num_windows.argtypes = [C.c_int, C.c_int]
    L.pgorb_fit_velocity_windows.argtypes = [vp] + series + [C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.pgorb_calibrator_eval.argtypes = [vp] + series + [vp, C.c_int, vp, vp]
    L.pgorb_fit_motion_velocities.argtypes = [vp] + series + [vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                              vp, vp, i32p, vp]
    L.pgorb_host_alloc.restype = vp
    L.pgorb_host_alloc.argtypes = [C.c_int64]
    L.pgorb_host_free.restype = None
    L.pgorb_host_free.argtypes = [vp]
    L.pgorb_host_register.argtypes = [vp, C.c_int64]
    L.pgorb_host_unregister.argtypes = [vp]
    L.pgorb_vocab_load_text.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.pgorb_vocab_load_cached.argtypes = [C.c_char_p, C.POINTER(vp), i32p]
    L.pgorb_vocab_from_blob.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    L.pgorb_vocab_blob.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int64)]
    L.pgorb_vocab_info.argtypes = [vp] + [i32p] * 6
    L.pgorb_vocab_free.restype = None
    L.pgorb_vocab_free.argtypes = [vp]
    L.pgorb_vocab_upload.argtypes = [vp, vp]
    L.pgorb_vocab_upload_device.argtypes = [vp, vp, C.c_int64, vp]
    L.pgorb_bow_transform.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp]
    L.pgorb_bow_transform_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.pgorb_bow_vectors.argtypes = [C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, i32p, vp, vp, vp, i32p]
    L.pgorb_comm_library.argtypes = [C.c_char_p, C.c_int, i32p]
    L.pgorb_comm_unique_id.argtypes = [vp]
    L.pgorb_comm_create_local.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(vp)]
    L.pgorb_comm_create_rank.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.pgorb_comm_ranks.argtypes = [vp]
    L.pgorb_vocab_broadcast.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_double)]
    L.pgorb_comm_destroy.restype = None
    L.pgorb_comm_destroy.argtypes = [vp]
    L.pgorb_bow_score_l1.restype = C.c_double
    L.pgorb_bow_score_l1.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
    for name in SYMBOLS:
        if name not in ("pgorb_destroy", "pgorb_last_error", "pgorb_vocab_free", "pgorb_bow_score_l1",
                        "pgorb_host_alloc", "pgorb_host_free", "pgorb_stream_destroy", "pgorb_stream_input", "pgorb_log_f", "pgorb_log_scale_factor",
                        "pgorb_comm_destroy"):
            getattr(L, name).restype = C.c_int
    L.pgorb_log_f.restype = C.c_float
    L.pgorb_log_scale_factor.restype = C.c_float
    _lib = L
    return L
