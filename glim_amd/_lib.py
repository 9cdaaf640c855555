"""ctypes loader for libglim_amd.so (the C-ABI shared library declared in include/glim_amd.h).

The product path fails loudly when the HIP extension is missing: there is NO CPU fallback anywhere in this package.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLIM_AMD_LIB") or os.path.join(_HERE, "libglim_amd.so")  # GLIM_AMD_LIB: profiling builds only
CSRC = os.path.join(_HERE, "csrc")

COMPACT_DOUBLES = 29
FACTOR_BINARY = 0x1
FACTOR_SURFACE_VALIDATION = 0x2

ERR = {
    0: "ok", -1: "invalid argument", -2: "HIP runtime error", -3: "no HIP device", -4: "voxel coordinate out of key range",
    -5: "invalid state for this call", -6: "unsupported", -7: "out of memory",
}


class Linearized6(C.Structure):
    _fields_ = [
        ("num_inliers", C.c_int64),
        ("error", C.c_double),
        ("H_tt", C.c_double * 36),
        ("H_ss", C.c_double * 36),
        ("H_ts", C.c_double * 36),
        ("b_t", C.c_double * 6),
        ("b_s", C.c_double * 6),
    ]


class PreprocessParams(C.Structure):
    """glim_amd_preprocess_params (include/glim_amd.h) == CloudPreprocessorParams (cloud_preprocessor.cpp:20-61)."""

    _fields_ = [
        ("distance_near_thresh", C.c_double),
        ("distance_far_thresh", C.c_double),
        ("use_random_grid_downsampling", C.c_int32),
        ("downsample_target", C.c_int32),
        ("downsample_resolution", C.c_double),
        ("downsample_rate", C.c_double),
        ("global_shutter", C.c_int32),
        ("enable_outlier_removal", C.c_int32),
        ("outlier_removal_k", C.c_int32),
        ("outlier_std_mul_factor", C.c_double),
        ("enable_cropbox_filter", C.c_int32),
        ("crop_bbox_frame_imu", C.c_int32),
        ("crop_bbox_min", C.c_double * 3),
        ("crop_bbox_max", C.c_double * 3),
        ("T_imu_lidar", C.c_double * 12),
        ("k_correspondences", C.c_int32),
        ("voxelgrid_block_size", C.c_int32),
        ("seed", C.c_uint64),
    ]


class GlimAmdError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {ERR.get(code, 'unknown error')} ({code}){(' -- ' + detail) if detail else ''}")


def build(force=False):
    """Compile every HIP source for gfx950 with hipcc (glim_amd/csrc/Makefile) into glim_amd/libglim_amd.so."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH


# every symbol include/glim_amd.h declares: name -> (restype, argtypes)
_vp, _i, _i32, _i64, _u32, _d, _sz = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_uint32, C.c_double, C.c_size_t
_dp, _fp, _ip, _lp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
_pp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "glim_amd_version": (_i, []),
    "glim_amd_error_string": (C.c_char_p, [_i]),
    "glim_amd_last_hip_error": (C.c_char_p, []),
    "glim_amd_device_count": (_i, []),
    "glim_amd_ctx_create": (_i, [_i, _i, _vp, _pp]),
    "glim_amd_ctx_create_ex": (_i, [_i, _i, _vp, _i, _pp]),
    "glim_amd_ctx_destroy": (_i, [_vp]),
    "glim_amd_ctx_synchronize": (_i, [_vp]),
    "glim_amd_ctx_set_diag": (_i, [_vp, C.c_char_p]),
    "glim_amd_ctx_get_diag": (_i, [_vp, C.c_char_p, _sz]),
    "glim_amd_device_info": (_i, [_vp, C.c_char_p, _sz, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i)]),
    "glim_amd_cloud_create": (_i, [_vp, _i64, _dp, _dp, _dp, _pp]),
    "glim_amd_cloud_create_f32": (_i, [_vp, _i64, _fp, _fp, _fp, _pp]),
    "glim_amd_cloud_create_deskewed": (_i, [_vp, _i64, _dp, _dp, _dp, _i32, _dp, _dp, _d, _dp, _dp, _i32, _pp]),
    "glim_amd_preprocess_default_params": (_i, [C.POINTER(PreprocessParams)]),
    "glim_amd_preprocess": (_i, [_vp, _i64, _dp, _dp, _dp, C.POINTER(PreprocessParams), _pp]),
    "glim_amd_cloud_download_frame": (_i, [_vp, _dp, _dp, _dp, _ip]),
    "glim_amd_factor_set_profile_fresh_samples": (_i, [_vp, _i32, _pp, _pp, C.POINTER(C.c_uint32), _dp, _i, _d, _fp]),
    "glim_amd_factor_set_trip_stats": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _i]),
    "glim_amd_multi_last_timing": (_i, [_vp, _fp, _fp]),
    "glim_amd_multi_last_breakdown": (_i, [_vp, _i32, _dp, _i32]),
    "glim_amd_debug_plan_stats": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _ip]),
    "glim_amd_debug_ctx_query_streams": (_i, [_vp, _ip]),
    "glim_amd_debug_pool_stats": (_i, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "glim_amd_debug_frame_stages": (_i, [_dp, _i32]),
    "glim_amd_multi_set_one_rank_collective": (_i, [_vp, _i32]),
    "glim_amd_multi_set_host_records": (_i, [_vp, _i32]),
    "glim_amd_factor_set_linearize_repeat": (_i, [_vp, _dp, _i, _i, _vp]),
    "glim_amd_multi_set_split": (_i, [_vp, _i32]),
    "glim_amd_multi_records": (_i, [_vp, _i64, _i64, _dp]),
    "glim_amd_voxelmap_set_lru_horizon": (_i, [_vp, _i32, _i32]),
    "glim_amd_cloud_create_exact": (_i, [_vp, _i64, _dp, _pp]),
    "glim_amd_frame_create": (_i, [_vp, _i64, _dp, _dp, _dp, _i32, _dp, _pp, _pp]),
    "glim_amd_shard_layout": (_i, [_lp, _i32, _i32, _lp, _lp, _ip, _lp]),
    "glim_amd_debug_resident_stop": (_i, [_i]),
    "glim_amd_debug_resident_stats": (_i, [_i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _ip]),
    "glim_amd_debug_deskew_table": (_i, [_i64, _dp, _dp, _i32, _dp, _dp, _d, _dp, _dp, _ip, _dp, _i32, _ip]),
    "glim_amd_cloud_deskew": (_i, [_vp, _dp, _i32, _dp, _dp, _d, _dp, _dp, _i32, _pp]),
    "glim_amd_cloud_save_compact": (_i, [_vp, C.c_char_p]),
    "glim_amd_cloud_load_compact": (_i, [_vp, C.c_char_p, _pp]),
    "glim_amd_nn_index_create": (_i, [_vp, _d, _pp]),
    "glim_amd_nn_index_destroy": (_i, [_vp]),
    "glim_amd_gicp_linearize": (_i, [_vp, _vp, _dp, _d, _u32, C.POINTER(Linearized6)]),
    "glim_amd_gicp_error": (_i, [_vp, _vp, _dp, _d, _dp, _lp]),
    "glim_amd_gicp_correspondences": (_i, [_vp, _vp, _dp, _d, _ip]),
    "glim_amd_merge_frames": (_i, [_vp, _i32, _dp, C.POINTER(_dp), C.POINTER(_dp), _lp, _d, _i32, _i32, C.c_uint64, _pp]),
    "glim_amd_cloud_download_merged": (_i, [_vp, _dp, _dp]),
    "glim_amd_debug_sort_pairs": (_i, [_vp, _i64, _i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "glim_amd_cloud_destroy": (_i, [_vp]),
    "glim_amd_cloud_size": (_i, [_vp, _lp]),
    "glim_amd_cloud_memory_usage": (_i, [_vp, C.POINTER(_sz)]),
    "glim_amd_cloud_download": (_i, [_vp, _fp, _fp, _fp, _ip]),
    "glim_amd_cloud_find_neighbors": (_i, [_vp, _i, _ip]),
    "glim_amd_cloud_set_neighbors": (_i, [_vp, _i, _ip]),
    "glim_amd_cloud_estimate_covariances": (_i, [_vp, _i]),
    "glim_amd_voxelmap_create": (_i, [_vp, _d, _i, _i, _d, _pp]),
    "glim_amd_voxelmap_insert": (_i, [_vp, _vp]),
    "glim_amd_voxelmap_destroy": (_i, [_vp]),
    "glim_amd_voxelmap_info": (_i, [_vp, _ip, _ip, _dp, C.POINTER(_sz)]),
    "glim_amd_voxelmap_download": (_i, [_vp, _ip, _ip, _fp, _fp]),
    "glim_amd_factor_set_create": (_i, [_vp, _pp]),
    "glim_amd_factor_set_destroy": (_i, [_vp]),
    "glim_amd_factor_set_add": (_i, [_vp, _vp, _vp, _u32, _ip]),
    "glim_amd_factor_set_clear": (_i, [_vp]),
    "glim_amd_factor_set_size": (_i, [_vp, _ip]),
    "glim_amd_factor_set_linearize": (_i, [_vp, _dp, C.POINTER(Linearized6)]),
    "glim_amd_factor_set_error": (_i, [_vp, _dp, _dp, _dp, _lp]),
    "glim_amd_factor_set_correspondences": (_i, [_vp, _i32, _dp, _ip]),
    "glim_amd_factor_set_linearize_device_async": (_i, [_vp, _dp, _vp, _i64]),
    "glim_amd_expand_compact": (_i, [_dp, _dp, _u32, C.POINTER(Linearized6)]),
    "glim_amd_factor_set_profile": (_i, [_vp, _dp, _i, _fp, _fp]),
    "glim_amd_factor_set_profile_sync": (_i, [_vp, _dp, _i, _fp]),
    "glim_amd_factor_set_profile_lm": (_i, [_vp, _dp, _i, _fp, _fp]),
    "glim_amd_overlap": (_i, [_vp, _i32, _pp, _dp, _vp, _dp]),
    "glim_amd_overlap_batch": (_i, [_vp, _i32, _ip, _pp, _dp, _pp, _dp]),
    "glim_amd_overlap_profile": (_i, [_vp, _i32, _ip, _pp, _dp, _pp, _i, _fp]),
    "glim_amd_factor_set_profile_fresh": (_i, [_vp, _i32, _pp, _pp, C.POINTER(C.c_uint32), _dp, _i, _fp]),
    "glim_amd_multi_create": (_i, [_ip, _i32, _pp]),
    "glim_amd_multi_destroy": (_i, [_vp]),
    "glim_amd_multi_info": (_i, [_vp, _ip, _ip, _lp]),
    "glim_amd_multi_add_cloud": (_i, [_vp, _i64, _dp, _dp, _dp, _ip]),
    "glim_amd_multi_add_cloud_f32": (_i, [_vp, _i64, _fp, _fp, _fp, _ip]),
    "glim_amd_multi_cloud_estimate_covariances": (_i, [_vp, _i32, _i]),
    "glim_amd_multi_add_voxelmap": (_i, [_vp, _i32, _d, _ip]),
    "glim_amd_multi_set_factors": (_i, [_vp, _i64, _ip, _ip, C.POINTER(C.c_uint32)]),
    "glim_amd_multi_shard": (_i, [_vp, _lp]),
    "glim_amd_multi_linearize": (_i, [_vp, _dp, C.POINTER(Linearized6), _dp]),
    "glim_amd_multi_profile": (_i, [_vp, _dp, _i, _fp]),
    "glim_amd_shard_bounds": (_i, [_dp, _i64, _i32, _lp]),
    "glim_amd_debug_scratch_poke": (_i, [_vp, _i32, _u32]),
    "glim_amd_debug_resident_timeline": (_i, [_i, _i, _dp, _i32]),
    "glim_amd_cloud_profile_neighbors": (_i, [_vp, _i, _i, _fp, _fp]),
    "glim_amd_factor_set_cull_stats": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _i]),
    "glim_amd_multi_set_gather_mode": (_i, [_vp, _i32]),
    "glim_amd_multi_wait_gather": (_i, [_vp]),
    "glim_amd_multi_gathered_device": (_i, [_vp, _i32, _pp, _lp]),
    "glim_amd_debug_multi_create_virtual": (_i, [_ip, _i32, _pp]),
    "glim_amd_debug_multi_inject_failure": (_i, [_vp, _i32, _i32]),
    "glim_amd_debug_multi_set_diag": (_i, [_vp, C.c_char_p]),
    "glim_amd_debug_multi_gathered_download": (_i, [_vp, _i32, _i64, _i64, _dp]),
}

_lib = None


def lib():
    """Load libglim_amd.so; raises (never falls back) when it is missing or lacks a declared symbol."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C glim_amd/csrc).  glim_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared entry point
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code, where):
    if code != 0:
        detail = lib().glim_amd_last_hip_error().decode() if code == -2 else ""
        raise GlimAmdError(code, where, detail)
