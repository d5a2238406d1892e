"""ctypes view of include/lvi_hotpath.h.

One binding, parameterised by the shared-library path: the product library
(``liblvi_hip.so``) and — from tests/bench only — the CPU oracle export the same
symbols, so parity tests drive both through this module.  Nothing in this
package knows where the oracle lives.
"""
import ctypes as C
import os

import numpy as np

# ---- status codes -----------------------------------------------------------
LVI_OK = 0
LVI_ERR_INVALID_ARG = -1
LVI_ERR_NO_DEVICE = -2
LVI_ERR_HIP = -3
LVI_ERR_CAPACITY = -4
LVI_ERR_STATE = -5
LVI_ERR_UNSUPPORTED = -6
LVI_TOO_FEW_FEATURES = 1
LVI_TOO_FEW_CORRESPONDENCES = 2
LVI_NO_MAP = 3
LVI_ICP_MAX_ITERS = 64

# lvi_debug_get items
DBG_CURVATURE, DBG_PICKED_OCCL, DBG_LABEL, DBG_PICKED_FINAL, DBG_CORNER_INDEX = 1, 2, 3, 4, 5
DBG_VOXEL_KEYS, DBG_VOXEL_CELLS, DBG_VOXEL_COUNTS, DBG_ICP_JTJ, DBG_ICP_POSE_TRACE = 6, 7, 8, 9, 10
DBG_FEAT_CYCLES = 11
DBG_MAP_CORNER_RAW = 13
DBG_MAP_SURF_RAW = 14
DBG_ICP_CYCLES = 12
TDBG_PYRAMID_L1, TDBG_PYRAMID_L2, TDBG_PYRAMID_L3, TDBG_MINEIG, TDBG_GFTT_NCAND, TDBG_MASK = 1, 2, 3, 4, 5, 6

PT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4")])
LIVOX_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "u1"), ("tag", "u1"),
                        ("line", "u1"), ("_pad", "u1"), ("offset_time", "<u4")])
assert PT_DTYPE.itemsize == 16 and LIVOX_DTYPE.itemsize == 20


class LidarParams(C.Structure):
    _fields_ = [("N_SCAN", C.c_int32), ("Horizon_SCAN", C.c_int32), ("downsampleRate", C.c_int32),
                ("lidarMinRange", C.c_float), ("lidarMaxRange", C.c_float),
                ("edgeThreshold", C.c_float), ("surfThreshold", C.c_float),
                ("edgeFeatureMinValidNum", C.c_int32), ("surfFeatureMinValidNum", C.c_int32),
                ("odometrySurfLeafSize", C.c_float), ("mappingCornerLeafSize", C.c_float), ("mappingSurfLeafSize", C.c_float),
                ("z_tollerance", C.c_float), ("rotation_tollerance", C.c_float), ("imuRPYWeight", C.c_float),
                ("numberOfCores", C.c_int32), ("icp_max_iters", C.c_int32), ("icp_disable_break", C.c_int32),
                ("max_raw_points", C.c_int32), ("max_map_points", C.c_int32), ("voxel_mode", C.c_int32),
                ("max_keyframes", C.c_int32), ("max_keyframe_points", C.c_int32), ("map_on_main_stream", C.c_int32),
                ("sector_handover_wait_us", C.c_int32), ("batch_scans", C.c_int32), ("map_plan_cache", C.c_int32)]


class ScanInfo(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n", C.c_int32),
                ("start_ring_index", C.POINTER(C.c_int32)), ("end_ring_index", C.POINTER(C.c_int32)),
                ("point_col_ind", C.POINTER(C.c_int32)), ("point_range", C.POINTER(C.c_float)),
                ("cloud_deskewed", C.c_void_p)]


class DeskewInfo(C.Structure):
    _fields_ = [("imu_available", C.c_int32), ("imu_pointer_cur", C.c_int32), ("time_scan_cur", C.c_double),
                ("imu_time", C.POINTER(C.c_double)), ("imu_rot_x", C.POINTER(C.c_double)),
                ("imu_rot_y", C.POINTER(C.c_double)), ("imu_rot_z", C.POINTER(C.c_double))]


class MeiParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("xi", "k1", "k2", "p1", "p2", "gamma1", "gamma2", "u0", "v0")]


class Cloud(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n", C.c_int32), ("pts", C.c_void_p)]


class ImuHint(C.Structure):
    _fields_ = [("imu_available", C.c_int32), ("imu_roll_init", C.c_float), ("imu_pitch_init", C.c_float), ("imu_yaw_init", C.c_float)]


class IcpResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("iters", C.c_int32), ("converged", C.c_int32), ("degenerate", C.c_int32),
                ("n_corner_ds", C.c_int32), ("n_surf_ds", C.c_int32),
                ("n_sel", C.c_int32 * LVI_ICP_MAX_ITERS), ("pose", C.c_float * 6)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double), ("bytes_alg", C.c_double)]


class TrackerParams(C.Structure):
    _fields_ = [("max_width", C.c_int32), ("max_height", C.c_int32), ("max_cnt", C.c_int32), ("min_dist", C.c_double),
                ("lk_win", C.c_int32), ("lk_max_level", C.c_int32), ("lk_max_iters", C.c_int32), ("lk_eps", C.c_double),
                ("lk_min_eig_threshold", C.c_float), ("gftt_quality", C.c_double), ("max_features", C.c_int32)]


_P = C.POINTER
_vp, _i32, _i64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes).  This table is also what tests use to check that the
# product library exports every symbol the header declares.
SIGNATURES = {
    "lvi_abi_version": (_i32, []),
    "lvi_backend": (C.c_char_p, []),
    "lvi_last_error": (C.c_char_p, []),
    "lvi_lidar_params_default": (None, [_P(LidarParams)]),
    "lvi_lidar_create": (_i32, [_P(LidarParams), _i32, _P(_vp)]),
    "lvi_lidar_destroy": (None, [_vp]),
    "lvi_lidar_sync": (_i32, [_vp]),
    "lvi_lidar_mark": (_i32, [_vp, _i32]),
    "lvi_lidar_wait_mark": (_i32, [_vp, _i32]),
    "lvi_organize_scan": (_i32, [_vp, _vp, _i32, _P(ScanInfo)]),
    "lvi_extract_features": (_i32, [_vp, _P(ScanInfo), _P(Cloud), _P(Cloud)]),
    "lvi_voxel_downsample": (_i32, [_vp, _vp, _i32, _f32, _vp, _i32, _P(_i32)]),
    "lvi_map_set": (_i32, [_vp, _vp, _i32, _vp, _i32]),
    "lvi_scan_to_map": (_i32, [_vp, _vp, _i32, _vp, _i32, _P(ImuHint), _P(_f32), _P(IcpResult)]),
    "lvi_transform_cloud": (_i32, [_vp, _vp, _i32, _P(_f32), _vp]),
    "lvi_scan_upload": (_i32, [_vp, _vp, _i32]),
    "lvi_scan_organize": (_i32, [_vp]),
    "lvi_scan_set_deskew": (_i32, [_vp, _vp]),
    "lvi_keyframe_add": (_i32, [_vp, _vp, _i32, _vp, _i32, _P(C.c_float), _P(_i32)]),
    "lvi_keyframe_add_current": (_i32, [_vp, _P(C.c_float), _P(_i32)]),
    "lvi_keyframe_set_pose": (_i32, [_vp, _i32, _P(C.c_float)]),
    "lvi_keyframe_count": (_i32, [_vp, _P(_i32), _P(_i32)]),
    "lvi_keyframes_clear": (_i32, [_vp]),
    "lvi_map_assemble": (_i32, [_vp, _P(_i32), _i32]),
    "lvi_map_update": (_i32, [_vp, _P(_i32), _i32]),
    "lvi_organize_scan_deskew": (_i32, [_vp, _vp, _i32, _vp, _P(ScanInfo)]),
    "lvi_scan_extract": (_i32, [_vp]),
    "lvi_scan_downsample": (_i32, [_vp]),
    "lvi_map_upload": (_i32, [_vp, _vp, _i32, _vp, _i32]),
    "lvi_map_build": (_i32, [_vp]),
    "lvi_scan_match": (_i32, [_vp, _P(ImuHint), _P(_f32), _P(IcpResult)]),
    "lvi_scan_match_async": (_i32, [_vp, _P(_f32), _vp]),
    "lvi_scan_upload_device": (_i32, [_vp, _vp, _i32]),
    "lvi_map_upload_device": (_i32, [_vp, _vp, _i32, _vp, _i32]),
    "lvi_map_share": (_i32, [_vp, _vp]),
    "lvi_scan_replay_enqueue": (_i32, [_vp, _vp, _i32, _P(_f32), _vp, _i32]),
    "lvi_scan_batch_bind_device": (_i32, [_vp, _i32, _P(_vp), _P(_i32)]),
    "lvi_scan_batch_upload": (_i32, [_vp, _i32, _P(_vp), _P(_i32)]),
    "lvi_scan_batch_run": (_i32, [_vp, _i32, _P(_f32), _vp, _i32]),
    "lvi_scan_batch_get_records": (_i32, [_vp, _i32, _vp]),
    "lvi_batch_select": (_i32, [_vp, _i32]),
    "lvi_get_scan_info": (_i32, [_vp, _P(ScanInfo)]),
    "lvi_get_features": (_i32, [_vp, _P(Cloud), _P(Cloud)]),
    "lvi_get_scan_ds": (_i32, [_vp, _P(Cloud), _P(Cloud)]),
    "lvi_get_map_ds": (_i32, [_vp, _P(Cloud), _P(Cloud)]),
    "lvi_get_counts": (_i32, [_vp, _P(_i32)]),
    "lvi_get_pose_record": (_i32, [_vp, _vp]),
    "lvi_debug_get": (_i32, [_vp, _i32, _vp, _i64, _P(_i64)]),
    "lvi_debug_knn": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp]),
    "lvi_debug_residuals": (_i32, [_vp, _i32, _P(_f32), _vp, _vp, _i32, _P(_i32)]),
    "lvi_prof_enable": (_i32, [_vp, _i32]),
    "lvi_prof_reset": (_i32, [_vp]),
    "lvi_prof_read": (_i32, [_vp, _P(KernelStat), _i32, _P(_i32)]),
    "lvi_tracker_params_default": (None, [_P(TrackerParams)]),
    "lvi_tracker_create": (_i32, [_P(TrackerParams), _i32, _P(_vp)]),
    "lvi_tracker_destroy": (None, [_vp]),
    "lvi_tracker_sync": (_i32, [_vp]),
    "lvi_lk_track": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "lvi_good_features": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f64, _f64, _vp, _i32, _P(_i32)]),
    "lvi_tracker_push_image": (_i32, [_vp, _vp, _i32, _i32, _i32]),
    "lvi_clahe": (_i32, [_vp, _vp, _i32, _i32, _i32, C.c_double, _i32, _i32, _vp, _i32]),
    "lvi_tracker_set_equalize": (_i32, [_vp, _i32, C.c_double, _i32, _i32]),
    "lvi_undistort_points": (_i32, [_vp, _P(MeiParams), _vp, _i32, _vp]),
    "lvi_tracker_set_points": (_i32, [_vp, _vp, _i32]),
    "lvi_tracker_run_lk": (_i32, [_vp]),
    "lvi_tracker_get_lk": (_i32, [_vp, _vp, _vp, _vp, _i32, _P(_i32)]),
    "lvi_tracker_set_mask": (_i32, [_vp, _vp, _i32, _i32, _i32]),
    "lvi_tracker_run_gftt": (_i32, [_vp, _i32]),
    "lvi_tracker_get_gftt": (_i32, [_vp, _vp, _i32, _P(_i32)]),
    "lvi_tracker_set_mask_circles": (_i32, [_vp, _vp, _i32, _i32]),
    "lvi_tracker_run_gftt_async": (_i32, [_vp, _i32]),
    "lvi_tracker_finish_frame": (_i32, [_vp, _P(MeiParams), _vp, _i32, _vp, _i32, _P(_i32), _vp]),
    "lvi_tracker_debug_get": (_i32, [_vp, _i32, _vp, _i64, _P(_i64)]),
    "lvi_tracker_prof_enable": (_i32, [_vp, _i32]),
    "lvi_tracker_prof_reset": (_i32, [_vp]),
    "lvi_tracker_prof_read": (_i32, [_vp, _P(KernelStat), _i32, _P(_i32)]),
}


class LviError(RuntimeError):
    def __init__(self, code, where, text):
        super().__init__(f"{where}: status {code} ({text})")
        self.code = code


class Library:
    """A loaded C-ABI library (product or oracle)."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found — build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                "there is no CPU fallback for the HIP path")
        self.path = path
        self.dll = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)          # AttributeError = missing export
            fn.restype = res
            fn.argtypes = args

    @property
    def backend(self):
        return self.dll.lvi_backend().decode()

    def check(self, code, where, soft_ok=True):
        if code < 0 or (code > 0 and not soft_ok):
            raise LviError(code, where, self.dll.lvi_last_error().decode(errors="replace"))
        return code


def as_pts(a):
    """any (n,4) float array or PT_DTYPE array -> contiguous PT_DTYPE array"""
    a = np.asarray(a)
    if a.dtype == PT_DTYPE:
        return np.ascontiguousarray(a)
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
    return a.view(PT_DTYPE).reshape(-1)


def pts_xyzi(a):
    """PT_DTYPE array -> (n,4) float32 view"""
    return np.ascontiguousarray(a).view(np.float32).reshape(-1, 4)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None
