"""ctypes view of host/lvi_seq_capi.cpp: the C++ host mirror (host/lvi_host.hpp — the reference's node classes over the
C-ABI) flattened to C for the replay harness: the sequential lidar_odometry loop and the feature_tracker node callback.

One binding, parameterised by the shared-library path: ``host/liblvi_host_hip.so`` (built by build.py, linked against
the product library) in deployment and bench; the CPU-tier tests build the same source against the CPU library they check with.
"""
import ctypes as C
import os
import subprocess
from shutil import which

import numpy as np

from . import _abi as A

HOST_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")
HOST_HIP_LIB = os.path.join(HOST_DIR, "liblvi_host_hip.so")


class SeqParams(C.Structure):
    _fields_ = [("incremental_map", C.c_int32), ("use_imu_heading_initialization", C.c_int32), ("mapping_process_interval", C.c_double),
                ("keyframe_adding_dist", C.c_float), ("keyframe_adding_angle", C.c_float), ("keyframe_density", C.c_float),
                ("keyframe_search_radius", C.c_float)]


class SeqResult(C.Structure):
    _fields_ = [("processed", C.c_int32), ("status", C.c_int32), ("iters", C.c_int32), ("converged", C.c_int32), ("degenerate", C.c_int32),
                ("saved_keyframe", C.c_int32), ("n_keyframes", C.c_int32), ("n_keys", C.c_int32), ("pose", C.c_float * 6)]


FUNDAMENTAL_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_double, C.POINTER(C.c_uint8), C.c_void_p)


def build_host_library(out_path, link_dir, link_name, extra=()):
    """g++ -shared of host/lvi_seq_capi.cpp against lib<link_name>.so in link_dir (rpath set)"""
    src = os.path.join(HOST_DIR, "lvi_seq_capi.cpp")
    deps = [src, os.path.join(HOST_DIR, "lvi_host.hpp"), os.path.join(HOST_DIR, "..", "..", "include", "lvi_hotpath.h")]
    if os.path.exists(out_path) and all(os.path.getmtime(d) <= os.path.getmtime(out_path) for d in deps):
        return out_path
    cxx = which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", out_path, src, "-L" + link_dir, "-l" + link_name,
           "-Wl,-rpath," + link_dir, *extra]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("host library build failed:\n" + r.stdout + r.stderr)
    return out_path


class HostLibrary:
    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found — build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.dll = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
        d = self.dll
        d.lvh_last_error.restype = C.c_char_p
        d.lvh_seq_params_default.argtypes = [C.POINTER(SeqParams)]
        d.lvh_seq_create.restype = C.c_void_p
        d.lvh_seq_create.argtypes = [C.POINTER(A.LidarParams), C.c_int32, C.POINTER(SeqParams)]
        d.lvh_seq_destroy.argtypes = [C.c_void_p]
        d.lvh_seq_handle.restype = C.c_void_p
        d.lvh_seq_handle.argtypes = [C.c_void_p]
        scan_args = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_float, C.c_float, C.c_float, C.POINTER(SeqResult)]
        d.lvh_seq_scan.argtypes = scan_args
        d.lvh_seq_scan_device.argtypes = scan_args
        d.lvh_seq_seed_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.c_double]
        d.lvh_seq_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        d.lvh_seq_keyposes.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        d.lvh_trk_create.restype = C.c_void_p
        d.lvh_trk_create.argtypes = [C.POINTER(A.TrackerParams), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(A.MeiParams)]
        d.lvh_trk_destroy.argtypes = [C.c_void_p]
        d.lvh_trk_set_fundamental_hook.argtypes = [C.c_void_p, FUNDAMENTAL_FN, C.c_void_p]
        d.lvh_trk_image.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_int32,
                                    C.POINTER(C.c_int32)]
        d.lvh_trk_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]

    def check(self, code, where):
        if code < 0:
            raise A.LviError(code, where, self.dll.lvh_last_error().decode(errors="replace"))
        return code


class _BorrowedHandle:
    """a LidarHotpath-like view of the lvi_lidar owned by the C++ side (never destroyed from here)"""

    def __init__(self, lidar_cls, lib, ptr, params):
        self._obj = lidar_cls.__new__(lidar_cls)
        self._obj.lib, self._obj.params, self._obj._h = lib, params, C.c_void_p(ptr)
        self._obj._cap_scan = int(params.N_SCAN) * int(params.Horizon_SCAN)
        self._obj.close = lambda: None

    def __getattr__(self, k):
        return getattr(self._obj, k)


class SequentialMapper:
    """MapOptimizationNode of host/lvi_host.hpp (updateInitialGuess, extractNearby, extractCloud, scan matching, saveFrame,
    key-pose push; mapOptimization.cpp:298-333, 806-999, 1315-1412, 1529-1603) fed scan by scan"""

    def __init__(self, hostlib, abi_lib, lidar_params, device=0, **seq):
        from .lidar import LidarHotpath
        self.hl = hostlib
        sp = SeqParams()
        hostlib.dll.lvh_seq_params_default(C.byref(sp))
        for k, v in seq.items():
            if not hasattr(sp, k):
                raise AttributeError(f"lvh_seq_params has no field {k}")
            setattr(sp, k, v)
        self.seq_params = sp
        self._s = hostlib.dll.lvh_seq_create(C.byref(lidar_params), int(device), C.byref(sp))
        if not self._s:
            raise A.LviError(-3, "lvh_seq_create", hostlib.dll.lvh_last_error().decode(errors="replace"))
        self.handle = _BorrowedHandle(LidarHotpath, abi_lib, hostlib.dll.lvh_seq_handle(self._s), lidar_params)

    def close(self):
        if self._s:
            self.hl.dll.lvh_seq_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _res(r):
        return dict(processed=bool(r.processed), status=r.status, iters=r.iters, converged=bool(r.converged), degenerate=bool(r.degenerate),
                    saved_keyframe=bool(r.saved_keyframe), n_keyframes=r.n_keyframes, n_keys=r.n_keys, pose=np.array(r.pose[:], np.float32))

    def scan(self, livox_pts, stamp, imu=None):
        pts = np.ascontiguousarray(livox_pts, dtype=A.LIVOX_DTYPE)
        r = SeqResult()
        ia, ro, pi, ya = (1, imu[0], imu[1], imu[2]) if imu is not None else (0, 0.0, 0.0, 0.0)
        self.hl.check(self.hl.dll.lvh_seq_scan(self._s, A._ptr(pts), len(pts), float(stamp), ia, ro, pi, ya, C.byref(r)), "lvh_seq_scan")
        return self._res(r)

    def scan_device(self, d_ptr, n_raw, stamp, imu=None):
        r = SeqResult()
        ia, ro, pi, ya = (1, imu[0], imu[1], imu[2]) if imu is not None else (0, 0.0, 0.0, 0.0)
        self.hl.check(self.hl.dll.lvh_seq_scan_device(self._s, C.c_void_p(int(d_ptr)), int(n_raw), float(stamp), ia, ro, pi, ya, C.byref(r)), "lvh_seq_scan_device")
        return self._res(r)

    def seed_keyframe(self, corner, surf, pose, time):
        """a keyframe of an earlier session: DS clouds (sensor frame) into the device store, pose into the key poses"""
        c, s = A.as_pts(corner), A.as_pts(surf)
        pose_c = (C.c_float * 6)(*[float(v) for v in pose])
        return self.hl.check(self.hl.dll.lvh_seq_seed_keyframe(self._s, A._ptr(c), len(c), A._ptr(s), len(s), pose_c, float(time)), "lvh_seq_seed_keyframe")

    def keys(self):
        n = C.c_int32(0)
        self.hl.check(self.hl.dll.lvh_seq_keys(self._s, None, 0, C.byref(n)), "lvh_seq_keys")
        out = np.zeros(max(n.value, 1), np.int32)
        self.hl.check(self.hl.dll.lvh_seq_keys(self._s, A._ptr(out), len(out), C.byref(n)), "lvh_seq_keys")
        return out[:n.value].copy()

    def keyposes(self):
        n = C.c_int32(0)
        self.hl.check(self.hl.dll.lvh_seq_keyposes(self._s, None, 0, C.byref(n)), "lvh_seq_keyposes")
        out = np.zeros((max(n.value, 1), 8), np.float64)
        self.hl.check(self.hl.dll.lvh_seq_keyposes(self._s, A._ptr(out), len(out), C.byref(n)), "lvh_seq_keyposes")
        return out[:n.value].copy()


class TrackerNode:
    """FeatureTrackerNode of host/lvi_host.hpp: img_callback of feature_tracker_node.cpp:37-231 without ROS"""
    OUTCOMES = ("first_image", "restart", "not_published", "first_publish_suppressed", "published")

    def __init__(self, hostlib, tracker_params, row, col, freq, equalize=False, cam=None, device=0):
        self.hl = hostlib
        c = A.MeiParams(*[float(cam[k]) for k in ("xi", "k1", "k2", "p1", "p2", "gamma1", "gamma2", "u0", "v0")]) if cam is not None else None
        self._t = hostlib.dll.lvh_trk_create(C.byref(tracker_params), int(device), int(row), int(col), int(freq), 1 if equalize else 0,
                                             C.byref(c) if c is not None else None)
        if not self._t:
            raise A.LviError(-3, "lvh_trk_create", hostlib.dll.lvh_last_error().decode(errors="replace"))
        self.cap = int(tracker_params.max_features)
        self._hook = None

    def close(self):
        if self._t:
            self.hl.dll.lvh_trk_destroy(self._t)
            self._t = None

    def set_fundamental_hook(self, fn):
        """fn(un_cur [n,2], un_forw [n,2], f_threshold) -> status [n] (uint8): the node's cv::findFundamentalMat"""
        if fn is None:
            self._hook = None
            self.hl.dll.lvh_trk_set_fundamental_hook(self._t, C.cast(None, FUNDAMENTAL_FN), None)
            return

        def tramp(a, b, n, thr, status, _user):
            ua = np.ctypeslib.as_array(a, shape=(n, 2)).copy() if n else np.zeros((0, 2), np.float32)
            ub = np.ctypeslib.as_array(b, shape=(n, 2)).copy() if n else np.zeros((0, 2), np.float32)
            st = np.asarray(fn(ua, ub, thr), np.uint8)
            for i in range(n):
                status[i] = int(st[i])
        self._hook = FUNDAMENTAL_FN(tramp)
        self.hl.dll.lvh_trk_set_fundamental_hook(self._t, self._hook, None)

    def image(self, img, stamp):
        img = np.ascontiguousarray(img, np.uint8)
        oc, n = C.c_int32(0), C.c_int32(0)
        pts = np.zeros((self.cap, 3), np.float32)
        ch = np.zeros((6, self.cap), np.float32)
        info = (C.c_int32 * 4)()
        self.hl.check(self.hl.dll.lvh_trk_image(self._t, A._ptr(img), float(stamp), C.byref(oc), C.byref(n), A._ptr(pts), A._ptr(ch), self.cap, info),
                      "lvh_trk_image")
        m = n.value
        return dict(outcome=self.OUTCOMES[oc.value], points=pts[:m].copy(), channels=ch[:, :m].copy(), pub_this_frame=bool(info[0]),
                    rejectWithF_skipped=info[1], n_cur_pts=info[2], pub_count=info[3])

    def points(self):
        n = C.c_int32(0)
        rows = np.zeros((self.cap, 4), np.float32)
        self.hl.check(self.hl.dll.lvh_trk_points(self._t, A._ptr(rows), self.cap, C.byref(n)), "lvh_trk_points")
        return rows[:n.value].copy()
