"""Python handle over the lidar half of the C-ABI (include/lvi_hotpath.h).

Method names follow the reference call sites they stand for
(imageProjection / featureExtraction / mapOptimization); see the header for the
file:line of each.  This is harness plumbing for tests and bench — the product
is the shared library underneath.
"""
import ctypes as C

import numpy as np

from . import _abi as A


def default_params(lib, **overrides):
    p = A.LidarParams()
    lib.dll.lvi_lidar_params_default(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"lvi_lidar_params has no field {k}")
        setattr(p, k, v)
    return p


class LidarHotpath:
    def __init__(self, lib, params=None, device=0, **overrides):
        self.lib = lib
        self.params = params if params is not None else default_params(lib, **overrides)
        self._h = C.c_void_p()
        lib.check(lib.dll.lvi_lidar_create(C.byref(self.params), int(device), C.byref(self._h)), "lvi_lidar_create")
        self._cap_scan = int(self.params.N_SCAN) * int(self.params.Horizon_SCAN)

    def close(self):
        if self._h:
            self.lib.dll.lvi_lidar_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ----------------------------------------------------------
    def _new_scan_info(self, cap=None):
        cap = int(cap if cap is not None else self._cap_scan)
        ns = int(self.params.N_SCAN)
        bufs = dict(start=np.zeros(ns, np.int32), end=np.zeros(ns, np.int32), col=np.zeros(cap, np.int32),
                    rng=np.zeros(cap, np.float32), pts=np.zeros(cap, A.PT_DTYPE))
        si = A.ScanInfo(cap, 0, bufs["start"].ctypes.data_as(C.POINTER(C.c_int32)), bufs["end"].ctypes.data_as(C.POINTER(C.c_int32)),
                        bufs["col"].ctypes.data_as(C.POINTER(C.c_int32)), bufs["rng"].ctypes.data_as(C.POINTER(C.c_float)),
                        bufs["pts"].ctypes.data_as(C.c_void_p))
        return si, bufs

    @staticmethod
    def _scan_info_dict(si, bufs):
        n = si.n
        return dict(n=n, start_ring_index=bufs["start"].copy(), end_ring_index=bufs["end"].copy(),
                    point_col_ind=bufs["col"][:n].copy(), point_range=bufs["rng"][:n].copy(), cloud_deskewed=bufs["pts"][:n].copy())

    @staticmethod
    def _new_cloud(cap):
        buf = np.zeros(max(int(cap), 1), A.PT_DTYPE)
        return A.Cloud(int(cap), 0, buf.ctypes.data_as(C.c_void_p)), buf

    # ---- one-call seams ---------------------------------------------------
    def organize_scan(self, livox_pts):
        pts = np.ascontiguousarray(livox_pts, dtype=A.LIVOX_DTYPE)
        si, bufs = self._new_scan_info()
        self.lib.check(self.lib.dll.lvi_organize_scan(self._h, A._ptr(pts), len(pts), C.byref(si)), "lvi_organize_scan")
        return self._scan_info_dict(si, bufs)

    def organize_scan_deskew(self, livox_pts, time_scan_cur, imu_time, imu_rot):
        """a-0 + f-1: imu_time [m] / imu_rot [m,3] are imuDeskewInfo's table (imageProjection.cpp:354-410)"""
        pts = np.ascontiguousarray(livox_pts, dtype=A.LIVOX_DTYPE)
        t = np.ascontiguousarray(imu_time, np.float64)
        r = np.ascontiguousarray(np.asarray(imu_rot, np.float64).T)
        dp = C.POINTER(C.c_double)
        info = A.DeskewInfo(1, len(t) - 1, float(time_scan_cur), t.ctypes.data_as(dp), r[0].ctypes.data_as(dp),
                            r[1].ctypes.data_as(dp), r[2].ctypes.data_as(dp))
        si, bufs = self._new_scan_info()
        self.lib.check(self.lib.dll.lvi_organize_scan_deskew(self._h, A._ptr(pts), len(pts), C.byref(info), C.byref(si)), "lvi_organize_scan_deskew")
        return self._scan_info_dict(si, bufs)

    def extract_features(self, info):
        n = int(info["n"])
        keep = dict(start=np.ascontiguousarray(info["start_ring_index"], np.int32), end=np.ascontiguousarray(info["end_ring_index"], np.int32),
                    col=np.ascontiguousarray(info["point_col_ind"], np.int32), rng=np.ascontiguousarray(info["point_range"], np.float32),
                    pts=A.as_pts(info["cloud_deskewed"]))
        si = A.ScanInfo(n, n, keep["start"].ctypes.data_as(C.POINTER(C.c_int32)), keep["end"].ctypes.data_as(C.POINTER(C.c_int32)),
                        keep["col"].ctypes.data_as(C.POINTER(C.c_int32)), keep["rng"].ctypes.data_as(C.POINTER(C.c_float)),
                        keep["pts"].ctypes.data_as(C.c_void_p))
        cc, cb = self._new_cloud(max(n, 1))
        sc, sb = self._new_cloud(max(n, 1))
        self.lib.check(self.lib.dll.lvi_extract_features(self._h, C.byref(si), C.byref(cc), C.byref(sc)), "lvi_extract_features")
        return cb[:cc.n].copy(), sb[:sc.n].copy()

    def voxel_downsample(self, pts, leaf):
        p = A.as_pts(pts)
        out = np.zeros(max(len(p), 1), A.PT_DTYPE)
        n_out = C.c_int32(0)
        self.lib.check(self.lib.dll.lvi_voxel_downsample(self._h, A._ptr(p), len(p), float(leaf), A._ptr(out), len(out), C.byref(n_out)),
                       "lvi_voxel_downsample")
        return out[:n_out.value].copy()

    def map_set(self, corner_raw, surf_raw):
        c, s = A.as_pts(corner_raw), A.as_pts(surf_raw)
        self.lib.check(self.lib.dll.lvi_map_set(self._h, A._ptr(c), len(c), A._ptr(s), len(s)), "lvi_map_set")

    @staticmethod
    def _imu(imu):
        if imu is None:
            return None
        return A.ImuHint(int(imu.get("imu_available", 1)), float(imu.get("roll", 0.0)), float(imu.get("pitch", 0.0)), float(imu.get("yaw", 0.0)))

    @staticmethod
    def _result_dict(st, res, pose):
        it = res.iters
        return dict(status=st, iters=it, converged=bool(res.converged), degenerate=bool(res.degenerate),
                    n_corner_ds=res.n_corner_ds, n_surf_ds=res.n_surf_ds, n_sel=list(res.n_sel[:it]),
                    pose=np.array(pose[:], np.float32))

    def scan_to_map(self, corner, surf, pose, imu=None):
        c, s = A.as_pts(corner), A.as_pts(surf)
        pose_c = (C.c_float * 6)(*[float(v) for v in pose])
        res = A.IcpResult()
        hint = self._imu(imu)
        st = self.lib.check(self.lib.dll.lvi_scan_to_map(self._h, A._ptr(c), len(c), A._ptr(s), len(s),
                                                        C.byref(hint) if hint else None, pose_c, C.byref(res)), "lvi_scan_to_map")
        return self._result_dict(st, res, pose_c)

    def transform_cloud(self, pts, pose6):
        p = A.as_pts(pts)
        out = np.zeros(len(p), A.PT_DTYPE)
        pose_c = (C.c_float * 6)(*[float(v) for v in pose6])
        self.lib.check(self.lib.dll.lvi_transform_cloud(self._h, A._ptr(p), len(p), pose_c, A._ptr(out)), "lvi_transform_cloud")
        return out

    # ---- staged, device-resident form ---------------------------------------
    def scan_upload(self, livox_pts):
        pts = np.ascontiguousarray(livox_pts, dtype=A.LIVOX_DTYPE)
        self.lib.check(self.lib.dll.lvi_scan_upload(self._h, A._ptr(pts), len(pts)), "lvi_scan_upload")

    def scan_upload_device(self, d_ptr, n_raw):
        self.lib.check(self.lib.dll.lvi_scan_upload_device(self._h, C.c_void_p(int(d_ptr)), int(n_raw)), "lvi_scan_upload_device")

    def map_upload_device(self, d_corner, nc, d_surf, ns):
        self.lib.check(self.lib.dll.lvi_map_upload_device(self._h, C.c_void_p(int(d_corner)), int(nc), C.c_void_p(int(d_surf)), int(ns)),
                       "lvi_map_upload_device")

    def map_share(self, owner):
        """read `owner`'s raw local map in place (one copy per GPU; lvi_map_share)"""
        self.lib.check(self.lib.dll.lvi_map_share(self._h, owner._h), "lvi_map_share")

    def scan_replay_enqueue(self, d_scan_ptr, n_raw, pose, d_record_ptr, rebuild_map=True):
        pose_c = (C.c_float * 6)(*[float(v) for v in pose])
        self.lib.check(self.lib.dll.lvi_scan_replay_enqueue(self._h, C.c_void_p(int(d_scan_ptr)), int(n_raw), pose_c,
                                                            C.c_void_p(int(d_record_ptr)), 1 if rebuild_map else 0), "lvi_scan_replay_enqueue")

    # ---- f-4: keyframe store / map assembly ---------------------------------
    def keyframe_add(self, corner, surf, pose):
        c, s = A.as_pts(corner), A.as_pts(surf)
        pose_c = (C.c_float * 6)(*[float(v) for v in pose]); idx = C.c_int32(-1)
        self.lib.check(self.lib.dll.lvi_keyframe_add(self._h, A._ptr(c), len(c), A._ptr(s), len(s), pose_c, C.byref(idx)), "lvi_keyframe_add")
        return idx.value

    def keyframe_add_current(self, pose):
        pose_c = (C.c_float * 6)(*[float(v) for v in pose]); idx = C.c_int32(-1)
        self.lib.check(self.lib.dll.lvi_keyframe_add_current(self._h, pose_c, C.byref(idx)), "lvi_keyframe_add_current")
        return idx.value

    def keyframe_set_pose(self, index, pose):
        pose_c = (C.c_float * 6)(*[float(v) for v in pose])
        self.lib.check(self.lib.dll.lvi_keyframe_set_pose(self._h, int(index), pose_c), "lvi_keyframe_set_pose")

    def keyframe_count(self):
        n, p = C.c_int32(0), C.c_int32(0)
        self.lib.check(self.lib.dll.lvi_keyframe_count(self._h, C.byref(n), C.byref(p)), "lvi_keyframe_count")
        return n.value, p.value

    def keyframes_clear(self):
        self.lib.check(self.lib.dll.lvi_keyframes_clear(self._h), "lvi_keyframes_clear")

    def map_assemble(self, key_indices):
        k = np.ascontiguousarray(key_indices, np.int32)
        self.lib.check(self.lib.dll.lvi_map_assemble(self._h, k.ctypes.data_as(C.POINTER(C.c_int32)), len(k)), "lvi_map_assemble")

    def map_update(self, key_indices):
        """lvi_map_assemble's result maintained incrementally (keyframes entering / leaving the list)"""
        k = np.ascontiguousarray(key_indices, np.int32)
        self.lib.check(self.lib.dll.lvi_map_update(self._h, k.ctypes.data_as(C.POINTER(C.c_int32)), len(k)), "lvi_map_update")

    def scan_organize(self):
        self.lib.check(self.lib.dll.lvi_scan_organize(self._h), "lvi_scan_organize")

    def scan_set_deskew(self, time_scan_cur=0.0, imu_time=None, imu_rot=None):
        """f-1: imuDeskewInfo's table (imu_time [m], imu_rot [m,3], seconds / radians); None switches deskew off"""
        if imu_time is None:
            self.lib.check(self.lib.dll.lvi_scan_set_deskew(self._h, None), "lvi_scan_set_deskew")
            return
        t = np.ascontiguousarray(imu_time, np.float64)
        r = np.ascontiguousarray(np.asarray(imu_rot, np.float64).T)          # rows: x, y, z
        dp = C.POINTER(C.c_double)
        info = A.DeskewInfo(1, len(t) - 1, float(time_scan_cur), t.ctypes.data_as(dp), r[0].ctypes.data_as(dp),
                            r[1].ctypes.data_as(dp), r[2].ctypes.data_as(dp))
        self.lib.check(self.lib.dll.lvi_scan_set_deskew(self._h, C.byref(info)), "lvi_scan_set_deskew")

    def scan_extract(self):
        self.lib.check(self.lib.dll.lvi_scan_extract(self._h), "lvi_scan_extract")

    def scan_downsample(self):
        self.lib.check(self.lib.dll.lvi_scan_downsample(self._h), "lvi_scan_downsample")

    def map_upload(self, corner_raw, surf_raw):
        c, s = A.as_pts(corner_raw), A.as_pts(surf_raw)
        self.lib.check(self.lib.dll.lvi_map_upload(self._h, A._ptr(c), len(c), A._ptr(s), len(s)), "lvi_map_upload")

    def map_build(self):
        self.lib.check(self.lib.dll.lvi_map_build(self._h), "lvi_map_build")

    def scan_match(self, pose, imu=None):
        pose_c = (C.c_float * 6)(*[float(v) for v in pose])
        res = A.IcpResult()
        hint = self._imu(imu)
        st = self.lib.check(self.lib.dll.lvi_scan_match(self._h, C.byref(hint) if hint else None, pose_c, C.byref(res)), "lvi_scan_match")
        return self._result_dict(st, res, pose_c)

    # ---- batched form (lvi_scan_batch_*) -------------------------------------------
    def batch_bind_device(self, d_ptrs, n_raws):
        n = len(d_ptrs)
        arr = (C.c_void_p * n)(*[int(p) for p in d_ptrs]); nr = (C.c_int32 * n)(*[int(v) for v in n_raws])
        self.lib.check(self.lib.dll.lvi_scan_batch_bind_device(self._h, n, arr, nr), "lvi_scan_batch_bind_device")

    def batch_upload(self, scans):
        keep = [np.ascontiguousarray(s, dtype=A.LIVOX_DTYPE) for s in scans]
        n = len(keep)
        arr = (C.c_void_p * n)(*[k.ctypes.data for k in keep]); nr = (C.c_int32 * n)(*[len(k) for k in keep])
        self.lib.check(self.lib.dll.lvi_scan_batch_upload(self._h, n, arr, nr), "lvi_scan_batch_upload")

    def batch_run(self, poses, d_records_ptr=0, rebuild_map=True):
        p = np.ascontiguousarray(poses, np.float32).reshape(-1, 6)
        self.lib.check(self.lib.dll.lvi_scan_batch_run(self._h, len(p), p.ctypes.data_as(C.POINTER(C.c_float)),
                                                       C.c_void_p(int(d_records_ptr)) if d_records_ptr else None, 1 if rebuild_map else 0), "lvi_scan_batch_run")

    def batch_get_records(self, n):
        rec = np.zeros((int(n), 8), np.float32)
        self.lib.check(self.lib.dll.lvi_scan_batch_get_records(self._h, int(n), A._ptr(rec)), "lvi_scan_batch_get_records")
        return rec

    def batch_select(self, slot):
        self.lib.check(self.lib.dll.lvi_batch_select(self._h, int(slot)), "lvi_batch_select")

    def scan_match_async(self, pose, d_record_ptr):
        pose_c = (C.c_float * 6)(*[float(v) for v in pose])
        self.lib.check(self.lib.dll.lvi_scan_match_async(self._h, pose_c, C.c_void_p(int(d_record_ptr))), "lvi_scan_match_async")

    def sync(self):
        self.lib.check(self.lib.dll.lvi_lidar_sync(self._h), "lvi_lidar_sync")

    def mark(self, slot):
        """record a point in the handle's stream order (everything enqueued so far)"""
        self.lib.check(self.lib.dll.lvi_lidar_mark(self._h, int(slot)), "lvi_lidar_mark")

    def wait_mark(self, slot):
        """block until the point recorded by mark(slot) has been reached (at once if never marked)"""
        self.lib.check(self.lib.dll.lvi_lidar_wait_mark(self._h, int(slot)), "lvi_lidar_wait_mark")

    def get_scan_info(self):
        si, bufs = self._new_scan_info()
        self.lib.check(self.lib.dll.lvi_get_scan_info(self._h, C.byref(si)), "lvi_get_scan_info")
        return self._scan_info_dict(si, bufs)

    def counts(self):
        c = (C.c_int32 * 8)()
        self.lib.check(self.lib.dll.lvi_get_counts(self._h, c), "lvi_get_counts")
        return dict(n=c[0], corner=c[1], surf=c[2], corner_ds=c[3], surf_ds=c[4], map_corner_ds=c[5], map_surf_ds=c[6])

    def get_pose_record(self):
        rec = np.zeros(8, np.float32)
        self.lib.check(self.lib.dll.lvi_get_pose_record(self._h, A._ptr(rec)), "lvi_get_pose_record")
        return dict(pose=rec[:6].copy(), status=int(rec[6:7].view(np.int32)[0]), iters=int(rec[7:8].view(np.int32)[0]))

    def _get_pair(self, fn, name, cap_a, cap_b):
        ca, ba = self._new_cloud(cap_a)
        cb, bb = self._new_cloud(cap_b)
        self.lib.check(fn(self._h, C.byref(ca), C.byref(cb)), name)
        return ba[:ca.n].copy(), bb[:cb.n].copy()

    def get_features(self):
        c = self.counts()
        return self._get_pair(self.lib.dll.lvi_get_features, "lvi_get_features", c["corner"], c["surf"])

    def get_scan_ds(self):
        c = self.counts()
        return self._get_pair(self.lib.dll.lvi_get_scan_ds, "lvi_get_scan_ds", c["corner_ds"], c["surf_ds"])

    def get_map_ds(self):
        c = self.counts()
        return self._get_pair(self.lib.dll.lvi_get_map_ds, "lvi_get_map_ds", c["map_corner_ds"], c["map_surf_ds"])

    # ---- inspection ---------------------------------------------------------
    def debug_get(self, what, dtype):
        nb = C.c_int64(0)
        self.lib.check(self.lib.dll.lvi_debug_get(self._h, int(what), None, 0, C.byref(nb)), "lvi_debug_get(size)")
        out = np.zeros(nb.value // np.dtype(dtype).itemsize, dtype)
        if out.size:
            self.lib.check(self.lib.dll.lvi_debug_get(self._h, int(what), A._ptr(out), out.nbytes, C.byref(nb)), "lvi_debug_get")
        return out

    def debug_knn(self, which, queries):
        q = A.as_pts(queries)
        idx = np.zeros((len(q), 5), np.int32)
        sqd = np.zeros((len(q), 5), np.float32)
        self.lib.check(self.lib.dll.lvi_debug_knn(self._h, int(which), A._ptr(q), len(q), A._ptr(idx), A._ptr(sqd)), "lvi_debug_knn")
        return idx, sqd

    def debug_residuals(self, which, pose):
        c = self.counts()
        cap = max(c["corner_ds"] if which == 0 else c["surf_ds"], 1)
        coeff = np.zeros(cap, A.PT_DTYPE)
        flag = np.zeros(cap, np.uint8)
        n = C.c_int32(0)
        pose_c = (C.c_float * 6)(*[float(v) for v in pose])
        self.lib.check(self.lib.dll.lvi_debug_residuals(self._h, int(which), pose_c, A._ptr(coeff), A._ptr(flag), cap, C.byref(n)), "lvi_debug_residuals")
        return coeff[:n.value].copy(), flag[:n.value].copy()

    # ---- kernel timing --------------------------------------------------------
    def prof_enable(self, on=True):
        self.lib.check(self.lib.dll.lvi_prof_enable(self._h, 1 if on else 0), "lvi_prof_enable")

    def prof_reset(self):
        self.lib.check(self.lib.dll.lvi_prof_reset(self._h), "lvi_prof_reset")

    def prof_read(self):
        stats = (A.KernelStat * 128)()
        n = C.c_int32(0)
        self.lib.check(self.lib.dll.lvi_prof_read(self._h, stats, 128, C.byref(n)), "lvi_prof_read")
        return [dict(name=stats[i].name.decode(), launches=stats[i].launches, total_ms=stats[i].total_ms, bytes_alg=stats[i].bytes_alg)
                for i in range(n.value)]
