"""Python handle over the feature_tracker half of the C-ABI (include/lvi_hotpath.h):
cv::calcOpticalFlowPyrLK (feature_tracker.cpp:113) and cv::goodFeaturesToTrack (:166)."""
import ctypes as C

import numpy as np

from . import _abi as A


def default_tracker_params(lib, **overrides):
    p = A.TrackerParams()
    lib.dll.lvi_tracker_params_default(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"lvi_tracker_params has no field {k}")
        setattr(p, k, v)
    return p


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


class TrackerHotpath:
    def __init__(self, lib, params=None, device=0, **overrides):
        self.lib = lib
        self.params = params if params is not None else default_tracker_params(lib, **overrides)
        self._t = C.c_void_p()
        lib.check(lib.dll.lvi_tracker_create(C.byref(self.params), int(device), C.byref(self._t)), "lvi_tracker_create")

    def close(self):
        if self._t:
            self.lib.dll.lvi_tracker_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one-call seams ---------------------------------------------------
    def lk_track(self, prev, nxt, prev_xy):
        prev, nxt = _img(prev), _img(nxt)
        xy = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
        n = len(xy)
        out = np.zeros((max(n, 1), 2), np.float32)
        status = np.zeros(max(n, 1), np.uint8)
        err = np.zeros(max(n, 1), np.float32)
        h, w = prev.shape
        self.lib.check(self.lib.dll.lvi_lk_track(self._t, A._ptr(prev), A._ptr(nxt), w, h, prev.strides[0], A._ptr(xy), n,
                                                 A._ptr(out), A._ptr(status), A._ptr(err)), "lvi_lk_track")
        return out[:n], status[:n], err[:n]

    def good_features(self, img, max_corners, quality=0.01, min_dist=20.0, mask=None):
        img = _img(img)
        h, w = img.shape
        m = _img(mask) if mask is not None else None
        cap = int(self.params.max_features)
        xy = np.zeros((cap, 2), np.float32)
        n = C.c_int32(0)
        self.lib.check(self.lib.dll.lvi_good_features(self._t, A._ptr(img), A._ptr(m) if m is not None else None, w, h, img.strides[0],
                                                      int(max_corners), float(quality), float(min_dist), A._ptr(xy), cap, C.byref(n)),
                       "lvi_good_features")
        return xy[:n.value].copy()

    # ---- f-2 / f-3 ----------------------------------------------------------
    def clahe(self, img, clip_limit=3.0, tiles=(8, 8)):
        """cv::createCLAHE(clip_limit, tiles)->apply (feature_tracker.cpp:86-90)"""
        img = _img(img)
        h, w = img.shape
        out = np.zeros((h, w), np.uint8)
        self.lib.check(self.lib.dll.lvi_clahe(self._t, A._ptr(img), w, h, img.strides[0], float(clip_limit), int(tiles[0]), int(tiles[1]),
                                              A._ptr(out), out.strides[0]), "lvi_clahe")
        return out

    def set_equalize(self, on, clip_limit=3.0, tiles=(8, 8)):
        self.lib.check(self.lib.dll.lvi_tracker_set_equalize(self._t, 1 if on else 0, float(clip_limit), int(tiles[0]), int(tiles[1])),
                       "lvi_tracker_set_equalize")

    def undistort_points(self, cam, xy):
        """cam: dict with xi,k1,k2,p1,p2,gamma1,gamma2,u0,v0 (MEI); returns liftProjective(x,y) / z as f32 pairs"""
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.zeros_like(xy)
        c = A.MeiParams(*[float(cam[k]) for k in ("xi", "k1", "k2", "p1", "p2", "gamma1", "gamma2", "u0", "v0")])
        self.lib.check(self.lib.dll.lvi_undistort_points(self._t, C.byref(c), A._ptr(xy), len(xy), A._ptr(out)), "lvi_undistort_points")
        return out

    # ---- staged form --------------------------------------------------------
    def push_image(self, img):
        img = _img(img)
        h, w = img.shape
        self.lib.check(self.lib.dll.lvi_tracker_push_image(self._t, A._ptr(img), w, h, img.strides[0]), "lvi_tracker_push_image")

    def set_points(self, xy):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        self.lib.check(self.lib.dll.lvi_tracker_set_points(self._t, A._ptr(xy), len(xy)), "lvi_tracker_set_points")

    def run_lk(self):
        self.lib.check(self.lib.dll.lvi_tracker_run_lk(self._t), "lvi_tracker_run_lk")

    def get_lk(self):
        cap = int(self.params.max_features)
        xy = np.zeros((cap, 2), np.float32)
        st = np.zeros(cap, np.uint8)
        err = np.zeros(cap, np.float32)
        n = C.c_int32(0)
        self.lib.check(self.lib.dll.lvi_tracker_get_lk(self._t, A._ptr(xy), A._ptr(st), A._ptr(err), cap, C.byref(n)), "lvi_tracker_get_lk")
        return xy[:n.value].copy(), st[:n.value].copy(), err[:n.value].copy()

    def set_mask(self, mask):
        if mask is None:
            self.lib.check(self.lib.dll.lvi_tracker_set_mask(self._t, None, 0, 0, 0), "lvi_tracker_set_mask")
            return
        m = _img(mask)
        h, w = m.shape
        self.lib.check(self.lib.dll.lvi_tracker_set_mask(self._t, A._ptr(m), w, h, m.strides[0]), "lvi_tracker_set_mask")

    def run_gftt(self, max_corners):
        self.lib.check(self.lib.dll.lvi_tracker_run_gftt(self._t, int(max_corners)), "lvi_tracker_run_gftt")

    def get_gftt(self):
        cap = int(self.params.max_features)
        xy = np.zeros((cap, 2), np.float32)
        n = C.c_int32(0)
        self.lib.check(self.lib.dll.lvi_tracker_get_gftt(self._t, A._ptr(xy), cap, C.byref(n)), "lvi_tracker_get_gftt")
        return xy[:n.value].copy()

    def set_mask_circles(self, centers_xy, radius):
        """setMask's raster half (feature_tracker.cpp:64-66): mask = 255 with filled circles of `radius` around the points"""
        c = np.ascontiguousarray(centers_xy, np.float32).reshape(-1, 2)
        self.lib.check(self.lib.dll.lvi_tracker_set_mask_circles(self._t, A._ptr(c) if len(c) else None, len(c), int(radius)), "lvi_tracker_set_mask_circles")

    def run_gftt_async(self, max_corners):
        self.lib.check(self.lib.dll.lvi_tracker_run_gftt_async(self._t, int(max_corners)), "lvi_tracker_run_gftt_async")

    def finish_frame(self, kept_xy, cam=None):
        """the end of readImage in one read: (new corners of a pending run_gftt_async, undistorted [kept ; new] or None)"""
        k = np.ascontiguousarray(kept_xy, np.float32).reshape(-1, 2)
        cap = int(self.params.max_features)
        new = np.zeros((cap, 2), np.float32)
        un = np.zeros((cap, 2), np.float32)
        n = C.c_int32(0)
        c = A.MeiParams(*[float(cam[q]) for q in ("xi", "k1", "k2", "p1", "p2", "gamma1", "gamma2", "u0", "v0")]) if cam is not None else None
        self.lib.check(self.lib.dll.lvi_tracker_finish_frame(self._t, C.byref(c) if c is not None else None, A._ptr(k) if len(k) else None, len(k),
                                                             A._ptr(new), cap, C.byref(n), A._ptr(un) if c is not None else None), "lvi_tracker_finish_frame")
        return new[:n.value].copy(), (un[:len(k) + n.value].copy() if c is not None else None)

    def sync(self):
        self.lib.check(self.lib.dll.lvi_tracker_sync(self._t), "lvi_tracker_sync")

    def debug_get(self, what, dtype):
        nb = C.c_int64(0)
        self.lib.check(self.lib.dll.lvi_tracker_debug_get(self._t, int(what), None, 0, C.byref(nb)), "lvi_tracker_debug_get(size)")
        out = np.zeros(nb.value // np.dtype(dtype).itemsize, dtype)
        if out.size:
            self.lib.check(self.lib.dll.lvi_tracker_debug_get(self._t, int(what), A._ptr(out), out.nbytes, C.byref(nb)), "lvi_tracker_debug_get")
        return out

    def prof_enable(self, on=True):
        self.lib.check(self.lib.dll.lvi_tracker_prof_enable(self._t, 1 if on else 0), "lvi_tracker_prof_enable")

    def prof_reset(self):
        self.lib.check(self.lib.dll.lvi_tracker_prof_reset(self._t), "lvi_tracker_prof_reset")

    def prof_read(self):
        stats = (A.KernelStat * 128)()
        n = C.c_int32(0)
        self.lib.check(self.lib.dll.lvi_tracker_prof_read(self._t, stats, 128, C.byref(n)), "lvi_tracker_prof_read")
        return [dict(name=stats[i].name.decode(), launches=stats[i].launches, total_ms=stats[i].total_ms, bytes_alg=stats[i].bytes_alg)
                for i in range(n.value)]
