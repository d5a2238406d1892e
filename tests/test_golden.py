"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the CPU oracle).
CPU tier: the oracle still reproduces them bit for bit (regression pin).  GPU tier: the HIP path
matches them — integers / fixed-order f32 exactly, voxel centroids and the pose within tolerance.
The fixtures are oracle outputs, not reference outputs (PARITY UNPINNED)."""
import os

import numpy as np
import pytest

from helpers import bits, xyzi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LIDAR_PARAMS = dict(N_SCAN=4, Horizon_SCAN=2048, max_raw_points=8192, max_map_points=65536)


def _run_lidar(pkg, lib, g):
    A = pkg._abi
    L = pkg.LidarHotpath(lib, **LIDAR_PARAMS)
    L.map_set(g["map_corner"], g["map_surf"])
    L.scan_upload(g["scan"]); L.scan_organize(); L.scan_extract(); L.scan_downsample()
    info = L.get_scan_info()
    corner, surf = L.get_features()
    cds, sds = L.get_scan_ds()
    mcds, msds = L.get_map_ds()
    out = dict(info=info, surf=xyzi(surf), corner_ds=xyzi(cds), surf_ds=xyzi(sds), map_corner_ds=xyzi(mcds), map_surf_ds=xyzi(msds),
               curvature=L.debug_get(A.DBG_CURVATURE, np.float32)[:info["n"]],
               picked_occl=L.debug_get(A.DBG_PICKED_OCCL, np.int32),
               corner_index=L.debug_get(A.DBG_CORNER_INDEX, np.int32),
               icp=L.scan_match(g["guess"]))
    L.close()
    return out


def _check_lidar(g, o, exact_floats):
    n = len(g["point_range"])
    assert o["info"]["n"] == n
    np.testing.assert_array_equal(o["info"]["start_ring_index"], g["start_ring_index"])
    np.testing.assert_array_equal(o["info"]["end_ring_index"], g["end_ring_index"])
    np.testing.assert_array_equal(o["info"]["point_col_ind"], g["point_col_ind"])
    np.testing.assert_array_equal(bits(o["info"]["point_range"]), bits(g["point_range"]))
    np.testing.assert_array_equal(bits(o["curvature"][5:n - 5]), bits(g["curvature"][5:n - 5]))
    np.testing.assert_array_equal(o["picked_occl"][5:n - 5], g["picked_occl"][5:n - 5])
    np.testing.assert_array_equal(o["corner_index"], g["corner_index"])
    for k in ("surf", "corner_ds", "surf_ds", "map_corner_ds", "map_surf_ds"):
        assert o[k].shape == g[k].shape, k
        if exact_floats:
            np.testing.assert_array_equal(o[k], g[k])
        else:
            np.testing.assert_allclose(o[k], g[k], rtol=0, atol=3e-4, err_msg=k)      # voxel centroid sum order
    assert o["icp"]["status"] == int(g["icp_status"]) and o["icp"]["iters"] == int(g["icp_iters"])
    assert bool(o["icp"]["degenerate"]) == bool(g["icp_degenerate"])
    if exact_floats:
        np.testing.assert_array_equal(o["icp"]["pose"], g["icp_pose"])
        np.testing.assert_array_equal(o["icp"]["n_sel"], g["icp_n_sel"])
    else:
        dp = np.abs(o["icp"]["pose"] - g["icp_pose"])
        assert dp[:3].max() < 1e-4 and dp[3:].max() < 1e-4, dp                         # north_star tolerance
        assert np.abs(np.array(o["icp"]["n_sel"]) - g["icp_n_sel"]).max() <= 3


def _run_tracker(pkg, lib, g):
    A = pkg._abi
    h, w = g["img0"].shape
    T = pkg.TrackerHotpath(lib, max_width=w, max_height=h)
    pts = T.good_features(g["img0"], 60, 0.01, 10.0)
    eig = T.debug_get(A.TDBG_MINEIG, np.float32).reshape(h, w)
    ncand = int(T.debug_get(A.TDBG_GFTT_NCAND, np.int32)[0])
    xy, st, err = T.lk_track(g["img0"], g["img1"], g["gftt_xy"])
    T.push_image(g["img1"])
    l1 = T.debug_get(A.TDBG_PYRAMID_L1, np.uint8)
    T.close()
    return dict(pts=pts, eig=eig, ncand=ncand, xy=xy, st=st, err=err, l1=l1)


def _check_tracker(g, o):
    np.testing.assert_array_equal(o["pts"], g["gftt_xy"])
    assert o["ncand"] == int(g["gftt_ncand"])
    np.testing.assert_array_equal(bits(o["eig"]), bits(g["mineig"]))
    np.testing.assert_array_equal(o["l1"], g["pyr_l1"])
    np.testing.assert_array_equal(o["st"], g["lk_status"])
    np.testing.assert_array_equal(bits(o["xy"]), bits(g["lk_xy"]))
    np.testing.assert_array_equal(bits(o["err"]), bits(g["lk_err"]))


def test_oracle_reproduces_lidar_golden(pkg, oracle):
    g = np.load(os.path.join(GOLD, "lidar_small.npz"))
    _check_lidar(g, _run_lidar(pkg, oracle, g), exact_floats=True)
    # and the fixture is meaningful: the pose it pins is the ground truth up to sensor noise
    assert np.abs(g["icp_pose"][3:] - g["pose_truth"][3:]).max() < 0.05


def test_oracle_reproduces_tracker_golden(pkg, oracle):
    g = np.load(os.path.join(GOLD, "tracker_small.npz"))
    _check_tracker(g, _run_tracker(pkg, oracle, g))
    gt = pkg.synth.apply_homography(g["homography"], g["gftt_xy"])
    ok = g["lk_status"] == 1
    assert np.median(np.linalg.norm(g["lk_xy"][ok] - gt[ok], axis=1)) < 0.15


@pytest.mark.gpu
def test_hip_matches_lidar_golden(pkg, hip):
    g = np.load(os.path.join(GOLD, "lidar_small.npz"))
    _check_lidar(g, _run_lidar(pkg, hip, g), exact_floats=False)


@pytest.mark.gpu
def test_hip_matches_tracker_golden(pkg, hip):
    g = np.load(os.path.join(GOLD, "tracker_small.npz"))
    _check_tracker(g, _run_tracker(pkg, hip, g))
