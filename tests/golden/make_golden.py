"""Generates the golden fixtures in this directory with the CPU oracle (oracle/).

The reference ships no test vectors and cannot be built or run here (ROS 2 / PCL / OpenCV are
absent), so these vectors come from this repo's restatement of it: they pin the oracle against
regressions and give the GPU tests fixed inputs and expected outputs; they do NOT pin the
oracle to the reference ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
from oracle import loader  # noqa: E402

LIDAR_PARAMS = dict(N_SCAN=4, Horizon_SCAN=2048, max_raw_points=8192, max_map_points=65536)


def lidar(pkg, lib):
    S, A = pkg.synth, pkg._abi
    L = pkg.LidarHotpath(lib, **LIDAR_PARAMS)
    mc, ms = S.make_map(L, 6, 6001, seed=31)
    pose = S.loop_pose(0.5, 0.01, -0.01)
    scan = S.make_scan(6001, pose, 2024)
    guess = S.perturbed_guess(pose, 3)
    L.map_set(mc, ms)
    L.scan_upload(scan); L.scan_organize(); L.scan_extract(); L.scan_downsample()
    info = L.get_scan_info()
    corner, surf = L.get_features()
    cds, sds = L.get_scan_ds()
    mcds, msds = L.get_map_ds()
    res = L.scan_match(guess)
    np.savez_compressed(
        os.path.join(HERE, "lidar_small.npz"),
        scan=scan, map_corner=mc, map_surf=ms, pose_truth=pose, guess=guess,
        start_ring_index=info["start_ring_index"], end_ring_index=info["end_ring_index"],
        point_col_ind=info["point_col_ind"], point_range=info["point_range"],
        curvature=L.debug_get(A.DBG_CURVATURE, np.float32)[:info["n"]],
        picked_occl=L.debug_get(A.DBG_PICKED_OCCL, np.int32).astype(np.int8),
        corner_index=L.debug_get(A.DBG_CORNER_INDEX, np.int32),
        surf=A.pts_xyzi(surf), corner_ds=A.pts_xyzi(cds), surf_ds=A.pts_xyzi(sds),
        map_corner_ds=A.pts_xyzi(mcds), map_surf_ds=A.pts_xyzi(msds),
        icp_pose=res["pose"], icp_iters=res["iters"], icp_n_sel=np.array(res["n_sel"], np.int32),
        icp_status=res["status"], icp_degenerate=int(res["degenerate"]))
    print("lidar_small:", L.counts(), res["iters"], res["pose"])


def tracker(pkg, lib):
    S, A = pkg.synth, pkg._abi
    w, h = 200, 150
    img0 = S.make_texture(w, h, 77)
    Hm = S.small_motion_homography(w, h, 5, max_px=4.0)
    img1 = S.warp_homography(img0, Hm)
    T = pkg.TrackerHotpath(lib, max_width=w, max_height=h)
    pts = T.good_features(img0, 60, 0.01, 10.0)
    eig = T.debug_get(A.TDBG_MINEIG, np.float32)
    ncand = int(T.debug_get(A.TDBG_GFTT_NCAND, np.int32)[0])
    xy, st, err = T.lk_track(img0, img1, pts)
    T.push_image(img1)
    l1 = T.debug_get(A.TDBG_PYRAMID_L1, np.uint8)
    np.savez_compressed(os.path.join(HERE, "tracker_small.npz"), img0=img0, img1=img1, homography=Hm,
                        gftt_xy=pts, gftt_ncand=ncand, mineig=eig.reshape(h, w), pyr_l1=l1,
                        lk_xy=xy, lk_status=st, lk_err=err)
    print("tracker_small:", len(pts), ncand, st.mean())


if __name__ == "__main__":
    pkg = graft.import_package()
    lib = loader.load(pkg)
    lidar(pkg, lib)
    tracker(pkg, lib)
