"""clock64 phase stamps of the residual kernel (LVI_DBG_ICP_CYCLES) at the bench configuration, bounded search on / off.
Run on the GPU box: python tools/diag/icp_cycles.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.import_package()
hip = pkg.load_hip()
A, S = pkg._abi, pkg.synth
P = dict(N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=1_200_000, icp_max_iters=10, icp_disable_break=1)
g0 = pkg.LidarHotpath(hip, **P)
mc, ms = S.make_map(g0, 60, 30001, seed=4711, target_surf=1_100_000)
g0.close()
pose = S.loop_pose(0.37, 0.0, -0.02)
scan = S.make_scan(100001, pose, 12345)
guess = S.perturbed_guess(pose, 0)
names = ["phase0", "knn_total", "merge", "fit_rows", "reduce_ticket", "total", "row_setup", "first_batch", "s_take", "s_solve", "rounds_direct", "rounds_tiled", "s_total", "T", "bounded", "searches"]
for nb, g1 in (("0", os.environ.get("LVI_ICP_G1", "4")),):
  for sit in os.environ.get("STAMP_ITERS", "-1").split(","):
    os.environ["LVI_ICP_STAMP_ITER"] = sit
    os.environ["LVI_ICP_G1"] = g1
    os.environ["LVI_KNN_NO_BOUND"] = nb
    g = pkg.LidarHotpath(hip, **P)
    g.map_upload(mc, ms); g.map_build()
    for rep in range(3):
        g.scan_upload(scan); g.scan_organize(); g.scan_extract(); g.scan_downsample()
        r = g.scan_match(guess)
    c = g.debug_get(A.DBG_ICP_CYCLES, np.int64)
    print("stamp iter", sit, "no_bound" if nb == "1" else "bounded G1=" + g1, r["n_sel"][-1])
    print("   ", {n: int(v) for n, v in zip(names, c) if n}, 'of', (g.counts()['corner_ds'] + g.counts()['surf_ds']) * r['iters'])
    g.prof_enable(True)
    g.scan_upload(scan); g.scan_organize(); g.scan_extract(); g.scan_downsample(); g.scan_match(guess)
    st = {s["name"]: (s["launches"], round(1e3 * s["total_ms"] / s["launches"], 2)) for s in g.prof_read()}
    print("   ", {k: st[k] for k in ("icp_gn", "set_pose_init") if k in st})
    g.close()
