"""diagnostic (not a test): phase cycles of the residual kernel at bench size"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.import_package(); hip = pkg.load_hip(); S = pkg.synth
P = dict(N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=3_000_000, icp_max_iters=10, icp_disable_break=1)
L = pkg.LidarHotpath(hip, **P)
mc, ms = S.make_map(L, 60, 30001, seed=4711)
L.map_set(mc, ms)
pose = S.loop_pose(0.37, 0.01, -0.02)
scan = S.make_scan(100001, pose, 12345)
L.scan_upload(scan); L.scan_organize(); L.scan_extract(); L.scan_downsample()
print(L.counts())
for rep in range(2):
    r = L.scan_match(S.perturbed_guess(pose, 0))
    c = L.debug_get(pkg._abi.DBG_ICP_CYCLES, np.int64)
    names = ["pose+transform", "knn", "knn_last_batch", "math", "reduce", "total", "knn_bounds", "knn_batches",
             "solve:partials", "solve:combine", "solve:qr", "solve:pose", "solve:total"]
    print(r["iters"], {n: int(v) for n, v in zip(names, c) if n != "-"})
