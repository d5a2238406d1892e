"""clock64 phase stamps of the sector kernel (LVI_DBG_FEAT_CYCLES: workgroup 0 = ring 0, sector 0) at the bench scan size.
Run on the GPU box: python tools/diag/sector_cycles.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.import_package()
hip = pkg.load_hip()
A, S = pkg._abi, pkg.synth
g = pkg.LidarHotpath(hip, N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=65536)
pose = S.loop_pose(0.37, 0.0, -0.02)
names = ["load+reach", "compact", "rank", "wait+corner walk", "surf setup", "surf rounds", "rounds", "apply+write"]
for k in range(3):
    g.scan_upload(S.make_scan(100001, pose, 12345 + k)); g.scan_organize(); g.scan_extract(); g.sync()
    c = g.debug_get(A.DBG_FEAT_CYCLES, np.int64)
    print({n: int(v) for n, v in zip(names, c)}, "corners", g.counts()["corner"])
g.prof_enable(True)
g.scan_upload(S.make_scan(100001, pose, 777)); g.scan_organize(); g.scan_extract(); g.sync()
print({s["name"]: round(1e3 * s["total_ms"] / s["launches"], 1) for s in g.prof_read() if s["name"].startswith("feat")})
