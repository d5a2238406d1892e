"""diagnostic (not a test): phase cycles of the sector kernel on a 100k-pt scan"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.import_package(); hip = pkg.load_hip(); S = pkg.synth
L = pkg.LidarHotpath(hip, N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=1 << 20)
scan = S.make_scan(100001, S.loop_pose(0.37, 0.01, -0.02), 12345)
for rep in range(2):
    L.scan_upload(scan); L.scan_organize(); L.scan_extract(); L.sync()
    c = L.debug_get(pkg._abi.DBG_FEAT_CYCLES, np.int64)
    names = ["load", "compact", "rank", "walk", "fp_init", "fp_rounds_cyc", "n_rounds", "apply+store"]
    print({n: int(v) for n, v in zip(names, c)})
