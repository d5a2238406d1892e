"""per-kernel HIP-event times of the map re-voxelisation chain at the bench map size, one handle, nothing else running.
python tools/diag/map_chain.py  (GPU box);  LVI_VB_BINS=pts,max to sweep the bin geometry"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.import_package()
hip = pkg.load_hip()
S = pkg.synth
P = dict(N_SCAN=4, Horizon_SCAN=32768, max_raw_points=131072, max_map_points=5_065_536)
cache = "/tmp/lvi_bench_map.npz"
if os.path.exists(cache):
    z = np.load(cache); mc, ms = z["mc"], z["ms"]
else:
    g0 = pkg.LidarHotpath(hip, **dict(P, max_map_points=65536))
    mc, ms = S.make_map(g0, 250, 30001, seed=4711, target_surf=5_000_000)
    g0.close()
    np.savez(cache, mc=mc, ms=ms)
for bins in sys.argv[1:] or ["2048,1024"]:
    os.environ["LVI_VB_BINS"] = bins
    g = pkg.LidarHotpath(hip, **P)
    g.map_upload(mc, ms)
    for _ in range(3):
        g.map_build()
    g.sync(); g.prof_enable(True)
    for _ in range(5):
        g.map_build()
    st = sorted(g.prof_read(), key=lambda s: -s["total_ms"])
    tot = sum(s["total_ms"] for s in st) / 5 * 1e3
    print(bins, g.counts()["map_surf_ds"], "chain us %.1f:" % tot, ", ".join("%s %.1f" % (s["name"], 1e3 * s["total_ms"] / s["launches"]) for s in st[:10]))
    g.close()
