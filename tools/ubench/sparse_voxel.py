"""experiment (not a test): binned vs sorted voxel grid on sparse outdoor-sized grids, kernel time from the library profiler"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.import_package(); hip = pkg.load_hip()
rng = np.random.default_rng(1)
n = 4_000_000
# points on "surfaces": ground plane + scattered vertical structures over a 200 m x 200 m x 20 m extent
xy = rng.uniform(-100, 100, (n, 2))
z = np.where(rng.random(n) < 0.6, rng.normal(0, 0.05, n), rng.uniform(0, 20, n))
pts = np.zeros((n, 4), np.float32); pts[:, :2] = xy; pts[:, 2] = z; pts[:, 3] = rng.uniform(0, 255, n)
for leaf in (0.4, 0.2, 0.1):
    for mode in (1, 2):
        L = pkg.LidarHotpath(hip, N_SCAN=4, Horizon_SCAN=1024, max_raw_points=4096, max_map_points=n + 16, voxel_mode=mode, max_keyframes=0)
        L.voxel_downsample(pts[:1000], leaf)
        L.prof_enable(True); L.prof_reset()
        out = L.voxel_downsample(pts, leaf)
        st = L.prof_read()
        tot = sum(s["total_ms"] for s in st)
        print(f"leaf {leaf} mode {'sorted' if mode == 1 else 'binned'}: {len(out)} voxels, kernels {1e3 * tot:.0f} us", flush=True)
        L.close()
