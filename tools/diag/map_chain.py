"""per-kernel HIP-event times of the map re-voxelisation chain at the bench map size, one handle, nothing else running.
python tools/diag/map_chain.py [bins ...]  (GPU box);  LVI_VB_BINS=pts,max to sweep the bin geometry; LVI_DIAG_BATCH=S builds S slots per launch"""
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
if os.environ.get("LVI_DIAG_BINSTAT"):
    for nm, cloud, leaf in (("corner", mc, 0.2), ("surf", ms, 0.4)):
        xyz = np.stack([cloud["x"], cloud["y"], cloud["z"]], 1).astype(np.float32)
        inv = np.float32(1.0) / np.float32(leaf)
        cell = np.floor(xyz * inv).astype(np.int64)
        mn, mx = cell.min(0), cell.max(0)
        d = mx - mn + 1
        key = (cell[:, 0] - mn[0]) + (cell[:, 1] - mn[1]) * d[0] + (cell[:, 2] - mn[2]) * d[0] * d[1]
        ncells = int(d[0] * d[1] * d[2])
        cnt = np.bincount(key >> 10)
        occ = cnt[cnt > 0]
        nv = np.array([len(np.unique(key[(key >> 10) == b])) for b in np.argsort(-cnt)[:5]])
        print(nm, "div", d, "ncells", ncells, "bins", len(cnt), "occupied", len(occ), "points/bin pct 10/50/90/99/max", np.percentile(occ, [10, 50, 90, 99, 100]).astype(int),
              "bins<=1024:", int((occ <= 1024).sum()), "pts in them", int(occ[occ <= 1024].sum()), "bins<=2048:", int((occ <= 2048).sum()), "pts", int(occ[occ <= 2048].sum()),
              "bins>4096:", int((occ > 4096).sum()), "pts", int(occ[occ > 4096].sum()), "chunks", int(np.ceil(occ / 4096).sum()), "voxels in 5 heaviest bins", nv)
        uk, uc = np.unique(key, return_counts=True)
        vb = np.bincount(uk >> 10)
        print("   voxels per occupied bin pct 50/90/99/max", np.percentile(vb[vb > 0], [50, 90, 99, 100]).astype(int))
    sys.exit(0)
for bins in sys.argv[1:] or ["2048,1024"]:
    os.environ["LVI_VB_BINS"] = bins
    g = pkg.LidarHotpath(hip, **dict(P, batch_scans=int(os.environ.get("LVI_DIAG_BATCH", "1"))))
    g.map_upload(mc, ms)
    for _ in range(3):
        g.map_build()
    g.sync(); g.prof_enable(True)
    best = {}
    for rep in range(4):                                     # four groups of 8 builds: the smallest group mean per kernel (a stall of the box hits one group)
        g.prof_reset()
        for _ in range(8):
            g.map_build()
        for s in g.prof_read():
            v = 1e3 * s["total_ms"] / s["launches"]
            best[s["name"]] = min(best.get(s["name"], 1e9), v)
    st = sorted(best.items(), key=lambda kv: -kv[1])
    print(bins, g.counts()["map_surf_ds"], "chain us %.1f:" % sum(v for _, v in st), ", ".join("%s %.1f" % kv for kv in st[:10]))
    g.close()
