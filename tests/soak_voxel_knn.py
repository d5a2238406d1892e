"""Voxel-grid / KNN parity soak (test infrastructure, by hand on the GPU box: `python tests/soak_voxel_knn.py [N] [seed]`).

N random cases through lvi_voxel_downsample (keys, distinct cells, counts and output order bit for bit, centroids inside
count * 2^-23 * max|coord|, the overflow rule) over cloud shapes the fixed tests only sample: 1 .. 60 k points, uniform /
clustered / planar / scan-line ordered / duplicated points, points placed exactly on voxel faces, offsets up to +-2 km,
leaf sizes 0.02 .. 5 m, every voxel mode; then, on a map with one point per voxel (identical centroids on both sides),
the 5 nearest neighbours of random queries (indices and squared-distance bits, queries with tied distances excepted).
The exit code is the number of cases with a difference.  PARITY UNPINNED (PCL / FLANN are not in the reference tree):
the checker is the CPU restatement."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402
from helpers import centroid_tol, xyzi  # noqa: E402


def make_cloud(rng, n):
    kind = int(rng.integers(0, 6))
    pts = np.zeros((n, 4), np.float32)
    if kind == 0:                                    # uniform box, flat in z like a lidar map
        pts[:, :3] = rng.uniform(-1, 1, (n, 3)) * rng.choice([5.0, 40.0, 150.0]) * [1, 1, 0.1]
    elif kind == 1:                                  # a few tight clusters: many points per voxel
        c = rng.uniform(-30, 30, (int(rng.integers(1, 8)), 3))
        pts[:, :3] = c[rng.integers(0, len(c), n)] + rng.normal(0, rng.choice([0.01, 0.3, 2.0]), (n, 3))
    elif kind == 2:                                  # planes (walls / ground)
        pts[:, :3] = rng.uniform(-40, 40, (n, 3))
        pts[:, int(rng.integers(0, 3))] = rng.choice([-3.0, 0.0, 7.25]) + rng.normal(0, 0.01, n)
    elif kind == 3:                                  # scan-line order: long runs of consecutive points in one voxel
        t = np.sort(rng.uniform(0, 1, n))
        pts[:, 0] = 30 * np.cos(12 * t); pts[:, 1] = 30 * np.sin(12 * t); pts[:, 2] = 3 * t
    elif kind == 4:                                  # exact duplicates and points exactly on voxel faces (multiples of 0.2)
        base = rng.integers(-100, 100, (max(n // 4, 1), 3)).astype(np.float32) * np.float32(0.2)
        pts[:, :3] = base[rng.integers(0, len(base), n)]
    else:                                            # far from the origin: large |coord|, coarse float spacing
        pts[:, :3] = rng.uniform(-20, 20, (n, 3)) + rng.choice([-2000.0, 500.0, 2000.0], 3)
    pts[:, 3] = rng.uniform(0, 255, n) if rng.random() < 0.8 else 7.0
    return pts


def main(n_cases=None, seed=None):
    n_cases = n_cases if n_cases is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 200)
    seed = seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 31)
    pkg = graft.import_package()
    from oracle import loader
    oracle, hip = loader.load(pkg), pkg.load_hip()
    A = pkg._abi
    rng = np.random.default_rng(seed)
    kw = dict(N_SCAN=4, Horizon_SCAN=1000, max_raw_points=4096, max_map_points=65536)
    o = pkg.LidarHotpath(oracle, **kw)
    gs = [pkg.LidarHotpath(hip, voxel_mode=m, **kw) for m in (0, 1, 2)]
    bad, report, t0 = 0, [], time.time()
    for case in range(n_cases):
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 1024, 4097, int(rng.integers(1, 60000))]))
        pts = make_cloud(rng, n)
        leaf = float(rng.choice([0.02, 0.1, 0.2, 0.4, 1.0, 5.0]))
        diffs = []
        vo = o.voxel_downsample(pts, leaf)
        ko, co, cnt = o.debug_get(A.DBG_VOXEL_KEYS, np.int32), o.debug_get(A.DBG_VOXEL_CELLS, np.int32), o.debug_get(A.DBG_VOXEL_COUNTS, np.int32)
        for mode, g in enumerate(gs):
            vg = g.voxel_downsample(pts, leaf)
            if not np.array_equal(ko, g.debug_get(A.DBG_VOXEL_KEYS, np.int32)):
                diffs.append(f"mode{mode} keys")
            if not np.array_equal(co, g.debug_get(A.DBG_VOXEL_CELLS, np.int32)):
                diffs.append(f"mode{mode} cells")
            if not np.array_equal(cnt, g.debug_get(A.DBG_VOXEL_COUNTS, np.int32)):
                diffs.append(f"mode{mode} counts")
            if len(vo) != len(vg):
                diffs.append(f"mode{mode} n {len(vo)} vs {len(vg)}")
            elif len(vo) and len(cnt) == 0:              # PCL's overflow rule: output = input
                if not np.array_equal(xyzi(vo).view(np.uint32), xyzi(vg).view(np.uint32)):
                    diffs.append(f"mode{mode} overflow output")
            elif len(vo) and len(cnt) == len(vo):
                d = np.abs(xyzi(vo).astype(np.float64) - xyzi(vg))
                if not np.all(d <= centroid_tol(cnt, vo)):
                    diffs.append(f"mode{mode} centroid off by {float((d / centroid_tol(cnt, vo)).max()):.2f} x tolerance")
                # a voxel holding one point returns that point up to the input quantum of the fixed-point sums
                # (2^-38 of the power of two above the bounding-box extent): far below one f32 ulp of anything
                # but coordinates within millimetres of zero
                single = cnt == 1
                ext = float(np.max(pts[:, :3].max(axis=0) - pts[:, :3].min(axis=0))) if n else 0.0
                quantum = 2.0 ** (np.ceil(np.log2(max(ext, 1e-3))) + 1 - 38)
                if single.any() and float(np.abs(xyzi(vo)[single, :3].astype(np.float64) - xyzi(vg)[single, :3]).max()) > max(quantum, 1e-30) * 1.01 + 0.5 * float(np.spacing(np.float32(np.abs(xyzi(vo)[single, :3]).max()))):
                    diffs.append(f"mode{mode} single-point voxels off by more than the input quantum")
        if diffs:
            bad += 1
            report.append(dict(case=case, part="voxel", n=n, leaf=leaf, diffs=diffs))
            print("DIFF", report[-1], flush=True)
        if case % 20 == 19:
            print(f"[{case + 1}/{n_cases}] differing={bad} {time.time() - t0:.0f}s", flush=True)
    o.close()
    for g in gs:
        g.close()

    # ---- KNN: maps with at most one point per voxel of either leaf, so that both sides hold identical centroids
    P = dict(N_SCAN=4, Horizon_SCAN=8192, max_raw_points=40000, max_map_points=400000)
    knn_cases, knn_queries, knn_ties = max(n_cases // 20, 3), 0, 0
    for case in range(knn_cases):
        o = pkg.LidarHotpath(oracle, **P); g = pkg.LidarHotpath(hip, **P)
        span = int(rng.choice([60, 200, 400]))
        m = np.zeros((int(rng.integers(2000, 80000)), 4), np.float32)
        m[:, :3] = (rng.integers(-span, span, (len(m), 3)) * 0.25 + 0.125) * [1, 1, float(rng.choice([0.2, 1.0]))]
        m = np.unique(m, axis=0)
        _, first = np.unique(np.floor(m[:, :3] / np.float32(0.4)).astype(np.int64), axis=0, return_index=True)
        m = m[np.sort(first)]
        for h in (o, g):
            h.map_set(m, m)
        diffs = []
        for (a, b) in zip(o.get_map_ds(), g.get_map_ds()):
            if not np.array_equal(xyzi(a).view(np.uint32), xyzi(b).view(np.uint32)):
                diffs.append("map ds differs")
        nq = 6000
        q = np.zeros((nq, 4), np.float32)
        q[:, :3] = rng.uniform(-span * 0.27, span * 0.27, (nq, 3)) * [1, 1, 0.2] + rng.normal(0, 1e-3, (nq, 3))
        q[: nq // 10, :3] = xyzi(o.get_map_ds()[0])[: nq // 10, :3]          # queries sitting exactly on map points
        for which in (0, 1):
            io, do = o.debug_knn(which, q)
            ig, dg = g.debug_knn(which, q)
            ties = np.array([len(np.unique(r[np.isfinite(r)])) < np.isfinite(r).sum() for r in do])
            ok = ~ties
            knn_queries += int(ok.sum()); knn_ties += int(ties.sum())
            if not np.array_equal(do[ok].view(np.uint32), dg[ok].view(np.uint32)):
                diffs.append(f"knn{which} distances")
            rows = np.nonzero(ok & (io != ig).any(axis=1))[0]
            if len(rows):
                # same distance bits at every rank but another index: only legitimate when the 5th and the (unreturned)
                # 6th neighbour are equally far — check that each side's indices really have the distances it reports
                mp = xyzi(o.get_map_ds()[which])[:, :3]
                def sqd(idx):
                    d = q[rows, None, :3] - mp[idx[rows]]
                    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]).astype(np.float32)
                genuine = np.array_equal(sqd(io).view(np.uint32), do[rows].view(np.uint32)) and np.array_equal(sqd(ig).view(np.uint32), dg[rows].view(np.uint32))
                only_last = bool(((io[rows] != ig[rows])[:, :4].sum() == 0) or np.array_equal(do[rows].view(np.uint32), dg[rows].view(np.uint32)))
                if genuine and only_last:
                    knn_ties += len(rows); knn_queries -= len(rows)
                else:
                    diffs.append(f"knn{which} indices ({len(rows)} queries)")
            # tied queries: the same distances must come back, whatever the order of the tied indices
            if not np.array_equal(np.sort(do[ties], axis=1).view(np.uint32), np.sort(dg[ties], axis=1).view(np.uint32)):
                diffs.append(f"knn{which} distances of tied queries")
        o.close(); g.close()
        if diffs:
            bad += 1
            report.append(dict(case=case, part="knn", map_points=len(m), diffs=diffs))
            print("DIFF", report[-1], flush=True)
    print(json.dumps(dict(voxel_cases=n_cases, knn_maps=knn_cases, knn_queries_compared=knn_queries, knn_queries_with_tied_distances=knn_ties,
                          seed=seed, differing=bad, report=report)))
    return bad


if __name__ == "__main__":
    sys.exit(min(main(), 100))
