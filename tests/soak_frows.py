"""Soak of the widened rows f-1 / f-3 / f-4 (test infrastructure, by hand on the GPU box: `python tests/soak_frows.py [N] [seed]`).

  f-1  organise with IMU deskew: random angular rates up to 6 rad/s, IMU tables of 2 .. 400 samples whose span starts
       before / inside the scan and ends inside / after it (findRotation's clamps, imageProjection.cpp:650-672), sample
       times that coincide with point times.  Ring starts / ends, columns, ranges: bit for bit.  Points: bit for bit where
       the device's double-rounded sin / cos equal libm's sinf / cosf, never more than 4e-6 of the coordinate scale apart.
  f-3  MEI liftProjective with random intrinsics / distortion / mirror parameter: bit for bit (NaN where the reference
       gives NaN).
  f-4  keyframe store + map assembly: random keyframe clouds, poses, index lists (repeats, any order): the fused raw
       clouds bit for bit (host-libm matrices on both sides), downstream counts equal.
The exit code is the number of differing cases.  PARITY UNPINNED: the checker is the CPU restatement."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402
from helpers import xyzi  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    pkg = graft.import_package()
    from oracle import loader
    oracle, hip = loader.load(pkg), pkg.load_hip()
    A, S = pkg._abi, pkg.synth
    rng = np.random.default_rng(seed)
    bad, report, t0 = 0, [], time.time()
    stats = dict(deskew_points=0, deskew_points_bit_equal=0, deskew_worst_rel=0.0, mei_points=0, assemble_points=0)

    # ------------------------------------------------------------------ f-1
    P = dict(N_SCAN=4, Horizon_SCAN=8192, max_raw_points=40000, max_map_points=400000)
    o = pkg.LidarHotpath(oracle, **P); g = pkg.LidarHotpath(hip, **P)
    for case in range(n_cases):
        n_raw = int(rng.integers(2, 30000))
        scan = S.make_scan(n_raw, S.loop_pose(rng.uniform(0, 6.28), 0.01, -0.02), int(rng.integers(1 << 30)))
        t_scan = float(rng.uniform(0, 2000))
        span = float(scan["offset_time"].max()) * 1e-9 if "offset_time" in scan.dtype.names else 0.1
        n_imu = int(rng.choice([2, 3, 20, 400]))
        start = t_scan + float(rng.choice([-0.01, -0.0005, 0.0, 0.3 * span]))
        stop = t_scan + float(rng.choice([0.5 * span, span, span + 0.02]))
        t = np.linspace(start, max(stop, start + 1e-3), n_imu)
        w = np.cumsum(rng.normal(0, 0.5, (n_imu, 3)), axis=0) * 0.1 + rng.uniform(-6, 6, 3)
        rot = np.concatenate([[np.zeros(3)], np.cumsum(w[1:] * np.diff(t)[:, None], axis=0)])
        diffs = []
        a, b = o.organize_scan_deskew(scan, t_scan, t, rot), g.organize_scan_deskew(scan, t_scan, t, rot)
        if a["n"] != b["n"]:
            diffs.append(f"n {a['n']} vs {b['n']}")
        else:
            for k in ("start_ring_index", "end_ring_index", "point_col_ind"):
                if not np.array_equal(a[k], b[k]):
                    diffs.append(k)
            if not np.array_equal(a["point_range"].view(np.uint32), b["point_range"].view(np.uint32)):
                diffs.append("point_range")
            pa, pb = xyzi(a["cloud_deskewed"]), xyzi(b["cloud_deskewed"])
            if len(pa):
                if not np.array_equal(pa[:, 3], pb[:, 3]):
                    diffs.append("intensity")
                rel = float(np.abs(pa[:, :3] - pb[:, :3]).max() / max(float(np.abs(pa[:, :3]).max()), 1e-6))
                stats["deskew_worst_rel"] = max(stats["deskew_worst_rel"], rel)
                stats["deskew_points"] += len(pa)
                stats["deskew_points_bit_equal"] += int((pa[:, :3].view(np.uint32) == pb[:, :3].view(np.uint32)).all(axis=1).sum())
                if rel > 4e-6:
                    diffs.append(f"points off by {rel:.2e} of the coordinate scale")
        if diffs:
            bad += 1
            report.append(dict(part="deskew", case=case, n_raw=n_raw, n_imu=n_imu, diffs=diffs)); print("DIFF", report[-1], flush=True)
    o.close(); g.close()
    print(f"deskew done, differing={bad}, {time.time() - t0:.0f}s", flush=True)

    # ------------------------------------------------------------------ f-3
    kw = dict(max_width=1280, max_height=720, max_features=1024)
    to, tg = pkg.TrackerHotpath(oracle, **kw), pkg.TrackerHotpath(hip, **kw)
    for case in range(n_cases):
        cam = dict(xi=float(rng.choice([0.0, 1.0, rng.uniform(0.5, 2.5)])), k1=float(rng.normal(0, 0.1)), k2=float(rng.normal(0, 0.1)),
                   p1=float(rng.normal(0, 1e-3)), p2=float(rng.normal(0, 1e-3)), gamma1=float(rng.uniform(200, 900)), gamma2=float(rng.uniform(200, 900)),
                   u0=float(rng.uniform(300, 700)), v0=float(rng.uniform(200, 400)))
        if rng.random() < 0.2:
            cam.update(k1=0.0, k2=0.0, p1=0.0, p2=0.0)
        n = int(rng.choice([1, 2, 150, 1000]))
        xy = np.stack([rng.uniform(-50, 1330, n), rng.uniform(-50, 770, n)], axis=1).astype(np.float32)
        a, b = to.undistort_points(cam, xy), tg.undistort_points(cam, xy)
        stats["mei_points"] += n
        ok = ~np.isnan(a)
        if not np.array_equal(np.isnan(a), np.isnan(b)) or not np.array_equal(a.view(np.uint32)[ok], b.view(np.uint32)[ok]):
            bad += 1
            report.append(dict(part="mei", case=case, cam=cam, differing_points=int((a.view(np.uint32)[ok] != b.view(np.uint32)[ok]).sum()))); print("DIFF", report[-1], flush=True)
    to.close(); tg.close()
    print(f"mei done, differing={bad}, {time.time() - t0:.0f}s", flush=True)

    # ------------------------------------------------------------------ f-4
    for case in range(max(n_cases // 5, 3)):
        o = pkg.LidarHotpath(oracle, **P); g = pkg.LidarHotpath(hip, **P)
        n_kf = int(rng.integers(1, 30))
        for k in range(n_kf):
            nc, ns = int(rng.integers(0, 600)), int(rng.integers(1, 9000))
            c = np.zeros((nc, 4), np.float32); c[:, :3] = rng.uniform(-40, 40, (nc, 3)); c[:, 3] = rng.uniform(0, 255, nc)
            s = np.zeros((ns, 4), np.float32); s[:, :3] = rng.uniform(-40, 40, (ns, 3)) * [1, 1, 0.1]; s[:, 3] = rng.uniform(0, 255, ns)
            pose = np.concatenate([rng.normal(0, 0.05, 2), rng.uniform(-3.2, 3.2, 1), rng.uniform(-30, 30, 2), rng.normal(0, 0.5, 1)]).astype(np.float32)
            for h in (o, g):
                h.keyframe_add(c, s, pose)
        if rng.random() < 0.5:                                   # loop-closure style correction of a stored pose
            k = int(rng.integers(0, n_kf))
            pose = np.concatenate([rng.normal(0, 0.05, 2), rng.uniform(-3.2, 3.2, 1), rng.uniform(-30, 30, 2), rng.normal(0, 0.5, 1)]).astype(np.float32)
            for h in (o, g):
                h.keyframe_set_pose(k, pose)
        keys = rng.integers(0, n_kf, int(rng.integers(1, min(2 * n_kf, 40) + 1))).astype(np.int32)
        if rng.random() < 0.5:
            keys = np.unique(keys)
        diffs = []
        for h in (o, g):
            h.map_assemble(keys)
        for what, name in ((A.DBG_MAP_CORNER_RAW, "corner raw"), (A.DBG_MAP_SURF_RAW, "surf raw")):
            ra, rb = o.debug_get(what, np.float32), g.debug_get(what, np.float32)
            stats["assemble_points"] += len(ra) // 4
            if ra.shape != rb.shape or not np.array_equal(ra.view(np.uint32), rb.view(np.uint32)):
                diffs.append(name)
        co, cg = o.counts(), g.counts()
        if (co["map_corner_ds"], co["map_surf_ds"]) != (cg["map_corner_ds"], cg["map_surf_ds"]):
            diffs.append(f"map ds counts {co['map_corner_ds']}/{co['map_surf_ds']} vs {cg['map_corner_ds']}/{cg['map_surf_ds']}")
        o.close(); g.close()
        if diffs:
            bad += 1
            report.append(dict(part="assemble", case=case, n_kf=n_kf, keys=[int(k) for k in keys], diffs=diffs)); print("DIFF", report[-1], flush=True)
    print(json.dumps(dict(cases_per_part=n_cases, seed=seed, differing=bad, stats=stats, report=report)))
    return bad


if __name__ == "__main__":
    sys.exit(min(main(), 100))
