"""Parity soak (test infrastructure, run by hand on the GPU box: `python tests/soak_parity.py [N] [seed] [cases,to,print,in,detail]`).

The randomised sweep of tests/test_gpu_lidar.py::test_parity_sweep_over_scans_and_settings, longer and wider: N cases
over ring counts (4 = pipelined sector kernel, 6 / 16 = its sequential form), scan sizes, noise, thresholds, leaf sizes,
voxel modes, deskew on / off and fresh / reused handles.  Nothing is asserted case by case: every difference from
"indices / labels exact, first-stage counts exact, status and iteration count equal, pose within 1e-4 m / 1e-4 rad" is
counted, printed and put in one of the classes the reference itself leaves open:
  ties        label / index differences in a scan that holds equal curvature values inside one sector: the reference's
              order between them is whatever libstdc++'s introsort leaves (featureExtraction.cpp:171), the HIP path's is
              ascending index
  second-ds   the count of the second-stage scan grid differs by a few points: its inputs are first-stage centroids,
              equal only within the centroid tolerance, and a centroid on a voxel face changes voxel
  gn          iteration count or pose differ in the staged path although the GN path alone agrees (selected counts equal,
              pose within the bar) when both libraries get bit-identical inputs — the oracle's downsampled map and scan
              through lvi_map_set + lvi_scan_to_map: the centroid tolerance again, amplified by the break test of the GN
              loop (mapOptimization.cpp:1293-1301 stops below 0.05 deg / 0.05 cm: one more or one fewer step)
  knife       with bit-identical inputs the GN path selects one or two features more or fewer out of thousands in some
              iteration (later iterations, which start from poses that then differ by ~1e-7, up to six; final poses within
              1e-5), pose still within the bar: a feature whose test value (plane distance, weight s, 5th-neighbour
              distance) sits on its threshold, decided by the last bit of the pose — the reference accumulates A^T A in
              f32 inside cv::gemm in an order it does not specify, the HIP path in f64
Anything else is "unexplained"; the exit code is the number of unexplained cases.  PARITY UNPINNED (see oracle/
headers): the checker is the CPU restatement."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402
from helpers import make_small_scene  # noqa: E402


def sector_ties(info, curv, n_scan):
    """True when two points of one sector (featureExtraction.cpp:165-166) hold the same curvature bits"""
    st, en = info["start_ring_index"], info["end_ring_index"]
    for ring in range(n_scan):
        for j in range(6):
            sp = (int(st[ring]) * (6 - j) + int(en[ring]) * j) // 6
            ep = (int(st[ring]) * (5 - j) + int(en[ring]) * (j + 1)) // 6 - 1
            if sp < ep:
                c = curv[sp:ep + 1].view(np.uint32)
                if len(np.unique(c)) != len(c):
                    return True
    return False


def tied_sectors(info, curv, n_scan):
    """[(ring, sector, sp, ep)] of the sectors that hold two equal curvature values"""
    st, en = info["start_ring_index"], info["end_ring_index"]
    out = []
    for ring in range(n_scan):
        for j in range(6):
            sp = (int(st[ring]) * (6 - j) + int(en[ring]) * j) // 6
            ep = (int(st[ring]) * (5 - j) + int(en[ring]) * (j + 1)) // 6 - 1
            if sp < ep:
                c = curv[sp:ep + 1].view(np.uint32)
                if len(np.unique(c)) != len(c):
                    out.append((ring, j, sp, ep))
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    detail = set(int(c) for c in sys.argv[3].split(",")) if len(sys.argv) > 3 else None
    summary = run(n_cases, seed, only=detail, verbose=detail is not None)
    print(json.dumps(summary))
    return summary["classes"]["unexplained"]


def run(n_cases, seed, only=None, verbose=False):
    """execute the cases in `only` (all when None) of the seeded stream of n_cases cases; returns the summary dict"""
    pkg = graft.import_package()
    from oracle import loader
    oracle, hip = loader.load(pkg), pkg.load_hip()
    A, S = pkg._abi, pkg.synth
    rng = np.random.default_rng(seed)
    scenes = {}
    bad, worst, seam_worst, report = 0, 0.0, 0.0, []
    classes = dict(ties=0, second_ds=0, gn=0, knife=0, unexplained=0)
    t0 = time.time()
    detail = set(only) if only is not None else None
    executed = 0
    for case in range(n_cases):
        # ---- draw the whole case first (the random stream does not depend on which cases are executed)
        n_scan = int(rng.choice([4, 4, 4, 6, 16]))
        kw = dict(N_SCAN=n_scan, Horizon_SCAN=8192, max_raw_points=40000, max_map_points=400000)
        if rng.random() < 0.4:
            kw.update(edgeThreshold=float(rng.uniform(0.3, 2.0)), surfThreshold=float(rng.uniform(0.05, 0.3)))
        if rng.random() < 0.4:
            kw.update(odometrySurfLeafSize=float(rng.choice([0.2, 0.3, 0.4])), mappingSurfLeafSize=float(rng.choice([0.3, 0.4, 0.5])),
                      mappingCornerLeafSize=float(rng.choice([0.15, 0.2, 0.25])))
        mseed = int(rng.integers(0, 3))
        voxel_mode = int(rng.integers(0, 3))
        reps = []
        for rep in range(1 + int(rng.random() < 0.3)):          # sometimes a second scan on the same handles
            r = dict(pose_args=(rng.uniform(0, 6.28), rng.normal(0, 0.02), rng.normal(0, 0.02)), n_raw=int(rng.integers(6000, 30000)),
                     scan_seed=int(rng.integers(1 << 30)), noise=float(rng.choice([0.0, 0.01, 0.02, 0.05])), line_seed=int(rng.integers(1 << 30)),
                     guess_seed=int(rng.integers(1 << 20)), deskew=bool(rng.random() < 0.3), w=rng.normal(0, 0.4, 3))
            reps.append(r)
        if detail is not None and case not in detail:
            continue
        executed += 1
        if mseed not in scenes:
            scenes[mseed] = make_small_scene(pkg, oracle, seed=4711 + mseed)
        scene = scenes[mseed]
        o = pkg.LidarHotpath(oracle, **kw); g = pkg.LidarHotpath(hip, voxel_mode=voxel_mode, **kw)
        diffs, kinds = [], set()
        cascade_seen = False
        for r in reps:
            pose = S.loop_pose(*r["pose_args"])
            scan = S.make_scan(r["n_raw"], pose, r["scan_seed"], noise=r["noise"])
            if n_scan != 4:
                scan["line"] = np.random.default_rng(r["line_seed"]).integers(0, n_scan, len(scan))
            guess = S.perturbed_guess(pose, r["guess_seed"])
            deskew = r["deskew"]
            if deskew:
                t_scan = 100.0 + case
                t = np.arange(t_scan - 0.004, t_scan + 0.12, 1.0 / 200.0)
                rot = (t - t[0])[:, None] * r["w"][None, :]
            for h in (o, g):
                h.map_set(scene["map_corner"], scene["map_surf"])
                if deskew:
                    h.scan_set_deskew(t_scan, t, rot)
                h.scan_upload(scan); h.scan_organize(); h.scan_extract(); h.scan_downsample()
            n = o.counts()["n"]
            if not deskew:                                       # deskewed points may differ in the last bit (device sin/cos), which can move a label
                lo, lg = o.debug_get(A.DBG_LABEL, np.int32), g.debug_get(A.DBG_LABEL, np.int32)
                if not np.array_equal(o.debug_get(A.DBG_CORNER_INDEX, np.int32), g.debug_get(A.DBG_CORNER_INDEX, np.int32)) or not np.array_equal(lo[5:n - 5], lg[5:n - 5]):
                    tied = sector_ties(o.get_scan_info(), o.debug_get(A.DBG_CURVATURE, np.float32), n_scan)
                    diffs.append("label/index" + (" (curvature ties in a sector)" if tied else ""))
                    kinds.add("ties" if tied else "unexplained")
                if not np.array_equal(lo[5:n - 5], lg[5:n - 5]):
                    if verbose:
                        idx = np.nonzero(lo[5:n - 5] != lg[5:n - 5])[0] + 5
                        cu_o, cu_g = o.debug_get(A.DBG_CURVATURE, np.float32), g.debug_get(A.DBG_CURVATURE, np.float32)
                        po, pg = o.debug_get(A.DBG_PICKED_FINAL, np.int32), g.debug_get(A.DBG_PICKED_FINAL, np.int32)
                        oc, gc = o.debug_get(A.DBG_PICKED_OCCL, np.int32), g.debug_get(A.DBG_PICKED_OCCL, np.int32)
                        info = o.get_scan_info()
                        print("label diffs at", idx[:20], "n", n, "rings", info["start_ring_index"], info["end_ring_index"])
                        for i in idx[:8]:
                            sl = slice(max(i - 6, 0), i + 7)
                            print(" i", i, "label o/g", lo[sl], lg[sl]); print("   curv o", cu_o[sl]); print("   curv g", cu_g[sl])
                            print("   picked o/g", po[sl], pg[sl], "occl o/g", oc[sl], gc[sl]); print("   col", info["point_col_ind"][sl], "range", info["point_range"][sl])
            co, cg = o.counts(), g.counts()
            loose = abs(co.pop("surf_ds") - cg.pop("surf_ds"))
            if co != cg and not deskew and "ties" not in kinds:
                diffs.append(f"first-stage counts {co} {cg}"); kinds.add("unexplained")
            if loose > 3:
                # cause: the second-stage grid sees first-stage centroids that differ by the centroid tolerance.  With ONE input
                # (the oracle's first-stage cloud) both libraries must return the same voxel set
                s_in = o.get_features()[1]
                leaf2 = float(kw.get("mappingSurfLeafSize", 0.4))
                same_input_agrees = len(o.voxel_downsample(s_in, leaf2)) == len(g.voxel_downsample(s_in, leaf2))
                diffs.append(f"surf_ds±{loose}" + ("" if same_input_agrees else " and the counts differ on identical input"))
                kinds.add("second_ds" if loose <= 12 and same_input_agrees else "unexplained")
            ro, rg = o.scan_match(guess), g.scan_match(guess)
            # the GN path alone: the oracle's downsampled map (one point per voxel: re-voxelising it returns the same bits on
            # both sides) and the oracle's downsampled scan through the one-call seam of both libraries
            (mc, ms), (fc, fs) = o.get_map_ds(), o.get_scan_ds()
            p0 = np.asarray(guess, np.float32)
            o2 = pkg.LidarHotpath(oracle, **kw); g2 = pkg.LidarHotpath(hip, voxel_mode=voxel_mode, **kw)
            for h in (o2, g2):
                h.map_set(mc, ms)
            so, sg = o2.scan_to_map(fc, fs, p0), g2.scan_to_map(fc, fs, p0)
            if verbose:
                jo, jg = o2.debug_get(A.DBG_ICP_JTJ, np.float32).reshape(-1, 27), g2.debug_get(A.DBG_ICP_JTJ, np.float32).reshape(-1, 27)
                to, tg = o2.debug_get(A.DBG_ICP_POSE_TRACE, np.float32).reshape(-1, 6), g2.debug_get(A.DBG_ICP_POSE_TRACE, np.float32).reshape(-1, 6)
                np.set_printoptions(linewidth=250, precision=9)
                for it in range(min(so["iters"], sg["iters"])):
                    print(" iter", it, "pose o", to[it], "\n         pose g", tg[it], "\n   |d pose|", np.abs(to[it] - tg[it]))
                    print("   JtJ/Jtb words that differ:", int((jo[it].view(np.uint32) != jg[it].view(np.uint32)).sum()), "max rel", float(np.max(np.abs(jo[it] - jg[it]) / (np.abs(jo[it]) + 1e-30))))
                    if it == 0:
                        M = np.zeros((6, 6)); k = 0
                        for r_ in range(6):
                            for c_ in range(r_, 6):
                                M[r_, c_] = M[c_, r_] = jo[it][k]; k += 1
                        w = np.linalg.eigvalsh(M); print("   eig(JtJ)", w, "cond", w[-1] / w[0])
            o2.close(); g2.close()
            seam_diff = float(np.abs(so["pose"] - sg["pose"]).max())
            seam_ok = so["status"] == sg["status"] and so["iters"] == sg["iters"] and (so["status"] != 0 or seam_diff < 1e-4)
            gaps = [abs(x - y) for x, y in zip(so["n_sel"], sg["n_sel"])] if seam_ok else []
            first_gap = next((gp for gp in gaps if gp), 0)                   # the knife edge itself: same pose on both sides up to here
            # later iterations start from poses that already differ by ~1e-7 and may flip a few more near-threshold features:
            # allowed up to 6 of thousands as long as the first difference is 1 - 2 features and the final poses agree to 1e-5
            cascade_bad = bool(gaps) and (first_gap > 2 or max(gaps) > 6 or (max(gaps) > 2 and seam_diff >= 1e-5))
            nsel_gap = 0 if not gaps else (3 if cascade_bad else min(max(gaps), 2))
            cascade = bool(gaps) and max(gaps) > 2 and nsel_gap <= 2
            cascade_seen = cascade_seen or cascade
            if cascade:
                print("NOTE", case, f"knife edge with a cascade: selected counts differ by {gaps} (first {first_gap}), pose diff {seam_diff:.3e}", flush=True)
            if so["status"] == 0 and so["iters"] == sg["iters"]:
                seam_worst = max(seam_worst, seam_diff)
                if seam_diff > 1e-5:
                    print("NOTE", case, f"identical-input GN pose diff {seam_diff:.3e}", np.abs(so["pose"] - sg["pose"]), "iters", so["iters"], "degenerate", so["degenerate"], sg["degenerate"], "n_sel", so["n_sel"][-1], flush=True)
            if not seam_ok or nsel_gap > 2:
                diffs.append(f"identical-input GN: {so['status']}/{so['iters']}/{so['n_sel']} vs {sg['status']}/{sg['iters']}/{sg['n_sel']} pose diff {seam_diff:.3e}")
                kinds.add("unexplained")
            elif nsel_gap:
                diffs.append(f"identical-input GN: selected counts {so['n_sel']} vs {sg['n_sel']}, pose diff {seam_diff:.3e}")
                kinds.add("knife")
            if verbose:
                print("pose o", ro["pose"], "iters", ro["iters"], "n_sel", ro["n_sel"]); print("pose g", rg["pose"], "iters", rg["iters"], "n_sel", rg["n_sel"])
                print("diff  ", np.abs(ro["pose"] - rg["pose"]), "noise", r["noise"], "deskew", deskew, "n_raw", r["n_raw"], "voxel_mode", voxel_mode)
            if ro["status"] != rg["status"] or ro["iters"] != rg["iters"]:
                diffs.append(f"status/iters {ro['status']}/{ro['iters']} vs {rg['status']}/{rg['iters']}"); kinds.add("gn" if seam_ok else "unexplained")
            elif ro["status"] == 0:
                dp = float(np.abs(ro["pose"] - rg["pose"]).max())
                worst = max(worst, dp)
                if dp >= 1e-4:
                    diffs.append(f"pose {dp:.3e}"); kinds.add("gn" if seam_ok else "unexplained")
        o.close(); g.close()
        if diffs:
            bad += 1
            kind = next(k for k in ("unexplained", "ties", "second_ds", "gn", "knife") if k in kinds)
            classes[kind] += 1
            report.append(dict(case=case, kind=kind, params=kw, diffs=diffs, cascade=bool(cascade_seen)))
            print("DIFF", case, kind, kw, diffs, flush=True)
        if case % 10 == 9:
            print(f"[{case + 1}/{n_cases}] differing={bad} {classes} worst_pose_diff={worst:.3e} seam_worst={seam_worst:.3e} {time.time() - t0:.0f}s", flush=True)
    return dict(cases=n_cases, executed=executed, seed=seed, differing=bad, classes=classes, worst_pose_diff_staged=worst, worst_pose_diff_seam=seam_worst, report=report)


if __name__ == "__main__":
    sys.exit(min(main(), 100))
