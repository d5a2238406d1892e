"""Tracker parity soak (test infrastructure, run by hand on the GPU box: `python tests/soak_tracker.py [N] [seed]`).

N random cases of the feature_tracker path through both libraries: image sizes from 64x48 to 1280x720 (odd sizes too),
textures, motions from sub-pixel to 25 px, masks, quotas, min distances, CLAHE on / off, points on and outside the
border.  Everything on this path is integer or fixed-order f32, so every output is compared bit for bit (pyramid levels,
min-eigenvalue map, GFTT corners, LK positions / status / err of the tracked points, CLAHE image, undistorted points);
the exit code is the number of cases with any difference.  PARITY UNPINNED (OpenCV is not in the reference tree): the
checker is the CPU restatement."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as graft  # noqa: E402


def main(n_cases=None, seed=None):
    n_cases = n_cases if n_cases is not None else (int(sys.argv[1]) if len(sys.argv) > 1 else 60)
    seed = seed if seed is not None else (int(sys.argv[2]) if len(sys.argv) > 2 else 77)
    pkg = graft.import_package()
    from oracle import loader
    oracle, hip = loader.load(pkg), pkg.load_hip()
    A, S = pkg._abi, pkg.synth
    rng = np.random.default_rng(seed)
    kw = dict(max_width=1280, max_height=720, max_features=1024)
    o = pkg.TrackerHotpath(oracle, **kw); g = pkg.TrackerHotpath(hip, **kw)
    bad, report, t0 = 0, [], time.time()
    sizes = [(64, 48), (97, 61), (320, 240), (333, 251), (640, 480), (752, 480), (1280, 720), (1279, 719)]
    for case in range(n_cases):
        w, h = sizes[int(rng.integers(0, len(sizes)))]
        img0 = S.make_texture(w, h, int(rng.integers(1 << 30)))
        if rng.random() < 0.2:                                    # low-contrast / saturated regions
            img0 = np.clip(img0.astype(np.int32) * int(rng.integers(1, 4)) - int(rng.integers(0, 200)), 0, 255).astype(np.uint8)
        max_px = float(rng.choice([0.3, 2.0, 5.0, 12.0, 25.0]))
        Hm = S.small_motion_homography(w, h, int(rng.integers(1 << 20)), max_px=max_px)
        img1 = S.warp_homography(img0, Hm)
        equalize = bool(rng.random() < 0.3) and w >= 64 and h >= 64
        quota = int(rng.choice([0, 10, 150, 500, 1000]))
        min_dist = float(rng.choice([0.5, 3.0, 10.0, 20.0, 30.0]))
        quality = float(rng.choice([0.001, 0.01, 0.1]))
        mask = None
        if rng.random() < 0.4:
            mask = np.full((h, w), 255, np.uint8)
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            mask[y0:y0 + int(rng.integers(1, h)), x0:x0 + int(rng.integers(1, w))] = 0
        diffs = []
        for t in (o, g):
            t.set_equalize(equalize)
        if equalize:
            if not np.array_equal(o.clahe(img0), g.clahe(img0)):
                diffs.append("clahe")
        def corners(t):
            try:
                return t.good_features(img0, quota, quality, min_dist, mask)
            except A.LviError:                                    # unlimited quota, small distance: more corners than max_features, on both
                return None
        po, pg = corners(o), corners(g)
        if (po is None) != (pg is None):
            diffs.append("gftt capacity error on one side only")
        if po is None or pg is None:
            po = pg = np.zeros((0, 2), np.float32)
        else:
            eo, eg = o.debug_get(A.TDBG_MINEIG, np.float32), g.debug_get(A.TDBG_MINEIG, np.float32)
            if not np.array_equal(eo.view(np.uint32), eg.view(np.uint32)):
                diffs.append(f"mineig ({int((eo.view(np.uint32) != eg.view(np.uint32)).sum())} px)")
            if po.shape != pg.shape or not np.array_equal(po, pg):
                diffs.append(f"gftt {len(po)} vs {len(pg)}")
        pts = po.astype(np.float32) if len(po) else np.zeros((0, 2), np.float32)
        extra = np.stack([rng.uniform(-20, w + 20, 12), rng.uniform(-20, h + 20, 12)], axis=1).astype(np.float32)   # sub-pixel, some outside
        pts = np.concatenate([pts + rng.uniform(-0.5, 0.5, pts.shape).astype(np.float32), extra])[:1000]
        xo, so, er_o = o.lk_track(img0, img1, pts)
        xg, sg, er_g = g.lk_track(img0, img1, pts)
        if not np.array_equal(so, sg):
            diffs.append(f"lk status ({int((so != sg).sum())} of {len(so)})")
        else:
            k = so == 1
            if not np.array_equal(xo[k].view(np.uint32), xg[k].view(np.uint32)):
                diffs.append(f"lk xy ({int((xo[k].view(np.uint32) != xg[k].view(np.uint32)).any(axis=1).sum())} of {int(k.sum())})")
            if not np.array_equal(er_o[k].view(np.uint32), er_g[k].view(np.uint32)):
                diffs.append("lk err")
        # staged form, as FeatureTracker::readImage drives it (CLAHE inside push_image when EQUALIZE is set)
        res = []
        for t in (o, g):
            t.push_image(img0); t.push_image(img1); t.set_points(pts); t.run_lk()
            xy, st, er = t.get_lk()
            t.set_mask(mask)
            try:
                t.run_gftt(quota)
                gf = t.get_gftt()
            except A.LviError:                                    # unlimited quota on a large image: more corners than max_features, on both
                gf = np.full((1, 2), -1.0, np.float32)
            res.append((xy, st, er, gf))
            t.set_mask(None)
        for what, name in ((A.TDBG_PYRAMID_L1, "pyr1"), (A.TDBG_PYRAMID_L2, "pyr2"), (A.TDBG_PYRAMID_L3, "pyr3")):
            lv = []
            for t in (o, g):                                      # small images stop the pyramid early: then both must refuse
                try:
                    lv.append(t.debug_get(what, np.uint8))
                except Exception:                                 # noqa: BLE001
                    lv.append(None)
            if (lv[0] is None) != (lv[1] is None) or (lv[0] is not None and not np.array_equal(lv[0], lv[1])):
                diffs.append(name)
        (xo, so, er_o, go), (xg, sg, er_g, gg) = res
        if not np.array_equal(so, sg):
            diffs.append(f"staged lk status ({int((so != sg).sum())} of {len(so)})")
        else:
            k = so == 1
            if not np.array_equal(xo[k].view(np.uint32), xg[k].view(np.uint32)) or not np.array_equal(er_o[k].view(np.uint32), er_g[k].view(np.uint32)):
                diffs.append("staged lk xy/err")
        if go.shape != gg.shape or not np.array_equal(go, gg):
            diffs.append(f"staged gftt {len(go)} vs {len(gg)}")
        if diffs:
            bad += 1
            spec = dict(case=case, w=w, h=h, max_px=max_px, equalize=equalize, quota=quota, min_dist=min_dist, quality=quality, mask=mask is not None, diffs=diffs)
            report.append(spec)
            print("DIFF", spec, flush=True)
        if case % 10 == 9:
            print(f"[{case + 1}/{n_cases}] differing={bad} {time.time() - t0:.0f}s", flush=True)
    o.close(); g.close()
    print(json.dumps(dict(cases=n_cases, seed=seed, differing=bad, report=report)))
    return bad


if __name__ == "__main__":
    sys.exit(min(main(), 100))
