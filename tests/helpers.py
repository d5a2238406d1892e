"""shared helpers of the test-suite"""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def xyzi(pts):
    return np.ascontiguousarray(pts).view(np.float32).reshape(-1, 4)


def small_params(**kw):
    d = dict(N_SCAN=4, Horizon_SCAN=8192, max_raw_points=40000, max_map_points=400000)
    d.update(kw)
    return d


def make_small_scene(pkg, lib, n_raw=20001, n_kf=12, seed=4711, **kw):
    """scan + frozen map generated through `lib` (normally the oracle)"""
    S = pkg.synth
    L = pkg.LidarHotpath(lib, **small_params(**kw))
    mc, ms = S.make_map(L, n_kf, n_raw, seed=seed)
    pose = S.loop_pose(0.37, 0.01, -0.02)
    scan = S.make_scan(n_raw, pose, 12345)
    guess = S.perturbed_guess(pose, 0)
    L.close()
    return dict(scan=scan, map_corner=mc, map_surf=ms, pose=pose, guess=guess)


def centroid_tol(counts, pts):
    """|Δ| <= count * 2^-23 * max|coord| per voxel"""
    m = np.abs(xyzi(pts)).max(axis=1)
    return (np.asarray(counts, np.float64) * 2.0 ** -23 * np.maximum(m, 1.0))[:, None] * 1.5 + 1e-7
