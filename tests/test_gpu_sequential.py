"""GPU tier: SURVEY §8 f-4 — the incremental local map (lvi_map_update) and the sequential caller loop (host/lvi_host.hpp
MapOptimizationNode over liblvi_hip.so) against the full assembly and against the CPU oracle.  PARITY UNPINNED (oracle/)."""
import os

import numpy as np
import pytest

from helpers import bits, small_params, xyzi

pytestmark = pytest.mark.gpu

KF_P = dict(max_keyframes=64, max_keyframe_points=600000)


def _keyframes(pkg, oracle, n, n_raw=16001):
    S = pkg.synth
    o = pkg.LidarHotpath(oracle, **small_params())
    out = []
    for k in range(n):
        pose = S.loop_pose(0.2 + 0.11 * k, 0.01 * np.sin(k), -0.01 * np.cos(k)).astype(np.float32)
        o.scan_upload(S.make_scan(n_raw, pose, 800 + k)); o.scan_organize(); o.scan_extract(); o.scan_downsample()
        c, s = o.get_scan_ds()
        out.append((c.copy(), s.copy(), pose))
    o.close()
    return out


def test_incremental_map_equals_full_assembly(pkg, oracle, hip):
    """whole keyframes entering and leaving persistent per-voxel sums give, at every step, the DS maps of the full fuse +
    VoxelGrid bit for bit (voxel set, order, centroids), and the same scan-matching record; duplicates in the list, keys
    coming back, a corrected pose and a cleared store included"""
    S = pkg.synth
    kfs = _keyframes(pkg, oracle, 12)
    a = pkg.LidarHotpath(hip, **small_params(**KF_P))          # incremental
    b = pkg.LidarHotpath(hip, **small_params(**KF_P))          # full assembly every time
    for h in (a, b):
        for c, s, pose in kfs[:10]:
            h.keyframe_add(c, s, pose)
    pose_q = S.loop_pose(0.6, 0.0, 0.01)
    scan = S.make_scan(16001, pose_q, 4242)
    guess = S.perturbed_guess(pose_q, 3)
    lists = [[0, 1, 2, 3], [0, 1, 2, 3, 4, 4, 3], [2, 3, 4, 5, 6, 5, 6], [6], [6, 7, 8, 9, 0, 1, 9, 8, 7], [9, 8, 7, 6, 5, 4, 3, 2, 1, 0], [3, 3, 3], [0, 1, 2, 3, 4, 5]]
    step = 0
    for keys in lists:
        a.map_update(keys); b.map_assemble(keys)
        assert a.counts()["map_surf_ds"] == b.counts()["map_surf_ds"] > 500 and a.counts()["map_corner_ds"] == b.counts()["map_corner_ds"]
        for x, y in zip(a.get_map_ds(), b.get_map_ds()):
            np.testing.assert_array_equal(xyzi(x).view(np.uint32), xyzi(y).view(np.uint32), err_msg=f"step {step} keys {keys}")
        recs = []
        for h in (a, b):
            h.scan_upload(scan); h.scan_organize(); h.scan_extract(); h.scan_downsample()
            recs.append(h.scan_match(guess))
        np.testing.assert_array_equal(bits(recs[0]["pose"]), bits(recs[1]["pose"]))
        assert recs[0]["n_sel"] == recs[1]["n_sel"] and recs[0]["status"] == recs[1]["status"]
        step += 1
        if step == 3:
            # correctPoses (:1650-1660): the pose of a key that is in the list changes → the sums are rebuilt
            newp = kfs[6][2].copy(); newp[3] += 0.07; newp[2] -= 0.004
            for h in (a, b):
                h.keyframe_set_pose(6, newp)
        if step == 5:
            # two more keyframes arrive (the scan just matched + one from the host)
            for h in (a, b):
                assert h.keyframe_add_current(recs[0]["pose"]) == 10
                assert h.keyframe_add(kfs[10][0], kfs[10][1], kfs[10][2]) == 11
            a.map_update([10, 11, 9]); b.map_assemble([10, 11, 9])
            for x, y in zip(a.get_map_ds(), b.get_map_ds()):
                np.testing.assert_array_equal(xyzi(x).view(np.uint32), xyzi(y).view(np.uint32))
    # an empty list, then a cleared store
    a.map_update([]); b.map_assemble([])
    assert a.counts()["map_surf_ds"] == b.counts()["map_surf_ds"] == 0
    for h in (a, b):
        h.keyframes_clear()
        h.keyframe_add(kfs[1][0], kfs[1][1], kfs[1][2])
    a.map_update([0]); b.map_assemble([0])
    for x, y in zip(a.get_map_ds(), b.get_map_ds()):
        np.testing.assert_array_equal(xyzi(x).view(np.uint32), xyzi(y).view(np.uint32))
    a.close(); b.close()


def test_incremental_map_falls_back(pkg, oracle, hip):
    """what the tables do not take — coordinates beyond +-2^20 voxels, PCL's overflow rule — goes the full way with the same result"""
    kfs = _keyframes(pkg, oracle, 2, n_raw=8001)
    a = pkg.LidarHotpath(hip, **small_params(**KF_P)); b = pkg.LidarHotpath(hip, **small_params(**KF_P))
    far = kfs[1][2].copy(); far[3] += 4.0e5                       # 400 km away: outside the voxel range, and the bbox overflows int32 cells
    for h in (a, b):
        h.keyframe_add(kfs[0][0], kfs[0][1], kfs[0][2]); h.keyframe_add(kfs[1][0], kfs[1][1], far)
    a.map_update([0, 1]); b.map_assemble([0, 1])
    assert a.counts() == b.counts()
    for x, y in zip(a.get_map_ds(), b.get_map_ds()):
        np.testing.assert_array_equal(xyzi(x).view(np.uint32), xyzi(y).view(np.uint32))
    a.map_update([0]); b.map_assemble([0])                        # and back to the incremental form
    for x, y in zip(a.get_map_ds(), b.get_map_ds()):
        np.testing.assert_array_equal(xyzi(x).view(np.uint32), xyzi(y).view(np.uint32))
    a.close(); b.close()


# --------------------------------------------------------------------------------------------- sequential replay
SEQ_P = dict(N_SCAN=4, Horizon_SCAN=8192, max_raw_points=20000, max_map_points=600000, max_keyframes=64, max_keyframe_points=600000)


@pytest.fixture(scope="module")
def hostlibs(pkg, oracle, hip, tmp_path_factory):
    H = pkg.host_api
    out = tmp_path_factory.mktemp("hostlib") / "liblvi_host_oracle.so"
    H.build_host_library(str(out), os.path.dirname(oracle.path), "lvi_oracle", extra=("-fopenmp",))
    return H.HostLibrary(str(out)), pkg.load_host()


def test_sequential_replay_hip_vs_oracle(pkg, oracle, hip, hostlibs):
    """raw Livox stream → pose → keyframe → next scan, 16 scans along the loop, through the same C++ node code over both
    libraries: the same keyframe decisions and key lists, poses within 1e-4 m / 1e-4 rad of the oracle's at every scan, and the
    incremental map (HIP) bit-identical to the full assembly (HIP) along the way"""
    H, S = pkg.host_api, pkg.synth
    h_ora, h_hip = hostlibs
    n = 16
    poses = [S.loop_pose(0.3 + 0.05 * k, 0.004 * np.sin(k), -0.004 * np.cos(k)) for k in range(n)]
    scans = [S.make_scan(16001, poses[k], 3000 + k) for k in range(n)]
    runs = {}
    for name, hl, lib, inc in (("oracle", h_ora, oracle, 0), ("hip_inc", h_hip, hip, 1), ("hip_full", h_hip, hip, 0)):
        m = H.SequentialMapper(hl, lib, pkg.default_params(lib, **SEQ_P), incremental_map=inc)
        rows = []
        for k, sc in enumerate(scans):
            r = m.scan(sc, 20.0 + 0.2 * k)
            mapds = [xyzi(c).view(np.uint32).copy() for c in m.handle.get_map_ds()] if k > 0 and name != "oracle" else None
            rows.append((r, m.keys().copy(), mapds))
        runs[name] = rows
        m.close()
    worst = 0.0
    for k in range(n):
        ro, ri, rf = runs["oracle"][k], runs["hip_inc"][k], runs["hip_full"][k]
        assert ro[0]["status"] == ri[0]["status"] == rf[0]["status"], k
        assert ro[0]["saved_keyframe"] == ri[0]["saved_keyframe"] == rf[0]["saved_keyframe"], k
        np.testing.assert_array_equal(ro[1], ri[1]); np.testing.assert_array_equal(ri[1], rf[1])
        np.testing.assert_array_equal(bits(ri[0]["pose"]), bits(rf[0]["pose"]))           # incremental == full, bit for bit
        if k > 0:
            for x, y in zip(ri[2], rf[2]):
                np.testing.assert_array_equal(x, y)
            assert ro[0]["iters"] == ri[0]["iters"], (k, ro[0]["iters"], ri[0]["iters"])
        dp = np.abs(ro[0]["pose"] - ri[0]["pose"])
        assert dp[:3].max() < 1e-4 and dp[3:].max() < 1e-4, (k, dp)
        worst = max(worst, float(dp.max()))
    assert sum(r[0]["saved_keyframe"] for r in runs["hip_inc"]) >= 5
    print(f"sequential replay: worst |pose_hip - pose_oracle| over {n} scans = {worst:.2e}")
