"""CPU tier: known-answer and property tests of the lidar oracle (SURVEY §8 a-0 … a-10).
Ground truth here is hand-computed or derived independently in numpy from the reference source
lines quoted in each test — not from the oracle itself."""
import numpy as np
import pytest

from helpers import make_small_scene, small_params, xyzi


@pytest.fixture()
def L(pkg, oracle):
    h = pkg.LidarHotpath(oracle, **small_params())
    yield h
    h.close()


def _livox(pkg, xyz, line, refl=None):
    p = np.zeros(len(xyz), pkg._abi.LIVOX_DTYPE)
    p["x"], p["y"], p["z"] = np.asarray(xyz, np.float32).T
    p["line"] = line
    p["reflectivity"] = 7 if refl is None else refl
    return p


# ----------------------------------------------------------------------------- a-0
def test_organize_hand_case(pkg, L):
    # 9 message points; the LAST one is dropped (imageProjection.cpp:249); one too near, one too far, one bad line
    xyz = [[5, 0, 0], [0, 6, 0], [0.2, 0, 0], [0, 0, 7], [200, 0, 0], [3, 4, 0], [1, 1, 1], [8, 0, 0], [9, 9, 9]]
    line = [1, 0, 0, 1, 2, 0, 7, 3, 0]
    info = L.organize_scan(_livox(pkg, xyz, line, refl=np.arange(9)))
    # survivors in message order: i0(r1) i1(r0) i3(r1) i5(r0) i7(r3); ring-major, stable inside a ring
    assert info["n"] == 5
    np.testing.assert_array_equal(xyzi(info["cloud_deskewed"])[:, :3], [[0, 6, 0], [3, 4, 0], [5, 0, 0], [0, 0, 7], [8, 0, 0]])
    np.testing.assert_array_equal(xyzi(info["cloud_deskewed"])[:, 3], [1, 5, 0, 3, 7])          # intensity = reflectivity
    np.testing.assert_array_equal(info["point_range"], [6, 5, 5, 7, 8])
    np.testing.assert_array_equal(info["point_col_ind"], [0, 1, 0, 1, 0])                        # dense per-ring counter
    # start = count-1+5, end = count-1-5 (imageProjection.cpp:630,645)
    np.testing.assert_array_equal(info["start_ring_index"], [4, 6, 8, 8])
    np.testing.assert_array_equal(info["end_ring_index"], [-4, -2, -2, -1])


def test_organize_range_gate_is_inclusive(pkg, L):
    # range < min || range > max is rejected: exactly 1.0 and exactly 100.0 survive
    info = L.organize_scan(_livox(pkg, [[1, 0, 0], [100, 0, 0], [0.99999, 0, 0], [100.001, 0, 0], [0, 0, 0]], [0] * 5))
    assert info["n"] == 2


def test_organize_horizon_truncation(pkg, oracle):
    h = pkg.LidarHotpath(oracle, N_SCAN=2, Horizon_SCAN=3, max_raw_points=64, max_map_points=64)
    xyz = [[2 + i, 0, 0] for i in range(10)] + [[0, 0, 0]]
    info = h.organize_scan(_livox(pkg, xyz, [0] * 5 + [1] * 5 + [0]))
    assert info["n"] == 6                                                       # 3 columns per ring
    np.testing.assert_array_equal(info["point_range"], [2, 3, 4, 7, 8, 9])
    h.close()


# ----------------------------------------------------------------------------- f-1 (IMU deskew)
def _imu_table(t0, t1, rate_hz, omega):
    """imuDeskewInfo (imageProjection.cpp:354-410) for a constant angular velocity omega [rad/s]: rot = omega * (t - t[0])"""
    t = np.arange(t0, t1 + 1e-9, 1.0 / rate_hz)
    return t, (t - t[0])[:, None] * np.asarray(omega, np.float64)[None, :]


def test_deskew_zero_rotation_is_identity(pkg, L):
    S = pkg.synth
    scan = S.make_scan(3001, S.loop_pose(0.3), 5)
    t, rot = _imu_table(99.99, 100.12, 200.0, [0, 0, 0])
    a = L.organize_scan(scan)
    b = L.organize_scan_deskew(scan, 100.0, t, rot)
    np.testing.assert_array_equal(xyzi(a["cloud_deskewed"]).view(np.uint32), xyzi(b["cloud_deskewed"]).view(np.uint32))
    np.testing.assert_array_equal(a["point_col_ind"], b["point_col_ind"])
    c = L.organize_scan(scan)                      # the plain entry switches deskew off again
    np.testing.assert_array_equal(xyzi(a["cloud_deskewed"]).view(np.uint32), xyzi(c["cloud_deskewed"]).view(np.uint32))


def test_deskew_constant_yaw_rate_against_float64(pkg, L):
    """deskewPoint (:538-568): p' = R(rot(t_first))^-1 R(rot(t)) p with rot interpolated in the table (:495-520)"""
    S = pkg.synth
    scan = S.make_scan(4001, S.loop_pose(1.1), 9)
    t0 = 250.0
    omega = np.array([0.0, 0.0, 0.8])
    t, rot = _imu_table(t0 - 0.004, t0 + 0.12, 400.0, omega)
    info = L.organize_scan_deskew(scan, t0, t, rot)
    plain = L.organize_scan(scan)
    assert info["n"] == plain["n"] > 3000
    np.testing.assert_array_equal(info["point_range"].view(np.uint32), plain["point_range"].view(np.uint32))   # range is taken before the deskew (:582,:616)
    # independent float64 model: per-point time, linear interpolation of the table, yaw-only rotation
    kept = scan[:-1]
    rng = np.sqrt(kept["x"].astype(np.float64) ** 2 + kept["y"].astype(np.float64) ** 2 + kept["z"].astype(np.float64) ** 2)
    ok = (rng >= 1.0) & (rng <= 100.0)
    tt = t0 + (kept["offset_time"].astype(np.float64) * 1e-9).astype(np.float32).astype(np.float64)
    yaw = np.interp(tt, t, rot[:, 2])
    first = np.flatnonzero(ok)[0]
    dy = yaw - yaw[first]
    x = kept["x"] * np.cos(dy) - kept["y"] * np.sin(dy)
    y = kept["x"] * np.sin(dy) + kept["y"] * np.cos(dy)
    want = []
    for ring in range(4):
        m = ok & (kept["line"] == ring)
        want.append(np.stack([x[m], y[m], kept["z"][m].astype(np.float64)], axis=1))
    want = np.concatenate(want)
    got = xyzi(info["cloud_deskewed"])[:, :3].astype(np.float64)
    assert np.abs(got - want).max() < 2e-5
    assert np.abs(got - xyzi(plain["cloud_deskewed"])[:, :3]).max() > 0.05       # the deskew did something


def test_deskew_table_edges(pkg, L):
    """point times before the first / after the last table entry take that entry (:507-511)"""
    xyz = [[5, 0, 0], [0, 6, 0], [4, 4, 0], [0, 0, 0]]
    p = _livox(pkg, xyz, [0, 0, 0, 0])
    p["offset_time"] = [0, 50_000_000, 100_000_000, 0]
    t = np.array([10.02, 10.04, 10.06])                       # all three point times: 10.00 (before), 10.05 (inside), 10.10 (after)
    rot = np.array([[0, 0, 0.0], [0, 0, 0.1], [0, 0, 0.3]])
    info = L.organize_scan_deskew(p, 10.0, t, rot)
    got = xyzi(info["cloud_deskewed"])[:, :3].astype(np.float64)
    for k, yaw in enumerate([0.0, 0.2, 0.3]):                 # first point: entry 0 (yaw 0) is the reference frame
        c, s = np.cos(yaw), np.sin(yaw)
        want = [xyz[k][0] * c - xyz[k][1] * s, xyz[k][0] * s + xyz[k][1] * c, 0.0]
        assert np.abs(got[k] - want).max() < 1e-5, (k, got[k], want)
    with pytest.raises(Exception):
        L.organize_scan_deskew(p, 10.0, t[:1], rot[:1])       # imuPointerCur <= 0 is "not available" in the reference (:406)


# ----------------------------------------------------------------------------- a-1, a-2
def test_smoothness_and_occlusion_against_numpy(pkg, L):
    A = pkg._abi
    S = pkg.synth
    scan = S.make_scan(12001, S.loop_pose(0.9), 3)
    L.scan_upload(scan); L.scan_organize(); L.scan_extract()
    info = L.get_scan_info()
    r = info["point_range"]; col = info["point_col_ind"]; n = info["n"]
    curv = L.debug_get(A.DBG_CURVATURE, np.float32)
    i = np.arange(5, n - 5)
    d = (((r[i - 2] + r[i - 1]) - r[i] * np.float32(4)) + r[i + 1]) + r[i + 2]                # featureExtraction.cpp:99-101, f32 left to right
    np.testing.assert_array_equal(curv[5:n - 5], d * d)
    # featureExtraction.cpp:113-148, written as the scatter it is
    pk = np.zeros(n, np.int32)
    for k in range(5, n - 6):
        if abs(int(col[k + 1]) - int(col[k])) < 10:
            if float(r[k] - r[k + 1]) > 0.3:
                pk[k - 1] = pk[k] = 1
            elif float(r[k + 1] - r[k]) > 0.3:
                pk[k + 1] = pk[k + 2] = 1
        if abs(r[k - 1] - r[k]) > 0.1 * float(r[k]) and abs(r[k + 1] - r[k]) > 0.1 * float(r[k]):
            pk[k] = 1
    got = L.debug_get(A.DBG_PICKED_OCCL, np.int32)
    np.testing.assert_array_equal(got[5:n - 5], pk[5:n - 5])


# ----------------------------------------------------------------------------- a-3
def test_extract_features_invariants(pkg, L):
    A = pkg._abi
    S = pkg.synth
    scan = S.make_scan(20001, S.loop_pose(0.37, 0.01, -0.02), 12345)
    L.scan_upload(scan); L.scan_organize(); L.scan_extract()
    info = L.get_scan_info()
    n = info["n"]
    curv = L.debug_get(A.DBG_CURVATURE, np.float32)
    occl = L.debug_get(A.DBG_PICKED_OCCL, np.int32)
    label = L.debug_get(A.DBG_LABEL, np.int32)
    cidx = L.debug_get(A.DBG_CORNER_INDEX, np.int32)
    corner, surf = L.get_features()
    assert len(cidx) == len(corner) > 50
    np.testing.assert_array_equal(xyzi(corner), xyzi(info["cloud_deskewed"][cidx]))
    assert (curv[cidx] > 1.0).all() and (occl[cidx] == 0).all()                                 # :177
    assert (label[cidx] == 1).all() and (label == 1).sum() == len(cidx)
    # per sector: at most 40, descending curvature except that position ep is visited first (:171-185)
    pos = 0
    for ring in range(4):
        s, e = int(info["start_ring_index"][ring]), int(info["end_ring_index"][ring])
        for j in range(6):
            sp = (s * (6 - j) + e * j) // 6
            ep = (s * (5 - j) + e * (j + 1)) // 6 - 1
            mine = [k for k in cidx[pos:] if sp <= k <= ep]
            take = []
            for k in cidx[pos:]:
                if sp <= k <= ep:
                    take.append(k)
                else:
                    break
            pos += len(take)
            assert len(take) <= 40 and take == mine[:len(take)]
            body = [k for k in take if k != ep]
            assert all(curv[a] >= curv[b] for a, b in zip(body, body[1:]))
            # neighbour suppression: two corners of one sector are more than 5 apart
            t = np.sort(np.array(take))
            assert len(t) < 2 or np.diff(t).min() > 5
    assert pos == len(cidx)
    # surf candidates = every non-corner point inside a sector, then a 0.4 m voxel grid per ring (:231-243)
    assert 0 < len(surf) < n


def test_forty_corner_cap(pkg, oracle):
    """a ring made only of isolated spikes: every sector hits the cap of 40 (featureExtraction.cpp:180)"""
    A = pkg._abi
    h = pkg.LidarHotpath(oracle, N_SCAN=1, Horizon_SCAN=8192, max_raw_points=16384, max_map_points=64)
    n = 6001
    r = np.full(n, 10.0, np.float32)
    r[::12] = 10.2                                   # spike every 12 points: curvature (4*0.2)^2... make it larger
    r[::12] = 10.29                                  # d = -4*0.29 -> curv 1.35 > 1.0, depth jumps stay <= 0.3 (not occluded)
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    xyz = np.stack([r * np.cos(ang), r * np.sin(ang), np.zeros(n)], 1)
    p = _livox(pkg, xyz, [0] * n)
    info = h.organize_scan(p)
    c, s = h.extract_features(info)
    assert len(c) == 6 * 40
    h.close()


# ----------------------------------------------------------------------------- a-4
def test_voxel_grid_hand_case(pkg, L):
    A = pkg._abi
    # leaf 0.5 -> inverse 2.0; min = (-0.6, 0.1, 0) -> min_b = (-2, 0, 0)
    pts = np.array([[-0.6, 0.1, 0.0, 1], [-0.55, 0.2, 0.1, 3], [0.6, 0.1, 0.0, 5], [0.2, 0.9, 0.3, 7], [0.3, 0.8, 0.4, 9]], np.float32)
    out = xyzi(L.voxel_downsample(pts, 0.5))
    keys = L.debug_get(A.DBG_VOXEL_KEYS, np.int32)
    # ijk = floor(p*2) - min_b ; div_b = (4, 2, 1) ; idx = i + 4*j
    np.testing.assert_array_equal(keys, [0, 0, 3, 6, 6])
    np.testing.assert_array_equal(L.debug_get(A.DBG_VOXEL_CELLS, np.int32), [0, 3, 6])
    np.testing.assert_array_equal(L.debug_get(A.DBG_VOXEL_COUNTS, np.int32), [2, 1, 2])
    np.testing.assert_allclose(out, [[-0.575, 0.15, 0.05, 2], [0.6, 0.1, 0.0, 5], [0.25, 0.85, 0.35, 8]], rtol=1e-6)


def test_voxel_grid_properties(pkg, L):
    A = pkg._abi
    rng = np.random.default_rng(9)
    pts = np.zeros((20000, 4), np.float32)
    pts[:, :3] = rng.uniform(-30, 30, (20000, 3)) * [1, 1, 0.05]
    pts[:, 3] = rng.uniform(0, 255, 20000)
    out = xyzi(L.voxel_downsample(pts, 0.4))
    cells = L.debug_get(A.DBG_VOXEL_CELLS, np.int32)
    counts = L.debug_get(A.DBG_VOXEL_COUNTS, np.int32)
    keys = L.debug_get(A.DBG_VOXEL_KEYS, np.int32)
    assert np.all(np.diff(cells) > 0), "output order is ascending voxel idx"
    assert counts.sum() == len(pts) and len(out) == len(cells)
    np.testing.assert_array_equal(np.unique(keys), cells)
    # centroid of centroids weighted by count = centroid of the input (checksum of checksums)
    np.testing.assert_allclose((out * counts[:, None]).sum(0) / len(pts), pts.mean(0), rtol=1e-4, atol=1e-4)
    # every centroid lies inside its voxel: same key when re-voxelised alone
    inv = np.float32(1.0) / np.float32(0.4)
    mn = np.floor(pts[:, :3].min(0) * inv)
    ijk = np.floor(out[:, :3] * inv) - mn
    div = np.floor(pts[:, :3].max(0) * inv) - mn + 1
    np.testing.assert_array_equal((ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]).astype(np.int32), cells)
    # overflow rule and empty input
    far = np.array([[0, 0, 0, 1], [3000, 3000, 3000, 2]], np.float32)
    np.testing.assert_array_equal(xyzi(L.voxel_downsample(far, 0.01)), far)
    assert len(L.voxel_downsample(np.zeros((0, 4), np.float32), 0.4)) == 0


# ----------------------------------------------------------------------------- a-5
def test_transform_cloud_matches_float64(pkg, L):
    rng = np.random.default_rng(5)
    pts = rng.uniform(-20, 20, (1000, 4)).astype(np.float32)
    pose = np.array([0.03, -0.05, 1.1, 2.0, -3.0, 0.7])
    out = xyzi(L.transform_cloud(pts, pose))
    R = pkg.synth.rot_zyx(*pose[:3])
    np.testing.assert_allclose(out[:, :3], pts[:, :3].astype(np.float64) @ R.T + pose[3:], atol=2e-5)
    np.testing.assert_array_equal(out[:, 3], pts[:, 3])


# ----------------------------------------------------------------------------- f-4 (keyframe store, map assembly)
def _keyframes(pkg, L, n=5, n_raw=6001):
    S = pkg.synth
    kfs = []
    for k in range(n):
        pose = S.loop_pose(0.2 + 0.11 * k, 0.01 * k, -0.005 * k)
        L.scan_upload(S.make_scan(n_raw, pose, 900 + k)); L.scan_organize(); L.scan_extract(); L.scan_downsample()
        c, s = L.get_scan_ds()
        kfs.append((c.copy(), s.copy(), pose.astype(np.float32)))
    return kfs


def test_map_assemble_is_transform_and_concatenate(pkg, L):
    """extractCloud (mapOptimization.cpp:931-957): listed keyframes, in list order, each through transformPointCloud"""
    A = pkg._abi
    kfs = _keyframes(pkg, L)
    for i, (c, s, pose) in enumerate(kfs):
        assert L.keyframe_add(c, s, pose) == i
    assert L.keyframe_count() == (len(kfs), sum(len(c) + len(s) for c, s, _ in kfs))
    order = [3, 0, 4, 0, 1]                                     # any order, duplicates allowed
    L.map_assemble(order)
    raw_c, raw_s = L.debug_get(A.DBG_MAP_CORNER_RAW, A.PT_DTYPE), L.debug_get(A.DBG_MAP_SURF_RAW, A.PT_DTYPE)
    want_c = np.concatenate([xyzi(pkg.synth.transform_points(kfs[k][0], kfs[k][2].astype(np.float64))) for k in order])
    want_s = np.concatenate([xyzi(pkg.synth.transform_points(kfs[k][1], kfs[k][2].astype(np.float64))) for k in order])
    assert len(raw_c) == len(want_c) and len(raw_s) == len(want_s) > 1000
    np.testing.assert_allclose(xyzi(raw_c), want_c, atol=3e-5)          # f32 transform vs the float64 model
    np.testing.assert_allclose(xyzi(raw_s), want_s, atol=3e-5)
    # … followed by exactly what lvi_map_set does with those clouds
    ds_a = [x.copy() for x in L.get_map_ds()]
    L.map_set(raw_c, raw_s)
    for a, b in zip(ds_a, L.get_map_ds()):
        np.testing.assert_array_equal(xyzi(a).view(np.uint32), xyzi(b).view(np.uint32))
    # a corrected pose (correctPoses :1650-1660) moves that keyframe's points only
    p2 = kfs[0][2].copy(); p2[3] += 1.0
    L.keyframe_set_pose(0, p2); L.map_assemble([0, 1])
    moved = xyzi(L.debug_get(A.DBG_MAP_SURF_RAW, A.PT_DTYPE))
    n0 = len(kfs[0][1])
    np.testing.assert_allclose(moved[:n0, 0], want_s[len(kfs[3][1]):len(kfs[3][1]) + n0, 0] + 1.0, atol=3e-5)
    with pytest.raises(pkg.LviError):
        L.map_assemble([7])
    L.keyframes_clear()
    assert L.keyframe_count() == (0, 0)


def test_keyframe_add_current_uses_the_scan_ds_clouds(pkg, L):
    S = pkg.synth
    pose = S.loop_pose(0.5)
    L.scan_upload(S.make_scan(6001, pose, 31)); L.scan_organize(); L.scan_extract(); L.scan_downsample()
    c, s = L.get_scan_ds()
    assert L.keyframe_add_current(pose) == 0
    assert L.keyframe_count() == (1, len(c) + len(s))
    L.map_assemble([0])
    A = pkg._abi
    np.testing.assert_allclose(xyzi(L.debug_get(A.DBG_MAP_SURF_RAW, A.PT_DTYPE)), xyzi(S.transform_points(s, pose)), atol=3e-5)


# ----------------------------------------------------------------------------- a-6 … a-10
def test_knn_debug_is_masked_exact_knn(pkg, L):
    rng = np.random.default_rng(6)
    m = np.zeros((20000, 4), np.float32)
    m[:, :3] = rng.uniform(-15, 15, (20000, 3)) * [1, 1, 0.1]
    L.map_set(m, m)
    mc, ms = L.get_map_ds()
    q = np.zeros((500, 4), np.float32)
    q[:, :3] = rng.uniform(-16, 16, (500, 3)) * [1, 1, 0.1]
    idx, sqd = L.debug_knn(1, q)
    ref = xyzi(ms)[:, :3]
    for i in range(len(q)):
        d = ((q[i, :3][None] - ref) ** 2).sum(1)
        order = np.argsort(d, kind="stable")[:5]
        near = d[order] < 1.0
        assert ((idx[i] >= 0) == near).all()
        np.testing.assert_allclose(sqd[i][near], d[order][near], rtol=1e-5)
        assert set(idx[i][near]) == set(order[near])


def test_residual_closed_forms(pkg, oracle):
    """5 collinear map points -> point-to-line residual; a flat patch -> point-to-plane residual
    (mapOptimization.cpp:1054-1092, 1130-1163)"""
    h = pkg.LidarHotpath(oracle, **small_params(edgeFeatureMinValidNum=0, surfFeatureMinValidNum=0))
    # corner map: points on the vertical line x=2,y=3 ; surf map: plane z=-2 (grid 0.41: one point per 0.4 voxel).
    # (LOAM's plane model a x + b y + c z + 1 = 0 cannot represent a plane through the origin.)
    zs = np.arange(-2, 2.01, 0.21, dtype=np.float32)
    line = np.stack([np.full_like(zs, 2), np.full_like(zs, 3), zs, np.zeros_like(zs)], 1)
    g = np.arange(-3, 3.01, 0.41, dtype=np.float32)
    gx, gy = np.meshgrid(g, g)
    plane = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, -2, np.float32), np.zeros(gx.size, np.float32)], 1)
    h.map_set(line, plane)
    corner = np.array([[2.3, 3.4, 0.1, 0]], np.float32)              # 0.5 m from the line
    surf = np.array([[0.1, 0.2, -1.7, 0]], np.float32)               # 0.3 m above the plane
    h.scan_to_map(corner, surf, np.zeros(6, np.float32))             # fills the DS scan clouds (returns TOO_FEW… soft status)
    cc, fc = h.debug_residuals(0, np.zeros(6, np.float32))
    cs, fs = h.debug_residuals(1, np.zeros(6, np.float32))
    assert fc[0] == 1 and fs[0] == 1
    s = 1 - 0.9 * 0.5
    np.testing.assert_allclose(xyzi(cc)[0], [s * 0.6, s * 0.8, 0.0, s * 0.5], atol=2e-4)
    s2 = 1 - 0.9 * 0.3 / np.sqrt(np.sqrt(0.01 + 0.04 + 1.7 * 1.7))
    got = xyzi(cs)[0]
    np.testing.assert_allclose(np.abs(got[:3]), [0, 0, s2], atol=2e-4)
    np.testing.assert_allclose(got[2] * got[3], s2 * s2 * 0.3, atol=2e-4)      # sign of normal and distance agree
    h.close()


def test_scan_to_map_recovers_injected_pose(pkg, oracle):
    sc = make_small_scene(pkg, oracle, n_raw=20001, n_kf=12)
    h = pkg.LidarHotpath(oracle, **small_params())
    h.map_set(sc["map_corner"], sc["map_surf"])
    h.scan_upload(sc["scan"]); h.scan_organize(); h.scan_extract(); h.scan_downsample()
    r = h.scan_match(sc["guess"])
    assert r["status"] == 0 and r["converged"] and not r["degenerate"] and 2 <= r["iters"] <= 20
    assert np.abs(r["pose"][3:] - sc["pose"][3:]).max() < 0.03          # 2 cm range noise, 1 cm map pose noise
    assert np.abs(r["pose"][:3] - sc["pose"][:3]).max() < 0.003
    assert np.abs(sc["guess"][3:] - sc["pose"][3:]).max() > 0.05         # the guess really was off
    # deterministic
    r2 = h.scan_match(sc["guess"])
    np.testing.assert_array_equal(r["pose"], r2["pose"])
    # IMU roll/pitch blending with weight 0.01 (mapOptimization.cpp:1345-1367): moves 1 % of the way
    imu = dict(imu_available=1, roll=float(r["pose"][0]) + 0.1, pitch=float(r["pose"][1]) - 0.2, yaw=0.0)
    r3 = h.scan_match(sc["guess"], imu)
    assert abs((r3["pose"][0] - r["pose"][0]) - 0.001) < 2e-5 and abs((r3["pose"][1] - r["pose"][1]) + 0.002) < 2e-5
    # clamps (mapOptimization.cpp:1370-1372)
    hc = pkg.LidarHotpath(oracle, **small_params(z_tollerance=0.5))
    hc.map_set(sc["map_corner"], sc["map_surf"])
    hc.scan_upload(sc["scan"]); hc.scan_organize(); hc.scan_extract(); hc.scan_downsample()
    assert hc.scan_match(sc["guess"])["pose"][5] == pytest.approx(0.5)
    h.close(); hc.close()


def test_soft_outcomes_and_errors(pkg, oracle):
    A = pkg._abi
    h = pkg.LidarHotpath(oracle, **small_params())
    with pytest.raises(pkg.LviError) as e:
        h.scan_organize()
    assert e.value.code == A.LVI_ERR_STATE
    with pytest.raises(pkg.LviError) as e:
        h.scan_upload(np.zeros(10 ** 6, A.LIVOX_DTYPE))
    assert e.value.code == A.LVI_ERR_CAPACITY
    few = np.zeros((5, 4), np.float32)
    assert h.scan_to_map(few, few, np.zeros(6))["status"] == A.LVI_NO_MAP                      # mapOptimization.cpp:1317
    h.map_set(np.ones((30, 4), np.float32), np.ones((30, 4), np.float32))
    assert h.scan_to_map(few, few, np.zeros(6))["status"] == A.LVI_TOO_FEW_FEATURES            # :1320
    h.close()


def test_batch_entry_points_are_a_loop(pkg, oracle):
    """lvi_scan_batch_* on the oracle: the reference processes scans one by one, a batch is exactly that"""
    from helpers import make_small_scene, small_params
    S = pkg.synth
    sc = make_small_scene(pkg, oracle, n_raw=6001, n_kf=5, Horizon_SCAN=2048)
    P = small_params(Horizon_SCAN=2048, max_raw_points=8192, icp_max_iters=6, icp_disable_break=1, batch_scans=3)
    poses = [S.loop_pose(0.37 + 0.5 * k) for k in range(3)]
    scans = [S.make_scan(6001 - 500 * k, poses[k], 70 + k) for k in range(3)]
    guesses = np.stack([S.perturbed_guess(poses[k], k) for k in range(3)])
    h = pkg.LidarHotpath(oracle, **P)
    h.map_upload(sc["map_corner"], sc["map_surf"]); h.map_build()
    ref = []
    for k in range(3):
        h.map_build(); h.scan_upload(scans[k]); h.scan_organize(); h.scan_extract(); h.scan_downsample()
        ref.append(h.scan_match(guesses[k]))
    h.batch_upload(scans); h.batch_run(guesses, 0, rebuild_map=True)
    rec = h.batch_get_records(3)
    for k in range(3):
        np.testing.assert_array_equal(rec[k, :6], ref[k]["pose"])
        assert int(rec[k, 7:8].view(np.int32)[0]) == 6
    out = np.zeros((3, 8), np.float32)
    h.batch_run(guesses, out.ctypes.data, rebuild_map=False)          # "device" records of the CPU library are host memory
    np.testing.assert_array_equal(out, rec)
    with pytest.raises(pkg.LviError):
        h.batch_upload(scans + scans)
    h.close()


def test_map_share_is_the_owners_map(pkg, oracle):
    """lvi_map_share on the oracle: the sharing handle matches against the owner's clouds (copied here, aliased on the GPU)"""
    from helpers import make_small_scene, small_params
    S = pkg.synth
    sc = make_small_scene(pkg, oracle, n_raw=6001, n_kf=5, Horizon_SCAN=2048)
    P = small_params(Horizon_SCAN=2048, max_raw_points=8192, icp_max_iters=6, icp_disable_break=1)
    pose = S.loop_pose(0.37)
    scan, guess = S.make_scan(6001, pose, 70), S.perturbed_guess(pose, 1)
    a, b = pkg.LidarHotpath(oracle, **P), pkg.LidarHotpath(oracle, **P)
    with pytest.raises(pkg.LviError):
        b.map_share(a)                                                 # the owner holds no map yet
    a.map_upload(sc["map_corner"], sc["map_surf"]); a.map_build()
    b.map_share(a); b.map_build()
    res = []
    for h in (a, b):
        h.scan_upload(scan); h.scan_organize(); h.scan_extract(); h.scan_downsample()
        res.append(h.scan_match(guess))
    np.testing.assert_array_equal(res[0]["pose"], res[1]["pose"])
    assert a.counts()["map_surf_ds"] == b.counts()["map_surf_ds"] > 0
    with pytest.raises(pkg.LviError):
        a.map_share(a)
    a.close(); b.close()
