"""CPU tier: the node-level host code (host/lvi_host.hpp through host/lvi_seq_capi.cpp) linked with the CPU oracle:
the feature_tracker node callback (frequency control, /vins/feature/feature assembly, first-publish suppression,
rejectWithF hook; feature_tracker_node.cpp:37-231, feature_tracker.cpp:150-242) and the sequential mapOptimization
caller loop (updateInitialGuess, extractNearby, saveFrame, key-pose push; mapOptimization.cpp:298-333, 806-999,
1387-1412, 1529-1603).  The same source links liblvi_hip.so in deployment (GPU tier: test_gpu_sequential.py)."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def hostlib(pkg, oracle, tmp_path_factory):
    out = tmp_path_factory.mktemp("hostlib") / "liblvi_host_oracle.so"
    H = pkg.host_api
    H.build_host_library(str(out), os.path.dirname(oracle.path), "lvi_oracle", extra=("-fopenmp",))
    return H.HostLibrary(str(out))


CAM = dict(xi=1.9926618269451453, k1=-0.0399258932468764, k2=0.15160828121223818, p1=0.00017756967825777937, p2=-0.0011531239076798612,
           gamma1=669.8940458885896, gamma2=669.1450614220616, u0=120.0, v0=90.0)


def _frames(pkg, n, w=240, h=180):
    S = pkg.synth
    img0 = S.make_texture(w, h, 9)
    return [img0] + [S.warp_homography(img0, S.small_motion_homography(w, h, 10 + i, 2.0)) for i in range(n - 1)]


def test_frequency_control_and_message_assembly(pkg, oracle, hostlib):
    H = pkg.host_api
    w, h, FREQ = 240, 180, 10
    tp = pkg.default_tracker_params(oracle, max_width=w, max_height=h, max_cnt=60, min_dist=12.0)
    node = H.TrackerNode(hostlib, tp, h, w, FREQ, equalize=False, cam=CAM)
    frames = _frames(pkg, 6)
    dt = 1.0 / 30.0                                    # 30 Hz camera, 10 Hz publishing
    T = pkg.TrackerHotpath(oracle, max_width=w, max_height=h)
    # reference restatement of the frequency rule (feature_tracker_node.cpp:101-112) in Python
    first_time, pub_count, init_pub = None, 1, False
    outcomes, msgs = [], []
    prev_un = None
    for k in range(40):
        t = 100.0 + k * dt
        r = node.image(frames[k % len(frames)], t)
        outcomes.append(r["outcome"])
        if k == 0:
            assert r["outcome"] == "first_image"
            first_time = t
            continue
        rate = round(1.0 * pub_count / (t - first_time))
        pub = rate <= FREQ
        if pub and abs(1.0 * pub_count / (t - first_time) - FREQ) < 0.01 * FREQ:
            first_time, pub_count = t, 0
        assert r["pub_this_frame"] == pub, (k, rate)
        pts = node.points()
        if not pub:
            assert r["outcome"] == "not_published"
            continue
        pub_count += 1
        assert r["pub_count"] == pub_count
        assert r["outcome"] == ("published" if init_pub else "first_publish_suppressed")
        init_pub = True
        # ---- the message: only features seen in more than one frame; channel order id, u, v, vx, vy, depth
        old = pts[pts[:, 3] > 1]
        ch, P3 = r["channels"], r["points"]
        assert ch.shape[1] == len(old) == len(P3)
        np.testing.assert_array_equal(ch[0], old[:, 2])                      # id * NUM_OF_CAM + 0
        np.testing.assert_array_equal(ch[1], old[:, 0]); np.testing.assert_array_equal(ch[2], old[:, 1])
        assert (ch[5] == -1.0).all() and (P3[:, 2] == 1.0).all()             # no depth cloud: DepthRegister's initial value
        un = T.undistort_points(CAM, old[:, :2])
        np.testing.assert_array_equal(P3[:, :2], un)
        assert (pts[:, 2] >= 0).all() and len(set(pts[:, 2].tolist())) == len(pts)   # updateID gave every feature a unique id
        msgs.append((t, {int(i): u for i, u in zip(old[:, 2], un)}, {int(i): (vx, vy) for i, vx, vy in zip(old[:, 2], ch[3], ch[4])}))
    assert outcomes.count("published") >= 8 and outcomes.count("not_published") >= 15
    assert outcomes.count("first_publish_suppressed") == 1
    # a discontinuity (> 1 s gap) restarts the node
    r = node.image(frames[0], 100.0 + 40 * dt + 5.0)
    assert r["outcome"] == "restart"
    assert node.image(frames[0], 100.0 + 40 * dt + 5.1)["outcome"] == "first_image"
    node.close(); T.close()


def test_velocity_channel_is_undistorted_motion_over_dt(pkg, oracle, hostlib):
    """pts_velocity (feature_tracker.cpp:313-347): (un_cur - un_prev) / dt for ids present in the previous frame's map.  A new
    feature carries id -1 while its first undistortedPoints runs (updateID comes after readImage), so it enters the map under
    -1 and its velocity is 0 on its second frame too; from the third frame on it is the undistorted motion over dt"""
    H = pkg.host_api
    w, h = 240, 180
    tp = pkg.default_tracker_params(oracle, max_width=w, max_height=h, max_cnt=50, min_dist=12.0)
    node = H.TrackerNode(hostlib, tp, h, w, 100, cam=CAM)          # FREQ 100 > camera rate: every frame is a PUB frame
    T = pkg.TrackerHotpath(oracle, max_width=w, max_height=h)
    frames = _frames(pkg, 5)
    prev = {}
    checked = zero = 0
    for k in range(5):
        t = 10.0 + 0.1 * k
        r = node.image(frames[k], t)
        pts = node.points()
        un = T.undistort_points(CAM, pts[:, :2])
        cur = {int(i): u for i, u, c in zip(pts[:, 2], un, pts[:, 3]) if c > 1}       # ids that were assigned when undistortedPoints ran
        allcur = {int(i): u for i, u in zip(pts[:, 2], un)}
        if r["outcome"] in ("published", "first_publish_suppressed") and k > 1:
            ids = r["channels"][0].astype(int)
            for j, i in enumerate(ids):
                if i in prev:
                    v = ((allcur[i] - prev[i]).astype(np.float64) / (t - tprev)).astype(np.float32)      # float difference, double division (:329-330)
                    np.testing.assert_array_equal(r["channels"][3:5, j], v)
                    checked += 1
                else:
                    np.testing.assert_array_equal(r["channels"][3:5, j], [0.0, 0.0])
                    zero += 1
        prev, tprev = cur, t
    assert checked > 40 and zero > 0
    node.close(); T.close()


def test_reject_with_f_hook(pkg, oracle, hostlib):
    """rejectWithF (feature_tracker.cpp:209-242): the node-side RANSAC sees FOCAL_LENGTH-scaled undistorted points of
    cur / forw and its status vector prunes every per-feature array before setMask; without a hook the skip is counted"""
    H = pkg.host_api
    w, h = 240, 180
    tp = pkg.default_tracker_params(oracle, max_width=w, max_height=h, max_cnt=50, min_dist=12.0)
    node = H.TrackerNode(hostlib, tp, h, w, 100, cam=CAM)
    frames = _frames(pkg, 4)
    seen = []

    def ransac(un_cur, un_forw, thr):
        seen.append((un_cur.copy(), un_forw.copy(), thr))
        st = np.ones(len(un_cur), np.uint8)
        st[::3] = 0                                            # reject every third track
        return st
    node.image(frames[0], 1.0)
    node.image(frames[1], 1.1)                                 # first tracked frame: cur_pts empty before it → no call
    before = node.points()
    assert node.image(frames[2], 1.2)["rejectWithF_skipped"] == 1      # >= 8 tracked points, no hook installed
    node.set_fundamental_hook(ransac)
    before = node.points()
    node.image(frames[3], 1.3)
    after = node.points()
    assert len(seen) == 1 and seen[0][2] == 1.0                # F_THRESHOLD
    un_cur, un_forw, _ = seen[0]
    assert len(un_cur) == len(un_forw) >= 8
    # undistorted, FOCAL_LENGTH-scaled pixels around the image centre (460 * x/z + COL/2)
    assert np.abs(un_cur[:, 0] - w / 2).max() < 460 and np.abs(un_forw - un_cur).max() < 40
    kept_ids = set(after[after[:, 3] > 1][:, 2].astype(int).tolist())
    tracked_ids = before[:, 2].astype(int)
    # the rejected tracks (every third of those that survived LK, in order) are gone; LK losses aside, the others survive
    assert len(kept_ids) <= len(tracked_ids) - len(un_cur[::3]) + 0
    assert len(kept_ids) >= len(un_cur) - len(un_cur[::3]) - 2
    node.close()


# --------------------------------------------------------------------------------------------- sequential mapOptimization loop
def _trajectory(pkg, n, step=0.055):
    S = pkg.synth
    poses = [S.loop_pose(0.3 + step * k, 0.004 * np.sin(k), -0.004 * np.cos(k)) for k in range(n)]
    scans = [S.make_scan(8001, poses[k], 3000 + k) for k in range(n)]
    return poses, scans


SEQ_P = dict(N_SCAN=4, Horizon_SCAN=4096, max_raw_points=9000, max_map_points=400000, max_keyframes=64, max_keyframe_points=400000)


def test_sequential_loop_with_the_oracle(pkg, oracle, hostlib):
    """raw Livox stream → pose → keyframe → next scan: 14 scans 0.52 m apart at 10 Hz.  Keyframe decisions follow saveFrame's
    thresholds, the key list follows extractNearby, poses track the ground truth, and the incremental map entry point gives
    the same trajectory as the full assembly"""
    H = pkg.host_api
    poses, scans = _trajectory(pkg, 14)
    out = {}
    for inc in (0, 1):
        m = H.SequentialMapper(hostlib, oracle, pkg.default_params(oracle, **SEQ_P), incremental_map=inc)
        rows = []
        for k, sc in enumerate(scans):
            # the first pose is the map origin: scans are expressed relative to pose 0 by giving the node the true first pose as IMU-free start
            r = m.scan(sc, 50.0 + 0.2 * k)
            rows.append((r, m.keys().copy()))
        out[inc] = (rows, m.keyposes())
        m.close()
    rows, kp = out[0]
    assert rows[0][0]["status"] == pkg._abi.LVI_NO_MAP and rows[0][0]["saved_keyframe"] and rows[0][0]["n_keys"] == 0
    assert all(r["processed"] for r, _ in rows)
    # the map frame is the first scan's frame: compare relative motion with the ground truth
    R0 = pkg.synth.rot_zyx(*poses[0][:3]); t0 = poses[0][3:]
    for k in range(1, len(scans)):
        r = rows[k][0]
        assert r["status"] == 0, (k, r)
        rel_t = R0.T @ (poses[k][3:] - t0)
        assert np.abs(r["pose"][3:] - rel_t).max() < 0.08, (k, r["pose"][3:], rel_t)
    # saveFrame: a keyframe whenever the pose moved >= 1 m or turned >= 0.2 rad since the last keyframe (the scans are 0.2 s apart, so the
    # LIVOX 1-s rule does not fire)
    last = np.zeros(6)
    n_kf = 1
    for k in range(1, len(scans)):
        p = rows[k][0]["pose"].astype(np.float64)
        moved = np.linalg.norm(p[3:] - last[3:]) >= 1.0 or np.abs(p[:3] - last[:3]).max() >= 0.2
        if rows[k][0]["saved_keyframe"]:
            n_kf += 1
            last = p
        # yaw changes 0.055 rad and position 0.52 m per scan: the distance rule decides; allow the rotation rule's frame mixing
        assert rows[k][0]["saved_keyframe"] == bool(moved) or np.abs(p[:3] - last[:3]).max() > 0.15, (k, p, last)
    assert n_kf == len(kp) >= 5
    # extractNearby: every key is within the search radius; the newest keyframes (10-s rule) are all listed
    for k in range(1, len(scans)):
        keys = rows[k][1]
        assert len(keys) >= 1 and keys.max() < rows[k - 1][0]["n_keyframes"]
    # incremental map entry point: identical trajectory (the oracle implements it as the full assembly)
    for (ra, ka), (rb, kb) in zip(out[0][0], out[1][0]):
        np.testing.assert_array_equal(ra["pose"], rb["pose"]); np.testing.assert_array_equal(ka, kb)


def test_mapping_interval_gate_and_livox_rule(pkg, oracle, hostlib):
    """mappingProcessInterval drops scans that arrive too early (:311-314); a keyframe is forced after 1 s (LIVOX, :1392-1396)"""
    H = pkg.host_api
    poses, scans = _trajectory(pkg, 6, step=0.002)             # almost standing still
    m = H.SequentialMapper(hostlib, oracle, pkg.default_params(oracle, **SEQ_P), mapping_process_interval=0.15)
    stamps = [0.0, 0.05, 0.2, 0.3, 0.5, 1.4]
    res = [m.scan(sc, 10.0 + t) for sc, t in zip(scans, stamps)]
    assert [r["processed"] for r in res] == [True, False, True, False, True, True]
    assert [r["saved_keyframe"] for r in res if r["processed"]] == [True, False, False, True]     # first scan; then only the > 1 s rule
    m.close()
