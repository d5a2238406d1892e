"""CPU tier: known-answer tests of the tracker oracle: pyramid arithmetic,
LK on images with known motion, min-eigenvalue / GFTT on synthetic corners."""
import numpy as np
import pytest


@pytest.fixture()
def T(pkg, oracle):
    t = pkg.TrackerHotpath(oracle, max_width=1280, max_height=720)
    yield t
    t.close()


def _pyrdown_numpy(img):
    """[1 4 6 4 1]x[1 4 6 4 1], REFLECT_101, (sum + 128) >> 8, size (w+1)/2 x (h+1)/2 — independent of the oracle"""
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(img.astype(np.int64), 2, mode="reflect")
    h, w = img.shape
    tmp = sum(k[i] * p[:, i:i + w] for i in range(5))
    full = sum(k[j] * tmp[j:j + h, :] for j in range(5))
    return ((full[::2, ::2] + 128) >> 8).astype(np.uint8)


def test_pyramid_levels(pkg, T):
    A = pkg._abi
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (121, 163), dtype=np.uint8)          # odd sizes exercise (w+1)/2
    T.push_image(img)
    l1 = T.debug_get(A.TDBG_PYRAMID_L1, np.uint8).reshape(61, 82)
    np.testing.assert_array_equal(l1, _pyrdown_numpy(img))
    l2 = T.debug_get(A.TDBG_PYRAMID_L2, np.uint8).reshape(31, 41)
    np.testing.assert_array_equal(l2, _pyrdown_numpy(l1))
    # next size would be 21x16 <= 21: buildOpticalFlowPyramid stops, level 3 does not exist
    with pytest.raises(pkg.LviError):
        T.debug_get(A.TDBG_PYRAMID_L3, np.uint8)
    flat = np.full((64, 64), 77, np.uint8)
    T.push_image(flat)
    assert (T.debug_get(A.TDBG_PYRAMID_L1, np.uint8) == 77).all()


def test_lk_recovers_known_translation(pkg, T):
    S = pkg.synth
    big = S.make_texture(400, 300, 11).astype(np.float64)
    # smooth the texture a little so that sub-pixel shifts by bilinear resampling are well defined
    big = (big + np.roll(big, 1, 0) + np.roll(big, 1, 1) + np.roll(np.roll(big, 1, 0), 1, 1)) / 4
    for dx, dy in ((2.0, -3.0), (0.5, 0.25), (-4.75, 1.5), (7.0, 6.0)):
        ys, xs = np.mgrid[20:260, 20:340].astype(np.float64)
        def samp(sx, sy):
            x0 = np.floor(sx).astype(int); y0 = np.floor(sy).astype(int)
            fx = sx - x0; fy = sy - y0
            return (big[y0, x0] * (1 - fx) * (1 - fy) + big[y0, x0 + 1] * fx * (1 - fy) + big[y0 + 1, x0] * (1 - fx) * fy + big[y0 + 1, x0 + 1] * fx * fy)
        a = (samp(xs, ys) + 0.5).astype(np.uint8)
        b = (samp(xs - dx, ys - dy) + 0.5).astype(np.uint8)               # content moves by (+dx, +dy)
        pts = T.good_features(a, 60, 0.01, 15.0)
        pts = pts[(pts[:, 0] > 30) & (pts[:, 0] < 290) & (pts[:, 1] > 30) & (pts[:, 1] < 210)]
        assert len(pts) > 20
        xy, st, err = T.lk_track(a, b, pts)
        assert st.mean() > 0.95
        flow = xy[st == 1] - pts[st == 1]
        assert np.abs(np.median(flow, 0) - [dx, dy]).max() < 0.05, (dx, dy, np.median(flow, 0))
        assert np.quantile(np.linalg.norm(flow - [dx, dy], axis=1), 0.9) < 0.2


def test_lk_status_rules(pkg, T):
    S = pkg.synth
    img = S.make_texture(200, 160, 5)
    flat = np.full_like(img, 100)
    pts = np.array([[100.0, 80.0], [-50.0, 10.0], [260.0, 10.0], [0.0, 0.0], [199.0, 159.0]], np.float32)
    xy, st, err = T.lk_track(img, img, pts)
    # a point whose window origin lies more than the window size outside the image is rejected at level 0
    assert st[1] == 0 and st[2] == 0 and st[0] == 1
    np.testing.assert_allclose(xy[0], pts[0], atol=0.02)                # identical images: no motion
    assert err[0] == 0.0
    xy, st, err = T.lk_track(flat, flat, pts)
    assert st.sum() == 0                                                # min eigenvalue below 1e-4 everywhere


def test_min_eig_and_gftt_on_checkerboard(pkg, T):
    A = pkg._abi
    sq = 24
    h, w = 8 * sq, 10 * sq
    yy, xx = np.mgrid[0:h, 0:w]
    img = (((yy // sq) + (xx // sq)) % 2 * 200 + 20).astype(np.uint8)
    pts = T.good_features(img, 0, 0.05, 10.0)
    eig = T.debug_get(A.TDBG_MINEIG, np.float32).reshape(h, w)
    assert eig.min() >= -1e-6
    # strong response only near the lattice crossings, ~zero along edges and inside squares
    assert eig[sq * 3 - 1: sq * 3 + 1, sq * 4 - 1: sq * 4 + 1].max() > 100 * eig[sq * 3 + sq // 2, sq * 4]
    assert eig[sq * 3 + sq // 2, sq * 4 + sq // 2] == 0.0
    inner = {(x, y) for x in range(1, 10) for y in range(1, 8)}
    got = set()
    for x, y in pts:
        gx, gy = round(x / sq), round(y / sq)
        assert abs(x - gx * sq) <= 1.5 and abs(y - gy * sq) <= 1.5, (x, y)
        got.add((gx, gy))
    assert got == inner
    # quota and descending quality
    p5 = T.good_features(img, 5, 0.05, 10.0)
    np.testing.assert_array_equal(p5, pts[:5])
    vals = eig[pts[:, 1].astype(int), pts[:, 0].astype(int)]
    assert np.all(np.diff(vals) <= 0)


def test_gftt_min_distance_mask_and_threshold(pkg, T):
    S = pkg.synth
    img = S.make_texture(320, 240, 4242)
    for md in (8.0, 20.0, 33.0):
        p = T.good_features(img, 0, 0.01, md)
        d = np.linalg.norm(p[:, None] - p[None], axis=2) + np.eye(len(p)) * 1e9
        assert d.min() >= md
    mask = np.zeros((240, 320), np.uint8)
    mask[60:180, 100:260] = 255
    p = T.good_features(img, 0, 0.01, 10.0, mask)
    assert len(p) > 5
    assert ((p[:, 0] >= 100) & (p[:, 0] < 260) & (p[:, 1] >= 60) & (p[:, 1] < 180)).all()
    # without the distance filter (minDistance < 1) a higher quality level keeps a subset
    big = pkg.TrackerHotpath(T.lib, max_width=320, max_height=240, max_features=4096)
    lo = {tuple(q) for q in big.good_features(img, 0, 0.05, 0.5)}
    hi = {tuple(q) for q in big.good_features(img, 0, 0.3, 0.5)}
    assert hi < lo
    # corners are never on the outermost pixel ring (featureselect.cpp scans 1..size-2)
    p = big.good_features(img, 0, 0.02, 0.5)
    assert p[:, 0].min() >= 1 and p[:, 1].min() >= 1 and p[:, 0].max() <= 318 and p[:, 1].max() <= 238
    # capacity is an error, not a silent truncation
    with pytest.raises(pkg.LviError):
        T.good_features(img, 0, 0.0001, 0.5)
    big.close()


# ----------------------------------------------------------------------------- f-2 CLAHE
def _clahe_numpy(img, clip, tiles):
    """independent vectorised restatement of cv::CLAHE::apply (8-bit): per-tile clipped histogram → LUT → f32 bilinear blend"""
    tx_n, ty_n = tiles
    H, W = img.shape
    ext = img
    if W % tx_n or H % ty_n:
        ext = np.pad(img, ((0, ty_n - H % ty_n), (0, tx_n - W % tx_n)), mode="reflect")
    th, tw = ext.shape[0] // ty_n, ext.shape[1] // tx_n
    area = tw * th
    scale = np.float32(255) / np.float32(area)
    limit = max(int(clip * area / 256), 1) if clip > 0 else 0
    lut = np.zeros((ty_n, tx_n, 256), np.uint8)
    for ty in range(ty_n):
        for tx in range(tx_n):
            h = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if limit > 0:
                clipped = int(np.maximum(h - limit, 0).sum())
                h = np.minimum(h, limit)
                batch, residual = divmod(clipped, 256)
                h += batch
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    h[idx] += 1
            cdf = np.cumsum(h).astype(np.float32)
            lut[ty, tx] = np.clip(np.rint(cdf * scale), 0, 255).astype(np.uint8)
    ys, xs = np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32)
    tyf = ys * (np.float32(1) / np.float32(th)) - np.float32(0.5)
    txf = xs * (np.float32(1) / np.float32(tw)) - np.float32(0.5)
    ty1, tx1 = np.floor(tyf).astype(int), np.floor(txf).astype(int)
    ya, xa = (tyf - ty1.astype(np.float32)), (txf - tx1.astype(np.float32))
    ya1, xa1 = np.float32(1) - ya, np.float32(1) - xa
    ty2, tx2 = np.minimum(ty1 + 1, ty_n - 1), np.minimum(tx1 + 1, tx_n - 1)
    ty1, tx1 = np.maximum(ty1, 0), np.maximum(tx1, 0)
    v = img.astype(int)
    f = lambda a, b: lut[a[:, None], b[None, :], v].astype(np.float32)
    res = (f(ty1, tx1) * xa1[None, :] + f(ty1, tx2) * xa[None, :]) * ya1[:, None] + (f(ty2, tx1) * xa1[None, :] + f(ty2, tx2) * xa[None, :]) * ya[:, None]
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def test_clahe_against_numpy(pkg, T):
    rng = np.random.default_rng(8)
    base = pkg.synth.texture_image(320, 240, seed=3) if hasattr(pkg.synth, "texture_image") else rng.integers(0, 256, (240, 320), dtype=np.uint8)
    for img, clip, tiles in ((base, 3.0, (8, 8)),
                             (rng.integers(0, 256, (250, 333), dtype=np.uint8), 3.0, (8, 8)),       # not a multiple of the grid → REFLECT_101 extension
                             (rng.integers(100, 140, (96, 128), dtype=np.uint8), 2.0, (4, 3)),      # narrow histogram → heavy clipping + residual path
                             (rng.integers(0, 256, (64, 64), dtype=np.uint8), 0.0, (2, 2))):        # clip 0 = plain tile equalisation
        got = T.clahe(img, clip, tiles)
        np.testing.assert_array_equal(got, _clahe_numpy(img, clip, tiles))


def test_clahe_periodic_image_is_a_single_lut(pkg, T):
    """identical tiles → identical LUTs → the blend returns lut[src] exactly: equalisation with the clip-limited CDF"""
    rng = np.random.default_rng(2)
    tile = rng.integers(0, 256, (30, 40), dtype=np.uint8)
    img = np.tile(tile, (8, 8))
    got = T.clahe(img, 3.0, (8, 8))
    h = np.bincount(tile.ravel(), minlength=256)
    limit = max(int(3.0 * tile.size / 256), 1)
    clipped = int(np.maximum(h - limit, 0).sum()); h = np.minimum(h, limit) + clipped // 256
    res = clipped % 256
    if res:
        h[np.arange(0, 256, max(256 // res, 1))[:res]] += 1
    lut = np.clip(np.rint(np.cumsum(h).astype(np.float32) * (np.float32(255) / np.float32(tile.size))), 0, 255).astype(np.uint8)
    np.testing.assert_array_equal(got, lut[img])
    # staged form: push_image equalises before the pyramid is built
    T.set_equalize(True, 3.0, (8, 8))
    T.push_image(img)
    l1 = T.debug_get(pkg._abi.TDBG_PYRAMID_L1, np.uint8).reshape(120, 160)
    np.testing.assert_array_equal(l1, _pyrdown_numpy(lut[img]))
    T.set_equalize(False)
    T.push_image(img)
    np.testing.assert_array_equal(T.debug_get(pkg._abi.TDBG_PYRAMID_L1, np.uint8).reshape(120, 160), _pyrdown_numpy(img))


# ----------------------------------------------------------------------------- f-3 MEI undistortion
MEI = dict(xi=1.9926618269451453, k1=-0.0399258932468764, k2=0.15160828121223818, p1=0.00017756967825777937, p2=-0.0011531239076798612,
           gamma1=669.8940458885896, gamma2=669.1450614220616, u0=377.9459252967363, v0=279.63655686698144)


def _mei_project(c, P):
    """CataCamera::spaceToPlane in float64 (the inverse of what is tested): unit-sphere + xi, distortion, K"""
    P = P / np.linalg.norm(P, axis=1, keepdims=True)
    z = P[:, 2] + c["xi"]
    mx, my = P[:, 0] / z, P[:, 1] / z
    r2 = mx * mx + my * my
    rad = c["k1"] * r2 + c["k2"] * r2 * r2
    dx = mx * rad + 2 * c["p1"] * mx * my + c["p2"] * (r2 + 2 * mx * mx)
    dy = my * rad + 2 * c["p2"] * mx * my + c["p1"] * (r2 + 2 * my * my)
    return np.stack([c["gamma1"] * (mx + dx) + c["u0"], c["gamma2"] * (my + dy) + c["v0"]], axis=1)


def test_mei_undistortion_round_trip(pkg, T):
    rng = np.random.default_rng(4)
    # within 230 px of the principal point the lifted ray looks forward (b.z > 0), so (un, 1) keeps its direction
    xy = np.stack([rng.uniform(MEI["u0"] - 160, MEI["u0"] + 160, 300), rng.uniform(MEI["v0"] - 160, MEI["v0"] + 160, 300)], axis=1).astype(np.float32)
    un = T.undistort_points(MEI, xy).astype(np.float64)
    # (un.x, un.y, 1) is the viewing ray: projecting it again must land on the pixel (8 fixed-point iterations → ~1e-6 px here)
    back = _mei_project(MEI, np.concatenate([un, np.ones((len(un), 1))], axis=1))
    assert np.abs(back - xy).max() < 2e-3
    # pinhole special case: no distortion, xi = 0 → (x - u0) / gamma
    pin = dict(MEI, xi=0.0, k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    un = T.undistort_points(pin, xy)
    want = np.stack([(xy[:, 0].astype(np.float64) - pin["u0"]) / pin["gamma1"], (xy[:, 1].astype(np.float64) - pin["v0"]) / pin["gamma2"]], axis=1)
    assert np.abs(un - want).max() < 1e-6
    # xi == 1 branch (:610-613)
    un1 = T.undistort_points(dict(pin, xi=1.0), xy).astype(np.float64)
    back = _mei_project(dict(pin, xi=1.0), np.concatenate([un1, np.ones((len(un1), 1))], axis=1))
    assert np.abs(back - xy).max() < 2e-3
    assert len(T.undistort_points(MEI, np.zeros((0, 2), np.float32))) == 0


def test_lk_accumulator_switch(pkg, T):
    """lvo_set_lk_accumulators: mode 1 = SURVEY App. A.6 literally (float sums, scalar row-major order), mode 0 = exact integer sums.
    On a textured pair the two agree to rounding; the switch is process-wide and must be put back"""
    import ctypes
    S = pkg.synth
    lib = T.lib
    lib.dll.lvo_set_lk_accumulators.argtypes = [ctypes.c_int]
    lib.dll.lvo_get_lk_accumulators.restype = ctypes.c_int
    assert lib.dll.lvo_get_lk_accumulators() == 0
    w, h = 320, 240
    img0 = S.make_texture(w, h, 11)
    img1 = S.warp_homography(img0, S.small_motion_homography(w, h, 3, max_px=3.0))
    pts = T.good_features(img0, 80, 0.01, 10.0)
    x0, s0, _ = T.lk_track(img0, img1, pts)
    try:
        lib.dll.lvo_set_lk_accumulators(1)
        assert lib.dll.lvo_get_lk_accumulators() == 1
        x1, s1, _ = T.lk_track(img0, img1, pts)
    finally:
        lib.dll.lvo_set_lk_accumulators(0)
    k = (s0 == 1) & (s1 == 1)
    assert k.sum() >= 0.9 * len(pts)
    assert np.abs(x0[k] - x1[k]).max() < 1e-2
    x2, s2, _ = T.lk_track(img0, img1, pts)
    np.testing.assert_array_equal(x0, x2)
