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
