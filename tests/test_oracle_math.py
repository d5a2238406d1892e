"""CPU tier: the oracle's restatements of the third-party small-matrix routines (OpenCV Jacobi /
QR solve / LU inverse, Eigen column-pivoted QR, PCL getTransformation, tf2 slerp, FLANN kd-tree)
against independent ground truth (numpy / LAPACK, closed forms, brute force).  These pin the
oracle to the mathematics, since the reference ships no vectors (PARITY UNPINNED)."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def dll(oracle):
    d = oracle.dll
    d.lvo_test_slerp_axis.restype = C.c_double
    d.lvo_test_slerp_axis.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
    d.lvo_test_get_transformation.argtypes = [C.c_float] * 6 + [C.c_void_p]
    return d


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("n", [3, 6])
def test_jacobi_matches_lapack(dll, n):
    rng = np.random.default_rng(n)
    for trial in range(50):
        M = rng.normal(size=(n, n)) * (10.0 ** rng.uniform(-2, 3))
        A = (M @ M.T).astype(np.float32)
        W = np.zeros(n, np.float32); V = np.zeros((n, n), np.float32)
        dll.lvo_test_jacobi(n, _p(A), _p(W), _p(V))
        w_ref = np.linalg.eigvalsh(A.astype(np.float64))[::-1]
        assert np.all(np.diff(W) <= 1e-6 * abs(W[0])), "eigenvalues must be descending"
        np.testing.assert_allclose(W, w_ref, rtol=2e-5, atol=2e-5 * abs(w_ref[0]))
        # rows of V are eigenvectors: A v = w v, orthonormal
        for i in range(n):
            r = A.astype(np.float64) @ V[i] - W[i] * V[i]
            assert np.linalg.norm(r) <= 5e-5 * abs(w_ref[0])
        np.testing.assert_allclose(V @ V.T, np.eye(n), atol=5e-6)


def test_solve_qr_and_inverse(dll):
    rng = np.random.default_rng(1)
    for trial in range(50):
        M = rng.normal(size=(6, 6))
        A = (M @ M.T + 0.5 * np.eye(6)).astype(np.float32)
        b = rng.normal(size=6).astype(np.float32)
        x = np.zeros(6, np.float32)
        assert dll.lvo_test_solve_qr6(_p(A), _p(b), _p(x)) == 1
        ref = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
        np.testing.assert_allclose(x, ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())
        inv = np.zeros((6, 6), np.float32)
        assert dll.lvo_test_inv6(_p(A), _p(inv)) == 1
        np.testing.assert_allclose(inv @ A, np.eye(6), atol=5e-3)
    # nearly singular (not exactly: an exactly zero column gives 0/0 in the Householder normalisation, as in OpenCV)
    sing = np.eye(6, dtype=np.float32); sing[5, 5] = 1e-9
    x = np.zeros(6, np.float32)
    assert dll.lvo_test_solve_qr6(_p(sing), _p(np.ones(6, np.float32)), _p(x)) == 0


def test_colpiv_least_squares_plane_fit(dll):
    rng = np.random.default_rng(2)
    for trial in range(100):
        nrm = rng.normal(size=3); nrm /= np.linalg.norm(nrm)
        d = rng.uniform(2, 30)
        # 5 points on the plane n.p + d = 0, slightly noisy — the exact use at mapOptimization.cpp:1128
        basis = np.linalg.svd(nrm[None])[2][1:]
        pts = (-d * nrm)[None] + rng.uniform(-0.5, 0.5, (5, 2)) @ basis + rng.normal(0, 0.005, (5, 3))
        A = pts.astype(np.float32); b = -np.ones(5, np.float32)
        x = np.zeros(3, np.float32)
        dll.lvo_test_colpiv_5x3(_p(A), _p(b), _p(x))
        ref = np.linalg.lstsq(A.astype(np.float64), b.astype(np.float64), rcond=None)[0]
        np.testing.assert_allclose(x, ref, rtol=5e-3, atol=5e-4 * np.abs(ref).max())
    # rank deficient: three identical columns -> minimum... Eigen returns a basic solution with zeros
    A = np.ones((5, 3), np.float32); b = -np.ones(5, np.float32); x = np.zeros(3, np.float32)
    dll.lvo_test_colpiv_5x3(_p(A), _p(b), _p(x))
    assert np.isclose(x.sum(), -1.0, atol=1e-5) and (x == 0).sum() == 2


def test_get_transformation_is_rz_ry_rx(dll, pkg):
    rng = np.random.default_rng(3)
    for trial in range(20):
        x, y, z = rng.uniform(-10, 10, 3)
        r, p, yw = rng.uniform(-1.2, 1.2, 3)
        m = np.zeros(12, np.float32)
        dll.lvo_test_get_transformation(x, y, z, r, p, yw, _p(m))
        m = m.reshape(3, 4)
        R = pkg.synth.rot_zyx(r, p, yw)
        np.testing.assert_allclose(m[:, :3], R, atol=3e-7)
        np.testing.assert_allclose(m[:, 3], [x, y, z], rtol=1e-7)


def test_slerp_moves_linearly_in_angle(dll):
    # tf2 slerp between two rotations about the same axis is linear interpolation of the angle
    for axis in (0, 1):
        for a, b, w in ((0.10, 0.30, 0.01), (-0.4, 0.2, 0.5), (0.0, 1.0, 0.25), (0.3, 0.3, 0.7)):
            got = dll.lvo_test_slerp_axis(axis, a, b, w)
            assert abs(got - (a + (b - a) * w)) < 1e-12


def test_kdtree_is_exact_knn(dll):
    rng = np.random.default_rng(4)
    for n in (3, 7, 40, 5000):
        pts = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
        if n == 5000:
            pts[:2000, 2] = 0.0                        # a plane: many equal coordinates
        q = rng.uniform(-22, 22, (300, 3)).astype(np.float32)
        idx = np.full((300, 5), -1, np.int32); sqd = np.zeros((300, 5), np.float32)
        dll.lvo_test_kdtree_knn(_p(pts), n, _p(q), 300, _p(idx), _p(sqd))
        k = min(5, n)
        for i in range(300):
            diff = q[i][None].astype(np.float32) - pts
            d = ((diff[:, 0] * diff[:, 0]) + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]      # f32, FLANN's order
            ref = np.sort(d)[:k]
            np.testing.assert_array_equal(sqd[i, :k], ref)
            assert np.all(np.diff(sqd[i, :k]) >= 0)
            np.testing.assert_array_equal(d[idx[i, :k]], sqd[i, :k])
