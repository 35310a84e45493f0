"""fsgm_epipolar_from_F (epipolar_geometry.m:40-96; SURVEY 8(f) N4, the in-tree part of the sparse half): host algebra, runs
without a GPU.  UNPINNED like every MATLAB-side row -- checked against a numpy restatement of the same lines that uses LAPACK's
SVD (what MATLAB's svd is), under every sign choice an SVD is free to make, and through the properties the outputs must have."""
import numpy as np
import pytest

import fsgm_amd
from fsgm_amd import synth


def _numpy_restatement(F, K, p1, p2, inl, flip=(1, 1, 1)):
    """epipolar_geometry.m:40-96 with numpy.linalg.svd; `flip`: signs applied to the singular-vector pairs of svd(E)."""
    _, _, Vt = np.linalg.svd(F.T)
    e = Vt[-1]
    epi = e[:2] / e[2]
    E = K.T @ F @ K
    U, _, Vt = np.linalg.svd(E)
    U, V = U * np.array(flip), Vt.T * np.array(flip)
    W = np.array([[0., -1, 0], [1, 0, 0], [0, 0, 1]])
    R1, R2 = U @ W @ V.T, U @ W.T @ V.T
    if np.linalg.det(R1) < 0:
        R1, R2 = -R1, -R2
    q1, q2 = bool((np.diag(R1) > 0).all()), bool((np.diag(R2) > 0).all())
    R = R1 if q1 else R2
    H = K @ R @ np.linalg.inv(K)
    exp = 0
    for (x1, y1), (x2, y2), ok in zip(p1, p2, inl):
        if not ok:
            continue
        pr = H @ np.array([x1, y1, 1.0])
        d1 = np.hypot(pr[0] / pr[2] - epi[0], pr[1] / pr[2] - epi[1])
        exp += np.hypot(x2 - epi[0], y2 - epi[1]) > d1
    return H, epi, 0 if exp / max(inl.sum(), 1) > 0.5 else 1, q1 == q2


def _scene(seed, forward=True, W=1242, H=375):
    """A camera with a small rotation moving forward (or backward): F, K and noise-free matches of random 3-D points."""
    rng = np.random.default_rng(seed)
    f = 0.58 * W
    K = np.array([[f, 0, W / 2 + 3.0], [0, f, H / 2 - 2.0], [0, 0, 1.0]])
    ax, ay, az = rng.uniform(-0.02, 0.02, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), -1.0 if forward else 1.0])     # X2 = R X1 + t: the scene moves towards the camera
    X = np.stack([rng.uniform(-8, 8, 200), rng.uniform(-3, 3, 200), rng.uniform(6, 40, 200)], 1)
    x1 = (K @ X.T).T
    X2 = (R @ X.T).T + t
    x2 = (K @ X2.T).T
    p1, p2 = x1[:, :2] / x1[:, 2:], x2[:, :2] / x2[:, 2:]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ tx @ R @ Kinv
    F = F / np.abs(F).max()
    return F, K, R, p1 + 1.0, p2 + 1.0                     # MATLAB pixel coordinates are 1-based ... any consistent offset does


@pytest.mark.parametrize("seed,forward", [(1, True), (2, False), (3, True), (4, False), (5, True)])
def test_epipolar_from_F_against_a_lapack_restatement(seed, forward):
    F, K, Rtrue, p1, p2 = _scene(seed, forward)
    # shift F to the 1-based coordinates of the points: x' = x + 1
    T = np.array([[1, 0, -1.0], [0, 1, -1.0], [0, 0, 1]])
    F1 = T.T @ F @ T
    K1 = np.array([[K[0, 0], 0, K[0, 2] + 1], [0, K[1, 1], K[1, 2] + 1], [0, 0, 1.0]])
    inl = (np.arange(len(p1)) % 7 != 0).astype(np.uint8)
    H, epi, direction, amb = fsgm_amd.epipolar_from_F(F1, K1, p1, p2, inl)
    assert not amb
    for flip in ((1, 1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, -1), (-1, -1, 1), (-1, -1, -1)):
        rH, repi, rdir, ramb = _numpy_restatement(F1, K1, p1, p2, inl.astype(bool), flip)
        assert not ramb
        np.testing.assert_allclose(H / H[2, 2], rH / rH[2, 2], rtol=0, atol=1e-7)
        np.testing.assert_allclose(epi, repi, rtol=1e-8, atol=1e-6)
        assert direction == rdir
    # properties: the epipole is the left null vector of F; H = K R K^-1 with R the scene's rotation; points moving away from
    # the epipole = expansion = direction 0 for a forward-moving camera
    e = np.array([epi[0], epi[1], 1.0])
    assert np.abs(F1.T @ e).max() < 1e-9 * np.abs(F1).max() * np.abs(e).max()
    R = np.linalg.inv(K1) @ H @ K1
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-9)
    assert abs(np.linalg.det(R) - 1) < 1e-9
    np.testing.assert_allclose(R, Rtrue, atol=1e-7)
    assert direction == (0 if forward else 1)


def test_epipolar_from_F_feeds_the_dense_half():
    """The helper's output is what epipolar_maps / epipolar_sgm_of take (and what synth.epi_geometry fabricates)."""
    F, K, _, p1, p2 = _scene(9)
    H, epi, direction, amb = fsgm_amd.epipolar_from_F(F, K, p1, p2)
    assert H.shape == (3, 3) and len(epi) == 2 and direction in (0, 1) and amb is False
    # no matches: 0 / 0 is NaN in MATLAB, not > 0.5 -> direction 1
    assert fsgm_amd.epipolar_from_F(F, K)[2] == 1


def test_epipolar_from_F_rejects_degenerate_input():
    F, K, _, p1, p2 = _scene(3)
    with pytest.raises(fsgm_amd.FsgmError, match="singular"):
        fsgm_amd.epipolar_from_F(F, np.zeros((3, 3)), p1, p2)
    with pytest.raises(fsgm_amd.FsgmError, match="rank"):
        fsgm_amd.epipolar_from_F(np.outer([1, 2, 3.0], [0, 0, 1.0]), K, p1, p2)
    with pytest.raises(fsgm_amd.FsgmError, match="finite"):
        fsgm_amd.epipolar_from_F(F * np.nan, K, p1, p2)
