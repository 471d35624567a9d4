"""CPU tests of oracle/orc_ransac.c with analytic known answers (independent of OpenCV fidelity)."""
import numpy as np
import pytest


def apply_h(H, p):
    q = np.c_[p, np.ones(len(p))] @ H.T
    return q[:, :2] / q[:, 2:3]


def rand_h(rng, persp=3e-4):
    th = np.deg2rad(rng.uniform(-15, 15))
    s = rng.uniform(0.8, 1.25)
    H = np.array([[s * np.cos(th), -s * np.sin(th), rng.uniform(-80, 80)],
                  [s * np.sin(th), s * np.cos(th), rng.uniform(-80, 80)],
                  [rng.uniform(-persp, persp), rng.uniform(-persp, persp), 1.0]])
    return H


def test_cv_rng_sequence(orc):
    """cv::RNG(0xFFFFFFFFFFFFFFFF): MWC state*4164903690 + carry; uniform(0,n) = next % n.
    First draws computed here independently in Python integers."""
    state = (1 << 64) - 1
    draws = []
    for _ in range(8):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & ((1 << 64) - 1)
        draws.append((state & 0xFFFFFFFF) % 1000)
    # with 1000 well-spread points no subset is rejected, so the first two subsets are draws[0:4], draws[4:8]
    # unless a duplicate index forces a redraw
    rng = np.random.default_rng(0)
    src = rng.uniform(0, 1000, (1000, 2)).astype(np.float32)
    H = rand_h(rng)
    dst = apply_h(H, src).astype(np.float32)
    sub = orc.ransac_sample(orc.MODEL_HOMOGRAPHY, src, dst, 2)
    assert len(set(draws[:4])) == 4 and len(set(draws[4:8])) == 4
    assert sub.tolist() == [draws[:4], draws[4:8]]
    sub2 = orc.ransac_sample(orc.MODEL_AFFINE_PARTIAL, src, dst, 4)
    assert sub2.reshape(-1).tolist() == draws


def test_exact_four_point_homography(orc):
    rng = np.random.default_rng(1)
    for _ in range(20):
        H = rand_h(rng, 8e-4)
        src = np.array([[10, 20], [600, 35], [580, 420], [25, 400]], np.float32) + rng.uniform(-5, 5, (4, 2)).astype(np.float32)
        dst = apply_h(H, src.astype(np.float64)).astype(np.float32)
        M = orc.fit_model(orc.MODEL_HOMOGRAPHY, src, dst)
        assert abs(M[2, 2] - 1.0) < 1e-12
        assert np.abs(apply_h(M, src.astype(np.float64)) - dst).max() < 2e-3
        # findHomography with exactly 4 points = direct solve, mask all ones
        M2, mask = orc.find_homography(src, dst)
        assert np.allclose(M2, M) and mask.all()


def test_exact_two_point_similarity(orc):
    rng = np.random.default_rng(2)
    for _ in range(20):
        th, s = rng.uniform(-3, 3), rng.uniform(0.5, 2)
        a, b, tx, ty = s * np.cos(th), s * np.sin(th), rng.uniform(-50, 50), rng.uniform(-50, 50)
        src = rng.uniform(0, 500, (2, 2)).astype(np.float32)
        dst = np.c_[a * src[:, 0] - b * src[:, 1] + tx, b * src[:, 0] + a * src[:, 1] + ty].astype(np.float32)
        M = orc.fit_model(orc.MODEL_AFFINE_PARTIAL, src, dst)
        assert np.allclose(M, [[a, -b, tx], [b, a, ty]], atol=2e-3, rtol=1e-4)
        M2, inl = orc.estimate_affine_partial(src, dst)
        assert np.array_equal(M2, M) and inl.all()


@pytest.mark.parametrize("n,outlier_frac", [(60, 0.0), (200, 0.3), (1000, 0.5)])
def test_homography_ransac_recovers_planted_model(orc, n, outlier_frac):
    rng = np.random.default_rng(n)
    H = rand_h(rng)
    src = rng.uniform(0, 1500, (n, 2)).astype(np.float32)
    dst = (apply_h(H, src.astype(np.float64)) + rng.normal(0, 0.4, (n, 2))).astype(np.float32)
    n_out = int(n * outlier_frac)
    dst[:n_out] = rng.uniform(0, 1500, (n_out, 2)).astype(np.float32)
    M, mask = orc.find_homography(src, dst)
    assert M is not None and abs(M[2, 2] - 1.0) < 1e-12
    inl = mask.ravel().astype(bool)
    assert inl[n_out:].mean() > 0.98 and (n_out == 0 or inl[:n_out].mean() < 0.05)
    rms = np.sqrt(((apply_h(M, src[n_out:].astype(np.float64)) - apply_h(H, src[n_out:].astype(np.float64))) ** 2).sum(1).mean())
    assert rms < 0.25


@pytest.mark.parametrize("n,outlier_frac", [(30, 0.0), (300, 0.4)])
def test_affine_partial_ransac_recovers_planted_model(orc, n, outlier_frac):
    rng = np.random.default_rng(100 + n)
    th, s, tx, ty = np.deg2rad(4.0), 1.03, 120.5, -33.25
    a, b = s * np.cos(th), s * np.sin(th)
    src = rng.uniform(0, 1900, (n, 2)).astype(np.float32)
    dst = (np.c_[a * src[:, 0] - b * src[:, 1] + tx, b * src[:, 0] + a * src[:, 1] + ty] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    n_out = int(n * outlier_frac)
    dst[:n_out] = rng.uniform(0, 1900, (n_out, 2)).astype(np.float32)
    M, inl = orc.estimate_affine_partial(src, dst)
    assert M is not None
    assert M[0, 0] == M[1, 1] and M[0, 1] == -M[1, 0]
    assert np.allclose(M, [[a, -b, tx], [b, a, ty]], atol=0.15, rtol=2e-3)
    assert inl.ravel()[n_out:].mean() > 0.98


def test_lm_refine_reaches_least_squares(orc):
    """Affine-partial refinement is a linear least-squares problem: LM must land on lstsq's answer."""
    rng = np.random.default_rng(7)
    src = rng.uniform(0, 1000, (150, 2)).astype(np.float32)
    a, b, tx, ty = 0.98, 0.07, 15.0, -8.0
    dst = (np.c_[a * src[:, 0] - b * src[:, 1] + tx, b * src[:, 0] + a * src[:, 1] + ty] + rng.normal(0, 1.0, (150, 2))).astype(np.float32)
    A = np.zeros((300, 4))
    A[0::2] = np.c_[src[:, 0], -src[:, 1], np.ones(150), np.zeros(150)]
    A[1::2] = np.c_[src[:, 1], src[:, 0], np.zeros(150), np.ones(150)]
    sol = np.linalg.lstsq(A, dst.reshape(-1).astype(np.float64), rcond=None)[0]
    M0 = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    M = orc.refine(orc.MODEL_AFFINE_PARTIAL, M0, src, dst, refit=False, max_iters=10)
    assert np.allclose([M[0, 0], M[1, 0], M[0, 2], M[1, 2]], sol, rtol=1e-6, atol=1e-5)


def test_score_threshold_and_float32_error(orc):
    H = np.eye(3)
    src = np.array([[0, 0], [10, 10], [20, 20], [30, 30]], np.float32)
    dst = src + np.array([[3, 4], [3.1, 4], [0, 5], [0, 5.0001]], np.float32)
    n, mask = orc.score(orc.MODEL_HOMOGRAPHY, H, src, dst, 5.0)
    # err = 25 (inlier, <=), 25.61 (out), 25 (in), >25 (out)
    assert mask.tolist() == [1, 0, 1, 0] and n == 2
    e = orc.reproj_err(orc.MODEL_HOMOGRAPHY, H, src, dst)
    assert e.dtype == np.float32 and e[0] == np.float32(25.0)


def test_degenerate_inputs(orc):
    src = np.zeros((3, 2), np.float32)
    assert orc.find_homography(src, src)[0] is None             # fewer than 4 points
    assert orc.estimate_affine_partial(src[:1], src[:1])[0] is None
    # all points identical: every subset is "collinear" -> no model
    p = np.full((20, 2), 5.0, np.float32)
    assert orc.find_homography(p, p)[0] is None
