"""CPU tests of oracle/orc_image.c: analytic known answers + the pure-numpy restatement (tests/np_ref.py)."""
import numpy as np
import pytest

import np_ref


def rand_img(rng, h, w, c, lo=0, hi=256):
    shape = (h, w) if c == 1 else (h, w, c)
    return rng.integers(lo, hi, shape, dtype=np.uint8)


def test_gray_and_bbox(orc):
    rng = np.random.default_rng(0)
    img = rand_img(rng, 37, 53, 3)
    assert np.array_equal(orc.bgr2gray(img), np_ref.gray(img))
    # [OCV-K] (4,0,0) and (0,0,1) are black under the 8-bit gray formula, (0,1,0) is not
    z = np.zeros((9, 11, 3), np.uint8)
    z[2, 3] = (4, 0, 0)
    z[7, 9] = (0, 0, 1)
    assert orc.bbox_nonzero(z) is False
    z[4, 5] = (0, 1, 0)
    assert orc.bbox_nonzero(z) == ((5, 4), (6, 5))
    for c in (1, 3):
        im = np.zeros((40, 60) + ((3,) if c == 3 else ()), np.uint8)
        im[5:22, 7:33] = rand_img(rng, 17, 26, c, 20, 256)
        assert orc.bbox_nonzero(im) == np_ref.corners(im) == ((7, 5), (33, 22))


def test_pad(orc):
    rng = np.random.default_rng(1)
    for c in (1, 3):
        img = rand_img(rng, 7, 10, c)
        ref = rand_img(rng, 9, 13, c)
        pi, pr = orc.get_padded_images(img, ref)
        assert pi.shape[:2] == (9 + 14, 13 + 20) == pr.shape[:2]
        assert np.array_equal(pi, np_ref.pad(img, 23, 33)) and np.array_equal(pr, np_ref.pad(ref, 23, 33))
        # ref lands at (img_w, img_h); img at ((ref_w+img_w)//2, (ref_h+img_h)//2)
        assert np.array_equal(pr[7:16, 10:23], ref)
        assert np.array_equal(pi[8:15, 11:21], img)


@pytest.mark.parametrize("c", [1, 3])
def test_warp_identity_and_integer_translation_are_copies(orc, c):
    rng = np.random.default_rng(2)
    img = rand_img(rng, 31, 45, c)
    eye = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float64)
    assert np.array_equal(orc.warp(img, eye[:2], (45, 31), False), img)
    assert np.array_equal(orc.warp(img, eye, (45, 31), True), img)
    T = eye.copy()
    T[0, 2], T[1, 2] = 7, -3
    exp = np.zeros_like(img)
    exp[0:28, 7:45] = img[3:31, 0:38]
    assert np.array_equal(orc.warp(img, T[:2], (45, 31), False), exp)
    assert np.array_equal(orc.warp(img, T, (45, 31), True), exp)
    # bigger destination than source: constant 0 border
    out = orc.warp(img, T[:2], (60, 40), False)
    assert out[:, :7].max() == 0 and out[28:, :].max() == 0 and np.array_equal(out[0:28, 7:52], img[3:31, 0:45])


@pytest.mark.parametrize("persp", [False, True])
def test_warp_within_1lsb_of_float64_bilinear(orc, persp):
    rng = np.random.default_rng(3)
    # smooth-ish image so that coordinate rounding cannot move a value by more than 1
    base = rng.integers(0, 256, (8, 10, 3)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8, 1)))
    img = (0.25 * (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, (1, 1), (0, 1)))).astype(np.uint8)
    th = np.deg2rad(7.0)
    M = np.array([[np.cos(th), -np.sin(th), 12.3], [np.sin(th), np.cos(th), -4.7], [0, 0, 1]])
    if persp:
        M[2, :2] = (1e-4, -2e-4)
    out = orc.warp(img, M if persp else M[:2], (100, 90), persp)
    ref = np_ref.warp_bilinear_f64(img, np.linalg.inv(M) if persp else np.linalg.inv(M)[:2], (100, 90))
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0
    assert np.abs(out.astype(np.float64) - np.rint(ref)).mean() < 0.02


@pytest.mark.parametrize("c", [1, 3])
def test_classic_warp_known_answers(orc, c):
    """arith=1, the fixed-point remap of OpenCV <= 4.10: exact-copy cases, round-half-UP at half pixels
    (the float32 kernels round half to even), 1/32-px coordinate quantisation."""
    rng = np.random.default_rng(12)
    img = rand_img(rng, 31, 45, c)
    eye = np.eye(3)
    T = eye.copy()
    T[0, 2], T[1, 2] = 7, -3
    exp = np.zeros_like(img)
    exp[0:28, 7:45] = img[3:31, 0:38]
    for persp in (False, True):
        assert np.array_equal(orc.warp(img, eye if persp else eye[:2], (45, 31), persp, arith=1), img)
        assert np.array_equal(orc.warp(img, T if persp else T[:2], (45, 31), persp, arith=1), exp)
    # half-pixel shift to the right: out[x] = (img[x-1] + img[x] + 1) >> 1   (weights 16384/16384, +16384 >> 15)
    H = eye.copy()
    H[0, 2] = 0.5
    a = img.astype(np.int32)
    left = np.zeros_like(a)
    left[:, 1:] = a[:, :-1]
    half_up = ((left + a + 1) >> 1).astype(np.uint8)
    for persp in (False, True):
        assert np.array_equal(orc.warp(img, H if persp else H[:2], (45, 31), persp, arith=1), half_up)
    half_even = np.rint((left + a) / 2.0).astype(np.uint8)   # numpy rint = half to even
    assert np.array_equal(orc.warp(img, H[:2], (45, 31), False, arith=0), half_even)
    assert (half_up != half_even).any()
    # coordinates are rounded to the 1/32 grid (the +16 of 1024): a shift of 1/64 px rounds to 0 (a copy),
    # 3/128 px rounds to 1/32 px = weights (1/32, 31/32) on the pixels (x-1, x)
    Q = eye.copy()
    Q[0, 2] = 1.0 / 64
    assert np.array_equal(orc.warp(img, Q[:2], (45, 31), False, arith=1), img)
    Q[0, 2] = 3.0 / 128
    out = orc.warp(img, Q[:2], (45, 31), False, arith=1)
    w1 = 1024
    expq = ((left * w1 + a * (32768 - w1) + 16384) >> 15).astype(np.uint8)
    assert np.array_equal(out, expq)


@pytest.mark.parametrize("persp", [False, True])
def test_classic_warp_within_2lsb_of_float64_bilinear(orc, persp):
    rng = np.random.default_rng(3)
    base = rng.integers(0, 256, (8, 10, 3)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8, 1)))
    img = (0.25 * (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, (1, 1), (0, 1)))).astype(np.uint8)
    th = np.deg2rad(7.0)
    M = np.array([[np.cos(th), -np.sin(th), 12.3], [np.sin(th), np.cos(th), -4.7], [0, 0, 1]])
    if persp:
        M[2, :2] = (1e-4, -2e-4)
    out = orc.warp(img, M if persp else M[:2], (100, 90), persp, arith=1)
    ref = np_ref.warp_bilinear_f64(img, np.linalg.inv(M) if persp else np.linalg.inv(M)[:2], (100, 90))
    # 1/64 px of coordinate quantisation x the steepest gradient of this image, + rounding
    grad = max(np.abs(np.diff(img.astype(float), axis=0)).max(), np.abs(np.diff(img.astype(float), axis=1)).max())
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0 + 2 * grad / 64
    assert np.abs(out.astype(np.float64) - ref).mean() < 0.5


def test_alpha_blend_known_answers(orc):
    # wrap-around outside the overlap: 250 + 8 -> 2 (ref bright, warped dark-but-nonzero, gray <= 10)
    w = np.full((1, 4), 8, np.uint8)
    r = np.full((1, 4), 250, np.uint8)
    assert orc.alpha_blend(w, r, 0.5).tolist() == [[2, 2, 2, 2]]
    # inside the overlap: trunc(c*ref) + trunc((1-c)*warped)
    w = np.array([[101, 11, 10, 255]], np.uint8)
    r = np.array([[51, 201, 200, 255]], np.uint8)
    out = orc.alpha_blend(w, r, 0.2)
    exp = [int(0.2 * 51) + int((1 - 0.2) * 101), int(0.2 * 201) + int((1 - 0.2) * 11), (10 + 200) % 256,
           int(0.2 * 255) + int((1 - 0.2) * 255)]
    assert out.tolist() == [exp]


@pytest.mark.parametrize("c", [1, 3])
@pytest.mark.parametrize("contrib", [0.5, 0.2, 0.65])
def test_alpha_blend_vs_numpy(orc, c, contrib):
    rng = np.random.default_rng(4)
    w = rand_img(rng, 33, 47, c)
    r = rand_img(rng, 33, 47, c)
    w[:10] = 0
    r[:, :12] = rng.integers(0, 12, r[:, :12].shape)
    assert np.array_equal(orc.alpha_blend(w, r, contrib), np_ref.alpha_merge(w, r, contrib))


def test_gradient_blend_1d_profile(orc):
    # a single 30-px-wide, 1-px-tall strip: the vertical pass zeroes it (top == bottom => alpha 0),
    # so the result equals ref; a tall block shows the horizontal ramp k/10
    h, w = 140, 200
    warped = np.zeros((h, w), np.uint8)
    warped[60, 80:110] = 200
    ref = np.full((h, w), 100, np.uint8)
    assert np.array_equal(orc.gradient_blend(warped, ref), ref)
    warped[:] = 0
    warped[40:100, 80:110] = 200
    out = orc.gradient_blend(warped, ref)
    y = 70  # far from the vertical feather zones (rows 40..49 and 90..99)
    ramp = [int(200 * (k / 10) + 100 * (1 - k / 10)) for k in range(10)]
    assert out[y, 80:90].tolist() == ramp
    assert out[y, 100:110].tolist() == [int(200 * ((9 - k) / 10) + 100 * (1 - (9 - k) / 10)) for k in range(10)]
    assert (out[y, 90:100] == 200).all() and (out[y, :80] == 100).all()
    # corner: both ramps multiply
    assert out[43, 82] == int(200 * ((2 / 10) * (3 / 10)) + 100 * (1 - (2 / 10) * (3 / 10)))


@pytest.mark.parametrize("c", [1, 3])
def test_gradient_blend_vs_numpy(orc, c):
    rng = np.random.default_rng(5)
    h, w = 150, 170
    for trial in range(4):
        warped = np.zeros((h, w) + ((3,) if c == 3 else ()), np.uint8)
        # ragged quadrilateral-ish support with holes and zero-valued channel-0 pixels
        y0, x0 = rng.integers(0, 60, 2)
        hh, ww = rng.integers(15, 80, 2)
        blk = rand_img(rng, hh, ww, c, 0, 256)
        blk[rng.random(blk.shape[:2]) < 0.05] = 0
        warped[y0:y0 + hh, x0:x0 + ww] = blk
        for yy in range(y0, y0 + hh):   # slanted left edge
            warped[yy, x0:x0 + (yy - y0) // 3] = 0
        ref = rand_img(rng, h, w, c)
        assert np.array_equal(orc.gradient_blend(warped, ref), np_ref.gradient_merge(warped, ref))
    # all-black warped returns ref unchanged
    z = np.zeros_like(ref)
    assert np.array_equal(orc.gradient_blend(z, ref), ref)
