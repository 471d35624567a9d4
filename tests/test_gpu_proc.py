"""The image helpers under their reference names (image_stitcher_amd.utils.proc mirrors the reference's
utils/proc.py) against the numpy restatement of the reference's own code (tests/np_ref.py) and the oracle:
same return contracts (False sentinels, tuples, views), same mutation of the gradient blender's second
argument."""
import numpy as np
import pytest

import np_ref

pytestmark = pytest.mark.gpu


def _img(rng, h, w, c, lo=0, hi=256):
    return rng.integers(lo, hi, (h, w) if c == 1 else (h, w, c), dtype=np.uint8)


@pytest.mark.parametrize("c", [1, 3])
def test_corners_crop_roi(cuda, c):
    from image_stitcher_amd.utils import proc
    rng = np.random.default_rng(c)
    im = np.zeros((60, 90) + ((3,) if c == 3 else ()), np.uint8)
    im[12:41, 20:77] = _img(rng, 29, 57, c, 1, 256)
    cs = proc.get_corners_from_image(im)
    assert cs == np_ref.corners(im) == ((20, 12), (77, 41))
    crop = proc.crop_image(im)
    assert np.array_equal(np.asarray(crop), im[12:41, 20:77])
    assert proc.get_roi_from_corners(im, cs[0], cs[1], img=False) == (12, 41, 20, 77)      # proc.py:331-333
    assert np.array_equal(np.asarray(proc.get_roi_from_corners(im, cs[0], cs[1])), im[12:41, 20:77])
    black = np.zeros_like(im)
    assert proc.get_corners_from_image(black) is False and proc.crop_image(black) is False


@pytest.mark.parametrize("c", [1, 3])
def test_padding(cuda, orc, c):
    from image_stitcher_amd.utils import proc
    rng = np.random.default_rng(10 + c)
    img, ref = _img(rng, 21, 33, c), _img(rng, 40, 55, c)
    pi, pr = proc.get_padded_images(img, ref)
    oi, orf = orc.get_padded_images(img, ref)
    assert np.array_equal(np.asarray(pi), oi) and np.array_equal(np.asarray(pr), orf)
    assert np.asarray(pr).shape[:2] == (40 + 2 * 21, 55 + 2 * 33)
    # ref pasted at (img_w, img_h); img centred with floor division (proc.py:286-291)
    assert np.array_equal(np.asarray(pr)[21:61, 33:88], ref)
    assert np.array_equal(np.asarray(pi)[(40 + 21) // 2:(40 + 21) // 2 + 21, (55 + 33) // 2:(55 + 33) // 2 + 33], img)
    with pytest.raises(ValueError):
        proc.pad_image(10, 10, 4, 4, np.zeros((4, 4, 4), np.uint8), channels=4)


@pytest.mark.parametrize("c", [1, 3])
def test_blenders_and_mutation(cuda, c):
    from image_stitcher_amd.device_image import DeviceImage
    from image_stitcher_amd.utils import proc
    rng = np.random.default_rng(20 + c)
    h, w = 150, 170
    ref = _img(rng, h, w, c, 0, 256)
    ref[:, :30] = 0
    warped = np.zeros_like(ref)
    warped[40:110, 60:150] = _img(rng, 70, 90, c, 0, 256)
    # alpha: returns a new image, inputs untouched (proc.py:48-82)
    for contrib in (0.2, 0.5):
        out = proc.seamless_merge(warped, ref, contrib)
        assert np.array_equal(np.asarray(out), np_ref.alpha_merge(warped, ref, contrib))
    # gradient: feathered paste inside bbox(warped) +- 50 px, IN PLACE on the second argument (proc.py:252-254)
    exp = np_ref.gradient_merge(warped, ref, 10)
    dref = DeviceImage.from_any(ref.copy(), cuda)
    ret = proc.seamless_gradient_merge(warped, dref)
    assert ret is dref and np.array_equal(np.asarray(dref), exp)
    assert (exp != ref).any() and np.array_equal(exp[:, :9], ref[:, :9])        # nothing beyond the +-50 px ROI
    # an all-black warped image leaves the reference unchanged (proc.py:250-251)
    d2 = DeviceImage.from_any(ref.copy(), cuda)
    assert proc.seamless_gradient_merge(np.zeros_like(ref), d2) is d2 and np.array_equal(np.asarray(d2), ref)


def test_transform_keypoints_from_roi(cuda):
    from image_stitcher_amd.detect import KeyPoint
    from image_stitcher_amd.utils import proc
    kps = [KeyPoint(1.5, 2.0), KeyPoint(10.0, 0.25)]
    out = proc.transform_keypoints_from_roi(kps, (100, 40))
    assert [k.pt for k in out] == [(101.5, 42.0), (110.0, 40.25)] and out[0] is kps[0]   # shifted in place (proc.py:315-322)
