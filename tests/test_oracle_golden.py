"""The CPU oracle against the committed vectors of tests/golden/oracle_golden.npz (regression pin of the
checker; made by tests/golden/make_oracle_golden.py).  cv2 parity itself is unpinned."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(G, allow_pickle=False)


def test_knn_and_ratio(orc, gold):
    idx, dist = orc.knn2(gold["sift_q"], gold["sift_t"])
    assert np.array_equal(idx, gold["sift_idx"]) and np.array_equal(dist.view(np.int32), gold["sift_dist"].view(np.int32))
    assert idx[3].tolist() == [17, 90] and dist[3].tolist() == [0.0, 0.0]     # planted exact matches: lower index first
    for r in (65, 80):
        gq, gt = orc.ratio_filter(idx, dist, r / 100)
        assert np.array_equal(gq, gold[f"sift_good_q_{r}"]) and np.array_equal(gt, gold[f"sift_good_t_{r}"])
    assert set(range(10, 30)) <= set(gold["sift_good_q_65"].tolist())          # the planted neighbours pass
    i2, d2 = orc.knn2(gold["orb_q"], gold["orb_t"])
    assert np.array_equal(i2, gold["orb_idx_l2"]) and np.array_equal(d2, gold["orb_dist_l2"])
    i3, d3 = orc.knn2(gold["orb_q"], gold["orb_t"], norm=orc.NORM_HAMMING)
    assert np.array_equal(i3, gold["orb_idx_ham"]) and np.array_equal(d3, gold["orb_dist_ham"])
    assert orc.match_count(gold["orb_q"], gold["orb_t"], 0.65) == int(gold["orb_count_l2"])
    assert orc.match_count(gold["orb_q"], gold["orb_t"], 0.65, norm=orc.NORM_HAMMING) == int(gold["orb_count_ham"])
    fq = gold["sift_q"].astype(np.float32) * np.float32(0.37)
    ft = gold["sift_t"].astype(np.float32) * np.float32(0.37)
    i4, d4 = orc.knn2(fq, ft)
    assert np.array_equal(i4, gold["f32_idx"]) and np.array_equal(d4, gold["f32_dist"])


def test_transforms(orc, gold):
    M, mask = orc.estimate_affine_partial(gold["aff_src"], gold["aff_dst"])
    assert np.allclose(M, gold["aff_M"], rtol=0, atol=1e-9) and np.array_equal(mask, gold["aff_mask"])
    H, hmask = orc.find_homography(gold["hom_src"], gold["hom_dst"])
    assert np.allclose(H, gold["hom_M"], rtol=0, atol=1e-9) and np.array_equal(hmask, gold["hom_mask"])
    assert 100 <= int(mask.sum()) <= 112 and 100 <= int(hmask.sum()) <= 112    # 48 of 160 are planted outliers


@pytest.mark.parametrize("c", [1, 3])
def test_image_chain(orc, gold, c):
    img, ref = gold[f"img{c}"], gold[f"ref{c}"]
    pimg, pref = orc.get_padded_images(img, ref)
    H, W = pref.shape[:2]
    wa = orc.warp(pimg, gold[f"M{c}_aff"], (W, H), False)
    wp = orc.warp(pimg, gold[f"M{c}_hom"], (W, H), True)
    assert np.array_equal(wa, gold[f"warp{c}_aff"]) and np.array_equal(wp, gold[f"warp{c}_hom"])
    assert np.array_equal(orc.warp(pimg, gold[f"M{c}_aff"], (W, H), False, arith=1), gold[f"warp{c}_aff_classic"])
    assert np.array_equal(orc.warp(pimg, gold[f"M{c}_hom"], (W, H), True, arith=1), gold[f"warp{c}_hom_classic"])
    assert np.array_equal(np.array(orc.bbox_nonzero(wa), np.int32), gold[f"bbox{c}_aff"])
    assert np.array_equal(orc.alpha_blend(wa, pref, 0.2), gold[f"alpha{c}_02"])
    assert np.array_equal(orc.alpha_blend(wp, pref, 0.5), gold[f"alpha{c}_05"])
    x0, y0, x1, y1 = gold[f"groi{c}"]
    assert np.array_equal(orc.gradient_blend(wa[y0:y1, x0:x1], pref[y0:y1, x0:x1], 10), gold[f"grad{c}"])
