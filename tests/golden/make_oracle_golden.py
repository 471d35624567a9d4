"""Generates tests/golden/oracle_golden.npz: seeded inputs and the CPU oracle's outputs for every stage of
the hot path.  The reference cannot run here (cv2 is absent: DESIGN.md section 2), so these vectors pin the
ORACLE (regression) and give the GPU tests fixed data to hit; parity with cv2 itself stays unpinned.
    python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orc  # noqa: E402


def texture(rng, h, w, c):
    base = rng.integers(12, 256, (h // 8 + 2, w // 8 + 2, c)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8, 1)))[:h, :w]
    img = (0.5 * img + 0.5 * np.roll(img, 3, 1)).astype(np.uint8)
    return img[..., 0] if c == 1 else img


def apply_m(M, p):
    q = np.c_[p, np.ones(len(p))] @ M.T
    return q[:, :2] / q[:, 2:3]


def main():
    rng = np.random.default_rng(20240611)
    out = {}
    # --- kNN-2 + ratio: SIFT-like (L2), ORB-like (L2 as the reference does, and Hamming), float32 ---
    q = rng.integers(0, 180, (80, 128), dtype=np.uint8)
    t = rng.integers(0, 180, (150, 128), dtype=np.uint8)
    t[17] = q[3]; t[90] = q[3]; t[40] = t[41]                      # exact match, tie between rows, duplicate rows
    q[10:30] = np.clip(t[100:120].astype(int) + rng.integers(-4, 5, (20, 128)), 0, 255).astype(np.uint8)  # planted neighbours
    out["sift_q"], out["sift_t"] = q, t
    out["sift_idx"], out["sift_dist"] = orc.knn2(q, t)
    for r in (65, 80):
        gq, gt = orc.ratio_filter(out["sift_idx"], out["sift_dist"], r / 100)
        out[f"sift_good_q_{r}"], out[f"sift_good_t_{r}"] = gq, gt
    oq = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    ot = rng.integers(0, 256, (90, 32), dtype=np.uint8)
    ot[5:25] = oq[30:50] ^ (1 << rng.integers(0, 8, (20, 32))).astype(np.uint8) * (rng.random((20, 32)) < 0.1)
    out["orb_q"], out["orb_t"] = oq, ot
    out["orb_idx_l2"], out["orb_dist_l2"] = orc.knn2(oq, ot)
    out["orb_idx_ham"], out["orb_dist_ham"] = orc.knn2(oq, ot, norm=orc.NORM_HAMMING)
    out["orb_count_l2"] = np.int32(orc.match_count(oq, ot, 0.65))
    out["orb_count_ham"] = np.int32(orc.match_count(oq, ot, 0.65, norm=orc.NORM_HAMMING))
    fq, ft = q.astype(np.float32) * np.float32(0.37), t.astype(np.float32) * np.float32(0.37)
    out["f32_idx"], out["f32_dist"] = orc.knn2(fq, ft)
    # --- robust transform ---
    for name, persp in (("aff", False), ("hom", True)):
        n = 160
        th, s = np.deg2rad(3.0), 1.02
        H = np.array([[s * np.cos(th), -s * np.sin(th), 140.0], [s * np.sin(th), s * np.cos(th), -60.0], [0, 0, 1.0]])
        if persp:
            H[2, :2] = (3e-5, -2e-5)
        src = rng.uniform(0, 1920, (n, 2)).astype(np.float32)
        dst = (apply_m(H, src.astype(np.float64)) + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
        dst[:48] = rng.uniform(0, 1920, (48, 2)).astype(np.float32)
        p = rng.permutation(n)
        src, dst = src[p], dst[p]
        M, mask = orc.find_homography(src, dst) if persp else orc.estimate_affine_partial(src, dst)
        out[f"{name}_src"], out[f"{name}_dst"], out[f"{name}_M"], out[f"{name}_mask"] = src, dst, M, mask
    # --- image space: pad -> warp -> bbox -> blenders (the order of Stitcher.stitch) ---
    for c in (1, 3):
        img, ref = texture(rng, 40, 56, c), texture(rng, 60, 90, c)
        pimg, pref = orc.get_padded_images(img, ref)
        Hh, Ww = pref.shape[:2]
        th = np.deg2rad(-4.0)
        M = np.array([[np.cos(th), -np.sin(th), 21.5], [np.sin(th), np.cos(th), -7.25], [0, 0, 1.0]])
        cx, cy = Ww / 2, Hh / 2
        M[0, 2] += cx - (M[0, 0] * cx + M[0, 1] * cy)
        M[1, 2] += cy - (M[1, 0] * cx + M[1, 1] * cy)
        Mp = M.copy()
        Mp[2, :2] = (8e-5, -5e-5)
        out[f"img{c}"], out[f"ref{c}"], out[f"M{c}_aff"], out[f"M{c}_hom"] = img, ref, M[:2].copy(), Mp
        wa = orc.warp(pimg, M[:2], (Ww, Hh), False)
        wp = orc.warp(pimg, Mp, (Ww, Hh), True)
        out[f"warp{c}_aff"], out[f"warp{c}_hom"] = wa, wp
        out[f"warp{c}_aff_classic"] = orc.warp(pimg, M[:2], (Ww, Hh), False, arith=1)   # OpenCV <= 4.10 fixed point
        out[f"warp{c}_hom_classic"] = orc.warp(pimg, Mp, (Ww, Hh), True, arith=1)
        out[f"bbox{c}_aff"] = np.array(orc.bbox_nonzero(wa), np.int32)
        out[f"alpha{c}_02"] = orc.alpha_blend(wa, pref, 0.2)
        out[f"alpha{c}_05"] = orc.alpha_blend(wp, pref, 0.5)
        (x0, y0), (x1, y1) = orc.bbox_nonzero(wa)
        gx0, gy0, gx1, gy1 = max(0, x0 - 50), max(0, y0 - 50), min(Ww, x1 + 50), min(Hh, y1 + 50)
        out[f"groi{c}"] = np.array([gx0, gy0, gx1, gy1], np.int32)
        out[f"grad{c}"] = orc.gradient_blend(wa[gy0:gy1, gx0:gx1], pref[gy0:gy1, gx0:gx1], 10)
    path = os.path.join(HERE, "oracle_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
