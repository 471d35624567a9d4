"""Randomised parity of the count kernels (ArbitraryStitcher.get_collections' inner loop) against the oracle's
knnMatch + ratio loop: random image counts, ragged keypoint counts (0 .. 1500), widths 32 / 61 / 64 / 128, value ranges that
force ties, planted near-duplicates; the default kernel, the two A/B kernels (flags bits 1 and 2) and the index kernel.
usage: python3 tools/fuzz_count.py [cases=40]   (exit status 1 on any mismatch)"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from image_stitcher_amd import ops
from oracle import orc


def main(N, seed0=90000):
    bad = 0
    for case in range(N):
        rng = np.random.default_rng(seed0 + case)
        d = int(rng.choice([32, 61, 64, 128]))
        hi = int(rng.choice([4, 60, 150])) if d > 32 else int(rng.choice([4, 60, 256]))
        n_img = int(rng.integers(2, 7))
        sizes = [int(rng.choice([0, 1, 2, int(rng.integers(3, 1500))], p=[0.05, 0.05, 0.05, 0.85])) for _ in range(n_img)]
        des = [rng.integers(0, hi, (s, d), dtype=np.uint8) for s in sizes]
        for _ in range(3):  # planted near-duplicates between two images
            i, j = rng.integers(0, n_img, 2)
            k = min(sizes[i], sizes[j]) // 3
            if i != j and k:
                des[i][:k] = np.clip(des[j][rng.permutation(sizes[j])[:k]].astype(np.int16) + rng.integers(-1, 2, (k, d)), 0, 255).astype(np.uint8)
        ratio = float(rng.choice([0.5, 0.65, 0.8, 1.0]))
        P = ops.prepare_descriptors(des)
        pairs = [(i, j) for i in range(n_img) for j in range(n_img) if i != j and P.counts[i] and P.counts[j]]
        if not pairs:
            continue
        items = np.array([(P.starts[i], P.counts[i], P.starts[j], P.counts[j]) for i, j in pairs], np.int32)
        got = [ops.knn2_l2_items(P, items, ratio, verify_all=f).cpu().numpy() for f in (0, 2, 4, 1)]
        got.append(ops.knn2_l2_items(P, items, ratio, want_matches=True)[0].cpu().numpy())
        exp = np.array([orc.match_count(des[i], des[j], ratio) for i, j in pairs], np.int32)
        if not all(np.array_equal(g, exp) for g in got):
            bad += 1
            print("MISMATCH", seed0 + case, d, hi, sizes, ratio, [int((g != exp).sum()) for g in got], flush=True)
    print("done", N, "cases,", bad, "mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40) else 0)
