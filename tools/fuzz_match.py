"""Randomised parity of isx_frame_match (train segments + per-query merge) against the oracle's knnMatch + ratio loop:
random sizes 1..3500, widths 16..128, value ranges 4 / 60 / 150 (many exact ties), duplicated train rows.
usage: python3 tools/fuzz_match.py [cases=200]   (exit status 1 on any mismatch)"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd")); sys.path.insert(0, ROOT)
import numpy as np
from image_stitcher_amd import ops
from oracle import orc


def main(N, seed0=70000):
    ctx = ops.FrameCtx("cuda:0")
    bad = 0
    for case in range(N):
        rng = np.random.default_rng(seed0 + case)
        bad += one(ctx, rng, case)
    print("done", N, "cases,", bad, "mismatches")
    return bad


def one(ctx, rng, case):
    d = int(rng.choice([32, 61, 64, 128, 16]))
    nq, nt = int(rng.integers(1, 2500)), int(rng.integers(1, 3500))
    hi = int(rng.choice([4, 60, 150]))
    t = rng.integers(0, hi, (nt, d), dtype=np.uint8)
    q = rng.integers(0, hi, (nq, d), dtype=np.uint8)
    k = min(nq, nt) // 2
    if k:
        src = rng.permutation(nt)[:k]
        q[:k] = np.clip(t[src].astype(np.int16) + rng.integers(-1, 2, (k, d)), 0, 255).astype(np.uint8)
    for _ in range(int(rng.integers(0, 12))):
        a, b = rng.integers(0, nt, 2)
        t[b] = t[a]
    ratio = float(rng.choice([0.5, 0.65, 0.8, 1.0]))
    ham = d in (16, 32, 61, 64) and rng.random() < 0.3  # the opt-in Hamming matcher on the same bytes
    gq, gt = ops.frame_match(ctx, q, t, ratio, hamming=ham)
    oq, ot = orc.ratio_filter(*orc.knn2(q, t, norm=orc.NORM_HAMMING if ham else orc.NORM_L2), ratio)
    if not (np.array_equal(gq, oq) and np.array_equal(gt, ot)):
        print("MISMATCH", case, nq, nt, d, hi, ratio, len(gq), len(oq), flush=True)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 200) else 0)
