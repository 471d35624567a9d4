"""Randomised parity of isx_knn2_l2_count_pairs (both directions of a pair from one distance matrix: max-only filter, exact
recount, overflow to the directed kernel, early exit) against the oracle's knnMatch + ratio loop (reference
stitcher.py:285-296) and against the directed count kernel: ragged 128-d (and 64- / 61-d) image sets up to 4096 rows (one to four passes, partly
filled last classes), value ranges that force ties, planted near-duplicates from a handful to most of an image (candidate lists
from empty to overflowing, in one direction or both), sorted and unsorted rows, several ratios, the every-query-is-a-candidate
hook.   usage: python3 tools/fuzz_pairs.py [cases=40] [seed0]   (exit status 1 on any mismatch)"""
import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "image-stitcher_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from image_stitcher_amd import ops, pairs as prs
from oracle import orc


def one_case(seed):
    rng = np.random.default_rng(seed)
    d = int(rng.choice([128, 128, 64, 61]))
    hi = int(rng.choice([4, 60, 150, 256]))
    n_img = int(rng.integers(2, 6))
    pick = lambda: int(rng.choice([1, 2, 16, 127, 128, 129, 1023, 1024, 1025, 2048, 2049, 3000, 4095, 4096,
                                   int(rng.integers(3, 4097))]))
    sizes = [pick() for _ in range(n_img)]
    if rng.random() < 0.2:  # images beyond 4096 rows: the two-group table (train side <= 8192 rows, query side <= 16384)
        sizes[int(rng.integers(0, n_img))] = int(rng.choice([4097, 6000, 8192, 8193, 12000, 16384, 16385]))
        if rng.random() < 0.5:
            sizes[int(rng.integers(0, n_img))] = int(rng.choice([4097, 5000, 8192, 9000]))
    if rng.random() < 0.5:  # SIFT-like rows (a wide norm range) instead of uniform ones
        from image_stitcher_amd.synthetic import sift_like_pool
        pool = sift_like_pool(rng, 4096, d)
        des = [np.clip(pool[rng.integers(0, len(pool), s)].astype(np.int16) + rng.integers(-3, 4, (s, d)), 0, 255).astype(np.uint8)
               for s in sizes]
    else:
        des = [rng.integers(0, hi, (s, d), dtype=np.uint8) for s in sizes]
    for _ in range(int(rng.integers(0, 5))):  # planted near-duplicates between two images
        i, j = rng.integers(0, n_img, 2)
        k = int(min(sizes[i], sizes[j]) * rng.choice([0.01, 0.05, 0.3, 0.9]))
        if i != j and k:
            rows_i = rng.permutation(sizes[i])[:k]
            des[i][rows_i] = np.clip(des[j][rng.permutation(sizes[j])[:k]].astype(np.int16)
                                     + rng.integers(-1, 2, (k, d)), 0, 255).astype(np.uint8)
    ratio = float(rng.choice([0.5, 0.65, 0.8, 1.0]))
    P = ops.prepare_descriptors(des)
    if rng.random() < 0.6:
        P = ops.sort_rows_by_norm(P)
    all_pairs = prs.unordered_pairs(n_img)
    items = prs.pair_items(all_pairs, np.asarray(P.starts), np.asarray(P.counts))
    st = {}
    got = ops.knn2_l2_pairs(P, items, ratio, stats=st).cpu().numpy()
    got1 = ops.knn2_l2_pairs(P, items, ratio, flags=1).cpu().numpy() if rng.random() < 0.3 else got
    # items are (q = image j?, t = image i?) as pairs.pair_items lays them out: take the directions from the items themselves
    starts = np.asarray(P.starts)
    img_of = {int(s): k for k, s in enumerate(starts)}
    exp = np.zeros_like(got)
    for e, (qs, nq, ts, nt) in enumerate(np.asarray(items).reshape(-1, 4)):
        a, b = des[img_of[int(qs)]].astype(np.float32), des[img_of[int(ts)]].astype(np.float32)
        exp[e, 0] = orc.match_count(a, b, ratio, nthreads=16)
        exp[e, 1] = orc.match_count(b, a, ratio, nthreads=16)
    ok = np.array_equal(got, exp) and np.array_equal(got1, exp)
    return ok, (seed, hi, sizes, ratio, int((got != exp).sum()), int((got1 != exp).sum()), st.get("overflow_directions"), int(exp.max()))


def main(N, seed0=170000):
    bad = n_over = 0
    for case in range(N):
        ok, info = one_case(seed0 + case)
        n_over += bool(info[6])
        if not ok:
            bad += 1
            print("MISMATCH", info, flush=True)
        elif case % 10 == 0:
            print("ok", info, flush=True)
    print("done", N, "cases,", bad, "mismatches,", n_over, "cases with overflowing directions")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 170000) else 0)
