"""CPU tests of the matching oracle (oracle/orc_match.c) against analytic known answers and a
straight numpy restatement of cv2.BFMatcher().knnMatch + stitcher.py:118-127."""
import numpy as np
import pytest


def np_knn2(q, t, hamming=False):
    """Tiny pure-numpy restatement: float32 sqrt distances, stable sort => lowest index wins ties."""
    q64 = q.astype(np.int64)
    t64 = t.astype(np.int64)
    if hamming:
        x = q[:, None, :] ^ t[None, :, :]
        d = np.unpackbits(x, axis=2).sum(2).astype(np.float32)
    else:
        s = ((q64[:, None, :] - t64[None, :, :]) ** 2).sum(2)
        d = np.sqrt(s.astype(np.float32)).astype(np.float32)
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    idx = np.full((q.shape[0], 2), -1, np.int32)
    dist = np.full((q.shape[0], 2), np.finfo(np.float32).max, np.float32)
    k = min(2, t.shape[0])
    idx[:, :k] = order[:, :k]
    dist[:, :k] = np.take_along_axis(d, order[:, :k], 1)
    return idx, dist


@pytest.mark.parametrize("nq,nt,d", [(1, 1, 32), (5, 1, 128), (7, 2, 61), (33, 70, 128), (64, 257, 32)])
def test_knn2_u8_vs_numpy(orc, nq, nt, d):
    rng = np.random.default_rng(nq * 1000 + nt)
    q = rng.integers(0, 256, (nq, d), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, d), dtype=np.uint8)
    idx, dist = orc.knn2(q, t)
    ridx, rdist = np_knn2(q, t)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)
    idx, dist = orc.knn2(q, t, norm=orc.NORM_HAMMING)
    ridx, rdist = np_knn2(q, t, hamming=True)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist, rdist)


def test_planted_ties_lowest_index_wins(orc):
    rng = np.random.default_rng(3)
    t = rng.integers(0, 200, (50, 128), dtype=np.uint8)
    q = t[[10]].copy()
    # rows 10, 20, 30 identical => distances 0,0,0: slots must be (10, 20)
    t[20] = t[10]
    t[30] = t[10]
    idx, dist = orc.knn2(q, t)
    assert idx.tolist() == [[10, 20]] and dist.tolist() == [[0.0, 0.0]]
    # a tie for the second slot only: best is row 5 (exact), rows 40 and 7 equally second
    t2 = rng.integers(0, 200, (50, 128), dtype=np.uint8)
    q2 = t2[[5]].copy()
    far = q2[0].copy()
    far[0] = np.uint8((int(far[0]) + 3) % 256) if far[0] < 250 else np.uint8(far[0] - 3)
    t2[40] = far
    t2[7] = far
    idx, dist = orc.knn2(q2, t2)
    assert idx.tolist() == [[5, 7]]
    assert dist[0, 1] == np.float32(3.0)


def test_float_input_matches_u8(orc):
    rng = np.random.default_rng(11)
    q = rng.integers(0, 256, (40, 128), dtype=np.uint8)
    t = rng.integers(0, 256, (90, 128), dtype=np.uint8)
    i8, d8 = orc.knn2(q, t)
    i32, d32 = orc.knn2(q.astype(np.float32), t.astype(np.float32))
    assert np.array_equal(i8, i32) and np.array_equal(d8, d32)


def test_ratio_filter_semantics(orc):
    # stitcher.py:118-127: strict <, float64 product, single-match rows kept unconditionally
    idx = np.array([[3, 4], [1, -1], [-1, -1], [2, 9], [5, 6]], np.int32)
    big = np.finfo(np.float32).max
    dist = np.array([[6.5, 10.0], [99.0, big], [big, big], [0.0, 0.0], [1.0, 2.0]], np.float32)
    gq, gt = orc.ratio_filter(idx, dist, 0.65)
    # row0: 6.5 < 0.65*10.0 -> 6.5 < 6.5 False ; row1 single -> kept; row2 absent; row3 0<0 False; row4 1<1.3
    assert gq.tolist() == [1, 4] and gt.tolist() == [1, 5]
    # float64 evaluation: d0 = float32(0.65) against ratio 0.65 (float64) * 1.0
    d0 = np.float32(0.65)
    dist = np.array([[d0, 1.0]], np.float32)
    gq, _ = orc.ratio_filter(np.array([[0, 1]], np.int32), dist, 0.65)
    assert (len(gq) == 1) == (float(d0) < 0.65 * 1.0)


def test_match_count(orc):
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    q = t[rng.permutation(300)[:100]].copy()
    q[:, 0] ^= 1
    idx, dist = orc.knn2(q, t)
    gq, _ = orc.ratio_filter(idx, dist, 0.65)
    assert orc.match_count(q, t, 0.65) == len(gq) == 100
    assert orc.match_count(q, t[:0], 0.65) == 0


def test_sqrtf_injective_below_limit():
    """ISX_L2_EXACT_LIMIT: integer ranking == float-sqrt ranking below it (exhaustive)."""
    s = np.arange(0, 4197200 + 2, dtype=np.int64)
    r = np.sqrt(s.astype(np.float32))
    eq = np.nonzero(r[1:] == r[:-1])[0]
    assert eq[0] + 1 == 4197201  # first collision is (4197200, 4197201)
