"""GPU parity: HIP kNN-2 + ratio (through the C ABI) vs the CPU oracle -- bit exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_pair(q, t, ratio=0.65):
    import torch
    from image_stitcher_amd import ops
    idx, dist, cnt = ops.knn2_l2(q, t, ratio)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), dist.cpu().numpy(), int(cnt.item())


def _check_pair(orc, q, t, ratio=0.65):
    idx, dist, cnt = _run_pair(q, t, ratio)
    qo = q if q.dtype == np.uint8 else q.astype(np.uint8)
    to = t if t.dtype == np.uint8 else t.astype(np.uint8)
    ridx, rdist = orc.knn2(qo, to)
    assert np.array_equal(idx, ridx), f"idx mismatch rows {np.nonzero((idx != ridx).any(1))[0][:10]}"
    assert np.array_equal(dist.view(np.int32), rdist.view(np.int32))
    gq, _ = orc.ratio_filter(ridx, rdist, ratio)
    assert cnt == len(gq)


@pytest.mark.parametrize("nq,nt,d", [
    (256, 128, 128), (64, 32, 128), (1, 1, 128), (5, 2, 128), (300, 517, 128), (1000, 1500, 128),
    (500, 500, 32), (37, 1100, 32), (700, 650, 61), (4096, 4096, 128),
])
def test_knn2_random_bytes_sift_range(cuda, orc, nq, nt, d):
    rng = np.random.default_rng(nq * 7 + nt + d)
    hi = 256 if d == 32 else 180  # keep 2*max||v||^2 below the exact-sqrt limit for d>32
    q = rng.integers(0, hi, (nq, d), dtype=np.uint8)
    t = rng.integers(0, hi, (nt, d), dtype=np.uint8)
    _check_pair(orc, q, t)


def test_knn2_sift_like_and_float_input(cuda, orc):
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, _ = overlapping_crops_descriptors(seed=7, n_scenes=1, imgs_per_scene=2, kp=1024, step=200)
    _check_pair(orc, des[0], des[1])
    _check_pair(orc, des[1].astype(np.float32), des[0].astype(np.float32))


def test_knn2_planted_ties(cuda, orc):
    rng = np.random.default_rng(99)
    t = rng.integers(0, 150, (2048, 128), dtype=np.uint8)
    q = t[rng.permutation(2048)[:512]].copy()
    # duplicate many train rows at various distances apart (same tile, other half-wave rows, other
    # tiles, other 512-row segments) so that equal distances occur in every merge stage
    for a, b in [(3, 4), (10, 14), (40, 41), (100, 900), (5, 1800), (511, 512), (1023, 1024), (31, 32)]:
        t[b] = t[a]
    q = np.concatenate([q, t[[3, 10, 40, 100, 5, 511, 1023, 31]]])
    _check_pair(orc, q, t)
    # all-identical train set: every distance ties -> slots must be (0, 1)
    t2 = np.repeat(t[:1], 700, 0)
    idx, dist, _ = _run_pair(q[:100], t2)
    assert (idx == np.array([0, 1])).all()


def test_knn2_float_noninteger_rejected(cuda):
    from image_stitcher_amd import ops
    q = np.full((8, 128), 0.5, np.float32)
    with pytest.raises(ops.IsxError):
        ops.knn2_l2(q, q)


def test_knn2_norm_limit_rejected(cuda):
    from image_stitcher_amd import ops
    q = np.full((8, 128), 255, np.uint8)
    with pytest.raises(ops.IsxError):
        ops.knn2_l2(q, q)


def test_knn2_items_batched_counts(cuda, orc):
    """get_collections loop body (stitcher.py:285-296): all ordered pairs of a small image set,
    ragged keypoint counts, XCD-aware item ordering on."""
    import torch
    from image_stitcher_amd import ops
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, _ = overlapping_crops_descriptors(seed=21, n_scenes=2, imgs_per_scene=4, kp=700, step=150)
    rng = np.random.default_rng(0)
    lst = [d[:int(n)] for d, n in zip(des, rng.integers(300, 701, len(des)))]
    P = ops.prepare_descriptors(lst)
    items, pairs = [], []
    for i in range(len(lst)):
        for j in range(len(lst)):
            if i != j:
                items.append((P.starts[i], P.counts[i], P.starts[j], P.counts[j]))
                pairs.append((i, j))
    counts = ops.knn2_l2_items(P, np.array(items, np.int32), 0.65)
    torch.cuda.synchronize()
    counts = counts.cpu().numpy()
    ref = np.array([orc.match_count(lst[i], lst[j], 0.65) for i, j in pairs])
    assert np.array_equal(counts, ref)
    assert ref.max() > 50 and ref.min() < 5  # both connected and unconnected pairs are exercised

    counts2, (idx, dist, starts) = ops.knn2_l2_items(P, np.array(items, np.int32), 0.65, want_matches=True)
    idx = idx.cpu().numpy()
    for e in (0, 17, len(items) - 1):
        i, j = pairs[e]
        ridx, _ = orc.knn2(lst[i], lst[j])
        assert np.array_equal(idx[starts[e]:starts[e] + len(lst[i])], ridx)
    assert np.array_equal(counts2.cpu().numpy(), ref)
