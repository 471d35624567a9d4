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


def test_knn2_norm_limit_routes_to_the_wide_kernel(cuda, orc):
    """Norms beyond ISX_L2_EXACT_LIMIT: the MFMA path refuses when asked to (wide_ok=False), the default routes the
    set to the float-distance-ranking kernel -- all-255 rows are all at distance 0: both slots = lowest indices."""
    from image_stitcher_amd import ops
    q = np.full((8, 128), 255, np.uint8)
    with pytest.raises(ops.NormLimitError):
        ops.prepare_descriptors([q, q], wide_ok=False)
    idx, dist, cnt = ops.knn2_l2(q, q)
    ridx, rdist = orc.knn2(q, q)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(idx.cpu().numpy(), np.tile([0, 1], (8, 1)))
    assert np.array_equal(dist.cpu().numpy(), rdist) and int(cnt.item()) == 0


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


# ---- count-only kernel (isx_knn2_l2_count_items): same counts as the index kernel and the oracle ----------
def _count_case(orc, des_list, ratio, verify_all):
    import torch
    from image_stitcher_amd import ops
    P = ops.prepare_descriptors(des_list)
    n = len(des_list)
    items = np.array([(P.starts[i], P.counts[i], P.starts[j], P.counts[j])
                      for i in range(n) for j in range(n) if i != j and P.counts[i] and P.counts[j]], np.int32)
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j and P.counts[i] and P.counts[j]]
    c = ops.knn2_l2_items(P, items, ratio, verify_all=verify_all)
    # flags bit 1: 128- / 64-byte rows on the 32x32x32 MFMA kernel; bit 2: 128-byte rows on the 16x16x64 kernel with 64
    # queries per wave (the default has 128): the A/B references must give the same counts
    c_alt = ops.knn2_l2_items(P, items, ratio, verify_all=int(verify_all) ^ 2)
    c_alt4 = ops.knn2_l2_items(P, items, ratio, verify_all=int(verify_all) ^ 4)
    c_idx, _ = ops.knn2_l2_items(P, items, ratio, want_matches=True)
    torch.cuda.synchronize()
    c, c_alt, c_idx = c.cpu().numpy(), c_alt.cpu().numpy(), c_idx.cpu().numpy()
    assert np.array_equal(c_alt4.cpu().numpy(), c), "the two 16x16x64 count kernels differ"
    exp = np.array([orc.match_count(des_list[i], des_list[j], ratio) for i, j in pairs], np.int32)
    assert np.array_equal(c_idx, exp)
    assert np.array_equal(c, exp), f"count kernel differs on {np.nonzero(c != exp)[0][:10]}"
    assert np.array_equal(c_alt, exp), f"count kernel (other MFMA shape) differs on {np.nonzero(c_alt != exp)[0][:10]}"


@pytest.mark.parametrize("verify_all", [False, 1])
def test_count_kernel_matches_oracle_sift_like(cuda, orc, verify_all):
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, _ = overlapping_crops_descriptors(seed=11, n_scenes=2, imgs_per_scene=3, kp=700, step=100)
    _count_case(orc, [d for d in des], 0.65, verify_all)


@pytest.mark.parametrize("d,hi", [(128, 180), (61, 256), (32, 256)])
def test_count_kernel_ragged_and_ties(cuda, orc, d, hi):
    rng = np.random.default_rng(5 + d)
    sizes = [1, 2, 33, 257, 700, 129, 64]
    des = [rng.integers(0, hi, (s, d), dtype=np.uint8) for s in sizes]
    des[4][10:40] = des[3][5:35]      # planted matches
    des[4][100] = des[4][101]         # duplicate train rows (ties)
    des[5][:] = des[5][0]             # all-identical image: every distance ties
    for ratio in (0.65, 0.9, 1.0):
        _count_case(orc, des, ratio, False)
    _count_case(orc, des, 0.65, 1)
    _count_case(orc, des, 0.65, 3)


def test_count_kernel_ratio_boundary_parity_bit(cuda, orc):
    """Queries whose ratio test sits within one unit of squared distance of the threshold (3k vs 4m
    below): the count kernel has to take its exact per-query path there and still agree with the oracle."""
    rng = np.random.default_rng(77)
    base = rng.integers(100, 140, (128,), dtype=np.uint8)
    far = rng.integers(0, 60, (200, 128), dtype=np.uint8)
    q = np.repeat(base[None], 40, 0)
    for k in range(1, 41):
        q[k - 1, :k] += 1              # ||q_k - base||^2 = k
    des = [q]
    for m in range(1, 32):
        t = np.concatenate([base[None], base[None], far])
        t[1, 64:64 + m] += 2           # ||q_k - t1||^2 = k + 4m ; ratio 0.5: good <=> 3k < 4m
        des.append(t[rng.permutation(len(t))])
    _count_case(orc, des[:12], 0.5, False)
    _count_case(orc, [des[0]] + des[12:22], 0.5, False)
    _count_case(orc, [des[0]] + des[22:], 0.5, False)


@pytest.mark.parametrize("nq,nt,d", [(96, 70001, 32), (70001, 33, 32), (40, 20000, 128), (64, 262143, 32)])
def test_long_train_and_query_sets(cuda, orc, nq, nt, d):
    """Sizes past every internal period: > 2^16 train rows (index packing, 512-row segments, many LDS
    chunks) and > 2^16 queries (grid y / query-block arithmetic), with duplicates planted far apart."""
    rng = np.random.default_rng(nq + nt)
    hi = 256 if d == 32 else 170
    q = rng.integers(0, hi, (nq, d), dtype=np.uint8)
    t = rng.integers(0, hi, (nt, d), dtype=np.uint8)
    if nt > 1000:
        t[nt - 1] = t[5]            # equal distances 70k rows apart: the lower index must win both slots
        t[nt // 2] = q[3]           # exact match in the middle
        q[7] = t[5]
    _check_pair(orc, q, t)
    _count_case(orc, [q, t], 0.8, False)


def test_large_ragged_set_sampled_against_the_oracle(cuda, orc):
    """A realistic get_collections launch: 96 images whose keypoint counts range from 300 to 6000 (SIFT gives
    what the image gives), all 9,120 ordered pairs in one launch with the XCD-aware item order; 48 items drawn
    at random -- biased towards the connected ones -- are recounted by the oracle."""
    import torch
    from image_stitcher_amd import ops
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, scene_of = overlapping_crops_descriptors(seed=77, n_scenes=3, imgs_per_scene=32, kp=6000, step=700)
    rng = np.random.default_rng(5)
    sizes = rng.integers(300, 6001, len(des))
    lst = [d[:int(s)] for d, s in zip(des, sizes)]
    P = ops.prepare_descriptors(lst)
    n = len(lst)
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
    items = np.array([(P.starts[i], P.counts[i], P.starts[j], P.counts[j]) for i, j in pairs], np.int32)
    counts = ops.knn2_l2_items(P, items, 0.65)
    torch.cuda.synchronize()
    counts = counts.cpu().numpy()
    strong = np.nonzero(counts >= 5)[0]
    assert len(strong) > 100 and (counts < 5).sum() > 1000
    pick = np.concatenate([rng.choice(strong, 24, replace=False), rng.choice(len(pairs), 24, replace=False)])
    for e in pick:
        i, j = pairs[e]
        assert counts[e] == orc.match_count(lst[i], lst[j], 0.65, nthreads=8), (i, j)


def test_item_lists_longer_than_one_grid(cuda, orc):
    """An item list whose grid would pass 2^31 threads (HIP forms gridDim*blockDim in 32 bits and silently
    truncates: at 2^24 items of one 256-thread workgroup the old single launch ran 5000 of them) is split into
    launches of at most 2^31 threads inside the entry points: 2^24 + 5000 small items, counts of every item and the
    indices of items in the first, second and last launch against the oracle."""
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(77)
    n_img = 24
    lst = [rng.integers(0, 90, (int(r), 32), dtype=np.uint8) for r in rng.integers(2, 9, n_img)]
    P = ops.prepare_descriptors(lst)
    n = (1 << 24) + 5000
    qi = rng.integers(0, n_img, n)
    ti = (qi + rng.integers(1, n_img, n)) % n_img
    st, ct = np.asarray(P.starts), np.asarray(P.counts)
    items = np.stack([st[qi], ct[qi], st[ti], ct[ti]], axis=1).astype(np.int32)
    table = np.array([[orc.match_count(lst[i], lst[j], 0.8) if i != j else -1 for j in range(n_img)] for i in range(n_img)])
    exp = table[qi, ti]
    for xcd in (False, True):
        c = ops.knn2_l2_items(P, items, 0.8, xcd_order=xcd).cpu().numpy()
        assert np.array_equal(c, exp), (xcd, int((c != exp).sum()), int(np.nonzero(c != exp)[0][0]))
    c2, (idx, dist, starts) = ops.knn2_l2_items(P, items, 0.8, want_matches=True, xcd_order=False)
    assert np.array_equal(c2.cpu().numpy(), exp)
    for e in (0, 12345, (1 << 23) - 1, 1 << 23, (1 << 24) - 1, 1 << 24, n - 1):
        ridx, rdist = orc.knn2(lst[qi[e]], lst[ti[e]])
        s = int(starts[e])
        assert np.array_equal(idx[s:s + ct[qi[e]]].cpu().numpy(), ridx)
        assert np.array_equal(dist[s:s + ct[qi[e]]].cpu().numpy(), rdist)
    assert exp.min() == 0 and exp.max() >= 3


@pytest.mark.parametrize("nq,nt,d", [(700, 1500, 32), (300, 130, 128), (257, 3000, 61), (5, 1, 32), (64, 128, 32), (1500, 2049, 32)])
def test_frame_match_train_segments_merge_exactly(cuda, orc, nq, nt, d):
    """isx_frame_match runs the index kernel on (query block x train segment) items and merges the per-segment results
    in the compaction kernel: the good-match list must be the oracle's (knn2 + ratio loop), including ties whose two
    rows lie in different segments (the lower train index wins, as in cv2) and queries whose two nearest rows come
    from two different segments."""
    from image_stitcher_amd import ops
    rng = np.random.default_rng(nq + 3 * nt + d)
    t = rng.integers(0, 150, (nt, d), dtype=np.uint8)
    q = rng.integers(0, 150, (nq, d), dtype=np.uint8)
    if nt > 400:
        # queries copied from train rows with a little noise (so that the ratio test passes for them), and duplicated
        # train rows 128, 256 ... rows apart: exact ties across segment boundaries
        src = rng.permutation(nt)[:nq // 2]
        q[:len(src)] = np.clip(t[src].astype(np.int16) + rng.integers(-2, 3, (len(src), d)), 0, 255).astype(np.uint8)
        for a, b in [(3, 131), (10, 266), (127, 128), (40, nt - 1), (300, 301), (5, 389)]:
            t[b] = t[a]
        q[-6:] = t[[3, 10, 127, 40, 300, 5]]
    ctx = ops.FrameCtx("cuda:0")
    for ratio in (0.65, 0.9, 1.0):
        gq, gt = ops.frame_match(ctx, q, t, ratio)
        ridx, rdist = orc.knn2(q, t)
        oq, ot = orc.ratio_filter(ridx, rdist, ratio)
        assert np.array_equal(gq, oq) and np.array_equal(gt, ot), (ratio, len(gq), len(oq))
    # float32 input (integer valued, as cv2.SIFT returns) takes the same path
    gq, gt = ops.frame_match(ctx, q.astype(np.float32), t.astype(np.float32), 0.8)
    oq, ot = orc.ratio_filter(*orc.knn2(q, t), 0.8)
    assert np.array_equal(gq, oq) and np.array_equal(gt, ot)
    # norm="hamming" (cv2.BFMatcher(NORM_HAMMING)): the same segments, the XOR / popcount index kernel.  Hamming distances
    # are small integers, so ties across segment boundaries are the rule here, in both slots
    if d <= 64:
        qb = rng.integers(0, 256, (nq, d), dtype=np.uint8)
        tb = rng.integers(0, 256, (nt, d), dtype=np.uint8)
        if nt > 400:
            src = rng.permutation(nt)[:nq // 2]
            qb[:len(src)] = tb[src] ^ (rng.random((len(src), d)) < 0.03).astype(np.uint8)
            for a, b in [(3, 131), (10, 266), (127, 128), (40, nt - 1), (300, 301), (5, 389)]:
                tb[b] = tb[a]
            qb[-6:] = tb[[3, 10, 127, 40, 300, 5]]
        for ratio in (0.65, 0.9, 1.0):
            gq, gt = ops.frame_match(ctx, qb, tb, ratio, hamming=True)
            ridx, rdist = orc.knn2(qb, tb, norm=orc.NORM_HAMMING)
            oq, ot = orc.ratio_filter(ridx, rdist, ratio)
            assert np.array_equal(gq, oq) and np.array_equal(gt, ot), ("hamming", ratio, len(gq), len(oq))
