"""isx_knn2_l2_count_pairs (both directions of an unordered image pair from one distance matrix, max-only filter +
exact recount of the candidates) against the oracle's knnMatch + ratio loop (reference stitcher.py:285-296) and against
the directed count kernel.  Counts must be the exact counts whatever the filter lets through: ragged image sizes, images
of one row, rows that tie, sorted and unsorted rows, the every-query-is-a-candidate hook, overlapping images whose
directions overflow into the directed kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_counts(orc, des, pairs, ratio):
    out = np.zeros((len(pairs), 2), np.int32)
    for k, (i, j) in enumerate(pairs):
        a, b = des[i].astype(np.float32), des[j].astype(np.float32)
        out[k, 0] = orc.match_count(a, b, ratio, nthreads=8)
        out[k, 1] = orc.match_count(b, a, ratio, nthreads=8)
    return out


def _ragged_set(seed, sizes, d=128, dup_frac=0.3):
    """SIFT-like integer descriptors; images share perturbed copies of pool rows so that counts are not all zero, and
    some rows are exact duplicates (ties in both slots)."""
    from image_stitcher_amd.synthetic import sift_like_pool
    rng = np.random.default_rng(seed)
    pool = sift_like_pool(rng, 3000, d).astype(np.int16)
    des = []
    for n in sizes:
        rows = pool[rng.integers(0, len(pool), n)]
        noise = np.rint(rng.normal(0, 5, rows.shape)).astype(np.int16)
        img = np.clip(rows + noise, 0, 255).astype(np.uint8)
        k = int(n * dup_frac)
        if k > 1:
            img[rng.integers(0, n, k)] = img[rng.integers(0, n, k)]
        des.append(img)
    return des


@pytest.mark.parametrize("sort_rows", [False, True])
def test_pairs_kernel_ragged_images_against_oracle(cuda, orc, sort_rows):
    from image_stitcher_amd import ops, pairs as prs
    sizes = [1, 2, 15, 16, 17, 127, 128, 129, 300, 640, 1023, 1024, 1025, 1500, 700, 33]
    des = _ragged_set(5, sizes)
    P = ops.prepare_descriptors(des)
    if sort_rows:
        P = ops.sort_rows_by_norm(P)
    all_pairs = prs.unordered_pairs(len(sizes))
    items = prs.pair_items(all_pairs, np.asarray(P.starts), np.asarray(P.counts))
    want = _oracle_counts(orc, des, all_pairs, 0.8)
    assert want.max() > 50 and (want == 0).any()
    st = {}
    got = ops.knn2_l2_pairs(P, items, 0.8, stats=st).cpu().numpy()
    assert np.array_equal(got, want), np.argwhere(got != want)[:10]
    # the hook that makes every query a candidate: small directions are recounted in the kernel, large ones overflow
    st1 = {}
    got1 = ops.knn2_l2_pairs(P, items, 0.8, flags=1, stats=st1).cpu().numpy()
    assert np.array_equal(got1, want)
    assert st1["overflow_directions"] > st["overflow_directions"]


def test_pairs_kernel_equals_directed_kernel_on_overlapping_crops(cuda, orc):
    """BASELINE configs[2] geometry at reduced image count (4096 kp x 128-d, 2 scenes x 12 images: neighbours share
    thousands of features -> their directions overflow, the others are settled by the filter)."""
    import torch
    from image_stitcher_amd import ops, pairs as prs
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, scene_of = overlapping_crops_descriptors(seed=21, n_scenes=2, imgs_per_scene=12, kp=4096, step=448)
    n, kp = des.shape[:2]
    starts, counts = np.arange(n, dtype=np.int64) * kp, np.full((n,), kp, np.int64)
    all_pairs = prs.unordered_pairs(n)
    src = torch.from_numpy(des.reshape(-1, 128)).cuda()
    P = ops.prepare_descriptor_tensor(src, starts, counts)
    want = ops.knn2_l2_items(P, prs.directed_items(all_pairs, starts, counts), 0.65).cpu().numpy().reshape(-1, 2)
    for Pk in (P, ops.sort_rows_by_norm(P)):
        st = {}
        got = ops.knn2_l2_pairs(Pk, prs.pair_items(all_pairs, starts, counts), 0.65, stats=st).cpu().numpy()
        assert np.array_equal(got, want), np.argwhere(got != want)[:10]
        assert 0 < st["overflow_directions"] < 2 * len(all_pairs)
    # three pairs against the oracle itself: neighbours, far apart in a scene, different scenes
    a = np.nonzero(scene_of == 0)[0]
    b = np.nonzero(scene_of == 1)[0]
    some = np.array([[a[0], a[1]], [a[0], a[-1]], [a[0], b[0]]])
    idx = [int(np.nonzero((all_pairs == np.sort(p)).all(1))[0][0]) for p in some]
    w = _oracle_counts(orc, des, all_pairs[idx], 0.65)
    assert np.array_equal(want[idx], w)


def test_sorted_rows_leave_counts_unchanged_and_shrink_the_candidate_lists(cuda):
    """Sorting an image's rows by norm is invisible in the counts (directed kernel on both layouts)."""
    from image_stitcher_amd import ops, pairs as prs
    des = _ragged_set(9, [900, 1100, 1300, 800])
    P = ops.prepare_descriptors(des)
    S = ops.sort_rows_by_norm(P)
    nrm = S.nrm.cpu().numpy()
    for s, c in zip(S.starts, S.counts):  # ascending, rotated by 64: the 64 smallest norms sit behind the largest
        assert (np.diff(nrm[s:s + c - 64]) >= 0).all() and (np.diff(nrm[s + c - 64:s + c]) >= 0).all()
        assert nrm[s + c - 64:s + c].max() <= nrm[s:s + c - 64].min()
    assert sorted(nrm.tolist()) == sorted(P.nrm.cpu().numpy().tolist())
    all_pairs = prs.unordered_pairs(4)
    it = prs.directed_items(all_pairs, np.asarray(P.starts), np.asarray(P.counts))
    assert np.array_equal(ops.knn2_l2_items(P, it, 0.7).cpu().numpy(), ops.knn2_l2_items(S, it, 0.7).cpu().numpy())


@pytest.mark.parametrize("d", [128, 61, 32])
def test_sort_rows_by_norm_is_the_stated_permutation(cuda, d):
    """isx_sort_rows_by_norm against numpy: per image the rows in (norm, row) order, rotated by 64 where the image has more than
    256 rows -- all four per-row arrays, ragged sets (empty, 1, 255 / 256 / 257 rows, beyond 4096 and at the 16384 limit) and
    the equal-count form; an image beyond the limit takes the torch route and gives the same norms."""
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(77 + d)
    for sizes in ([0, 1, 255, 256, 257, 1000, 0, 4096, 5000, 64], [16384, 3, 700], [512] * 7, [16385, 300]):
        des = [rng.integers(0, 40, (n, d)).astype(np.uint8) for n in sizes]  # (small values: many equal norms)
        P = ops.prepare_descriptors(des)
        S = ops.sort_rows_by_norm(P)
        nrm = P.nrm.cpu().numpy()
        exact = max(sizes) <= ops.SORT_MAX_ROWS
        for s, c in zip(P.starts, P.counts):
            s, c = int(s), int(c)
            order = np.lexsort((np.arange(c), nrm[s:s + c]))
            if c > 256:
                order = np.roll(order, -64)
            if exact:
                for a, b in ((P.desc, S.desc), (P.cbase, S.cbase), (P.nrm, S.nrm), (P.cin, S.cin)):
                    assert torch.equal(a[s:s + c][torch.from_numpy(order).to(a.device)], b[s:s + c]), (sizes, s, c)
            else:
                assert np.array_equal(nrm[s:s + c][order], S.nrm[s:s + c].cpu().numpy())


def test_pairs_kernel_refuses_what_it_cannot_run(cuda):
    from image_stitcher_amd import ops
    from image_stitcher_amd._lib import IsxError
    des = _ragged_set(1, [300, 10], d=32)
    P = ops.prepare_descriptors(des)
    with pytest.raises(IsxError):  # 32-byte rows
        ops.knn2_l2_pairs(P, np.array([[0, 300, 300, 10]], np.int32), 0.65)
    assert not ops.pairs_kernel_applies(P, np.array([[0, 300, 300, 10]] * 5000, np.int32))
    # images the two-group table cannot hold either: more than half of the pairs -> not worth the launch
    P128 = ops.prepare_descriptors(_ragged_set(2, [40, 10]))
    assert not ops.pairs_kernel_applies(P128, np.array([[0, 9000, 9000, 8500]] * 5000, np.int32))
    assert ops.pairs_kernel_applies(P128, np.array([[0, 9000, 9000, 8000]] * 5000, np.int32))


@pytest.mark.parametrize("sort_rows", [False, True])
def test_pairs_kernel_images_beyond_4096_rows(cuda, orc, sort_rows):
    """Images of more than 4096 rows (cv2.SIFT_create() does not cap the keypoints): the two-group table takes a smaller
    image of up to 8192 rows as the train side -- whichever side of the item it is -- and a larger one of up to 16384 rows in
    up to 16 passes; what it cannot hold (here 8300 x 9000) comes back as (-1, -1) and is re-run on the directed kernel by
    ops.knn2_l2_pairs.  Exact counts against the oracle, both directions, both item orientations."""
    from image_stitcher_amd import ops
    sizes = [4097, 8192, 700, 12000, 8300, 9000, 5000, 1]
    des = _ragged_set(9, sizes, dup_frac=0.05)
    P = ops.prepare_descriptors(des)
    if sort_rows:
        P = ops.sort_rows_by_norm(P)
    st, ct = np.asarray(P.starts), np.asarray(P.counts)
    pairs = [(0, 1), (1, 0), (2, 3), (3, 2), (4, 5), (1, 3), (3, 1), (6, 0), (7, 3), (3, 7), (6, 2), (4, 2)]
    items = np.array([(st[i], ct[i], st[j], ct[j]) for i, j in pairs], np.int32)
    want = _oracle_counts(orc, des, pairs, 0.8)
    assert want.max() > 20
    stt = {}
    got = ops.knn2_l2_pairs(P, items, 0.8, stats=stt).cpu().numpy()
    assert np.array_equal(got, want), (got.tolist(), want.tolist())
    assert stt["overflow_directions"] >= 2  # at least the 8300 x 9000 pair


@pytest.mark.parametrize("d", [61, 64])
def test_pairs_kernel_64_byte_rows(cuda, orc, d):
    """AKAZE-width descriptors (61 bytes, padded to 64-byte rows; 64 as well): one matrix instruction per tile and column
    block, the same filter and recount.  Ragged sizes incl. images beyond 4096 rows, sorted rows, against the oracle."""
    from image_stitcher_amd import ops, pairs as prs
    sizes = [1, 17, 128, 129, 1000, 1024, 1025, 2500, 4096, 5000, 640]
    des = _ragged_set(21, sizes, d=d)
    P = ops.sort_rows_by_norm(ops.prepare_descriptors(des))
    assert P.d_pad == 64
    all_pairs = prs.unordered_pairs(len(sizes))
    items = prs.pair_items(all_pairs, np.asarray(P.starts), np.asarray(P.counts))
    want = _oracle_counts(orc, des, all_pairs, 0.8)
    assert want.max() > 50 and (want == 0).any()
    got = ops.knn2_l2_pairs(P, items, 0.8).cpu().numpy()
    assert np.array_equal(got, want), np.argwhere(got != want)[:10]
    got1 = ops.knn2_l2_pairs(P, items, 0.8, flags=1).cpu().numpy()
    assert np.array_equal(got1, want)


def test_pairs_kernel_empty_images(cuda, orc):
    """Images without keypoints (update_all_keypoints_and_descriptors drops them, but a caller of the C ABI may pass them): every
    direction that involves one counts 0, with both tables, and the other pairs of the launch are unaffected."""
    from image_stitcher_amd import ops
    rng = np.random.default_rng(0)
    des = [rng.integers(0, 200, (n, 128), dtype=np.uint8) if n else None for n in (100, 0, 300, 5000, 0)]
    des[2][:60] = np.clip(des[0][:60].astype(np.int16) + rng.integers(-1, 2, (60, 128)), 0, 255).astype(np.uint8)
    P = ops.prepare_descriptors(des)
    st, ct = np.asarray(P.starts), np.asarray(P.counts)
    for pairs in ([(0, 1), (1, 0), (1, 4), (0, 2), (2, 1)], [(3, 1), (1, 3), (3, 0), (4, 3), (2, 0)]):
        items = np.array([(st[i], ct[i], st[j], ct[j]) for i, j in pairs], np.int32)
        got = ops.knn2_l2_pairs(P, items, 0.8).cpu().numpy()
        want = np.array([[0 if des[i] is None or des[j] is None else orc.match_count(des[i], des[j], 0.8),
                          0 if des[i] is None or des[j] is None else orc.match_count(des[j], des[i], 0.8)] for i, j in pairs], np.int32)
        assert np.array_equal(got, want), (got.tolist(), want.tolist())
    assert want.max() >= 50
