"""GPU parity of the two non-default matchers (through the C ABI) vs the CPU oracle:
Hamming (bit exact) and float32 L2 (1e-4 relative distance, BASELINE.json north_star)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planted(rng, nq, nt, d, flips=6):
    t = rng.integers(0, 256, (nt, d), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, d), dtype=np.uint8)
    k = min(nq, nt) // 2
    src = rng.permutation(nt)[:k]
    q[:k] = t[src]
    for i in range(k):  # flip a few bits of the planted copies
        for b in rng.integers(0, d * 8, flips):
            q[i, b // 8] ^= np.uint8(1 << (b % 8))
    return q, t


@pytest.mark.parametrize("nq,nt,d", [(500, 500, 32), (1, 1, 32), (3, 2, 32), (257, 1031, 32), (700, 650, 61), (64, 300, 64)])
def test_hamming_bit_exact(cuda, orc, nq, nt, d):
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(nq + 3 * nt + d)
    q, t = _planted(rng, nq, nt, d)
    if nt > 40:
        t[7] = t[3]       # duplicate rows: equal distances must resolve to the lower index
        t[nt - 1] = t[20]
    for ratio in (0.65, 0.8):
        idx, dist, cnt = ops.knn2_hamming(q, t, ratio)
        torch.cuda.synchronize()
        ridx, rdist = orc.knn2(q, t, norm=orc.NORM_HAMMING)
        assert np.array_equal(idx.cpu().numpy(), ridx)
        assert np.array_equal(dist.cpu().numpy().view(np.int32), rdist.view(np.int32))
        gq, _ = orc.ratio_filter(ridx, rdist, ratio)
        assert int(cnt.item()) == len(gq)


@pytest.mark.parametrize("nq,nt,d", [(300, 500, 128), (1, 1, 128), (5, 2, 64), (129, 257, 61), (1000, 1200, 128)])
def test_f32_within_tolerance(cuda, orc, nq, nt, d):
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(nq + nt + d)
    t = rng.gamma(1.0, 30.0, (nt, d)).astype(np.float32)  # non-integer, SIFT-like magnitudes
    q = rng.gamma(1.0, 30.0, (nq, d)).astype(np.float32)
    k = min(nq, nt) // 2
    q[:k] = t[rng.permutation(nt)[:k]] + rng.normal(0, 2.0, (k, d)).astype(np.float32)
    idx, dist, cnt = ops.knn2_f32(q, t, 0.65)
    torch.cuda.synchronize()
    idx, dist = idx.cpu().numpy(), dist.cpu().numpy()
    ridx, rdist = orc.knn2(q, t)
    K = min(2, nt)
    rel = np.abs(dist[:, :K] - rdist[:, :K]) / np.maximum(rdist[:, :K], 1e-6)
    assert rel.max() <= 1e-4, f"max relative distance error {rel.max():.3g}"   # the north_star gate
    # indices agree wherever the oracle's own neighbours are separated by more than the tolerance
    if nt >= 3:
        clear = (rdist[:, 1] - rdist[:, 0]) > 1e-4 * rdist[:, 1]
        assert np.array_equal(idx[clear, 0], ridx[clear, 0])
    else:
        assert np.array_equal(idx[:, :K], ridx[:, :K])
    if nt == 1:
        assert (idx[:, 1] == -1).all()
    gq, _ = orc.ratio_filter(ridx, rdist, 0.65)
    assert abs(int(cnt.item()) - len(gq)) <= max(1, len(gq) // 200)


def test_f32_integer_valued_matches_exact_path(cuda, orc):
    """On integer-valued floats every f32 operation is exact, so the fp32-MFMA path must agree bit for bit
    with the oracle (and with the int8 path)."""
    import torch
    from image_stitcher_amd import ops
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, _ = overlapping_crops_descriptors(seed=21, n_scenes=1, imgs_per_scene=2, kp=900, step=300)
    q, t = des[0].astype(np.float32), des[1].astype(np.float32)
    idx, dist, cnt = ops.knn2_f32(q, t, 0.65)
    idx2, dist2, cnt2 = ops.knn2_l2(q, t, 0.65)
    torch.cuda.synchronize()
    ridx, rdist = orc.knn2(q, t)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(idx2.cpu().numpy(), ridx)
    assert np.array_equal(dist.cpu().numpy().view(np.int32), rdist.view(np.int32))
    assert int(cnt.item()) == int(cnt2.item())


@pytest.mark.parametrize("d", [32, 16, 7])
def test_hamming_counts_on_the_fp4_matrix_pipe(cuda, orc, d):
    """Bit vectors as FP4 0/1 nibbles through v_mfma_scale_f32_32x32x64_f8f6f4: the good-match counts equal
    the XOR/popcount kernel's and the oracle's, on ragged sets with planted neighbours, duplicate rows
    (ties), a single-row image (single-match rows bypass the ratio test) and all-identical rows."""
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(100 + d)
    sizes = [1, 2, 33, 257, 700, 129, 64, 500]
    des = [rng.integers(0, 256, (s, d), dtype=np.uint8) for s in sizes]
    flips = (rng.random((30, d)) < 0.1) * (1 << rng.integers(0, 8, (30, d)))
    des[4][10:40] = des[3][5:35] ^ flips.astype(np.uint8)      # planted near neighbours
    des[4][100] = des[4][101]                                  # duplicate train rows
    des[5][:] = des[5][0]                                      # every distance ties
    des[6][:] = 0                                              # zero popcount rows
    n = len(des)
    pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
    items_of = lambda st, ct: np.array([(st[i], ct[i], st[j], ct[j]) for i, j in pairs], np.int32)  # noqa: E731
    for ratio in (0.65, 0.9):
        exp = np.array([orc.match_count(des[i], des[j], ratio, norm=orc.NORM_HAMMING) for i, j in pairs], np.int32)
        got = {impl: ops.alt_item_counts(des, items_of, "hamming", ratio, cuda, hamming_impl=impl).cpu().numpy()
               for impl in ("valu", "fp4")}
        assert np.array_equal(got["valu"], exp)
        assert np.array_equal(got["fp4"], exp), f"fp4 differs on items {np.nonzero(got['fp4'] != exp)[0][:10]}"
    assert exp.max() > 20
    with pytest.raises(ops.IsxError):
        ops.alt_item_counts([rng.integers(0, 256, (8, 61), dtype=np.uint8)] * 2, lambda st, ct: np.array([(0, 8, 8, 8)], np.int32),
                            "hamming", 0.65, cuda, hamming_impl="fp4")


@pytest.mark.parametrize("d", [61, 32, 128])
def test_saturated_byte_descriptors_stay_exact(cuda, orc, d):
    """Dense / saturated byte descriptors (AKAZE's 61 bytes, ORB's 32, 128): squared distances reach 8.3e6, where
    sqrtf is no longer injective on integers, so different integer distances collide as float32 and cv2 keeps the
    lower train index.  The MFMA kernels rank on integers; such sets go to isx_knn2_l2_wide_items, which ranks on the
    float distances -- indices and distances bit-identical to the oracle, planted float-collisions included."""
    import torch
    from image_stitcher_amd import ops
    rng = np.random.default_rng(17 + d)
    nq, nt = 300, 700
    q = rng.choice(np.array([0, 1, 254, 255], np.uint8), (nq, d))
    t = rng.choice(np.array([0, 1, 2, 253, 254, 255], np.uint8), (nt, d))
    q[0] = 255
    t[5] = 0                      # the largest possible distance: d * 255^2
    t[9] = 0
    t[9, 0] = 1                   # integer distance differs by 509 from row 5's: may or may not collide as float
    # planted float collisions: rows whose integer distances to q[1] differ by 1 at s ~ 6e6 (sqrtf spacing > 1 there)
    q[1] = 255
    base = np.zeros(d, np.uint8)
    t[20], t[21] = base, base.copy()
    t[21, 1] = 1                  # s differs from row 20's by 2*255 - 1
    P = ops.prepare_descriptors([q, t])
    if d * 255 * 255 * 2 >= 4197200:
        assert P.wide
    idx, dist, cnt = ops.knn2_l2(q, t, 0.65)
    ridx, rdist = orc.knn2(q, t)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(dist.cpu().numpy().view(np.int32), rdist.view(np.int32))
    gq, _ = orc.ratio_filter(ridx, rdist, 0.65)
    assert int(cnt.item()) == len(gq)
    # count-only form through the item interface (what get_collections uses)
    items = np.array([[0, nq, nq, nt], [nq, nt, 0, nq]], np.int32)
    c = ops.knn2_l2_items(P, items, 0.65).cpu().numpy()
    assert c[0] == len(gq) and c[1] == orc.match_count(t, q, 0.65)


def test_wide_kernel_equals_mfma_kernel_inside_the_limit(cuda, orc):
    """Inside the exact range both kernels must give the same answer: SIFT-like rows forced through the wide kernel."""
    import torch
    from image_stitcher_amd import ops
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, _ = overlapping_crops_descriptors(seed=5, n_scenes=1, imgs_per_scene=2, kp=900, step=200)
    P = ops.prepare_descriptors([des[0], des[1]])
    assert not P.wide
    items = np.array([[0, 900, 900, 900]], np.int32)
    c0, (i0, d0, _) = ops.knn2_l2_items(P, items, 0.65, want_matches=True, xcd_order=False)
    P.wide = True
    c1, (i1, d1, _) = ops.knn2_l2_items(P, items, 0.65, want_matches=True, xcd_order=False)
    assert torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32)) and torch.equal(c0, c1)
