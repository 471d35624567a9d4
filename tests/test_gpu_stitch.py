"""End-to-end GPU tests of the drop-in host layer (ConsecutiveStitcher / ArbitraryStitcher) against the
reference pipeline restated on the CPU oracle (tests/oracle_stitch.py).  Detection uses the same
deterministic numpy detector on both sides, so both paths see identical descriptors."""
import numpy as np
import pytest

from oracle_stitch import Cfg, oracle_stitch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    from image_stitcher_amd.synthetic import procedural_world
    return procedural_world(seed=42, h=800, w=2200)


def _frames(world, n, hw=(240, 320), seed=1, rot=1.0):
    from image_stitcher_amd.synthetic import crop_path
    return crop_path(world, n, hw, seed=seed, max_rot_deg=rot)[0]


def _apply(M, p):
    M = np.vstack([M, [0, 0, 1]]) if M.shape[0] == 2 else M
    q = np.c_[p, np.ones(len(p))] @ M.T
    return q[:, :2] / q[:, 2:3]


@pytest.mark.parametrize("type_,blender,contrib", [(0, 1, 0.2), (1, 0, 0.5), (0, 0, 0.2), (1, 1, 0.2)])
def test_consecutive_stream_matches_oracle_frame_by_frame(cuda, orc, world, tmp_path, type_, blender, contrib):
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 6)
    det = NumpyOrbLite(nfeatures=600)
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), type=type_, blender=blender,
                             ref_image_contrib=contrib, consecutive_range=1, algorithm=1, detector=det)
    cfg = Cfg(type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1)
    assert st.stitch_consecutive(frames[0]) is None and st.refs[-1] is frames[0]
    for k, f in enumerate(frames[1:]):
        ref_before = np.array(np.asarray(st.refs[-1]))
        corners_before = st.corners
        tf = st.stitch_consecutive(f)
        assert tf is not None, f"frame {k + 1} failed to stitch"
        got = np.asarray(st.refs[-1])
        # (a) the oracle's own run from the same inputs: identical good matches / points, M within tolerance
        o = oracle_stitch(ref_before, f, det, cfg, corners=corners_before)
        assert o not in (None, False)
        oM, osrc, odst = o[2]
        assert np.array_equal(tf.src_pts, osrc) and np.array_equal(tf.dst_pts, odst)
        assert tf.src_pts.dtype == np.float32 and tf.src_pts.shape[1:] == (1, 2)
        rms = np.sqrt(((_apply(tf.M, osrc.reshape(-1, 2).astype(np.float64)) - _apply(oM, osrc.reshape(-1, 2).astype(np.float64))) ** 2).sum(1).mean())
        assert rms < 1e-3  # gate: 0.5 px
        # (b) warp + blend + bbox + crop fed the SAME transform: bit exact mosaic and corners
        o2 = oracle_stitch(ref_before, f, det, cfg, tf=(tf.M, tf.src_pts, tf.dst_pts), corners=corners_before)
        assert got.shape == o2[0].shape
        assert np.array_equal(got, o2[0]), f"{(got != o2[0]).sum()} pixels differ"
        assert st.corners == o2[1]
    assert st.image_count == 5 and st.current_image_count == 5 and st.stitch_count == 0
    assert got.shape[1] > frames[0].shape[1] * 1.5  # the mosaic really grew


def test_rgb_plus_depth_reusing_tf(cuda, orc, world, tmp_path):
    """BASELINE config 4 (README 'synchronised stitching'): the depth stitcher reuses the RGB transform
    (warp/blend-only path, blender=0, ref_image_contrib=0.5); uint16 depth wraps to uint8 exactly as the
    reference's pad_image does (utils/proc.py:283-291)."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 4)
    rng = np.random.default_rng(0)
    depth = [(np.linspace(300, 3000, f.shape[1])[None, :] + 40 * k + rng.integers(0, 3, f.shape[:2])).astype(np.uint16)
             for k, f in enumerate(frames)]
    det = NumpyOrbLite(nfeatures=600)
    kw = dict(log_level=3, output_dir=str(tmp_path), type=0, blender=0, ref_image_contrib=0.5, consecutive_range=1)
    rgb = ConsecutiveStitcher(algorithm=1, detector=det, **kw)
    dep = ConsecutiveStitcher(algorithm=1, detector=det, **kw)
    cfg = Cfg(type=0, blender=0, ref_image_contrib=0.5, consecutive_range=1)
    for k in range(len(frames)):
        dref = None if dep.refs[-1] is None else np.array(np.asarray(dep.refs[-1])).astype(np.uint8)
        tf = rgb.stitch_consecutive(frames[k])
        tf2 = dep.stitch_consecutive(depth[k], tf)
        if k == 0:
            assert tf is None and tf2 is None
            continue
        assert tf2 is tf
        o = oracle_stitch(dref, depth[k].astype(np.uint8), det, cfg, tf=(tf.M, tf.src_pts, tf.dst_pts))
        assert np.array_equal(np.asarray(dep.refs[-1]), o[0])


def _oracle_depth16(orc, ref, img, M, contrib, persp):
    """stitcher.py:176-229 on uint16 canvases (the opt-in depth-preserving mode): pad both, warp, alpha merge, corners of the
    warped image, crop -- orc_warp16 / orc_alpha_blend16, bounding boxes by numpy."""
    def bbox(a):
        ys, xs = np.nonzero(a)
        return False if len(ys) == 0 else ((int(xs.min()), int(ys.min())), (int(xs.max()) + 1, int(ys.max()) + 1))
    ih, iw = img.shape
    rh, rw = ref.shape
    H, W = rh + 2 * ih, rw + 2 * iw
    cur, refc = np.zeros((H, W), np.uint16), np.zeros((H, W), np.uint16)
    cur[(H - ih) // 2:(H - ih) // 2 + ih, (W - iw) // 2:(W - iw) // 2 + iw] = img
    refc[ih:ih + rh, iw:iw + rw] = ref
    warped = orc.warp16(cur, M, (W, H), persp)
    merged = orc.alpha_blend16(warped, refc, contrib)
    (x0, y0), (x1, y1) = bbox(merged)
    return merged[y0:y1, x0:x1], bbox(warped)


@pytest.mark.parametrize("type_", [0, 1])
def test_rgb_plus_depth16_keeps_sixteen_bits(cuda, orc, world, tmp_path, type_):
    """BASELINE configs[3] in the opt-in depth-preserving mode (depth16=True): the depth stitcher reuses the RGB transform
    (warp/blend-only path, blender=0, ref_image_contrib=0.5) and the uint16 mosaic keeps its 16 bits -- every frame
    against the oracle's uint16 restatement of the same steps (warp in float32 linear arithmetic, round half to even;
    v > 10 masks; truncated float64 products; uint16 sum).  Values above 255, above 32767 and a saturating 65535 patch are in
    the frames; the default (depth16=False) wraps modulo 256 and is covered by test_rgb_plus_depth_reusing_tf."""
    from image_stitcher_amd import ConsecutiveStitcher, TransformDetails
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 4)
    rng = np.random.default_rng(0)
    depth = []
    for k, f in enumerate(frames):
        d = (np.linspace(300, 60000, f.shape[1])[None, :] + 40 * k + rng.integers(0, 3, f.shape[:2])).astype(np.uint16)
        d[5:25, 30:60] = 65535
        d[40:60, 10:40] = rng.integers(0, 12, (20, 30))  # values around the mask threshold (10) and zeros
        depth.append(d)
    det = NumpyOrbLite(nfeatures=600)
    kw = dict(log_level=3, output_dir=str(tmp_path), type=type_, blender=0, ref_image_contrib=0.5, consecutive_range=1)
    rgb = ConsecutiveStitcher(algorithm=1, detector=det, **kw)
    dep = ConsecutiveStitcher(algorithm=1, detector=det, depth16=True, **kw)
    for k in range(len(frames)):
        dref = None if dep.refs[-1] is None else np.array(np.asarray(dep.refs[-1]))
        tf = rgb.stitch_consecutive(frames[k])
        tf2 = dep.stitch_consecutive(depth[k], tf)
        if k == 0:
            assert tf is None and tf2 is None
            continue
        assert tf2 is tf and dref.dtype == np.uint16
        want, wc = _oracle_depth16(orc, dref, depth[k], tf.M, 0.5, type_ == 1)
        got = np.asarray(dep.refs[-1])
        assert got.dtype == np.uint16 and got.shape == want.shape and np.array_equal(got, want), (got != want).sum()
        assert dep.corners == wc
    assert got.max() == 65535 and (got > 32767).any()
    dep.save_last_stitch()  # blend_0.png, 16-bit
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "blend_0.png")).astype(np.uint16), got)
    # detection on depth maps is not offered, the gradient blender casts to uint8 in the reference: both refuse loudly
    from image_stitcher_amd._lib import IsxError
    with pytest.raises(IsxError):
        dep.stitch(depth[0], depth[1])
    with pytest.raises(IsxError):
        ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), blender=1, depth16=True).stitch(
            depth[0], depth[1], tf=TransformDetails(np.eye(2, 3), None, None))


def test_failed_frame_resets_the_mosaic(cuda, world, tmp_path):
    """stitcher.py:425-433: a frame that cannot be matched saves blend_N.png, starts a new mosaic and is
    itself dropped; the caller's tf (None) comes back."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 3)
    from image_stitcher_amd.synthetic import procedural_world
    other = procedural_world(seed=99, h=240, w=320)  # unrelated content: no overlap
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), consecutive_range=2, algorithm=1,
                             min_match_count=40, detector=NumpyOrbLite(nfeatures=400),
                             consecutive_volatility_threshold=0)
    st.stitch_consecutive(frames[0])
    assert st.stitch_consecutive(frames[1]) is not None
    assert st.stitch_consecutive(other) is None
    assert st.stitch_count == 1 and st.refs[-1] is None and st.corners is None and st.current_image_count == 0
    assert (tmp_path / "blend_0.png").exists()
    assert len(st.refs) == 2 and st.refs[0] is not None  # the ring keeps the old mosaic
    st.stitch_consecutive(frames[2])
    assert st.refs[-1] is frames[2]  # first frame of the new mosaic, no work


def test_get_collections_small(cuda, orc, world):
    """ArbitraryStitcher.get_collections (stitcher.py:280-304) incl. the phantom-singleton quirk."""
    from image_stitcher_amd import ArbitraryStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    a = _frames(world, 4, seed=3)
    from image_stitcher_amd.synthetic import crop_path
    b = crop_path(world, 3, (240, 320), seed=5, start=(1500.0, 400.0))[0]  # far from the a-chain: no overlap
    black = np.zeros_like(a[0])
    order = [a[0], b[0], a[1], black, b[1], a[2], b[2], a[3]]
    det = NumpyOrbLite(nfeatures=500)
    st = ArbitraryStitcher(images=list(order), with_tqdm=False, log_level=3, algorithm=1, detector=det,
                           min_match_count=40)  # unrelated crops reach ~20 chance matches with this detector
    cols = st.get_collections()
    # oracle: same loop on the CPU
    des = [det.detectAndCompute(im)[1] for im in order]
    keep = [i for i, d in enumerate(des) if d is not None]
    des = [des[i] for i in keep]
    n = len(des)
    from image_stitcher_amd.utils.structs import DynamicConnectivity
    dc = DynamicConnectivity(len(order))
    for i in range(n):
        for j in range(n):
            if i != j and orc.match_count(des[i], des[j], 0.65) >= 40:
                dc.union(i, j)
    exp = sorted(dc.get_connected_components(), key=len, reverse=True)
    assert cols == exp
    assert [len(c) for c in cols] == [4, 3, 1]  # a-chain, b-chain, phantom trailing singleton (index 7)
    assert len(st.des) == 7 and len(st.images) == 7


@pytest.mark.parametrize("width,dtype", [(128, np.float32), (61, np.uint8)])
def test_get_collections_through_the_pair_kernel(cuda, orc, width, dtype):
    """ArbitraryStitcher.get_collections (stitcher.py:280-304) on a set large enough (40 images, 780 pairs) to take the
    product's round-3 route: rows sorted by norm, isx_knn2_l2_count_pairs, overflowing directions re-run on the directed
    kernel.  Integer-valued float32 128-d descriptors as cv2.SIFT returns them, and 61-byte AKAZE-width ones (64-byte rows).
    Every pair's two counts and the components against the oracle's double loop."""
    from image_stitcher_amd import ArbitraryStitcher, ops, pairs as prs
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, scene_of = overlapping_crops_descriptors(seed=31, n_scenes=4, imgs_per_scene=10, kp=700, step=120, d=width)
    arrays = [d.astype(dtype) for d in des]

    class Stub:
        def __init__(self):
            self.k = 0

        def detectAndCompute(self, image, mask=None):
            d = arrays[self.k % len(arrays)]
            self.k += 1
            return [None] * len(d), d
    n = len(arrays)
    st = ArbitraryStitcher(images=[np.zeros((8, 8, 3), np.uint8)] * n, with_tqdm=False, log_level=3, detector=Stub())
    cols = st.get_collections()
    pairs, counts = st.good_counts
    assert len(pairs) == n * (n - 1) // 2 >= ops.PAIR_MIN_ITEMS
    want = np.array([[orc.match_count(arrays[i], arrays[j], 0.65, nthreads=8), orc.match_count(arrays[j], arrays[i], 0.65, nthreads=8)]
                     for i, j in pairs], np.int32)
    assert np.array_equal(np.asarray(counts), want), np.argwhere(np.asarray(counts) != want)[:10]
    assert want.max() > 100
    exp, _ = prs.components_from_counts(n, pairs, want[:, 0], want[:, 1], 5)
    assert cols == exp and sorted(map(sorted, cols)) == sorted(sorted(np.nonzero(scene_of == s)[0].tolist()) for s in range(4))


def test_no_gpu_free_cpu_path(cuda):
    """The product never falls back to the CPU: CPU tensors are rejected loudly."""
    import torch
    from image_stitcher_amd import ops
    with pytest.raises(ops.IsxError):
        ops.bbox_nonzero(torch.zeros((4, 4), dtype=torch.uint8))


def test_matcher_routing_hamming_and_f32(cuda, orc, world):
    """Extension keyword norm='hamming' (XOR/popcount) and automatic routing of non-integer float32
    descriptors to the fp32-MFMA matcher; both checked against the oracle."""
    from image_stitcher_amd import ArbitraryStitcher, Stitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    det = NumpyOrbLite(nfeatures=400)
    frames = _frames(world, 3, seed=8)
    des = [det.detectAndCompute(f)[1] for f in frames]
    st = Stitcher(log_level=3, algorithm=1, detector=det, norm="hamming")
    assert st._matcher_kind(des[0]) == "hamming"
    good = st.filter_matches(st.get_matches(des[0], des[1], 2))
    ridx, rdist = orc.knn2(des[0], des[1], norm=orc.NORM_HAMMING)
    gq, gt = orc.ratio_filter(ridx, rdist, 0.65)
    assert np.array_equal(good.good_q, gq) and np.array_equal(good.good_t, gt)
    # get_collections with the Hamming matcher
    ar = ArbitraryStitcher(images=list(frames), with_tqdm=False, log_level=3, algorithm=1, detector=det, norm="hamming",
                           min_match_count=10)
    ar.get_collections()
    pairs, counts = ar.good_counts
    for (i, j), (cij, cji) in zip(pairs, counts):
        assert cij == orc.match_count(des[i], des[j], 0.65, norm=orc.NORM_HAMMING)
        assert cji == orc.match_count(des[j], des[i], 0.65, norm=orc.NORM_HAMMING)
    # non-integer float descriptors -> fp32 path
    f0, f1 = des[0].astype(np.float32) * 0.37, des[1].astype(np.float32) * 0.37
    st2 = Stitcher(log_level=3, detector=det)
    assert st2._matcher_kind(f0) == "f32" and st2._matcher_kind(des[0].astype(np.float32)) == "int"
    m = st2.get_matches(f0, f1, 2)
    ridx, rdist = orc.knn2(f0, f1)
    got = m.dist.cpu().numpy()
    assert np.abs(got - rdist).max() <= 1e-4 * rdist.max()


def test_stitch_collections_greedy_merge(cuda, orc, world, tmp_path):
    """ArbitraryStitcher.stitch_collections (stitcher.py:306-374) on resident mosaics.  Every
    Stitcher.stitch call of the greedy loop is recorded and replayed on the oracle pipeline: call order,
    corner carry-over, None/tuple results, bit-exact mosaics given the same transform; one blend_i.png per
    collection; the return value is [] as in the reference."""
    from image_stitcher_amd import ArbitraryStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    from image_stitcher_amd.synthetic import crop_path
    from PIL import Image
    a = _frames(world, 3, seed=3)
    b = crop_path(world, 2, (240, 320), seed=5, start=(1500.0, 400.0))[0]
    order = [a[0], b[0], a[2], a[1], b[1]]  # two interleaved chains, the a-chain out of path order
    det = NumpyOrbLite(nfeatures=500)
    st = ArbitraryStitcher(images=list(order), with_tqdm=False, log_level=3, algorithm=1, detector=det,
                           min_match_count=40, output_dir=str(tmp_path))
    cols = st.get_collections()
    assert sorted(map(sorted, cols)) == [[0, 2, 3], [1, 4]]
    calls = []
    inner = st.stitch

    def recording_stitch(ref, img, tf=None, corners=None):
        ref_h = np.array(np.asarray(ref))
        res = inner(ref, img, tf=tf, corners=corners)
        calls.append((ref_h, np.asarray(img), corners, res))
        return res
    st.stitch = recording_stitch
    assert st.stitch_collections(cols) == []
    cfg = Cfg(type=0, blender=1, ref_image_contrib=0.5, min_match_count=40)
    # replay: the same greedy loop driven by the oracle pipeline must make the same calls in the same order
    it = iter(calls)
    finals = []
    for coll in cols:
        idxs = list(coll)
        ref_img = np.asarray(order[idxs.pop(0)])
        while idxs:
            matched, corners, done = False, None, []
            for k in idxs:
                ref_h, img_h, c_in, res = next(it)
                assert np.array_equal(ref_h, ref_img) and img_h is order[k] and c_in == corners
                o = oracle_stitch(ref_img, order[k], det, cfg, corners=corners)
                assert (o is None) == (res is None) and o is not False and res is not False
                if res is None:
                    corners = None
                    continue
                mosaic, corners, tf = res
                o2 = oracle_stitch(ref_img, order[k], det, cfg, tf=(tf.M, tf.src_pts, tf.dst_pts), corners=c_in)
                assert np.array_equal(tf.src_pts, o[2][1]) and np.array_equal(np.asarray(mosaic), o2[0]) and corners == o2[1]
                ref_img = o2[0]
                matched = True
                done.append(k)
            for k in done:
                idxs.remove(k)
            if not matched:
                break
        finals.append(ref_img)
    assert next(it, None) is None
    for i, f in enumerate(finals):
        png = np.asarray(Image.open(tmp_path / f"blend_{i}.png"))
        assert np.array_equal(png[..., ::-1] if png.ndim == 3 else png, f)
        assert f.shape[1] > 320 * 1.1  # every collection really grew


def test_async_save_writes_the_same_files(cuda, world, tmp_path):
    """async_save=True copies and encodes blend_N.png off the stitching stream; after
    flush_saves() the files are identical to the synchronous stitcher's, in the same order, and the
    stitching results themselves are unchanged."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    from image_stitcher_amd.synthetic import procedural_world
    from PIL import Image
    frames = _frames(world, 5)
    other = procedural_world(seed=99, h=240, w=320)
    seq = frames[:3] + [other] + frames[3:]          # the unrelated frame forces a save_and_reset in mid-stream
    outs = {}
    for mode in (False, True):
        d = tmp_path / ("async" if mode else "sync")
        st = ConsecutiveStitcher(log_level=3, output_dir=str(d), consecutive_range=1, algorithm=1, min_match_count=40,
                                 detector=NumpyOrbLite(nfeatures=500), backup_interval=1, async_save=mode,
                                 blend_processor=lambda a: a[::-1].copy())
        mosaics = []
        for f in seq:
            st.stitch_consecutive(f)
            mosaics.append(None if st.refs[-1] is None else np.array(np.asarray(st.refs[-1])))
        st.save_and_reset()
        st.flush_saves()
        files = sorted(p.name for p in d.iterdir())
        outs[mode] = (files, {n: np.asarray(Image.open(d / n)) for n in files}, mosaics, st.stitch_count)
    assert outs[False][0] == outs[True][0] and len(outs[True][0]) >= 2
    for n in outs[False][0]:
        assert np.array_equal(outs[False][1][n], outs[True][1][n]), n
    assert outs[False][3] == outs[True][3]
    for a, b in zip(outs[False][2], outs[True][2]):
        assert (a is None) == (b is None) and (a is None or np.array_equal(a, b))


@pytest.mark.parametrize("type_", [0, 1])
def test_consecutive_stream_classic_warp_arithmetic(cuda, orc, world, tmp_path, type_):
    """warp_arith="classic" (OpenCV <= 4.10 fixed-point remap) end to end: the mosaic is bit-identical to the
    oracle pipeline run with the same arithmetic, and differs from the float32-kernel mosaic."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 4)
    det = NumpyOrbLite(nfeatures=600)
    kw = dict(log_level=3, output_dir=str(tmp_path), type=type_, blender=1, consecutive_range=1, algorithm=1, detector=det)
    st = ConsecutiveStitcher(warp_arith="classic", **kw)
    cfg = Cfg(type=type_, blender=1, consecutive_range=1, warp_arith=1)
    st.stitch_consecutive(frames[0])
    differs = False
    for f in frames[1:]:
        ref_before = np.array(np.asarray(st.refs[-1]))
        corners_before = st.corners
        tf = st.stitch_consecutive(f)
        assert tf is not None
        o = oracle_stitch(ref_before, f, det, cfg, tf=(tf.M, tf.src_pts, tf.dst_pts), corners=corners_before)
        assert np.array_equal(np.asarray(st.refs[-1]), o[0]) and st.corners == o[1]
        o_b = oracle_stitch(ref_before, f, det, Cfg(type=type_, blender=1, consecutive_range=1),
                            tf=(tf.M, tf.src_pts, tf.dst_pts), corners=corners_before)
        differs |= o_b[0].shape != o[0].shape or bool((o_b[0] != o[0]).any())
    assert differs
    with pytest.raises(ValueError):
        ConsecutiveStitcher(warp_arith="nearest", **kw)


def test_stitch_return_contracts(cuda, world, tmp_path):
    """stitcher.py:196-198 (False: no descriptors in the reference image), :210-212 (None: too few good
    matches), :208/:216 (M is None is not guarded: cv2 raises; the build raises too), tf given => no
    detection at all (:213-214)."""
    from image_stitcher_amd import Stitcher, TransformDetails
    from image_stitcher_amd.detect import NumpyOrbLite

    class Counting(NumpyOrbLite):
        calls = 0

        def detectAndCompute(self, image, mask=None):
            Counting.calls += 1
            return super().detectAndCompute(image, mask)
    frames = _frames(world, 2)
    det = Counting(nfeatures=400)
    st = Stitcher(log_level=3, output_dir=str(tmp_path), algorithm=1, detector=det)
    black = np.zeros_like(frames[0])                        # nothing to detect: the detector returns des = None
    assert st.stitch(black, frames[1]) is False
    from image_stitcher_amd.synthetic import procedural_world
    other = procedural_world(seed=99, h=240, w=320)
    st.min_match_count = 40
    assert st.stitch(other, frames[1]) is None
    st.min_match_count = 5
    res = st.stitch(frames[0], frames[1])
    mosaic, corners, tf = res
    assert isinstance(tf, TransformDetails) and corners is None      # corners only with consecutive_range (:219-223)
    before = Counting.calls
    res2 = st.stitch(frames[0], frames[1], tf=tf)
    assert Counting.calls == before and np.array_equal(np.asarray(res2[0]), np.asarray(mosaic)) and res2[2] is tf
    with pytest.raises(RuntimeError):
        st.stitch(frames[0], frames[1], tf=TransformDetails(None, tf.src_pts, tf.dst_pts))


@pytest.mark.parametrize("blender", [0, 1])
def test_overridden_seams_are_honoured(cuda, world, tmp_path, blender):
    """A subclass that overrides the reference's seam methods (warp_target / merge /
    detectAndComputeFromCorners, stitcher.py:143-165) gets them called with padded canvases, as in the
    reference; with pass-through overrides the mosaic equals the default region path bit for bit."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    calls = {"warp": 0, "merge": 0, "detect": 0}

    class Sub(ConsecutiveStitcher):
        def warp_target(self, img, ref, M):
            calls["warp"] += 1
            assert img.shape == ref.shape                     # both are padded canvases
            return super().warp_target(img, ref, M)

        def merge(self, warped_image, ref):
            calls["merge"] += 1
            return super().merge(warped_image, ref)

        def detectAndComputeFromCorners(self, image, corners):
            calls["detect"] += 1
            return super().detectAndComputeFromCorners(image, corners)
    frames = _frames(world, 4)
    kw = dict(log_level=3, output_dir=str(tmp_path), blender=blender, consecutive_range=1, algorithm=1)
    a = ConsecutiveStitcher(detector=NumpyOrbLite(nfeatures=600), **kw)
    b = Sub(detector=NumpyOrbLite(nfeatures=600), **kw)
    for f in frames:
        ta, tb = a.stitch_consecutive(f), b.stitch_consecutive(f)
        assert (ta is None) == (tb is None)
        if ta is not None:
            assert np.array_equal(ta.src_pts, tb.src_pts) and np.array_equal(ta.M, tb.M)
        assert np.array_equal(np.asarray(a.refs[-1]), np.asarray(b.refs[-1])) and a.corners == b.corners
    assert calls["warp"] == 3 and calls["merge"] == 3 and calls["detect"] == 4   # ROI detection from the 2nd stitch on


def test_saturated_byte_descriptors_stay_on_an_exact_path(cuda, orc):
    """Dense binary descriptors read as bytes (AKAZE/ORB rows full of 0xFF) can push squared L2 distances
    past 2^22, where sqrtf stops being injective and the integer ranking is no longer provably cv2's float
    ranking: the MFMA path refuses such a set (wide_ok=False) and the host layer matches it with the kernel that
    ranks on the float distances cv2 compares -- indices AND distances bit-identical to the oracle."""
    from image_stitcher_amd import ArbitraryStitcher, Stitcher, ops
    rng = np.random.default_rng(61)
    base = np.where(rng.random((400, 61)) < 0.8, 255, rng.integers(0, 256, (400, 61))).astype(np.uint8)
    q = base[:300].copy()
    t = base[100:].copy()
    flip = rng.random(t.shape) < 0.05
    t[flip] = rng.integers(0, 256, int(flip.sum()), dtype=np.uint8)
    with pytest.raises(ops.NormLimitError):
        ops.prepare_descriptors([q, t], wide_ok=False)
    st = Stitcher(log_level=3)
    m = st.get_matches(q, t, 2)
    ridx, rdist = orc.knn2(q, t)
    assert np.array_equal(m.idx.cpu().numpy(), ridx)
    assert np.array_equal(m.dist.cpu().numpy().view(np.int32), rdist.view(np.int32))
    good = st.filter_matches(m)
    gq, gt = orc.ratio_filter(ridx, rdist, 0.65)
    assert np.array_equal(good.good_q, gq) and np.array_equal(good.good_t, gt)

    class Fixed:                                   # a detector that hands out the prepared descriptors
        def __init__(self, des):
            self.des, self.k = des, 0

        def detectAndCompute(self, image, mask=None):
            d = self.des[self.k]
            self.k += 1
            return [None] * len(d), d
    ar = ArbitraryStitcher(images=[np.zeros((8, 8, 3), np.uint8)] * 2, with_tqdm=False, log_level=3, detector=Fixed([q, t]))
    ar.get_collections()
    pairs, counts = ar.good_counts
    assert counts.tolist() == [[orc.match_count(q, t, 0.65), orc.match_count(t, q, 0.65)]]


def test_resident_false_returns_ndarrays(cuda, world, tmp_path):
    """resident=False: refs[-1] and the first element of stitch's result are plain ndarrays (what the reference
    hands to user code); same pixels as the resident mode; grayscale frames work too (2-D arrays)."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    from image_stitcher_amd.device_image import DeviceImage
    frames = _frames(world, 3)
    gray = [np.ascontiguousarray(f[..., 1]) for f in frames]
    for seq in (frames, gray):
        kw = dict(log_level=3, output_dir=str(tmp_path), consecutive_range=1, algorithm=1)
        a = ConsecutiveStitcher(detector=NumpyOrbLite(nfeatures=500), **kw)
        b = ConsecutiveStitcher(detector=NumpyOrbLite(nfeatures=500), resident=False, **kw)
        for f in seq:
            a.stitch_consecutive(f)
            b.stitch_consecutive(f)
        assert isinstance(a.refs[-1], DeviceImage) and isinstance(b.refs[-1], np.ndarray)
        assert b.refs[-1].dtype == np.uint8 and b.refs[-1].ndim == seq[0].ndim and b.refs[-1].flags["C_CONTIGUOUS"]
        assert np.array_equal(np.asarray(a.refs[-1]), b.refs[-1]) and b.refs[-1].shape[1] > seq[0].shape[1]


def test_get_collections_float_descriptors(cuda, orc):
    """cv2.SIFT hands out float32 descriptors with integer values: get_collections takes the exact int8 path
    for them without scanning the arrays on the host (the device-side preparation checks), and falls back to
    the fp32 matcher when the values are not integers (RootSIFT-like)."""
    from image_stitcher_amd import ArbitraryStitcher
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors
    des, _ = overlapping_crops_descriptors(seed=31, n_scenes=2, imgs_per_scene=2, kp=600, step=150)

    class Fixed:
        def __init__(self, arrays):
            self.arrays, self.k = arrays, 0

        def detectAndCompute(self, image, mask=None):
            d = self.arrays[self.k]
            self.k += 1
            return [None] * len(d), d
    n = len(des)
    exp = {(i, j): orc.match_count(des[i], des[j], 0.65) for i in range(n) for j in range(n) if i != j}
    for arrays, exact in (([d.astype(np.float32) for d in des], True),
                          ([np.sqrt(d.astype(np.float32)) for d in des], False)):
        ar = ArbitraryStitcher(images=[np.zeros((8, 8, 3), np.uint8)] * n, with_tqdm=False, log_level=3, detector=Fixed(arrays))
        cols = ar.get_collections()
        pairs, counts = ar.good_counts
        if exact:
            for (i, j), (cij, cji) in zip(pairs, counts):
                assert cij == exp[(i, j)] and cji == exp[(j, i)]
            assert sorted(map(sorted, cols)) == [[0, 1], [2, 3]] or len(cols) == 2
        else:
            ref = {(i, j): orc.match_count(arrays[i], arrays[j], 0.65) for i in range(n) for j in range(n) if i != j}
            for (i, j), (cij, cji) in zip(pairs, counts):
                assert abs(int(cij) - ref[(i, j)]) <= 2 and abs(int(cji) - ref[(j, i)]) <= 2   # 1e-4 distance tolerance band


@pytest.mark.gpu
@pytest.mark.parametrize("detector_arg", ["single", "pair"])
def test_parallel_detect_gives_the_same_stream(cuda, world, tmp_path, detector_arg):
    """parallel_detect=True runs the two CPU detections of a stitch call concurrently (frame on a worker
    thread): mosaics, corners and transforms are those of the serial stitcher, and a detector that raises on
    the worker thread is recovered from exactly as on the calling thread (stitcher.py:189-191)."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    import threading
    frames = _frames(world, 6)

    class Recording(NumpyOrbLite):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.threads, self.fail_at, self.calls = set(), None, 0

        def detectAndCompute(self, image, mask=None):
            self.threads.add(threading.current_thread().name)
            self.calls += 1
            if self.calls == self.fail_at:
                raise RuntimeError("planted detector failure")
            return super().detectAndCompute(image, mask)

    outs = {}
    for par in (False, True):
        a, b = Recording(nfeatures=500), Recording(nfeatures=500)
        det = a
        if detector_arg == "pair":
            # the frame's ROI detection of the second stitch raises: the serial stitcher makes it call 3 of its
            # one detector (frame, reference, frame, ...), the parallel one call 2 of the worker's detector
            if par:
                det, b.fail_at = (a, b), 2
            else:
                a.fail_at = 3
        st = ConsecutiveStitcher(log_level=0, output_dir=str(tmp_path / f"o{par}"), consecutive_range=2, algorithm=1,
                                 detector=det, parallel_detect=par)
        rec = []
        for f in frames:
            tf = st.stitch_consecutive(f)
            rec.append((np.array(np.asarray(st.refs[-1])), st.corners, None if tf is None else np.array(tf.M)))
        outs[par] = rec
        names = a.threads | b.threads
        assert any(n.startswith("isx-detect") for n in names) == par
    for (m0, c0, t0), (m1, c1, t1) in zip(outs[False], outs[True]):
        assert np.array_equal(m0, m1) and c0 == c1 and (t0 is None) == (t1 is None)
        assert t0 is None or np.array_equal(t0, t1)


@pytest.mark.gpu
def test_parallel_detect_with_device_frames_reads_a_private_region(cuda, world, tmp_path):
    """Frames handed over as device tensors have no host copy, so the frame's detector region is downloaded into the
    ctx's pinned staging block; with parallel_detect the worker thread reads it WHILE this thread downloads the
    reference's region into the same block.  The frame's region must therefore be a private copy: a slow detector on the
    worker thread makes the overlap certain, and the stream must equal the serial one."""
    import time
    import torch
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 5)

    class Slow(NumpyOrbLite):
        def detectAndCompute(self, image, mask=None):
            time.sleep(0.05)  # the main thread's download of the reference region lands in this window
            return super().detectAndCompute(image, mask)

    outs = {}
    for par in (False, True):
        det = (NumpyOrbLite(nfeatures=500), Slow(nfeatures=500)) if par else NumpyOrbLite(nfeatures=500)
        st = ConsecutiveStitcher(log_level=0, output_dir=str(tmp_path / f"d{par}"), consecutive_range=2, algorithm=1,
                                 detector=det, parallel_detect=par)
        rec = []
        for f in frames:
            tf = st.stitch_consecutive(torch.from_numpy(f).cuda())
            last = st.refs[-1]  # (the first frame is kept as handed over: a device tensor)
            rec.append((last.cpu().numpy() if isinstance(last, torch.Tensor) else np.array(np.asarray(last)), st.corners,
                        None if tf is None else np.array(tf.M)))
        outs[par] = rec
    assert outs[True][-1][2] is not None
    for (m0, c0, t0), (m1, c1, t1) in zip(outs[False], outs[True]):
        assert np.array_equal(m0, m1) and c0 == c1 and (t0 is None) == (t1 is None)
        assert t0 is None or np.array_equal(t0, t1)


@pytest.mark.gpu
@pytest.mark.parametrize("blender", [0, 1])
def test_rectangle_beyond_the_frame_pipeline_limit_takes_the_canvas_path(cuda, orc, world, tmp_path, blender, monkeypatch):
    """A transform whose warped quad exceeds ISX_FRAME_MAX_Q (a degenerate homography: the quad's bound falls back to the
    whole canvas) must not raise from the frame pipeline: the stitch falls back to materialised canvases, as the
    reference computes it.  The limit is lowered for the test; the mosaic must equal the oracle's."""
    from image_stitcher_amd import Stitcher, TransformDetails, ops
    frames = _frames(world, 2)
    ih, iw = frames[1].shape[:2]
    rh, rw = frames[0].shape[:2]
    ox, oy = (rw + iw) // 2, (rh + ih) // 2
    M = np.array([[1.0, 0.0, iw + 40.0 - ox], [0.0, 1.0, ih + 12.0 - oy]])
    pts = np.zeros((5, 1, 2), np.float32)
    monkeypatch.setattr(ops, "FRAME_MAX_Q", 64)
    st = Stitcher(log_level=3, output_dir=str(tmp_path), type=0, blender=blender, ref_image_contrib=0.5, consecutive_range=1,
                  detector=object())
    got = st.stitch(frames[0], frames[1], tf=TransformDetails(M, pts, pts))
    o = oracle_stitch(frames[0], frames[1], None, Cfg(type=0, blender=blender, ref_image_contrib=0.5, consecutive_range=1),
                      tf=(M, pts, pts))
    assert np.array_equal(np.asarray(got[0]), o[0]) and got[1] == o[1]


@pytest.mark.gpu
def test_arbitrary_stitcher_from_filepaths(cuda, world, tmp_path):
    """ArbitraryStitcher(filepaths=...) (stitcher.py:239-246): .png/.jpg paths are kept, others dropped, images
    are loaded lazily (cv2.imread, or PIL where cv2 is absent) and give the collections of the same arrays
    passed as images=; an image without descriptors is dropped from filepaths too (stitcher.py:270-275)."""
    from image_stitcher_amd import ArbitraryStitcher
    from image_stitcher_amd.async_save import write_image
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 4, seed=3) + [np.zeros((240, 320, 3), np.uint8)]
    paths = []
    for k, f in enumerate(frames):
        paths.append(str(tmp_path / f"f{k}.png"))
        write_image(paths[-1], f)
    (tmp_path / "notes.txt").write_text("not an image")
    kw = dict(with_tqdm=False, log_level=3, algorithm=1, min_match_count=40)
    a = ArbitraryStitcher(filepaths=paths + [str(tmp_path / "notes.txt")], detector=NumpyOrbLite(nfeatures=500), **kw)
    b = ArbitraryStitcher(images=list(frames), detector=NumpyOrbLite(nfeatures=500), **kw)
    assert len(a.filepaths) == 5
    assert np.array_equal(np.asarray(a.images[2]), frames[2])
    ca, cb = a.get_collections(), b.get_collections()
    assert ca == cb and [len(c) for c in ca] == [4, 1]
    assert a.filepaths == paths[:4] and len(a.images) == 4
    # the trailing singleton is the reference's phantom (union-find sized before the drop, stitcher.py:283-284):
    # it indexes past the remaining images in the reference too, so only the real collection is stitched
    out = a.stitch_collections(ca[:1])
    ob = b.stitch_collections(cb[:1])
    assert len(out) == len(ob) and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(out, ob))


@pytest.mark.gpu
def test_readme_synchronised_flow_through_the_alias_package(cuda, world, tmp_path, monkeypatch):
    """The reference README's synchronised RGB + depth recipe, written against `image_stitcher.stitcher`
    (resolved by image-stitcher_amd/compat): the depth stitcher follows the RGB stitcher's transforms and its
    save_and_reset when a frame cannot be matched, so both end with the same number of mosaics, of the same extent."""
    import os, sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-stitcher_amd")
    monkeypatch.syspath_prepend(os.path.join(root, "compat"))
    from image_stitcher.stitcher import ConsecutiveStitcher, TransformDetails
    from image_stitcher_amd.detect import NumpyOrbLite
    from image_stitcher_amd.synthetic import procedural_world
    frames = _frames(world, 5)
    other = procedural_world(seed=99, h=240, w=320)
    seq = frames[:3] + [other] + frames[3:]
    # uint16 depth whose low byte (what the uint8 canvases keep, utils/proc.py:283-291) is never black
    depth = [(256 * (k + 1) + 20 + (np.arange(f.shape[1])[None, :] + np.arange(f.shape[0])[:, None]) % 200).astype(np.uint16)
             for k, f in enumerate(seq)]
    rgb = ConsecutiveStitcher(output_dir=str(tmp_path / "rgb"), matcher=1, algorithm=1, backup_interval=2, consecutive_range=2,
                              blender=1, log_level=3, min_match_count=40, detector=NumpyOrbLite(nfeatures=500))
    dep = ConsecutiveStitcher(output_dir=str(tmp_path / "depth"), backup_interval=2, blender=0, ref_image_contrib=0.5,
                              log_level=3, detector=NumpyOrbLite(nfeatures=500))
    shapes = []
    for k, (f, d) in enumerate(zip(seq, depth)):
        first = rgb.stitch_count == 0
        tf = rgb.stitch_consecutive(f)
        if not first and rgb.stitch_count > dep.stitch_count:
            dep.save_and_reset()
        tf_d = dep.stitch_consecutive(d, tf)
        assert tf is None or isinstance(tf, TransformDetails)
        assert tf_d is tf
        if rgb.refs[-1] is not None and dep.refs[-1] is not None:
            shapes.append((rgb.refs[-1].shape[:2], dep.refs[-1].shape[:2]))
    rgb.save_and_reset()
    dep.save_and_reset()
    assert rgb.stitch_count == dep.stitch_count == 2
    # same mosaics up to the border pixels the feather blender (RGB, blender=1) blends to black and the alpha
    # blender (depth, blender=0) keeps
    assert len(shapes) >= 4 and all(abs(a[0] - b[0]) <= 2 and abs(a[1] - b[1]) <= 2 for a, b in shapes)
    assert max(a[1] for a, _ in shapes) > 400
    assert sorted(os.listdir(tmp_path / "rgb")) == sorted(os.listdir(tmp_path / "depth"))


def _stream_with_given_tf(cuda, tmp_path, frames, Ms, blender, type_=0, contrib=0.5):
    """Frames blended onto the resident arena with given transforms (the reference's tf path, stitcher.py:213-214);
    every mosaic is compared with the oracle pipeline fed the ORACLE's previous mosaic, and the arena invariant
    (zero outside the mosaic's rectangle) is checked after every frame."""
    import torch
    from image_stitcher_amd import ConsecutiveStitcher, TransformDetails
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), type=type_, blender=blender, ref_image_contrib=contrib,
                             consecutive_range=1, detector=object())
    cfg = Cfg(type=type_, blender=blender, ref_image_contrib=contrib, consecutive_range=1)
    st.stitch_consecutive(frames[0])
    ref, corners = frames[0], None
    pts = np.zeros((5, 1, 2), np.float32)
    infos = []
    for f, M in zip(frames[1:], Ms):
        tf = TransformDetails(np.asarray(M, np.float64), pts, pts)
        assert st.stitch_consecutive(f, tf) is tf
        o = oracle_stitch(ref, f, None, cfg, tf=(tf.M, pts, pts), corners=corners)
        got = np.asarray(st.refs[-1])
        assert got.shape == o[0].shape and np.array_equal(got, o[0]), f"{(got != o[0]).sum()} bytes differ"
        assert st.corners == o[1]
        ref, corners = o[0], o[1]
        ar = st.refs[-1]._arena
        outside = ar.t.clone()
        outside[ar.y:ar.y + ar.h, ar.x:ar.x + ar.w] = 0
        assert int(outside.max()) == 0, "arena holds non-zero bytes outside the mosaic"
        infos.append((ar.t.shape, ar.x, ar.y))
    return st, infos


def _shift(dx, dy):
    return [[1.0, 0.0, float(dx)], [0.0, 1.0, float(dy)]]


def test_resident_arena_grows_and_stays_exact(cuda, orc, world, tmp_path):
    """12 frames marching right (then down) with sub-pixel rotations: the arena is reallocated only a few times
    (amortised growth), mosaics stay bit-identical to the oracle's, both blenders."""
    from image_stitcher_amd.synthetic import crop_path
    frames = crop_path(world, 12, (200, 260), seed=4, max_rot_deg=0.0)[0]
    for blender in (1, 0):
        Ms = []
        rw = 260
        for k in range(1, 12):
            th = np.deg2rad(0.7 * (-1) ** k)
            c, s = np.cos(th), np.sin(th)
            # the frame sits at ((rw+260)//2, (rh+200)//2) of the canvas, the mosaic at (260, 200): move the frame
            # so that it overlaps the mosaic's right end by ~60 %
            tx, ty = (260 + rw - 160) - (rw + 260) // 2 + 0.3, 5 * (k % 3) - 3.4
            Ms.append([[c, -s, tx], [s, c, ty]])
            rw += 100  # approximate; exactness does not depend on it
        st, infos = _stream_with_given_tf(cuda, tmp_path, frames, Ms, blender)
        shapes = [i[0] for i in infos]
        assert len(set(shapes)) <= 5 and st.frame_stats["arena_allocs"] <= 5, f"arena reallocated too often: {shapes}"
        assert st.frame_stats["frames"] == 11 and st.frame_stats["rescans"] == 0 and st.frame_stats["rows_slow"] == 0
        assert np.asarray(st.refs[-1]).shape[1] > 260 * 3


def test_gradient_rows_outside_roi_take_the_exact_slow_path(cuda, orc, tmp_path):
    """A frame with a channel-0-only region ((3,0,0): gray == 0, mask > 0 in the reference's per-channel mask,
    utils/proc.py:207-208) more than 50 px away from its gray content: those pixels lie in the ROI's rows but
    outside its columns, so the row extents must be taken inside the ROI (result block rows_ok == 0)."""
    ref = np.full((220, 300, 3), 90, np.uint8)
    f = np.zeros((200, 300, 3), np.uint8)
    f[30:170, 20:140] = (120, 80, 200)
    f[90:110, 250:290] = (3, 0, 0)      # far right of the content, same rows
    f[5:9, 40:60] = (4, 0, 1)            # gray == 0 but channel 0 != 0, above the content (inside the ROI's columns)
    st, _ = _stream_with_given_tf(cuda, tmp_path, [ref, f], [_shift(-40, 6)], blender=1)
    assert st.frame_stats["rows_slow"] == 1
    f[90:110, 250:290] = 0               # without the far region the fast path is taken, same oracle check
    st, _ = _stream_with_given_tf(cuda, tmp_path, [ref, f], [_shift(-40, 6)], blender=1)
    assert st.frame_stats["rows_slow"] == 0 and st.frame_stats["rescans"] == 0


def test_alpha_wrap_to_zero_shrinks_the_crop(cuda, orc, tmp_path):
    """utils/proc.py:79: outside the overlap the alpha blender ADDS with uint8 wrap-around, so 250 + 6 = 0: a pixel
    of the mosaic that defined its bounding box can turn black, and the crop (stitcher.py:225) then shrinks.  The
    frame pipeline reports it (result block `zeroed`) and the host rescans; gray == 0 leftovers outside the new crop
    ((3,0,0) pixels of the frame, `dirty`) are cleared from the arena."""
    ref = np.zeros((120, 200, 3), np.uint8)
    ref[10:110, 20:200] = 70
    ref[60, 0] = (250, 250, 250)         # the only non-black pixel of columns 0..19
    f = np.zeros((100, 160, 3), np.uint8)
    f[50, 30] = 6                        # gray 6 <= 10: never "overlap", always added -> lands on ref (0, 60)
    f[20:80, 60:120] = 6                 # lands on the 70-valued area
    f[95:97, 60:80] = (3, 0, 0)          # gray == 0 bytes that land below the mosaic's content (ref rows 115..116)
    # frame pasted at ((200+160)//2, (120+100)//2) = (180, 110); the mosaic at (160, 100): canvas (160, 160) is ref (0, 60)
    # move frame pixel (30, 50) onto it: 180 + 30 + tx = 160, 110 + 50 + ty = 160
    M1 = _shift(-50, 0)
    st, _ = _stream_with_given_tf(cuda, tmp_path, [ref, f], [M1], blender=0, contrib=0.5)
    assert np.asarray(st.refs[-1]).shape[1] == 180     # columns 0..19 were cropped away
    assert st.frame_stats["rescans"] == 1 and st.frame_stats["clears"] >= 1
    # and a second frame lands where the (3,0,0) bytes were: the arena must have been cleared there
    f2 = np.zeros((100, 160, 3), np.uint8)
    f2[10:90, 10:150] = 40
    _stream_with_given_tf(cuda, tmp_path, [ref, f, f2], [M1, _shift(30, -5)], blender=0, contrib=0.5)
    _stream_with_given_tf(cuda, tmp_path, [ref, f, f2], [M1, _shift(30, -5)], blender=1)


def test_public_stitch_does_not_modify_the_reference_mosaic(cuda, world, tmp_path):
    """Stitcher.stitch called by user code is functional as in the reference: a resident mosaic passed as `ref`
    keeps its pixels (only ConsecutiveStitcher / stitch_collections, which replace the mosaic by the result, blend
    in place)."""
    from image_stitcher_amd import Stitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    frames = _frames(world, 3)
    st = Stitcher(log_level=3, output_dir=str(tmp_path), algorithm=1, detector=NumpyOrbLite(nfeatures=500))
    m1, _, _ = st.stitch(frames[0], frames[1])
    before = np.array(np.asarray(m1))
    m2, _, _ = st.stitch(m1, frames[2])
    assert m2.shape[1] > m1.shape[1]
    m1.invalidate()
    assert np.array_equal(np.asarray(m1), before)


@pytest.mark.gpu
@pytest.mark.parametrize("type_", [0, 1])
def test_sparse_masks_and_narrow_frames_through_the_frame_kernels(cuda, orc, tmp_path, type_):
    """Stress of the frame kernels' bookkeeping: frames whose channel 0 is zero on about half of the pixels (holes in
    the per-channel mask, rows whose first and last mask pixel coincide, empty rows and columns, many 0 -> 1
    transitions per column), frames 2, 3 and 5 pixels wide (the border path's wide loads need 8 bytes per row), and
    a frame that is one solid row -- each against the oracle pipeline, both blenders."""
    rng = np.random.default_rng(11)
    ref = rng.integers(1, 255, (150, 210, 3), dtype=np.uint8)
    frames, Ms = [ref], []

    def add(f, deg, tx, ty):
        th = np.deg2rad(deg)
        c, s = np.cos(th), np.sin(th)
        frames.append(f)
        Ms.append([[c, -s, tx], [s, c, ty]] if type_ == 0 else [[c, -s, tx], [s, c, ty], [1e-5, -2e-5, 1.0]])

    f = rng.integers(0, 255, (120, 170, 3), dtype=np.uint8)
    f[rng.random((120, 170)) < 0.5, 0] = 0            # holes in channel 0 only
    f[rng.random((120, 170)) < 0.2] = 0               # black pixels
    f[40:43] = 0                                      # empty rows
    f[:, 60:62] = 0                                   # empty columns
    f[70, :] = 0
    f[70, 33] = (9, 200, 30)                          # a row with exactly one mask pixel (L == R)
    add(f, 3.0, -20.3, 4.6)
    for w in (2, 3, 5):
        add(rng.integers(1, 255, (90, w, 3), dtype=np.uint8), -2.0, 7.5 * w, -3.25)
    row = np.zeros((40, 120, 3), np.uint8)
    row[17] = (60, 70, 80)                            # one solid row, everything else black
    add(row, 0.0, 12.0, 9.0)
    add(rng.integers(0, 3, (80, 100, 3), dtype=np.uint8), 1.5, -30.2, 20.1)   # gray == 0 almost everywhere
    for blender in (1, 0):
        st, _ = _stream_with_given_tf(cuda, tmp_path, frames, Ms, blender, type_=type_)
        assert st.frame_stats["frames"] == len(Ms)


@pytest.mark.gpu
def test_frames_wider_than_16_chunks_and_taller_than_one_extent_batch(cuda, orc, tmp_path):
    """Geometry limits of the frame kernels' loops: a frame 4500 pixels wide (18 chunks of 256 pixels for the 16 waves of
    a row workgroup) and a frame 1700 rows tall (two batches of the column-extent hierarchy, 1536 rows each), with
    black rows / columns inside so that the extents are not trivial; both blenders, against the oracle pipeline."""
    rng = np.random.default_rng(5)
    ref = rng.integers(1, 255, (120, 160, 3), dtype=np.uint8)
    wide = rng.integers(0, 255, (40, 4500, 3), dtype=np.uint8)
    wide[:, 1000:1003] = 0
    wide[:, 4200:, 0] = 0                      # channel 0 empty on the right: R of every row far from the gray extent
    tall = rng.integers(0, 255, (1700, 36, 3), dtype=np.uint8)
    tall[1535:1540] = 0                        # black rows across the batch boundary
    tall[:7] = 0
    th = np.deg2rad(0.4)
    c, s = np.cos(th), np.sin(th)
    frames = [ref, wide, tall]
    Ms = [[[c, -s, 30.5], [s, c, -6.25]], [[c, s, -12.0], [-s, c, 40.75]]]
    for blender in (1, 0):
        st, _ = _stream_with_given_tf(cuda, tmp_path, frames, Ms, blender)
        assert st.frame_stats["frames"] == 2
