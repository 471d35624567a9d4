"""BASELINE.json `configs`, each as stated, through the drop-in classes on the GPU and checked against the
reference pipeline restated on the CPU oracle (tests/oracle_stitch.py).  cv2 is not installed here or on the
GPU box, so detection uses the deterministic numpy stand-ins of image_stitcher_amd.detect on both sides
(both paths see identical descriptors, as north_star requires); parity is against the oracle, which is itself
unpinned at the cv2 boundary.

  configs[0]  ConsecutiveStitcher SIFT (algorithm=0) + BF (matcher=0), blender=1, 20 x 640x480 crops
  configs[1]  ConsecutiveStitcher 1080p stream, ORB + Hamming, consecutive_range=2
  configs[4]  (scale) ArbitraryStitcher all-pairs at 4096 images
configs[2] at full size lives in test_gpu_golden.py::test_full_size_all_pairs_properties, configs[3] in
test_gpu_stitch.py::test_rgb_plus_depth_reusing_tf."""
import numpy as np
import pytest

from oracle_stitch import Cfg, OracleConsecutive, oracle_stitch

pytestmark = pytest.mark.gpu


def _recording(st):
    """Wrap st.stitch so that every call is recorded as (ref ndarray, img ndarray, corners_in, result)."""
    calls = []
    inner = st.stitch

    def stitch(ref, img, tf=None, corners=None):
        ref_h = np.array(np.asarray(ref))
        res = inner(ref, img, tf=tf, corners=corners)
        calls.append((ref_h, np.asarray(img), corners, res))
        return res
    st.stitch = stitch
    return calls


def _same_ring(st, oc):
    assert len(st.refs) == len(oc.refs)
    for a, b in zip([st.refs[i] for i in range(len(st.refs))], oc.refs):  # (iterating the ring itself is stateful)
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(np.asarray(a), b)
    assert st.corners == oc.corners
    assert (st.stitch_count, st.image_count, st.current_image_count) == (oc.stitch_count, oc.image_count, oc.current_image_count)


def test_config0_sift_bf_gradient_20_frames_640x480(cuda, orc, tmp_path):
    """configs[0]: ConsecutiveStitcher(algorithm=0, matcher=0, blender=1) on 20 overlapping 640x480 crops
    (15 % translation per frame, +-1 degree).  The detector returns 128-d float32 descriptors that are integer
    valued in [0,255] like cv2.SIFT's, so get_matches runs the exact int8-MFMA L2 path on 128-byte rows
    (reference stitcher.py:200-209 with cv2.BFMatcher()); every frame is compared with the oracle pipeline:
    identical good matches and point lists, M within 1e-3 px RMS (gate 0.5 px), mosaic and corners bit exact."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpySiftLite
    from image_stitcher_amd.synthetic import crop_path, procedural_world
    world = procedural_world(seed=42, h=1000, w=3000)
    frames = crop_path(world, 20, (480, 640), seed=1, max_rot_deg=1.0)[0]
    det = NumpySiftLite(nfeatures=700)
    kp, des = det.detectAndCompute(frames[0])
    assert des.dtype == np.float32 and des.shape[1] == 128 and np.array_equal(des, np.rint(des)) and des.max() <= 255
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), algorithm=0, matcher=0, blender=1, detector=det)
    assert st._matcher_kind(des) == "int"
    calls = _recording(st)
    oc = OracleConsecutive(det, Cfg(type=0, blender=1, ref_image_contrib=0.2, consecutive_range=1))
    kinds = []
    for f in frames:
        del calls[:]
        st.stitch_consecutive(f)
        kinds.append(oc.step(f, calls))
        _same_ring(st, oc)
    assert kinds == ["first"] + ["stitched"] * 19
    assert st.image_count == 19 and st.stitch_count == 0
    assert np.asarray(st.refs[-1]).shape[1] > 640 * 3  # 19 x 15 % of 640 px


def test_config1_hamming_range2_stream_with_fallback(cuda, orc, tmp_path):
    """configs[1] (state machine, 320x240): norm='hamming' (the opt-in XOR/popcount matcher, parity target
    cv2.BFMatcher(NORM_HAMMING)), consecutive_range=2.  The stream is built so that every branch of
    stitcher.py:416-439 runs: two stitches, an unrelated frame (reset, frame dropped), an unrelated first frame of
    the new mosaic, then a frame of the OLD chain, which fails against refs[-1] and is stitched onto refs[-2]
    (the fallback of :426-432, refs.pop()), then a plain stitch onto the re-adopted mosaic."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import NumpyOrbLite
    from image_stitcher_amd.synthetic import crop_path, procedural_world
    world = procedural_world(seed=42, h=800, w=2200)
    a = crop_path(world, 5, (240, 320), seed=1, max_rot_deg=1.0)[0]
    x = procedural_world(seed=99, h=240, w=320)
    y = procedural_world(seed=123, h=240, w=320)
    seq = [a[0], a[1], a[2], x, y, a[3], a[4]]
    det = NumpyOrbLite(nfeatures=500)
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), algorithm=1, norm="hamming", consecutive_range=2,
                             min_match_count=40, detector=det)
    calls = _recording(st)
    oc = OracleConsecutive(det, Cfg(type=0, blender=1, ref_image_contrib=0.2, consecutive_range=2, norm="hamming",
                                    min_match_count=40))
    kinds = []
    for f in seq:
        del calls[:]
        st.stitch_consecutive(f)
        kinds.append(oc.step(f, calls))
        _same_ring(st, oc)
    assert kinds == ["first", "stitched", "stitched", "reset", "first", "fallback", "stitched"]
    assert len(st.refs) == 1 and st.stitch_count == 1 and (tmp_path / "blend_0.png").exists()


@pytest.mark.parametrize("type_,blender", [(0, 1), (1, 0)])
def test_config1_full_size_1080p_frames(cuda, orc, tmp_path, type_, blender):
    """configs[1] at full size: 1920x1080 BGR frames, consecutive_range=2.  Three frames are stitched with
    the path's ground-truth transforms handed in as tf (the reference's tf-given path, stitcher.py:213-214), so
    warp + blend + both bounding boxes + crop run on the full-size canvas (5760 x 3240 for the first pair) and
    are compared bit for bit with the oracle; the second stitch consumes the corners and the grown mosaic of the
    first."""
    from image_stitcher_amd import ConsecutiveStitcher, TransformDetails
    from image_stitcher_amd.synthetic import crop_path, procedural_world
    world = procedural_world(seed=42, h=1500, w=3700)
    frames, poses = crop_path(world, 3, (1080, 1920), seed=1, max_rot_deg=1.0)
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), algorithm=1, norm="hamming", consecutive_range=2,
                             type=type_, blender=blender, ref_image_contrib=0.5, detector=object())
    cfg = Cfg(type=type_, blender=blender, ref_image_contrib=0.5, consecutive_range=2)
    st.stitch_consecutive(frames[0])
    rng = np.random.default_rng(0)
    for k in (1, 2):
        ref = np.array(np.asarray(st.refs[-1]))
        corners = st.corners
        # a plausible transform into THIS call's padded canvas: the frame lands over the right part of the mosaic,
        # rotated by a degree and (type 1) with a slight perspective
        ih, iw = frames[k].shape[:2]
        rh, rw = ref.shape[:2]
        ox, oy = (rw + iw) // 2, (rh + ih) // 2
        th = np.deg2rad(rng.uniform(-1, 1))
        c, s = np.cos(th), np.sin(th)
        tx = iw + rw - int(0.85 * iw) - ox + rng.uniform(-3, 3)
        ty = ih - oy + rng.uniform(-20, 20)
        A = np.array([[c, -s, tx + ox - (c * ox - s * oy)], [s, c, ty + oy - (s * ox + c * oy)]], np.float64)
        M = A if type_ == 0 else np.vstack([A, [1e-6, -2e-6, 1.0]])
        pts = np.zeros((5, 1, 2), np.float32)
        tf = TransformDetails(M, pts, pts)
        assert st.stitch_consecutive(frames[k], tf) is tf
        o = oracle_stitch(ref, frames[k], None, cfg, tf=(M, pts, pts), corners=corners)
        got = np.asarray(st.refs[-1])
        assert got.shape == o[0].shape and np.array_equal(got, o[0]), f"{(got != o[0]).sum()} bytes differ"
        assert st.corners == o[1]
    assert got.shape[1] > 1920 * 1.2


def test_config4_scale_all_pairs_4096_images(cuda, orc):
    """configs[4] at its stated size: one pass of the all-pairs matcher over 4096 images x 4096 keypoints x 128-d
    (2 GiB of descriptors generated on the device, 8,386,560 unordered pairs, 16.8 M directed items: five launches,
    HIP forms gridDim*blockDim in 32 bits; about 23 s of kernel on the directed kernel and 13 s more on the pair kernel).  The oracle would need days for all of it, so
    (1) 36 directed items of this pass -- union edges, items within +-2 of min_match_count, random ones -- are
    recounted with the oracle (knnMatch float32 + ratio loop, reference stitcher.py:285-296) on the downloaded
    images, and (2) the whole count matrix is checked through the size-independent properties of
    test_full_size_all_pairs_properties: the planted scene structure (64 scenes x 64 images), counts below
    min_match between scenes, the components, and relabelling invariance on a sampled sub-matrix.
    The FLANN half of configs[4] needs cv2: unpinned."""
    import torch
    from image_stitcher_amd import ops, pairs as prs
    from image_stitcher_amd.synthetic import overlapping_crops_descriptors_device
    from test_gpu_golden import assert_counts_match_oracle, oracle_sample_of_directed_items
    n, kp, d = 4096, 4096, 128
    des, scene_of = overlapping_crops_descriptors_device(cuda, seed=77, n_scenes=64, imgs_per_scene=64, kp=kp, step=448, d=d)
    starts = np.arange(n, dtype=np.int64) * kp
    counts = np.full((n,), kp, np.int64)
    all_pairs = prs.unordered_pairs(n)
    assert len(all_pairs) == 8386560
    items = prs.directed_items(all_pairs, starts, counts)
    P = ops.prepare_descriptor_tensor(des.reshape(-1, d), starts, counts)
    c0 = ops.knn2_l2_items(P, items, 0.65).cpu().numpy().reshape(-1, 2)
    del items
    # the same pass on the kernel the product runs since round 3 (both directions of a pair from one distance matrix, rows
    # sorted by norm): all 16.8 M directions equal the directed kernel's, which the oracle sample below pins
    cp = ops.knn2_l2_pairs(ops.sort_rows_by_norm(P), prs.pair_items(all_pairs, starts, counts), 0.65).cpu().numpy()
    assert np.array_equal(cp, c0)
    del cp
    sample, _ = oracle_sample_of_directed_items(c0, all_pairs, 36, 5, np.random.default_rng(5))
    assert len(sample) >= 30
    assert_counts_match_oracle(orc, lambda i: des[i].cpu().numpy(), c0, all_pairs, sample, 0.65, nthreads=16)
    comps, _ = prs.components_from_counts(n, all_pairs, c0[:, 0], c0[:, 1], 5)
    truth = sorted(sorted(np.nonzero(scene_of == s)[0].tolist()) for s in range(64))
    assert sorted(sorted(c) for c in comps) == truth
    cross = scene_of[all_pairs[:, 0]] != scene_of[all_pairs[:, 1]]
    assert c0[cross].max() < 5 and c0[~cross].max() > 1000
    # relabelling invariance on 96 sampled images (4560 pairs): shuffled image order and row order
    rng = np.random.default_rng(3)
    sub = np.sort(rng.choice(n, 96, replace=False))
    perm = rng.permutation(96)
    g = torch.Generator(device=cuda)
    g.manual_seed(9)
    des2 = torch.stack([des[int(sub[perm[k]])][torch.randperm(kp, generator=g, device=cuda)] for k in range(96)])
    sp = prs.unordered_pairs(96)
    P2 = ops.prepare_descriptor_tensor(des2.reshape(-1, d), starts[:96], counts[:96])
    c1 = ops.knn2_l2_items(P2, prs.directed_items(sp, starts[:96], counts[:96]), 0.65).cpu().numpy().reshape(-1, 2)

    def at(i, j):  # all_pairs enumerates i<j row by row: pair (i, j) sits at i*n - i*(i+1)/2 + (j - i - 1)
        if i < j:
            return c0[i * n - i * (i + 1) // 2 + (j - i - 1), 0]
        return c0[j * n - j * (j + 1) // 2 + (i - j - 1), 1]
    img = sub[perm]
    assert tuple(all_pairs[5 * n - 15 + (9 - 5 - 1)]) == (5, 9)
    for k, (a, b) in enumerate(sp):
        assert c1[k, 0] == at(int(img[a]), int(img[b])) and c1[k, 1] == at(int(img[b]), int(img[a]))
