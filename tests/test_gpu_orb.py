"""the opt-in on-GPU ORB-style detector (detect.HipOrb -> isx_orb_candidates / isx_orb_describe) against its
numpy definition tests/np_orb.py, bit for bit: gray, blur, key map, keypoints in selection order, orientation bins and
all 256 descriptor bits; and a whole ConsecutiveStitcher stream whose detector runs on the GPU against the oracle
pipeline whose detector is the numpy definition.  (Parity with cv2.ORB itself is unpinned by construction: see np_orb.)"""
import numpy as np
import pytest

import np_orb
from oracle_stitch import Cfg, oracle_stitch

pytestmark = pytest.mark.gpu


def _world(h=800, w=2200):
    from image_stitcher_amd.synthetic import procedural_world
    return procedural_world(seed=42, h=h, w=w, speckle=4)


def _ws_views(det, h, w):
    """The workspace arrays of the last HipOrb call (layout of csrc/isx_orb.hip: g, blurred, hsum, key map)."""
    al = lambda v: (v + 255) & ~255
    n = h * w
    ws = det._ws.cpu().numpy()
    g = ws[:n].reshape(h, w)
    bl = ws[al(n):al(n) + n].reshape(h, w)
    key = ws[2 * al(n) + al(2 * n):2 * al(n) + al(2 * n) + 4 * n].view(np.uint32).reshape(h, w)
    return g, bl, key


def _check(det, img, **kw):
    kps, des = det.detectAndCompute(img)
    exp_xy, exp_resp, exp_ang, exp_des = np_orb.detect(img, nfeatures=det.nfeatures, fast_threshold=det.fast_threshold,
                                                      clean=det.clean, **kw)
    h, w = img.shape[:2]
    if h >= 33 and w >= 33:
        g, bl, key = _ws_views(det, h, w)
        eg = np_orb.gray(img)
        assert np.array_equal(g, eg), "gray differs"
        assert np.array_equal(bl, np_orb.blur(eg)), "blur differs"
        ek = np_orb.keymap(eg, det.fast_threshold, det.clean)
        assert np.array_equal(key, ek), f"key map differs at {np.argwhere(key != ek)[:5]}"
    if len(exp_xy) == 0:
        assert des is None and len(kps) == 0
        return 0
    assert len(kps) == len(exp_xy)
    assert np.array_equal(kps.xy, exp_xy.astype(np.float32)), "keypoints or their order differ"
    assert np.array_equal(kps.response.view(np.uint32), exp_resp.view(np.uint32))
    assert np.array_equal(kps.angle, exp_ang.astype(np.float32) * 11.25), "orientation bins differ"
    assert des.dtype == np.uint8 and np.array_equal(des, exp_des), f"{(des != exp_des).any(1).sum()} descriptors differ"
    assert [k.pt for k in kps][:3] == [tuple(map(float, p)) for p in exp_xy[:3]]
    return len(kps)


def test_detector_matches_its_definition(cuda):
    import torch
    from image_stitcher_amd.detect import HipOrb
    from image_stitcher_amd.synthetic import crop_path
    world = _world()
    frames = crop_path(world, 3, (240, 320), seed=1)[0]
    det = HipOrb(nfeatures=500)
    n = [_check(det, f) for f in frames]
    assert min(n) > 100
    # a truncating nfeatures (selection order and ties), another threshold, no clean margin
    assert _check(HipOrb(nfeatures=37), frames[0]) == 37
    assert _check(HipOrb(nfeatures=300, fast_threshold=9, clean=0), frames[1]) > 100
    # one channel; odd sizes that end inside a 64-pixel bit word / a 32-pixel tile; pitched device view
    assert _check(det, np_orb.gray(frames[2])) > 100
    assert _check(det, np.ascontiguousarray(world[100:389, 200:611])) > 100
    view = torch.from_numpy(world).to(cuda)[37:300, 501:900]
    kps, des = det.detectAndCompute(view)
    exy, _, _, edes = np_orb.detect(world[37:300, 501:900])
    assert np.array_equal(kps.xy, exy.astype(np.float32)) and np.array_equal(des, edes)
    # too small for a single patch: nothing, as cv2 returns ([], None)
    assert det.detectAndCompute(np.full((32, 200, 3), 90, np.uint8)) == ([], None)
    assert _check(det, np.full((33, 33, 3), 90, np.uint8)) == 0


def test_black_regions_and_plateaus(cuda):
    """A frame pasted on a black canvas (what the stitcher's ROI of a mosaic looks like): corners within `clean` pixels
    of a black pixel are dropped; equal responses (a periodic pattern) resolve to the lower index, on both sides."""
    from image_stitcher_amd.detect import HipOrb
    from image_stitcher_amd.synthetic import crop_path
    world = _world()
    f = crop_path(world, 1, (200, 260), seed=2)[0][0]
    canvas = np.zeros((330, 420, 3), np.uint8)
    canvas[60:260, 90:350] = f
    canvas[100:104, 200:230] = 0          # a black scratch inside the content
    det = HipOrb(nfeatures=400)
    assert _check(det, canvas) > 30
    kps, _ = det.detectAndCompute(canvas)
    assert kps.xy[:, 0].min() >= 90 + 16 and kps.xy[:, 0].max() < 350 - 16 and kps.xy[:, 1].min() >= 60 + 16
    tile = np.full((12, 12), 60, np.uint8)
    tile[4:7, 4:7] = 200
    per = np.tile(tile, (20, 26))          # identical blobs: identical responses everywhere
    assert _check(HipOrb(nfeatures=50), per) == 50
    assert _check(HipOrb(nfeatures=50, device_select=False), per) == 50      # the host-side selection, same answer


def test_more_survivors_than_the_device_sort_takes(cuda):
    """> 65536 NMS survivors: isx_orb_select_describe reports -1 and the keypoints are picked on the host from the tile
    slots -- same definition, same result."""
    from image_stitcher_amd.detect import HipOrb
    rng = np.random.default_rng(5)
    tile = np.full((8, 8), 60, np.uint8)
    tile[2:5, 2:5] = 200
    per = np.tile(tile, (268, 268)).astype(np.int16)
    per += rng.integers(0, 3, per.shape, dtype=np.int16)   # break the ties so that the order is by response
    per = per.astype(np.uint8)
    det = HipOrb(nfeatures=300)
    assert _check(det, per) == 300
    nt = ((per.shape[0] + 31) // 32) * ((per.shape[1] + 31) // 32)
    assert int(det._tiles_host[:nt].sum()) > 65536


def test_full_hd_frame(cuda):
    from image_stitcher_amd.detect import HipOrb
    world = _world(1300, 2300)
    assert _check(HipOrb(nfeatures=1500), np.ascontiguousarray(world[100:1180, 200:2120])) == 1500


class NpOrbDetector:
    """The numpy definition behind the detectAndCompute interface (for the oracle pipeline)."""

    def __init__(self, nfeatures):
        self.nfeatures = nfeatures

    def detectAndCompute(self, image, mask=None):
        from image_stitcher_amd.detect import KeyPoint
        xy, resp, ang, des = np_orb.detect(image, nfeatures=self.nfeatures)
        if len(xy) == 0:
            return [], None
        return [KeyPoint(x, y, r) for (x, y), r in zip(xy, resp)], des


@pytest.mark.parametrize("blender", [1, 0])
def test_stream_with_gpu_detection_matches_the_oracle_pipeline(cuda, orc, tmp_path, blender):
    """ConsecutiveStitcher(detector=HipOrb()): no CPU detector in the loop, the mosaic region goes to the detector
    without leaving the GPU.  Frame by frame against the oracle pipeline fed the numpy definition of the detector."""
    from image_stitcher_amd import ConsecutiveStitcher
    from image_stitcher_amd.detect import HipOrb
    from image_stitcher_amd.synthetic import crop_path
    frames = crop_path(_world(), 6, (240, 320), seed=1)[0]
    st = ConsecutiveStitcher(log_level=3, output_dir=str(tmp_path), type=0, blender=blender, consecutive_range=1, algorithm=1,
                             min_match_count=30, detector=HipOrb(nfeatures=500))
    ref_det = NpOrbDetector(500)
    cfg = Cfg(type=0, blender=blender, ref_image_contrib=0.2, consecutive_range=1, min_match_count=30)
    st.stitch_consecutive(frames[0])
    for k, f in enumerate(frames[1:]):
        ref_before, corners_before = np.array(np.asarray(st.refs[-1])), st.corners
        tf = st.stitch_consecutive(f)
        assert tf is not None, f"frame {k + 1} failed to stitch"
        o = oracle_stitch(ref_before, f, ref_det, cfg, corners=corners_before)
        assert o not in (None, False)
        assert np.array_equal(tf.src_pts, o[2][1]) and np.array_equal(tf.dst_pts, o[2][2]), "good matches differ"
        o2 = oracle_stitch(ref_before, f, ref_det, cfg, tf=(tf.M, tf.src_pts, tf.dst_pts), corners=corners_before)
        assert np.array_equal(np.asarray(st.refs[-1]), o2[0]) and st.corners == o2[1]
    assert np.asarray(st.refs[-1]).shape[1] > 320 * 1.5
