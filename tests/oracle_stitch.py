"""The reference's per-pair pipeline `Stitcher.stitch` (reference stitcher.py:176-229) restated on top of
the CPU oracle -- TEST INFRASTRUCTURE.  Used to check the GPU host layer stage by stage and end to end."""
import numpy as np

from oracle import orc


class Cfg:
    def __init__(self, type=0, blender=1, ratio=0.65, min_match_count=5, ref_image_contrib=0.5,
                 consecutive_range=None, warp_arith=0, norm=None, volatility_threshold=4):
        self.type, self.blender, self.ratio = type, blender, ratio
        self.min_match_count, self.ref_image_contrib = min_match_count, ref_image_contrib
        self.consecutive_range = consecutive_range
        self.warp_arith = warp_arith  # 0: OpenCV 4.11 float32 linear kernels, 1: classic fixed point
        self.norm = orc.NORM_L2 if norm in (None, "l2") else orc.NORM_HAMMING  # "hamming": the opt-in matcher
        self.volatility_threshold = volatility_threshold


def _kp_xy(kps):
    return np.array([kp.pt for kp in kps], np.float32).reshape(-1, 2)


def _detect_roi(detector, image, corners):
    (x0, y0), (x1, y1) = corners
    roi = image[y0:y1, x0:x1]
    kps, des = detector.detectAndCompute(np.ascontiguousarray(roi), None)
    for kp in kps:
        kp.pt = (kp.pt[0] + x0, kp.pt[1] + y0)
    return kps, des


def oracle_stitch(arg_ref, arg_img, detector, cfg, tf=None, corners=None):
    """Returns (mosaic, corners, (M, src_pts, dst_pts)) | None | False exactly like the reference."""
    current_image, reference_image = orc.get_padded_images(arg_img, arg_ref)          # :178
    current_corners = orc.bbox_nonzero(current_image)                                 # :179
    if current_corners is False:
        corners = None
    if tf is None:
        thisdes = None
        if corners is not None and cfg.consecutive_range is not None:                 # :185
            try:
                thiskp, thisdes = _detect_roi(detector, current_image, current_corners)
                refkp, refdes = _detect_roi(detector, reference_image, corners)
            except Exception:
                corners = None
        if corners is None or cfg.consecutive_range is None or thisdes is None:       # :192
            thiskp, thisdes = detector.detectAndCompute(current_image, None)
            refkp, refdes = detector.detectAndCompute(reference_image, None)
        if refdes is None:
            return False                                                              # :196-198
        if thisdes is None or len(thisdes) == 0 or len(refdes) == 0:
            good_q = good_t = np.zeros(0, np.int32)
        else:
            idx, dist = orc.knn2(thisdes, refdes, norm=cfg.norm)                                     # :200
            good_q, good_t = orc.ratio_filter(idx, dist, cfg.ratio)                   # :202
        if len(good_q) < cfg.min_match_count:
            return None                                                               # :210-212
        src_pts = _kp_xy(thiskp)[good_q].reshape(-1, 1, 2)                            # :207
        dst_pts = _kp_xy(refkp)[good_t].reshape(-1, 1, 2)
        if cfg.type == 0:                                                             # :208
            M, _ = orc.estimate_affine_partial(src_pts, dst_pts)
        else:
            M, _ = orc.find_homography(src_pts, dst_pts)
        tf = (M, src_pts, dst_pts)
    else:
        M, src_pts, dst_pts = tf
    H, W = reference_image.shape[:2]
    warped = orc.warp(current_image, M, (W, H), cfg.type == 1, arith=cfg.warp_arith)  # :216
    if cfg.blender == 0:                                                              # :217
        reference_image = orc.alpha_blend(warped, reference_image, cfg.ref_image_contrib)
    else:
        reference_image = orc.gradient_blend(warped, reference_image)
    if cfg.consecutive_range is not None:                                             # :219-223
        corners = orc.bbox_nonzero(warped)
        if corners is False:
            corners = None
    cc = orc.bbox_nonzero(reference_image)                                            # :225
    if cc is False:
        return None
    (x0, y0), (x1, y1) = cc
    return reference_image[y0:y1, x0:x1], corners, tf


class OracleConsecutive:
    """ConsecutiveStitcher.stitch_consecutive (reference stitcher.py:377-439) restated over oracle_stitch: the
    refs ring of at most consecutive_range mosaics, the retry against older mosaics while
    current_image_count < consecutive_volatility_threshold, save_and_reset, the dropped frame.

    `gpu_calls` is the list of Stitcher.stitch calls the product made for the same frame, in order:
    (ref ndarray, img ndarray, corners_in, result).  Every stitch this state machine makes is checked against
    the next recorded call: same inputs bit for bit, same outcome kind, identical point lists, M within
    `m_tol` px RMS; the mosaic is then produced by the oracle FROM THE PRODUCT'S TRANSFORM (so that one LSB of
    M cannot move a pixel) and must equal the product's mosaic bit for bit."""

    def __init__(self, detector, cfg, m_tol=1e-3):
        self.det, self.cfg, self.m_tol = detector, cfg, m_tol
        self.refs = [None]
        self.corners = None
        self.stitch_count = self.image_count = self.current_image_count = 0
        self.saved = []

    def _append(self, item):                                                    # structs.py:82-85
        self.refs.append(item)
        if len(self.refs) > self.cfg.consecutive_range:
            self.refs.pop(0)

    def _checked_stitch(self, ref, img, calls):
        g_ref, g_img, g_corners, g_res = calls.pop(0)
        assert np.array_equal(g_ref, ref), "the product stitched against a different reference image"
        assert np.array_equal(g_img, img) and g_corners == self.corners
        o = oracle_stitch(ref, img, self.det, self.cfg, corners=self.corners)
        assert (o is None) == (g_res is None) and (o is False) == (g_res is False), (type(o), type(g_res))
        if o is None or o is False:
            return o
        mosaic, corners, tf = g_res
        oM, osrc, odst = o[2]
        assert np.array_equal(tf.src_pts, osrc) and np.array_equal(tf.dst_pts, odst)
        p = osrc.reshape(-1, 2).astype(np.float64)

        def ap(M):
            M = np.vstack([M, [0, 0, 1]]) if M.shape[0] == 2 else M
            q = np.c_[p, np.ones(len(p))] @ M.T
            return q[:, :2] / q[:, 2:3]
        assert np.sqrt(((ap(tf.M) - ap(oM)) ** 2).sum(1).mean()) < self.m_tol
        o2 = oracle_stitch(ref, img, self.det, self.cfg, tf=(tf.M, tf.src_pts, tf.dst_pts), corners=self.corners)
        got = np.asarray(mosaic)
        assert got.shape == o2[0].shape and np.array_equal(got, o2[0]), "mosaic differs from the oracle's"
        assert corners == o2[1]
        return o2

    def _finalise(self, result):                                                # :407-414
        self.refs[-1], self.corners, _ = result
        self.image_count += 1
        self.current_image_count += 1

    def step(self, image, gpu_calls):
        """Returns 'first' | 'stitched' | 'fallback' | 'reset'."""
        calls = list(gpu_calls)
        if self.refs[-1] is None:                                               # :421-422
            self.refs[-1] = image
            assert not calls
            return "first"
        result = self._checked_stitch(self.refs[-1], image, calls)               # :424
        if result is None or result is False:
            if self.current_image_count < self.cfg.volatility_threshold:        # :426
                for exref in reversed(self.refs[:-1]):                          # :427
                    result = self._checked_stitch(exref, image, calls)
                    if not (result is None or result is False):
                        self.refs.pop()                                         # :430
                        self._finalise(result)
                        assert not calls
                        return "fallback"
            self.saved.append(self.refs[-1])                                    # save_and_reset :400-405
            self._append(None)
            self.stitch_count += 1
            self.corners = None
            self.current_image_count = 0
            assert not calls
            return "reset"
        self._finalise(result)
        assert not calls
        return "stitched"
