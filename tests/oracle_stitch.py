"""The reference's per-pair pipeline `Stitcher.stitch` (reference stitcher.py:176-229) restated on top of
the CPU oracle -- TEST INFRASTRUCTURE.  Used to check the GPU host layer stage by stage and end to end."""
import numpy as np

from oracle import orc


class Cfg:
    def __init__(self, type=0, blender=1, ratio=0.65, min_match_count=5, ref_image_contrib=0.5,
                 consecutive_range=None, warp_arith=0):
        self.type, self.blender, self.ratio = type, blender, ratio
        self.min_match_count, self.ref_image_contrib = min_match_count, ref_image_contrib
        self.consecutive_range = consecutive_range
        self.warp_arith = warp_arith  # 0: OpenCV 4.11 float32 linear kernels, 1: classic fixed point


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
            idx, dist = orc.knn2(thisdes, refdes)                                     # :200
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
