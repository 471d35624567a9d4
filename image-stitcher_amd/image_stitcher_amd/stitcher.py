"""Drop-in host layer: the reference's `Stitcher` / `ArbitraryStitcher` / `ConsecutiveStitcher`
(reference image_stitcher/stitcher.py) with the same constructor keywords, methods, return contracts and
state machine, driving the HIP kernels of libisx_hip.so through image_stitcher_amd.ops.

What runs where
  CPU  feature detection (cv2 or a `detector=` object), cv::RNG subset sampling, union-find, file I/O
  GPU  kNN-2 + Lowe ratio (+ stable compaction), RANSAC hypotheses/scoring/refinement, bbox, warp,
       alpha / gradient blend, crop; mosaics stay resident in HBM as DeviceImage

Deliberate differences from the reference:
  * log_level=0 switches logging off instead of raising AttributeError (stitcher.py:69-70);
  * dropping empty images does not touch self.filepaths when `images=` was given (stitcher.py:275);
  * matcher=1 (FLANN, approximate and randomised in cv2) is served by the exact brute-force kernel;
  * mosaics are returned as DeviceImage (array-like; np.asarray() downloads) unless resident=False;
  * keyword-only extensions: detector=, device=, resident=, norm= ('hamming' opts in to the XOR/popcount
    matcher; the default 'l2' is what the reference's cv2.BFMatcher() computes for every algorithm),
    async_save=, warp_arith=, parallel_detect= (see __init__).
"""
import copy
import logging
import os
import math
import time

import numpy as np
import torch

from . import ops
from .async_save import AsyncImageWriter, write_image
from .detect import make_detector
from .device_image import DepthImage, DeviceImage, MosaicArena
from .strips import StripMosaic
from .pairs import components_from_counts, directed_items, gather_counts, pair_items, shard_pairs, unordered_pairs
from .utils.logger import LogWrapper
from .utils.proc import (crop_image, get_corners_from_image, get_padded_images, get_roi_from_corners,  # noqa: F401
                         load_image, resize_image, seamless_gradient_merge, seamless_merge,
                         transform_keypoints_from_roi)
from .utils.structs import ConstantLengthList, DynamicConnectivity, LazyList  # noqa: F401
from .utils.utils import no_tqdm, timer, tqdm


class TransformDetails:
    """reference stitcher.py:23-33"""

    def __init__(self, M, src_pts, dst_pts):
        self.M = M
        self.src_pts = src_pts
        self.dst_pts = dst_pts

    def get(self):
        return self.M, self.src_pts, self.dst_pts

    def __str__(self):
        return f"M: {self.M}, src_pts: {self.src_pts}, dst_pts: {self.dst_pts}"


class DMatch:
    """cv2.DMatch look-alike, materialised only when user code iterates over matches."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, q, t, d):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = int(q), int(t), 0, float(d)


class KnnMatches:
    """Result of get_matches: device-resident SoA (idx (nq,2) int32, dist (nq,2) float32).  Iterating
    yields the reference's list-of-lists of DMatch (rows have min(2, nt) entries)."""

    def __init__(self, idx, dist, count):
        self.idx, self.dist, self.count = idx, dist, count

    def __len__(self):
        return self.idx.shape[0]

    def __iter__(self):
        idx, dist = self.idx.cpu().numpy(), self.dist.cpu().numpy()
        for q in range(len(idx)):
            yield [DMatch(q, idx[q, k], dist[q, k]) for k in range(2) if idx[q, k] >= 0]


class GoodMatches:
    """Result of filter_matches: query/train indices (host int32 arrays) in query order."""

    def __init__(self, good_q, good_t, dist=None):
        self.good_q, self.good_t, self.dist = good_q, good_t, dist

    def __len__(self):
        return len(self.good_q)

    def __iter__(self):
        for i in range(len(self.good_q)):
            d = float(self.dist[i]) if self.dist is not None else float("nan")
            yield [DMatch(self.good_q[i], self.good_t[i], d)]


def _kp_array(kps):
    """keypoints (cv2.KeyPoint-like objects or an (n,2) array) -> (n,2) float32"""
    if isinstance(kps, np.ndarray):
        return kps.astype(np.float32, copy=False).reshape(-1, 2)
    xy = getattr(kps, "xy", None)  # detect.KeyPointList: coordinates kept as an array beside the objects
    if xy is not None and len(xy) == len(kps):
        return np.asarray(xy, np.float32).reshape(-1, 2)
    if len(kps) and type(kps[0]).__module__.startswith("cv2"):
        import cv2
        return np.asarray(cv2.KeyPoint_convert(kps), np.float32).reshape(-1, 2)
    return np.array([kp.pt for kp in kps], dtype=np.float32).reshape(-1, 2)


def _kp_shifted(kps, origin):
    """transform_keypoints_from_roi (utils/proc.py:315-322) on coordinates only: (n,2) float32 of kp.pt + origin,
    summed in float64 as the reference's Python floats are and rounded once, as np.float32([...]) does in get_pts."""
    xy = _kp_array(kps).astype(np.float64)
    xy += np.asarray(origin, np.float64)
    return xy.astype(np.float32)


class Stitcher:
    def __init__(self,
                 log_level=2,
                 output_dir="stitches/",
                 min_match_count=5,
                 lowes_ratio_threshold=0.65,
                 algorithm=0,
                 matcher=0,
                 type=0,
                 resize=None,
                 consecutive_range=None,
                 blend_processor=lambda x: x,
                 blender=1,
                 ref_image_contrib=0.5,  # Used in alpha blending
                 *, detector=None, device=None, resident=True, norm="l2", async_save=False, warp_arith="f32",
                 parallel_detect=False, depth16=False):
        options = {
            "log_level": [logging.DEBUG, logging.INFO, logging.WARNING],
            "algorithm": ["sift", "orb", "akaze"],
            "matcher": ["bf", "flann"],
            "blender": ["alpha", "gradient"],
            "type": ["affine", "perspective"],
        }
        _logger = logging.getLogger("root")
        logging.basicConfig(format="[%(filename)s:%(lineno)s - %(funcName)s] %(message)s")
        self.logger = LogWrapper(_logger)
        if log_level != 0:
            self.log_level = options["log_level"][log_level - 1]
            self.logger.logger.setLevel(self.log_level)
        else:
            self.log_level = logging.CRITICAL
            self.logger.off()

        self.blend_processor = blend_processor
        self.output_dir = output_dir
        self.min_match_count = min_match_count
        self.lowes_ratio_threshold = lowes_ratio_threshold
        self.ref_image_contrib = ref_image_contrib
        self.type = type
        self.resize = resize
        self.consecutive_range = consecutive_range
        self.matcher = matcher
        self.logger.info(f"Initialising {options['matcher'][self.matcher].upper()} matcher")
        # matcher 0: cv2.BFMatcher() == NORM_L2 for every algorithm (stitcher.py:85); matcher 1 (FLANN) is
        # approximate in cv2 -- both are served by the exact GPU brute-force L2 kernel.
        self.matcher_obj = None
        self.algorithm = algorithm
        self.logger.info(f"Initialising {options['algorithm'][self.algorithm].upper()} algorithm")
        self._detector = detector
        self.logger.info(f"Using {options['type'][self.type].upper()} type stitching")
        self.blender = blender
        self.logger.info(f"Using {options['blender'][self.blender].upper()} blender")
        self.device = torch.device(device) if device is not None else None
        self.resident = resident
        if norm not in ("l2", "hamming"):
            raise ValueError("norm must be 'l2' (what the reference's cv2.BFMatcher() computes) or 'hamming'")
        self.norm = norm  # extension: 'hamming' = cv2.BFMatcher(cv2.NORM_HAMMING) for binary descriptors
        # extension: blend_N.png is copied out and encoded off the stitching stream (async_save.py); the
        # file is on disk after flush_saves() (also run at interpreter exit), not when save_* returns
        self.async_save = async_save
        # extension: which INTER_LINEAR arithmetic cv2.warp* is taken to run -- "f32" (OpenCV 4.11, the
        # reference's pinned version) or "classic" (<= 4.10 fixed point); see ops.warp
        if warp_arith not in ops.WARP_ARITH:
            raise ValueError("warp_arith must be 'f32' or 'classic'")
        self.warp_arith = warp_arith
        # extension: the two CPU detections of a `stitch` call (frame and reference) run concurrently, the
        # frame's on a worker thread.  Same results (they are independent); needs either no `detector=`
        # (two cv2 objects are built), a pair `detector=(a, b)`, or one object that is safe to call from
        # two threads at once.
        self.parallel_detect = parallel_detect
        # extension (opt-in): single-channel uint16 frames (depth maps) keep their 16 bits.  The reference allocates every
        # canvas as uint8 (utils/proc.py:283-291), so a uint16 frame wraps modulo 256 there -- and here by default.  With
        # depth16=True such frames take the warp / alpha-merge path of _stitch_depth16 below: the same steps on uint16.
        self.depth16 = depth16
        self._own_detector = False
        self._detector_b = None
        if isinstance(detector, (tuple, list)):
            self._detector, self._detector_b = detector
        self._pool = None
        self._writer = None
        self._ctx = None          # ops.FrameCtx, created on first use
        self._seams_plain_cache = None
        self.phase_seconds = {}   # diagnostics: wall seconds per phase of `stitch`
        self._inplace_ok = False  # set by the internal callers that replace the reference mosaic by the result
        self.frame_stats = {"frames": 0, "rows_slow": 0, "rescans": 0, "clears": 0, "arena_allocs": 0}  # diagnostics

    # ---- plumbing -------------------------------------------------------------------------------
    @property
    def algorithm_obj(self):
        if self._detector is None:
            self._detector = make_detector(self.algorithm)
            self._own_detector = True
        return self._detector

    def _detect_two(self, img_a, make_b):
        """(detectAndCompute(img_a), detectAndCompute(make_b())) -- img_a on a worker thread while this thread
        fetches and detects the second image when parallel_detect is set, one after the other otherwise."""
        det = self.algorithm_obj
        if not self.parallel_detect or getattr(det, "accepts_device", False):  # (a GPU detector serves one thread)
            ra = det.detectAndCompute(img_a, None)
            return ra, det.detectAndCompute(make_b(), None)
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(1, thread_name_prefix="isx-detect")
            if self._detector_b is None:
                # a caller-supplied single object is used from both threads; our own cv2 objects come in pairs
                self._detector_b = det if self._own_detector is False else make_detector(self.algorithm)
        fut = self._pool.submit(self._detector_b.detectAndCompute, img_a, None)
        try:
            rb = det.detectAndCompute(make_b(), None)
        finally:
            fut.exception()  # never leave the worker running behind an exception of this thread
        return fut.result(), rb

    def _dev(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise ops.IsxError("no HIP device: the stitching core has no CPU fallback")
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    # ---- the seven seams (stitcher.py:106-165) -----------------------------------------------------
    @timer
    def get_matches(self, descriptors_1, descriptors_2, k=2):
        if descriptors_1 is None or descriptors_2 is None:
            return list()
        if k != 2:
            raise ops.IsxError("only k=2 is implemented (the reference never asks for another k)")
        if len(descriptors_1) == 0 or len(descriptors_2) == 0:
            return list()  # cv2 knnMatch returns [] for an empty query or train set
        with torch.cuda.device(self._dev()):
            kind = self._route(descriptors_1)
            if kind == "hamming":
                idx, dist, count = ops.knn2_hamming(descriptors_1, descriptors_2, self.lowes_ratio_threshold, self._dev())
            elif kind == "f32":
                idx, dist, count = ops.knn2_f32(descriptors_1, descriptors_2, self.lowes_ratio_threshold, self._dev())
            else:
                try:
                    # (dense byte descriptors whose squared distances leave the range where integer order == cv2's float
                    # order -- saturated AKAZE/ORB rows -- are matched inside ops by the kernel that ranks on the float
                    # distances themselves, still exactly)
                    idx, dist, count = ops.knn2_l2(descriptors_1, descriptors_2, self.lowes_ratio_threshold)
                except ops.NotIntegerValuedError:
                    # float descriptors that are not integer valued (found by the device-side check of
                    # isx_desc_prep): the fp32 matcher computes what cv2 does
                    f1, f2 = (np.asarray(d, np.float32) for d in (descriptors_1, descriptors_2))
                    idx, dist, count = ops.knn2_f32(f1, f2, self.lowes_ratio_threshold, self._dev())
        return KnnMatches(idx, dist, count)

    def _route(self, des):
        """_matcher_kind without the host scan of float32 arrays: they are sent to the integer path, whose
        device-side preparation reports non-integer values (NotIntegerValuedError -> fp32 matcher)."""
        a = des if isinstance(des, np.ndarray) else np.asarray(des)
        if a.dtype == np.float32 and self.norm != "hamming":
            return "int"
        return self._matcher_kind(a)

    def _matcher_kind(self, des):
        """'int' = exact int8-MFMA L2 (uint8, or integer-valued float32 as cv2.SIFT returns), 'f32' = fp32-MFMA
        L2 for any other float descriptors, 'hamming' = opt-in XOR/popcount for uint8."""
        a = des if isinstance(des, np.ndarray) else np.asarray(des)
        if a.dtype == np.uint8:
            return "hamming" if self.norm == "hamming" else "int"
        if self.norm == "hamming":
            raise ValueError("norm='hamming' needs uint8 (binary) descriptors")
        if a.dtype == np.float32:
            integer = bool(np.all(a == np.rint(a))) and float(a.min(initial=0)) >= 0 and float(a.max(initial=0)) <= 255
            return "int" if integer else "f32"
        raise ValueError(f"unsupported descriptor dtype {a.dtype}")

    @timer
    def filter_matches(self, matches):
        if isinstance(matches, KnnMatches):
            gq, gt = ops.ratio_compact(matches.idx, matches.dist, self.lowes_ratio_threshold)
            gq_h = gq.cpu().numpy()
            return GoodMatches(gq_h, gt.cpu().numpy())
        good = list()  # user-supplied list of DMatch rows: the reference's loop verbatim in behaviour
        for match in matches:
            if len(match) > 1:
                if match[0].distance < self.lowes_ratio_threshold * match[1].distance:
                    good.append([match[0]])
            else:
                good.append([match[0]])
        return good

    @timer
    def get_pts(self, src_kp, dst_kp, good_matches):
        if isinstance(good_matches, GoodMatches):
            src = _kp_array(src_kp)[good_matches.good_q]
            dst = _kp_array(dst_kp)[good_matches.good_t]
        else:
            src = np.float32([src_kp[m[0].queryIdx].pt for m in good_matches])
            dst = np.float32([dst_kp[m[0].trainIdx].pt for m in good_matches])
        return src.reshape(-1, 1, 2), dst.reshape(-1, 1, 2)

    @timer
    def get_transform(self, src_pts, dst_pts):
        with torch.cuda.device(self._dev()):
            M, _mask = ops.estimate_transform(src_pts, dst_pts, perspective=(self.type == 1), device=self._dev())
        return M

    @timer
    def warp_target(self, img, ref, M):
        """Full-canvas warp (API compatibility; `stitch` uses the region path below)."""
        d = DeviceImage.from_any(img, self._dev())
        with torch.cuda.device(self._dev()):
            out = ops.warp(d.t, M, self.type == 1, (0, 0, ref.shape[1], ref.shape[0]), arith=self.warp_arith)
        return DeviceImage(out)

    @timer
    def merge(self, warped_image, ref):
        if self.blender == 0:
            return seamless_merge(warped_image, ref, self.ref_image_contrib, device=self._dev())
        return seamless_gradient_merge(warped_image, ref, device=self._dev())

    @timer
    def detectAndComputeFromCorners(self, image, corners):
        roi_img = get_roi_from_corners(image, corners[0], corners[1])
        self.logger.debug(roi_img.shape)
        keypoints_roi, thisdes = self.algorithm_obj.detectAndCompute(np.ascontiguousarray(roi_img), None)
        thiskp = transform_keypoints_from_roi(keypoints_roi, corners[0])
        return thiskp, thisdes

    @timer
    def save_last_stitch(self, stitch, filename):
        filepath = os.path.join(self.output_dir, filename)
        if isinstance(stitch, StripMosaic):
            full = np.asarray(stitch)  # collective: every rank takes part in the gather, rank 0 writes the file
            if stitch.rank != 0:
                return
            stitch = full
        if self.async_save:
            if self._writer is None:
                self._writer = AsyncImageWriter(self._dev())
            self._writer.submit(stitch, filepath, self.blend_processor)
            self.logger.info(f"Queued last stitch for {filepath}")
            return
        processed = self.blend_processor(np.asarray(stitch))
        if not os.path.exists(self.output_dir):
            os.makedirs(self.output_dir)
        write_image(filepath, processed)
        self.logger.info(f"Saved last stitch to {filepath}")

    def flush_saves(self):
        """Returns when every file queued by an async_save stitcher is on disk (no-op otherwise)."""
        if self._writer is not None:
            self._writer.flush()

    # ---- per-pair pipeline (stitcher.py:176-229) ------------------------------------------------------
    @staticmethod
    def _host_canvas(img_host, H, W, ox, oy):
        shape = (H, W) if img_host.ndim == 2 else (H, W, img_host.shape[2])
        out = np.zeros(shape, np.uint8)
        out[oy:oy + img_host.shape[0], ox:ox + img_host.shape[1]] = img_host
        return out

    def _host_region(self, img, ox, oy, H, W, p1, p2, stable=False):
        """get_roi_from_corners on the padded canvas of `img` (pasted at ox, oy) without building the
        canvas: numpy slice semantics (clipped to the canvas), zeros outside the image.
        A region downloaded from the device is a VIEW of the ctx's pinned staging block, valid until the next staging
        call on the ctx (frame_download / frame_match / frame_transform); stable=True returns a private copy instead
        (needed when another thread keeps reading it while this thread downloads the next region)."""
        keep = (lambda v: np.array(v)) if stable else (lambda v: v)
        if isinstance(img, DeviceImage) and img._host is not None:
            img = img._host  # the frame came from the host: no download needed
        x0, y0 = max(0, min(W, p1[0])), max(0, min(H, p1[1]))
        x1, y1 = max(x0, min(W, p2[0])), max(y0, min(H, p2[1]))
        if isinstance(img, StripMosaic):  # rows live on several ranks: assembled by one all_gather (zeros outside)
            return img.region(x0 - ox, y0 - oy, x1 - ox, y1 - oy)
        if x0 >= ox and y0 >= oy and x1 <= ox + img.shape[1] and y1 <= oy + img.shape[0] and x1 > x0 and y1 > y0:
            # the region lies inside the image (the usual case): a view of the host frame / a download of
            # just that part of the resident mosaic -- no zero canvas, no second copy
            if isinstance(img, DeviceImage) and self._seams_plain:
                # one pitched DMA into the ctx's pinned block; the view is consumed by the detector right away
                # (before the next staging call on the ctx)
                return keep(ops.frame_download(self._frame_ctx(), img.t[y0 - oy:y1 - oy, x0 - ox:x1 - ox]))
            return np.asarray(img[y0 - oy:y1 - oy, x0 - ox:x1 - ox])
        ix0, iy0 = max(x0, ox), max(y0, oy)
        ix1, iy1 = min(x1, ox + img.shape[1]), min(y1, oy + img.shape[0])
        inter = ix1 > ix0 and iy1 > iy0
        if isinstance(img, DeviceImage) and self._seams_plain and x1 > x0 and y1 > y0:
            # the slice reaches beyond the mosaic: zero frame + one pitched DMA of the part that exists, in the ctx
            src = img.t[iy0 - oy:iy1 - oy, ix0 - ox:ix1 - ox] if inter else img.t[0:0, 0:0]
            return keep(ops.frame_download(self._frame_ctx(), src, (y1 - y0, x1 - x0), (iy0 - y0, ix0 - x0) if inter else (0, 0)))
        c = () if img.ndim == 2 else (img.shape[2],)
        out = np.zeros((y1 - y0, x1 - x0) + c, np.uint8)
        if inter:
            src = img[iy0 - oy:iy1 - oy, ix0 - ox:ix1 - ox]
            out[iy0 - y0:iy1 - y0, ix0 - x0:ix1 - x0] = np.asarray(src)
        return out

    def _device_region(self, img, ox, oy, H, W, p1, p2):
        """_host_region for a detector that works in HBM (detect.HipOrb: `accepts_device`): the same pixels as a
        device tensor -- a view of the resident image when the region lies inside it, else zeros + one device copy."""
        if not isinstance(img, DeviceImage):
            return self._host_region(img, ox, oy, H, W, p1, p2)
        x0, y0 = max(0, min(W, p1[0])), max(0, min(H, p1[1]))
        x1, y1 = max(x0, min(W, p2[0])), max(y0, min(H, p2[1]))
        ih, iw = img.shape[:2]
        if x0 >= ox and y0 >= oy and x1 <= ox + iw and y1 <= oy + ih and x1 > x0 and y1 > y0:
            return img.t[y0 - oy:y1 - oy, x0 - ox:x1 - ox]
        out = torch.zeros((y1 - y0, x1 - x0) + tuple(img.t.shape[2:]), dtype=torch.uint8, device=img.t.device)
        ix0, iy0, ix1, iy1 = max(x0, ox), max(y0, oy), min(x1, ox + iw), min(y1, oy + ih)
        if ix1 > ix0 and iy1 > iy0:
            out[iy0 - y0:iy1 - y0, ix0 - x0:ix1 - x0] = img.t[iy0 - oy:iy1 - oy, ix0 - ox:ix1 - ox]
        return out

    @staticmethod
    def _device_canvas(img_t, H, W, ox, oy):
        out = torch.zeros((H, W) + tuple(img_t.shape[2:]), dtype=torch.uint8, device=img_t.device)
        out[oy:oy + img_t.shape[0], ox:ox + img_t.shape[1]] = img_t
        return out

    def _quad_region(self, M, ox, oy, iw, ih, W, H, margin):
        """Conservative canvas rectangle that contains every non-zero pixel of the warped frame."""
        # (plain Python floats: four points do not pay for numpy's per-call overhead)
        m = [[float(v) for v in row] for row in M]
        if len(m) == 2:
            m.append([0.0, 0.0, 1.0])
        xs, ys = [], []
        for px, py in ((ox - 1, oy - 1), (ox + iw, oy - 1), (ox + iw, oy + ih), (ox - 1, oy + ih)):
            qx = m[0][0] * px + m[0][1] * py + m[0][2]
            qy = m[1][0] * px + m[1][1] * py + m[1][2]
            qw = m[2][0] * px + m[2][1] * py + m[2][2]
            if not (math.isfinite(qx) and math.isfinite(qy) and math.isfinite(qw)) or qw <= 1e-9:
                return 0, 0, W, H
            x, y = qx / qw, qy / qw
            if not (math.isfinite(x) and math.isfinite(y)) or abs(x) > 1e8 or abs(y) > 1e8:
                return 0, 0, W, H
            xs.append(x)
            ys.append(y)
        x0 = int(math.floor(min(xs))) - 2 - margin
        y0 = int(math.floor(min(ys))) - 2 - margin
        x1 = int(math.ceil(max(xs))) + 3 + margin
        y1 = int(math.ceil(max(ys))) + 3 + margin
        return max(0, x0), max(0, y0), min(W, x1), min(H, y1)

    @property
    def _seams_plain(self):
        """True when no subclass replaced a matching / transform seam (stitcher.py:106-141): `stitch` then runs them as
        two composite library calls (ops.frame_match, ops.frame_transform) -- same kernels, same results."""
        v = self._seams_plain_cache
        if v is None:
            v = not any(self._overridden(n) for n in ("get_matches", "filter_matches", "get_pts", "get_transform"))
            self._seams_plain_cache = v
        return v

    def _good_matches_fast(self, des1, des2):
        """get_matches + filter_matches (stitcher.py:106-127) for one pair in one library call."""
        if des1 is None or des2 is None or len(des1) == 0 or len(des2) == 0:
            return GoodMatches(np.zeros(0, np.int32), np.zeros(0, np.int32))
        kind = self._route(des1)
        if kind != "f32":
            try:
                with torch.cuda.device(self._dev()):
                    gq, gt = ops.frame_match(self._frame_ctx(), des1, des2, self.lowes_ratio_threshold, hamming=(kind == "hamming"))
                return GoodMatches(gq, gt)
            except (ops.NormLimitError, ops.NotIntegerValuedError):
                pass  # the float32 matcher computes what cv2 does for such descriptors (see get_matches)
        return self.filter_matches(self.get_matches(des1, des2, 2))

    def _frame_ctx(self):
        if self._ctx is None:
            self._ctx = ops.FrameCtx(self._dev())
        return self._ctx

    def _blend_resident(self, img_d, ref_d, M, W, H, rox, roy, ox, oy):
        """stitcher.py:216-225 (warp_target, merge, corners of the warped image, crop_image) on a resident
        mosaic arena through ONE library call (ops.frame_blend -> isx_frame_blend).  Coordinates are those of the
        reference's padded canvas (W x H, mosaic pasted at (rox, roy), frame at (ox, oy)); the canvas itself is
        never built.  Returns (mosaic DeviceImage | None, corners-of-warped | False)."""
        ih, iw = img_d.shape[:2]
        rh, rw = ref_d.shape[:2]
        C = img_d.channels
        arena = ref_d._arena
        live = (arena is not None and self._inplace_ok and (arena.h, arena.w) == (rh, rw) and arena.t.device == ref_d.t.device
                and arena.view().data_ptr() == ref_d.t.data_ptr())
        if not live:
            # a host image, a plain tensor, or a mosaic the caller may still want untouched: start a new arena
            # shaped like the reference's canvas, so the first frame always fits
            arena = MosaicArena.around(ref_d.t, iw, ih)
            self.frame_stats["arena_allocs"] += 1
        if arena.bbox is None:
            bb = ops.bbox_nonzero(arena.view())[:5].cpu().numpy()
            arena.bbox = (int(bb[0]) + arena.x, int(bb[1]) + arena.y, int(bb[2]) + arena.x, int(bb[3]) + arena.y) if bb[4] else False
        qx0, qy0, qx1, qy1 = self._quad_region(M, ox, oy, iw, ih, W, H, 52)
        qw, qh = max(0, qx1 - qx0), max(0, qy1 - qy0)
        if qw == 0 or qh == 0:
            qx0 = qy0 = qw = qh = 0
        if qw > ops.FRAME_MAX_Q or qh > ops.FRAME_MAX_Q:
            # a degenerate transform (the quad's bound fell back to the whole canvas, or the frame is blown up beyond the
            # frame pipeline's rectangle limit): the reference just produces a bad warp; so does the canvas path
            return None, None
        cax, cay = arena.x - rox, arena.y - roy  # canvas -> arena
        if qw:
            t_before = arena.t
            dx, dy = arena.ensure(qx0 + cax, qy0 + cay, qx0 + qw + cax, qy0 + qh + cay, max(iw, arena.t.shape[1] // 2), max(ih, arena.t.shape[0] // 2))
            cax, cay = cax + dx, cay + dy
            self.frame_stats["arena_allocs"] += arena.t is not t_before
        old = arena.bbox
        res = ops.frame_blend(self._frame_ctx(), img_d.t, M, self.type == 1, (ox, oy), (W, H), arena.t, (cax, cay),
                              (qx0, qy0, qw, qh), self.blender, self.ref_image_contrib, 10, self.warp_arith)
        wb = res["wbbox"]
        wcorners = ((wb[0], wb[1]), (wb[2], wb[3])) if wb else False
        written = None  # canvas rectangle whose mosaic pixels were rewritten
        rescan = False
        if self.blender == 0:
            written = (qx0, qy0, qx0 + qw, qy0 + qh) if qw else None
        elif wb:
            written = res["roi"]
            if not res["done"]:
                # rows_ok == 0 (a channel-0-only pixel far outside the warped image's gray box): the extents of
                # utils/proc.py:213-218 have to be taken inside the ROI -- the four-kernel blender does that
                gx0, gy0, gx1, gy1 = written
                wv = ops.frame_warped_view(self._frame_ctx(), (qx0, qy0, qw, qh), C, img_d.ndim == 3)
                ops.gradient_blend_roi(wv[gy0 - qy0:gy1 - qy0, gx0 - qx0:gx1 - qx0],
                                       arena.t[gy0 + cay:gy1 + cay, gx0 + cax:gx1 + cax])
                rescan = True
                self.frame_stats["rows_slow"] += 1
        self.frame_stats["frames"] += 1
        # the rectangle that can hold non-zero bytes now: the old mosaic's and what was written (arena coordinates)
        ux0, uy0, ux1, uy1 = arena.x, arena.y, arena.x + arena.w, arena.y + arena.h
        if written is not None:
            ux0, uy0 = min(ux0, written[0] + cax), min(uy0, written[1] + cay)
            ux1, uy1 = max(ux1, written[2] + cax), max(uy1, written[3] + cay)
        if rescan or res["zeroed"]:
            self.frame_stats["rescans"] += 1
            bb = ops.bbox_nonzero(arena.t[uy0:uy1, ux0:ux1])[:5].cpu().numpy()
            new = (int(bb[0]) + ux0, int(bb[1]) + uy0, int(bb[2]) + ux0, int(bb[3]) + uy0) if bb[4] else False
        else:
            mb = res["mbbox"]
            new = (mb[0] + cax, mb[1] + cay, mb[2] + cax, mb[3] + cay) if res["mfound"] else False
            if old and new:
                new = (min(old[0], new[0]), min(old[1], new[1]), max(old[2], new[2]), max(old[3], new[3]))
            elif old:
                new = old
        if new is False:
            return None, wcorners
        if arena.dirty or res["dirty"] or rescan or res["zeroed"]:
            # bytes of gray == 0 pixels outside the crop: the reference's crop drops them
            self.frame_stats["clears"] += ops.clear_rects(arena.t, ((uy0, new[1], ux0, ux1), (new[3], uy1, ux0, ux1),
                                                                     (new[1], new[3], ux0, new[0]), (new[1], new[3], new[2], ux1)))
        arena.x, arena.y, arena.w, arena.h = new[0], new[1], new[2] - new[0], new[3] - new[1]
        arena.bbox, arena.dirty = new, False
        return DeviceImage(arena.view(), arena=arena), wcorners

    def _overridden(self, name):
        """True when a subclass replaced one of the reference's seam methods (stitcher.py:106-165)."""
        return getattr(type(self), name) is not getattr(Stitcher, name)

    def _stitch_tail_via_seams(self, img_d, ref_d, M, tf):
        """stitcher.py:216-229 through the overridable seams, on materialised padded canvases (resident in
        HBM).  Taken only when a subclass overrides warp_target or merge; the default path below produces
        the same mosaic from the warped frame's bounding region without building the canvases."""
        dev = self._dev()
        current_image, reference_image = get_padded_images(img_d, ref_d, device=dev)
        warped_image = self.warp_target(current_image, reference_image, M)
        reference_image = self.merge(warped_image, reference_image)
        corners = None
        if self.consecutive_range is not None:
            corners = get_corners_from_image(warped_image, device=dev)
            if corners is False:
                self.logger.warning("Could not get image corners. The merged image has no non-black regions.")
                corners = None
        reference_image = crop_image(reference_image, device=dev)
        if reference_image is False:
            return None
        if not self.resident:
            reference_image = np.ascontiguousarray(np.asarray(reference_image))
        return reference_image, corners, tf

    def _phase(self, name, t0):
        """Accumulate wall seconds per phase of `stitch` (diagnostics: bench.py reports them)."""
        t1 = time.perf_counter()
        self.phase_seconds[name] = self.phase_seconds.get(name, 0.0) + (t1 - t0)
        return t1

    @staticmethod
    def _is_u16(a):
        if isinstance(a, DepthImage):
            return True
        if isinstance(a, torch.Tensor):
            return a.dtype == torch.uint16 and a.ndim == 2
        return isinstance(a, np.ndarray) and a.dtype == np.uint16 and a.ndim == 2

    def _stitch_depth16(self, arg_ref, arg_img, tf, corners):
        """stitcher.py:176-229 for single-channel uint16 frames with the canvases in the FRAME's dtype (opt-in; the
        reference's uint8 canvases wrap the values modulo 256, utils/proc.py:283-291): get_padded_images, warp_target
        (INTER_LINEAR on CV_16U: float32 linear kernels), seamless_merge (blender=0) on uint16, corners of the warped
        image, crop_image.  It is the warp/blend-only path of synchronised RGB + depth stitching (README.md:90-103):
        the transform comes from the RGB stitcher (tf), a depth map is not something a feature detector is run on."""
        if tf is None:
            raise ops.IsxError("depth16=True stitches uint16 frames with a transform handed over (tf=...), as in the "
                               "reference's synchronised RGB + depth use; detection on depth maps is not offered")
        if self.blender != 0:
            raise ops.IsxError("depth16=True needs blender=0 (alpha): the reference's gradient blender casts its result "
                               "to uint8 (utils/proc.py:253)")
        if self.warp_arith != "f32":
            raise ops.IsxError("depth16=True runs the float32 linear warp only (warp_arith='f32')")
        dev = self._dev()
        img, ref = DepthImage.from_any(arg_img, dev), DepthImage.from_any(arg_ref, dev)
        ih, iw = img.shape
        rh, rw = ref.shape
        W, H = rw + 2 * iw, rh + 2 * ih                      # utils/proc.py:295-304
        ox, oy = (rw + iw) // 2, (rh + ih) // 2
        M, src_pts, dst_pts = tf.get()
        if M is None:
            raise RuntimeError("transform estimation failed (cv2 would raise on M=None)")
        M = np.asarray(M, np.float64)
        with torch.cuda.device(dev):
            canvas = torch.zeros((H, W), dtype=torch.uint16, device=dev)
            canvas[ih:ih + rh, iw:iw + rw] = ref.t
            qx0, qy0, qx1, qy1 = self._quad_region(M, ox, oy, iw, ih, W, H, 2)
            wcorners = False
            if qx1 > qx0 and qy1 > qy0:
                # outside this rectangle the warped image is zero and the merge leaves the mosaic as it is
                warped = ops.warp16(img.t, M, self.type == 1, (qx0, qy0, qx1 - qx0, qy1 - qy0), src_xy=(ox, oy))
                if self.consecutive_range is not None:
                    wc = ops.bbox_nonzero16(warped)
                    if wc:
                        wcorners = ((wc[0][0] + qx0, wc[0][1] + qy0), (wc[1][0] + qx0, wc[1][1] + qy0))
                ops.alpha_blend16(warped, canvas[qy0:qy1, qx0:qx1], self.ref_image_contrib)
            cc = ops.bbox_nonzero16(canvas)
        if self.consecutive_range is not None:
            corners = wcorners
            if corners is False:
                self.logger.warning("Could not get image corners. The merged image has no non-black regions.")
                corners = None
        if cc is False:
            return None
        (x0, y0), (x1, y1) = cc
        mosaic = DepthImage(canvas[y0:y1, x0:x1])
        if not self.resident:
            mosaic = mosaic.numpy()
        return mosaic, corners, tf

    @timer
    def stitch(self, arg_ref, arg_img, tf=None, corners=None):
        tp = time.perf_counter()
        dev = self._dev()
        if self.depth16 and self._is_u16(arg_img) and self._is_u16(arg_ref):
            return self._stitch_depth16(arg_ref, arg_img, tf, corners)
        img_d = DeviceImage.from_any(arg_img, dev)
        ref_d = arg_ref if isinstance(arg_ref, StripMosaic) else DeviceImage.from_any(arg_ref, dev)
        ih, iw = img_d.shape[:2]
        rh, rw = ref_d.shape[:2]
        C = img_d.channels
        if ref_d.channels != C or img_d.ndim != ref_d.ndim:
            raise ValueError("reference and image must have the same number of channels")
        # get_padded_images (utils/proc.py:295-304): canvas = ref + 2*img per axis, both pasted centred
        W, H = rw + 2 * iw, rh + 2 * ih
        rox, roy = iw, ih
        ox, oy = (rw + iw) // 2, (rh + ih) // 2
        bb = None
        host = img_d._host
        if host is not None and ih > 0 and iw > 0:
            # get_corners_from_image of the frame (stitcher.py:178-179): when each of the four border lines holds a pixel
            # with gray != 0 the box is the whole frame, and the host copy can tell (G != 0 implies gray != 0: its weight
            # alone, 19235, exceeds the rounding threshold 16384; for one channel the byte is the gray value) -- no
            # kernel, no read-back.  Any other frame goes to the device.
            g = host if host.ndim == 2 else (host[:, :, 1] if host.shape[2] == 3 else None)
            if g is not None and g[0].any() and g[ih - 1].any() and g[:, 0].any() and g[:, iw - 1].any():
                bb = (0, 0, iw, ih, 1)
        if bb is None:
            with torch.cuda.device(dev):
                bb = ops.bbox_nonzero(img_d.t).cpu().numpy()
        if bb[4]:
            current_corners = ((int(bb[0]) + ox, int(bb[1]) + oy), (int(bb[2]) + ox, int(bb[3]) + oy))
        else:
            current_corners = False
            self.logger.warning("Could not get image corners. The input image has no non-black regions.")
            corners = None
        tp = self._phase("upload+frame_bbox", tp)

        if tf is None:
            thisdes = None
            dev_det = (getattr(self.algorithm_obj, "accepts_device", False) and isinstance(ref_d, DeviceImage)
                       and not self._overridden("detectAndComputeFromCorners"))
            if corners is not None and self.consecutive_range is not None:
                try:
                    if self._overridden("detectAndComputeFromCorners"):
                        # a subclass's seam (stitcher.py:159-165): it gets the padded canvases, as in the reference
                        cur_c, ref_c = get_padded_images(img_d, ref_d, device=dev)
                        thiskp, thisdes = self.detectAndComputeFromCorners(cur_c, current_corners)
                        refkp, refdes = self.detectAndComputeFromCorners(ref_c, corners)
                    else:
                        # a detector that works in HBM gets device views: the mosaic region never visits the host
                        region = self._device_region if dev_det else self._host_region
                        # (parallel_detect: the frame's region is read by the worker thread while this thread downloads
                        # the reference's region into the same pinned block -- it must be a private copy then)
                        if dev_det:
                            cur_roi = region(img_d, ox, oy, H, W, current_corners[0], current_corners[1])
                        else:
                            cur_roi = region(img_d, ox, oy, H, W, current_corners[0], current_corners[1], stable=self.parallel_detect)
                        (kps, thisdes), (rkps, refdes) = self._detect_two(
                            cur_roi, lambda: region(ref_d, rox, roy, H, W, corners[0], corners[1]))
                        if self._seams_plain:
                            # coordinates only (the keypoint objects do not leave this call): no per-object loops
                            thiskp, refkp = _kp_shifted(kps, current_corners[0]), _kp_shifted(rkps, corners[0])
                        else:
                            thiskp = transform_keypoints_from_roi(kps, current_corners[0])
                            refkp = transform_keypoints_from_roi(rkps, corners[0])
                except Exception as e:  # same recovery as stitcher.py:189-191
                    self.logger.warning("Error in detecting keypoints from corners. Computing on whole image. Exception: ", str(e))
                    corners = None
            if corners is None or self.consecutive_range is None or thisdes is None:
                if dev_det:
                    with torch.cuda.device(dev):
                        (thiskp, thisdes), (refkp, refdes) = self._detect_two(
                            self._device_canvas(img_d.t, H, W, ox, oy), lambda: self._device_canvas(ref_d.t, H, W, rox, roy))
                else:
                    (thiskp, thisdes), (refkp, refdes) = self._detect_two(
                        self._host_canvas(img_d.numpy(), H, W, ox, oy), lambda: self._host_canvas(ref_d.numpy(), H, W, rox, roy))

            tp = self._phase("detect(cpu)+roi_download+kp", tp)
            if refdes is None:
                self.logger.warning("Key points not found in ref. Moving to next blend.")
                return False

            if self._seams_plain:
                good_matches = self._good_matches_fast(thisdes, refdes)
            else:
                matches = self.get_matches(thisdes, refdes, 2)
                good_matches = self.filter_matches(matches)
            tp = self._phase("match", tp)
            if len(good_matches) >= self.min_match_count:
                self.logger.debug("Enough matches - {}/{}".format(len(good_matches), self.min_match_count))
                src_pts, dst_pts = self.get_pts(thiskp, refkp, good_matches)
                tp = self._phase("get_pts", tp)
                if self._seams_plain:
                    with torch.cuda.device(dev):
                        M, _mask = ops.frame_transform(self._frame_ctx(), src_pts, dst_pts, self.type == 1)
                else:
                    M = self.get_transform(src_pts, dst_pts)
                tf = TransformDetails(M, src_pts, dst_pts)
                tp = self._phase("transform", tp)
            else:
                self.logger.debug("Cannot blend. Not enough matches found - {}/{}".format(len(good_matches), self.min_match_count))
                return None
        else:
            M, src_pts, dst_pts = tf.get()
        if M is None:
            # the reference passes None to cv2.warp* here, which raises (stitcher.py:208, :216)
            raise RuntimeError("transform estimation failed (cv2 would raise on M=None)")
        M = np.asarray(M, np.float64)
        if isinstance(ref_d, StripMosaic):
            if self._overridden("warp_target") or self._overridden("merge") or self._overridden("detectAndComputeFromCorners"):
                raise ops.IsxError("a strip-partitioned mosaic is blended by the library's frame pipeline only "
                                   "(overridden warp_target / merge / detectAndComputeFromCorners seams are not supported)")
            with torch.cuda.device(dev):
                mosaic, wcorners = ref_d.blend(self, img_d, M, W, H, rox, roy, ox, oy)
        elif self._overridden("warp_target") or self._overridden("merge"):
            return self._stitch_tail_via_seams(img_d, ref_d, M, tf)
        else:
            with torch.cuda.device(dev):
                mosaic, wcorners = self._blend_resident(img_d, ref_d, M, W, H, rox, roy, ox, oy)
            if mosaic is None and wcorners is None:  # degenerate transform: materialised canvases, as the reference
                return self._stitch_tail_via_seams(img_d, ref_d, M, tf)
        tp = self._phase("warp+blend", tp)

        if self.consecutive_range is not None:
            corners = wcorners
            if corners is False:
                self.logger.warning("Could not get image corners. The merged image has no non-black regions.")
                corners = None

        if mosaic is None:
            return None  # crop_image returned False (stitcher.py:226-227)
        if not self.resident and not isinstance(mosaic, StripMosaic):
            mosaic = np.ascontiguousarray(mosaic.numpy())
        return mosaic, corners, tf


class ArbitraryStitcher(Stitcher):
    def __init__(self, images=None, filepaths=None, with_tqdm=True, **kwargs):
        super().__init__(**kwargs)
        self.filepaths = None
        if images is not None:
            self.images = images
        elif filepaths is not None:
            self.filepaths = filepaths
            self.logger.debug("Input filepaths:", filepaths)
            self.filepaths = [f for f in self.filepaths if f[-4:] == ".png" or f[-4:] == ".jpg"]
            self.images = LazyList(lambda index: load_image(self.filepaths[index], to_resize=self.resize), len(self.filepaths))
        else:
            raise ValueError("No images specified")
        if with_tqdm == False:  # noqa: E712  (stitcher.py:250-255 compares with ==: 0 / 1 work too)
            self.tqdm = no_tqdm
        elif with_tqdm == True:  # noqa: E712
            self.tqdm = tqdm
        else:
            self.tqdm = with_tqdm
        self.dist_group = None  # torch.distributed group to shard get_collections over (None = default)
        # extension: strip_group = a torch.distributed process group (or True for the default group):
        # stitch_collections then keeps every mosaic partitioned into row strips over the group's ranks (strips.py); all
        # ranks must call stitch_collections with the same images; rank 0 writes the blend_i.png files
        self.strip_group = None

    @timer
    def update_all_keypoints_and_descriptors(self):
        kp, des, empty = list(), list(), list()
        for i in range(0, len(self.images)):
            thiskp, thisdes = self.algorithm_obj.detectAndCompute(np.asarray(self.images[i]), None)
            if thisdes is None:
                empty.append(i)
            kp.append(thiskp)
            des.append(thisdes)
        for index in sorted(empty, reverse=True):  # dropping images with no keypoints (stitcher.py:270-275)
            if isinstance(self.images, LazyList):
                self.images.length -= 1
            else:
                del self.images[index]
            del kp[index]
            del des[index]
            if self.filepaths is not None:
                del self.filepaths[index]
        self.kp = kp
        self.des = des

    def _pair_counts(self, n):
        """good-match counts of every unordered pair, both directions: (pairs (p,2), counts (p,2) numpy).
        Sharded over torch.distributed ranks when a process group is initialised."""
        import torch.distributed as dist
        dev = self._dev()
        world = dist.get_world_size(self.dist_group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(self.dist_group) if world > 1 else 0
        pairs = unordered_pairs(n, self.consecutive_range)
        with torch.cuda.device(dev):
            mine = shard_pairs(pairs, rank, world)
            kinds = {self._route(d) for d in self.des if d is not None and len(d)}
            kind = "f32" if "f32" in kinds else (kinds.pop() if kinds else "int")
            if kind == "int":
                try:
                    P = ops.prepare_descriptors(self.des, device=dev)
                    pitems = pair_items(mine, np.asarray(P.starts), np.asarray(P.counts))
                    if ops.pairs_kernel_applies(P, pitems):
                        # both directions of a pair from one distance matrix (isx_knn2_l2_count_pairs); counts do not
                        # depend on the row order, so the rows are sorted by norm first (short candidate lists)
                        c = ops.knn2_l2_pairs(ops.sort_rows_by_norm(P), pitems, self.lowes_ratio_threshold)
                    else:
                        items = directed_items(mine, P.starts, P.counts)
                        c = ops.knn2_l2_items(P, items, self.lowes_ratio_threshold).view(-1, 2)
                except ops.NotIntegerValuedError:
                    kind = "f32"  # see get_matches
            if kind != "int":
                des = self.des if kind == "hamming" else [None if d is None else np.asarray(d, np.float32) for d in self.des]
                c = ops.alt_item_counts(des, lambda st, ct: directed_items(mine, st, ct), kind,
                                        self.lowes_ratio_threshold, dev).view(-1, 2)
            allc = gather_counts(c, len(pairs), rank, world, self.dist_group)
            if world > 1:  # every rank runs the union-find: the counts go out as one tensor broadcast (RCCL on GPUs)
                if rank != 0:
                    allc = torch.empty((len(pairs), 2), dtype=torch.int32, device=c.device)
                allc = allc.contiguous()
                dist.broadcast(allc, src=dist.get_global_rank(self.dist_group, 0) if self.dist_group is not None else 0,
                               group=self.dist_group)
                counts = allc.cpu().numpy()
            else:
                counts = allc.cpu().numpy()
        return pairs, counts

    @timer
    def get_collections(self):
        self.logger.info("Calculating collections")
        n_before = len(self.images)  # the union-find is sized BEFORE empties are dropped (stitcher.py:283)
        self.update_all_keypoints_and_descriptors()
        n = len(self.images)
        if n >= 2:
            pairs, counts = self._pair_counts(n)
        else:
            pairs, counts = np.zeros((0, 2), np.int64), np.zeros((0, 2), np.int32)
        self.good_counts = (pairs, counts)
        collections, connections = components_from_counts(n, pairs, counts[:, 0], counts[:, 1],
                                                          self.min_match_count, n_union_find=n_before)
        self.logger.debug(collections)
        self.logger.info(len(collections), "possible blends found")
        self.connections = connections
        return collections

    @timer
    def stitch_collections(self, collections):
        """Greedy per-component merge (stitcher.py:306-374); writes blend_{i}.png, returns []."""
        self.logger.info("Stitching images")
        progress_bar = tqdm(total=len(self.images)) if self.tqdm is not False and self.tqdm is not no_tqdm else None
        unblended_collections = copy.deepcopy(collections)
        blended_collections = list()
        i = 0
        while i < len(unblended_collections):
            idxs = copy.deepcopy(unblended_collections[i])
            if len(idxs) < 1:
                self.logger.warning("No images found in unblended group. Moving to next blend.")
                i += 1  # (the reference `continue`s without advancing and would spin forever)
                continue
            ref = idxs[0]
            reference_image = self.images[ref]
            if self.strip_group is not None:
                import torch.distributed as dist
                group = None if self.strip_group is True else self.strip_group
                reference_image = StripMosaic.from_image(np.asarray(reference_image), group if group is not None else dist.group.WORLD,
                                                         self._dev())
            idxs.remove(ref)
            key_points_broken = False
            while len(idxs) != 0:
                self.logger.debug(idxs)
                matched_once = False
                remove_indexes = list()
                corners = None
                for k in idxs:
                    self._inplace_ok = True  # reference_image is replaced by the result below
                    try:
                        result = self.stitch(reference_image, self.images[k], corners=corners)
                    finally:
                        self._inplace_ok = False
                    if result is False:
                        unblended_collections.append(idxs[k:])
                        key_points_broken = True
                        corners = None
                        break
                    elif result is None:
                        corners = None
                    else:
                        reference_image, corners, tf = result
                        matched_once = True
                        remove_indexes.append(k)
                        if progress_bar is not None:
                            progress_bar.update(1)
                for r in remove_indexes:
                    idxs.remove(r)
                if matched_once is False or key_points_broken is True:
                    break
            self.save_last_stitch(reference_image, f"blend_{i}.png")
            if progress_bar is not None:
                progress_bar.update(1)
            i += 1
        if progress_bar is not None:
            progress_bar.close()
        return blended_collections


class ConsecutiveStitcher(Stitcher):
    """stitcher.py:377-439.  One difference a caller can see (resident=True, the default): `refs[-1]` is a DeviceImage
    that ALIASES the stitcher's mosaic arena in HBM, and the next stitch_consecutive call blends into that arena in
    place -- a mosaic kept from an earlier call (refs[-1], its `.t`, a cached host copy) changes or goes stale with it,
    where the reference hands out a fresh ndarray per frame.  Take `np.array(np.asarray(st.refs[-1]))` (or construct
    with resident=False: every result is then a private host array) to keep a frozen copy."""

    def __init__(self,
                 ref_image_contrib=0.2,
                 consecutive_range=1,
                 backup_interval=False,
                 consecutive_volatility_threshold=4,
                 **kwargs):
        super().__init__(**kwargs)
        self.consecutive_volatility_threshold = consecutive_volatility_threshold
        self.ref_image_contrib = ref_image_contrib
        self.stitch_count = 0
        self.consecutive_range = consecutive_range
        self.refs = ConstantLengthList(self.consecutive_range)
        self.refs.append(None)
        self.corners = None
        self.image_count = 0
        self.current_image_count = 0
        self.backup_interval = backup_interval

    def save_last_stitch(self):
        return super().save_last_stitch(self.refs[-1], f"blend_{self.stitch_count}.png")

    def save_and_reset(self):
        self.save_last_stitch()
        self.refs.append(None)
        self.stitch_count += 1
        self.corners = None
        self.current_image_count = 0

    def finalise_current_image(self, result):
        self.refs[-1], self.corners, tf = result
        self.image_count += 1
        self.current_image_count += 1
        if self.backup_interval != False:  # noqa: E712  (the reference's test: 0 and False both mean 'off', stitcher.py:411)
            if self.image_count % self.backup_interval == 0:
                self.save_last_stitch()
        return tf

    def _stitch_inplace(self, ref, image, **kw):
        self._inplace_ok = True
        try:
            return self.stitch(ref, image, **kw)
        finally:
            self._inplace_ok = False

    @timer
    def stitch_consecutive(self, input_image, tf=None):
        """stitcher.py:416-439: returns the TransformDetails on success, the caller's `tf` otherwise."""
        self.logger.info(f"Stitch {self.stitch_count + 1}: Stitching image {self.image_count}")
        image = resize_image(input_image, self.resize)
        if self.refs[-1] is None:
            self.refs[-1] = image
        else:
            # refs[-1] / exref are replaced by the result on success, so the resident mosaic is blended in place
            result = self._stitch_inplace(self.refs[-1], image, corners=self.corners, tf=tf)
            if result is False or result is None:
                if self.current_image_count < self.consecutive_volatility_threshold:
                    for exref in reversed(self.refs[:-1]):
                        result = self._stitch_inplace(exref, image, corners=self.corners, tf=tf)
                        if not (result is False or result is None):
                            self.refs.pop()
                            tf = self.finalise_current_image(result)
                            return tf
                self.save_and_reset()
            else:
                tf = self.finalise_current_image(result)
                return tf
        return tf
