"""Feature detection stays on the CPU (BASELINE.json north_star: "keypoint detection stays on the
reference's cv2 SIFT/ORB so both paths see identical descriptors"; reference stitcher.py:92-100).

  make_detector(algorithm)  -> cv2.SIFT_create() / ORB_create() / AKAZE_create() when cv2 is importable
  NumpyOrbLite              -> a small numpy/scipy corner + BRIEF-style detector used by the tests and
                               bench.py where cv2 is not installed (it is NOT part of the hot path)
  NumpySiftLite             -> the same corners with a 4x4x8 gradient-histogram descriptor post-processed like
                               cv2.SIFT's (normalise, clip 0.2, renormalise, x512, round): 128-d float32 that is
                               INTEGER VALUED in [0,255], the property the exact int8 matcher relies on
Any object with detectAndCompute(image, mask) -> (keypoints, descriptors) can be passed to the stitchers
as `detector=`; keypoints need a mutable `.pt` (x, y) like cv2.KeyPoint.
"""
import numpy as np


class KeyPoint:
    __slots__ = ("pt", "response")

    def __init__(self, x, y, response=0.0):
        self.pt = (float(x), float(y))
        self.response = response


class KeyPointList(list):
    """list of KeyPoint objects that also keeps the coordinates as an (n,2) float32 array (`xy`), so that the
    stitcher does not have to walk the objects (cv2 offers cv2.KeyPoint_convert for the same purpose)."""
    __slots__ = ("xy",)

    def __init__(self, kps, xy):
        super().__init__(kps)
        self.xy = xy


def make_detector(algorithm):
    try:
        import cv2
    except ImportError as e:  # pragma: no cover - cv2 is absent in the build image
        raise ImportError(
            "cv2 is not installed: pass detector=<object with detectAndCompute> to the stitcher "
            "(e.g. image_stitcher_amd.detect.NumpyOrbLite())") from e
    return [cv2.SIFT_create, cv2.ORB_create, cv2.AKAZE_create][algorithm]()


def _gray(img):
    if img.ndim == 2:
        return img
    b, g, r = (img[..., k].astype(np.int32) for k in range(3))
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


class NumpyOrbLite:
    """Harris corners + a fixed 256-pair BRIEF pattern on a blurred image -> (n,32) uint8 descriptors.
    Deterministic; translation- and small-rotation-tolerant, which is all the synthetic streams need."""

    def __init__(self, nfeatures=1500, patch=12, seed=7, nms=7):
        from scipy import ndimage  # noqa: F401  (fail early if scipy is missing)
        self.nfeatures, self.patch, self.nms = nfeatures, patch, nms
        rng = np.random.Generator(np.random.PCG64(seed))
        p = np.clip(np.rint(rng.normal(0, patch / 2.2, (256, 4))), -patch, patch).astype(np.int32)
        self.pairs = p

    def _corners(self, image):
        """Harris corners away from black (padding) regions -> (blurred float32 image, ys, xs, responses) or None"""
        from scipy import ndimage
        img = np.asarray(image)
        g8 = _gray(img)
        h, w = g8.shape
        m = self.patch + 4
        if h <= 2 * m + 2 or w <= 2 * m + 2:
            return None
        g = ndimage.gaussian_filter(g8.astype(np.float32), 2.0)
        ix = ndimage.sobel(g, axis=1)
        iy = ndimage.sobel(g, axis=0)
        sxx = ndimage.gaussian_filter(ix * ix, 2.0)
        syy = ndimage.gaussian_filter(iy * iy, 2.0)
        sxy = ndimage.gaussian_filter(ix * iy, 2.0)
        r = sxx * syy - sxy * sxy - 0.04 * (sxx + syy) ** 2
        # keep away from black (padding) regions: every pixel of the patch must be content
        content = ndimage.minimum_filter((g8 > 0).astype(np.uint8), size=2 * m + 1)
        r[content == 0] = 0
        r[:m] = 0
        r[-m:] = 0
        r[:, :m] = 0
        r[:, -m:] = 0
        peak = (r == ndimage.maximum_filter(r, size=self.nms)) & (r > 1e-6 * max(float(r.max()), 1e-30))
        ys, xs = np.nonzero(peak)
        if len(xs) == 0:
            return None
        resp = r[ys, xs]
        order = np.argsort(-resp, kind="stable")[: self.nfeatures]
        return g, ys[order], xs[order], resp[order], ix, iy

    def detectAndCompute(self, image, mask=None):
        found = self._corners(image)
        if found is None:
            return [], None
        g, ys, xs, resp = found[:4]
        p = self.pairs
        a = g[ys[:, None] + p[None, :, 1], xs[:, None] + p[None, :, 0]]
        b = g[ys[:, None] + p[None, :, 3], xs[:, None] + p[None, :, 2]]
        des = np.packbits(a < b, axis=1)
        kps = KeyPointList([KeyPoint(x, y, v) for x, y, v in zip(xs, ys, resp)], np.stack([xs, ys], 1).astype(np.float32))
        return kps, des


class NumpySiftLite(NumpyOrbLite):
    """Same corners; descriptor = 4x4 cells x 8 orientation bins of gradient magnitude over a 16x16 window,
    then cv2.SIFT's post-processing -> (n,128) float32, integer valued in [0,255], ||d|| ~ 512.  Stand-in for
    cv2.SIFT_create() (BASELINE configs[0], algorithm=0) where cv2 is not installed; NOT part of the hot path."""

    def __init__(self, nfeatures=1500, seed=7, nms=7):
        super().__init__(nfeatures=nfeatures, patch=12, seed=seed, nms=nms)
        d = np.arange(-8, 8)
        self._dy, self._dx = (a.reshape(-1) for a in np.meshgrid(d, d, indexing="ij"))
        self._cell = (((self._dy + 8) // 4) * 4 + (self._dx + 8) // 4) * 8

    def detectAndCompute(self, image, mask=None):
        found = self._corners(image)
        if found is None:
            return [], None
        g, ys, xs, resp, ix, iy = found
        py, px = ys[:, None] + self._dy[None, :], xs[:, None] + self._dx[None, :]
        gx, gy = ix[py, px], iy[py, px]
        mag = np.sqrt(gx * gx + gy * gy)
        ori = (np.floor((np.arctan2(gy, gx) + np.pi) * (8 / (2 * np.pi))).astype(np.int64)) % 8
        n = len(xs)
        flat = (np.arange(n)[:, None] * 128 + self._cell[None, :] + ori).reshape(-1)
        hist = np.bincount(flat, weights=mag.reshape(-1).astype(np.float64), minlength=n * 128).reshape(n, 128)
        hist = hist.astype(np.float32)
        hist /= np.maximum(np.linalg.norm(hist, axis=1, keepdims=True), 1e-12)
        np.minimum(hist, 0.2, out=hist)
        hist /= np.maximum(np.linalg.norm(hist, axis=1, keepdims=True), 1e-12)
        des = np.clip(np.rint(hist * 512.0), 0, 255).astype(np.float32)
        kps = KeyPointList([KeyPoint(x, y, v) for x, y, v in zip(xs, ys, resp)], np.stack([xs, ys], 1).astype(np.float32))
        return kps, des
