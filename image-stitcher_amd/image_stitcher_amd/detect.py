"""Feature detection stays on the CPU (BASELINE.json north_star: "keypoint detection stays on the
reference's cv2 SIFT/ORB so both paths see identical descriptors"; reference stitcher.py:92-100).

  make_detector(algorithm)  -> cv2.SIFT_create() / ORB_create() / AKAZE_create() when cv2 is importable
  NumpyOrbLite              -> a small numpy/scipy corner + BRIEF-style detector used by the tests and
                               bench.py where cv2 is not installed (it is NOT part of the hot path)
  NumpySiftLite             -> the same corners with a 4x4x8 gradient-histogram descriptor post-processed like
                               cv2.SIFT's (normalise, clip 0.2, renormalise, x512, round): 128-d float32 that is
                               INTEGER VALUED in [0,255], the property the exact int8 matcher relies on
  HipOrb                    -> OPT-IN on-GPU detector of ORB's structure (FAST-9 + Harris ranking + intensity-centroid
                               orientation + steered BRIEF, single scale; csrc/isx_orb.hip).  NOT cv2.ORB:
                               different keypoints and bits, so mosaics differ from a cv2 run (parity unpinned by
                               construction); it removes the CPU detector from the frame loop.
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


class LazyKeyPoints(list):
    """Keypoints of HipOrb: the coordinates live in `xy` ((n,2) float32), the KeyPoint objects are only built when
    somebody indexes or iterates the list (the stitchers' default path never does)."""
    __slots__ = ("xy", "response", "angle", "_n", "_built")

    def __init__(self, xy, response, angle):
        super().__init__()
        self.xy, self.response, self.angle, self._n, self._built = xy, response, angle, len(xy), False

    def _build(self):
        if not self._built:
            self._built = True
            for (x, y), r, a in zip(self.xy.tolist(), self.response.tolist(), self.angle.tolist()):
                kp = KeyPoint(x, y, r)
                list.append(self, kp)

    def __len__(self):
        return self._n

    def __iter__(self):
        self._build()
        return list.__iter__(self)

    def __getitem__(self, i):
        self._build()
        return list.__getitem__(self, i)


def orb_base_pattern(seed=11):
    """256 test pairs (x1, y1, x2, y2): isotropic Gaussian, sigma = 31/5, clipped to the radius-13 disc (rotation and
    rounding then stay inside the radius-15 patch)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pts = []
    while len(pts) < 512:
        p = np.rint(rng.normal(0, 31 / 5.0, 2))
        if p[0] * p[0] + p[1] * p[1] <= 13 * 13:
            pts.append(p)
    return np.array(pts, np.int64).reshape(256, 4)


def orb_rotated_patterns(seed=11):
    """int8 [32][256][4]: the base pairs rotated by k * 11.25 degrees, rounded once here on the host."""
    base = orb_base_pattern(seed).astype(np.float64)
    out = np.zeros((32, 256, 4), np.int8)
    for k in range(32):
        c, s = np.cos(k * 2 * np.pi / 32), np.sin(k * 2 * np.pi / 32)
        for j in (0, 2):
            x, y = base[:, j], base[:, j + 1]
            out[k, :, j] = np.rint(x * c - y * s)
            out[k, :, j + 1] = np.rint(x * s + y * c)
    return out


class HipOrb:
    """detectAndCompute on the GPU (see the module docstring).  Accepts a numpy image (uploaded), a uint8 torch tensor
    or a DeviceImage that already lives in HBM (`accepts_device`: the stitchers then skip the download of the mosaic
    region).  Returns (LazyKeyPoints, (n,32) uint8 ndarray) like cv2.ORB; `des_dev` of the last call stays on the device.
    One instance serves one thread."""
    accepts_device = True

    def __init__(self, nfeatures=500, fast_threshold=20, clean=16, seed=11, device=None, device_select=True):
        self.nfeatures, self.fast_threshold, self.clean = int(nfeatures), int(fast_threshold), int(clean)
        self.device_select = bool(device_select)  # False: pick the keypoints on the host (what happens beyond 65536 survivors)
        self.device = device
        self._pattern_host = orb_rotated_patterns(seed)
        self._pattern = None
        self._ws = self._tiles = self._tiles_host = self._out = self._out_host = self._sel = self._sel_host = None

    def _ensure(self, dev, h, w):
        import torch
        from . import ops
        if self._pattern is None or self._pattern.device != dev:
            self._pattern = torch.from_numpy(self._pattern_host).to(dev)
            self._ws = self._tiles = self._out = self._sel = None
        need = ops.orb_workspace_bytes(h, w)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need + need // 4,), dtype=torch.uint8, device=dev)
        nt = ((h + 31) // 32) * ((w + 31) // 32)
        if self._tiles is None or self._tiles.numel() < nt * 129:
            self._tiles = torch.empty((nt * 129 + nt * 16,), dtype=torch.int32, device=dev)
            self._tiles_host = torch.empty((self._tiles.numel(),), dtype=torch.int32).pin_memory()
        n = self.nfeatures
        if self._out is None:
            # one block, one download: [sel: 2 + 2n int32 | angle bins: n int32 | descriptors: n x 32 bytes]
            nbytes = (2 + 2 * n) * 4 + n * 4 + n * 32
            self._out = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            self._out_host = torch.empty((nbytes,), dtype=torch.uint8).pin_memory()
            self._sel = torch.empty((n,), dtype=torch.int32, device=dev)
            self._sel_host = torch.empty((n,), dtype=torch.int32).pin_memory()
        return nt

    def detectAndCompute(self, image, mask=None):
        import torch
        from . import ops
        from .device_image import DeviceImage
        if isinstance(image, DeviceImage):
            t = image.t
        elif isinstance(image, torch.Tensor):
            t = image
        else:
            a = np.ascontiguousarray(np.asarray(image))
            if a.dtype != np.uint8:
                a = a.astype(np.uint8)
            dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
            t = torch.from_numpy(a).to(dev)
        if not t.is_cuda:
            raise ops.IsxError("HipOrb needs a HIP device (there is no CPU path)")
        h, w = int(t.shape[0]), int(t.shape[1])
        if h < 2 * ops.ORB_EDGE + 1 or w < 2 * ops.ORB_EDGE + 1:
            return [], None
        dev = t.device
        with torch.cuda.device(dev):
            nt = self._ensure(dev, h, w)
            ops.orb_candidates(t, self._ws, self._tiles, self.fast_threshold, self.clean)
            N = self.nfeatures
            o_sel, o_ang, o_des = 0, (2 + 2 * N) * 4, (2 + 2 * N) * 4 + N * 4
            sel_d = self._out[o_sel:o_ang].view(torch.int32)
            ang_d = self._out[o_ang:o_des].view(torch.int32)
            des_d = self._out[o_des:o_des + N * 32].view(N, 32)
            if self.device_select and N <= 65536:
                # selection on the device: 6 kernels queued back to back, ONE download, one synchronisation
                ops.orb_select_describe(self._ws, h, w, self._tiles, N, self._pattern, sel_d, des_d, ang_d)
                self._out_host.copy_(self._out, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                oh = self._out_host.numpy()
                head = oh[:o_ang].view(np.int32)
                n = int(head[0])
            else:
                n = -1
            if n == 0:
                return [], None
            if n > 0:
                pos = head[2:2 + n].copy()
                key = head[2 + N:2 + N + n].view(np.uint32).copy()
                ang = oh[o_ang:o_ang + 4 * n].view(np.int32).copy()
                des = oh[o_des:o_des + 32 * n].reshape(n, 32).copy()
                des_d = des_d[:n]
            else:
                # more survivors than the device sort takes (or device_select off): select on the host from the tile slots
                th = self._tiles_host[:nt * 129]
                th.copy_(self._tiles[:nt * 129], non_blocking=True)
                torch.cuda.current_stream().synchronize()
                th = th.numpy()
                counts = th[:nt]
                total = int(counts.sum())
                if total == 0:
                    return [], None
                live = np.arange(64, dtype=np.int32)[None, :] < counts[:, None]
                pos = th[nt:nt * 65].reshape(nt, 64)[live]
                key = th[nt * 65:nt * 129].reshape(nt, 64)[live].view(np.uint32)
                idx = (pos >> 16).astype(np.uint64) * np.uint64(w) + (pos & 0xFFFF).astype(np.uint64)
                rank = ((~key).astype(np.uint64) << np.uint64(32)) | idx
                n = min(N, total)
                order = np.argpartition(rank, n - 1)[:n] if n < total else np.arange(total)
                order = order[np.argsort(rank[order], kind="stable")]
                pos, key = pos[order], key[order]
                self._sel_host[:n].copy_(torch.from_numpy(pos))
                sel = self._sel[:n]
                sel.copy_(self._sel_host[:n], non_blocking=True)
                des_d, ang_n = des_d[:n], ang_d[:n]
                ops.orb_describe(self._ws, h, w, sel, self._pattern, des_d, ang_n)
                self._out_host.copy_(self._out, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                oh = self._out_host.numpy()
                ang = oh[o_ang:o_ang + 4 * n].view(np.int32).copy()
                des = oh[o_des:o_des + 32 * n].reshape(n, 32).copy()
        self.des_dev = des_d
        xy = np.stack([pos & 0xFFFF, pos >> 16], 1).astype(np.float32)
        return LazyKeyPoints(xy, key.view(np.float32), ang.astype(np.float32) * 11.25), des
