"""DeviceImage: a uint8 image (H,W) or (H,W,C) resident in HBM, array-like enough to stand in for the
ndarray the reference hands around (`refs[-1]`, the first element of `stitch`'s result).

  np.asarray(img) / img.numpy()   download (cached until the device data changes)
  img[y0:y1, x0:x1]               zero-copy crop (a strided view, like numpy)
  img.shape / img.ndim / len()    as numpy
"""
import numpy as np
import torch


class MosaicArena:
    """A mosaic resident in HBM inside a larger zero-filled image, so that the next frame is blended IN PLACE
    (no per-frame canvas allocation, no copy of the mosaic: SURVEY 8a-7).  Invariant: every byte of `t` outside
    the rectangle [x, x+w) x [y, y+h) is 0, which is what the reference's padded canvas holds there
    (utils/proc.py:280-304).  Grows by reallocation with slack (amortised O(1) per frame).

      t            (Ha, Wa[, C]) uint8 tensor
      x, y, w, h   the mosaic's rectangle inside t
      bbox         (x0, y0, x1, y1) of the mosaic's gray != 0 pixels in ARENA coordinates, False if there is none,
                   None if not known yet (a raw first frame may have black borders)
      dirty        True if bytes outside `bbox` may be non-zero (gray == 0 pixels such as (4,0,0)): they have to
                   be cleared once the mosaic is cropped to its bbox
    """
    __slots__ = ("t", "x", "y", "w", "h", "bbox", "dirty")

    def __init__(self, t, x, y, w, h, bbox=None, dirty=True):
        self.t, self.x, self.y, self.w, self.h, self.bbox, self.dirty = t, x, y, w, h, bbox, dirty

    @classmethod
    def around(cls, img_t, pad_x, pad_y):
        """New arena holding a copy of img_t with pad_x / pad_y zero pixels on every side."""
        h, w = img_t.shape[:2]
        t = torch.zeros((h + 2 * pad_y, w + 2 * pad_x) + tuple(img_t.shape[2:]), dtype=torch.uint8, device=img_t.device)
        t[pad_y:pad_y + h, pad_x:pad_x + w] = img_t
        return cls(t, pad_x, pad_y, w, h)

    def view(self):
        return self.t[self.y:self.y + self.h, self.x:self.x + self.w]

    def ensure(self, x0, y0, x1, y1, slack_x, slack_y):
        """Make the arena rectangle [x0,x1) x [y0,y1) (arena coordinates, may be negative / beyond the size)
        addressable; returns the (dx, dy) by which arena coordinates shifted."""
        Ha, Wa = self.t.shape[:2]
        if x0 >= 0 and y0 >= 0 and x1 <= Wa and y1 <= Ha:
            return 0, 0
        # extend the sides that overflowed, with slack, and never shrink the others (a path that jitters up and
        # down must not trade one side's slack for the other's on every frame); only the mosaic is copied,
        # everything else is zero
        left = (-x0 + slack_x) if x0 < 0 else 0
        top = (-y0 + slack_y) if y0 < 0 else 0
        right = (x1 - Wa + slack_x) if x1 > Wa else 0
        bottom = (y1 - Ha + slack_y) if y1 > Ha else 0
        t = torch.zeros((top + Ha + bottom, left + Wa + right) + tuple(self.t.shape[2:]), dtype=torch.uint8,
                        device=self.t.device)
        t[top + self.y:top + self.y + self.h, left + self.x:left + self.x + self.w] = self.view()
        left, top = left + self.x, top + self.y
        dx, dy = left - self.x, top - self.y
        self.t, self.x, self.y = t, left, top
        if self.bbox:
            b = self.bbox
            self.bbox = (b[0] + dx, b[1] + dy, b[2] + dx, b[3] + dy)
        return dx, dy


class DeviceImage:
    __slots__ = ("t", "_host", "_arena")

    def __init__(self, tensor, arena=None):
        if tensor.dtype != torch.uint8 or tensor.ndim not in (2, 3):
            raise TypeError("DeviceImage wraps a uint8 (H,W) or (H,W,C) tensor")
        self.t = tensor
        self._host = None
        self._arena = arena  # set on mosaics returned by Stitcher.stitch: the next stitch blends into it in place

    @classmethod
    def from_any(cls, img, device):
        """ndarray / DeviceImage / tensor -> DeviceImage on `device`.  Like the reference's pad_image
        (utils/proc.py:283-291) the pixel type is uint8: wider integer inputs wrap modulo 256."""
        if isinstance(img, cls):
            return img if img.t.device == device else cls(img.t.to(device))
        if isinstance(img, torch.Tensor):
            return cls(img.to(device=device, dtype=torch.uint8))
        a = np.asarray(img)
        if a.dtype != np.uint8:
            a = a.astype(np.uint8)          # numpy's uint8 assignment semantics (wrap), as the reference
        d = cls(torch.from_numpy(np.ascontiguousarray(a)).to(device))
        d._host = a
        return d

    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return self.t.ndim

    @property
    def dtype(self):
        return np.dtype(np.uint8)

    @property
    def channels(self):
        return 1 if self.t.ndim == 2 else self.t.shape[2]

    def __len__(self):
        return self.t.shape[0]

    def numpy(self):
        if self._host is None:
            self._host = self.t.cpu().numpy()
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def invalidate(self):
        self._host = None

    def __getitem__(self, key):
        v = self.t[key]
        if isinstance(v, torch.Tensor) and v.ndim >= 2:
            return DeviceImage(v)
        return v.cpu().numpy()

    def __repr__(self):
        return f"DeviceImage(shape={self.shape}, device={self.t.device})"


class DepthImage:
    """A single-channel uint16 image resident in HBM: what Stitcher(depth16=True) keeps as its mosaic (the opt-in
    depth-preserving mode; the default wraps uint16 input modulo 256 as the reference's uint8 canvases do).  Array-like
    enough for the stitchers and for user code that takes refs[-1]: shape, ndim, dtype, np.asarray()."""

    def __init__(self, tensor):
        if tensor.dtype != torch.uint16 or tensor.ndim != 2:
            raise TypeError("DepthImage wraps a 2-D uint16 tensor")
        self.t = tensor

    @classmethod
    def from_any(cls, img, device):
        if isinstance(img, DepthImage):
            return img
        if isinstance(img, torch.Tensor):
            return cls(img.to(device))
        return cls(torch.from_numpy(np.ascontiguousarray(img)).to(device))

    @property
    def shape(self):
        return tuple(self.t.shape)

    @property
    def ndim(self):
        return 2

    @property
    def dtype(self):
        return np.dtype(np.uint16)

    def numpy(self):
        return self.t.contiguous().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.t.shape[0]
