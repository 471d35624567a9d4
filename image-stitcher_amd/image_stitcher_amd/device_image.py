"""DeviceImage: a uint8 image (H,W) or (H,W,C) resident in HBM, array-like enough to stand in for the
ndarray the reference hands around (`refs[-1]`, the first element of `stitch`'s result).

  np.asarray(img) / img.numpy()   download (cached until the device data changes)
  img[y0:y1, x0:x1]               zero-copy crop (a strided view, like numpy)
  img.shape / img.ndim / len()    as numpy
"""
import numpy as np
import torch


class DeviceImage:
    __slots__ = ("t", "_host")

    def __init__(self, tensor):
        if tensor.dtype != torch.uint8 or tensor.ndim not in (2, 3):
            raise TypeError("DeviceImage wraps a uint8 (H,W) or (H,W,C) tensor")
        self.t = tensor
        self._host = None

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
